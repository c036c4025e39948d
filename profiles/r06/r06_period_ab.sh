#!/bin/bash
# round 6: launches of a shape between two re-sorts of its dispatch order (SHRAY_DISPATCH_PERIOD), on the orbit one frame at a time
for rep in 1 2; do
for p in ${PERIODS:-8 4 2 1 16}; do
  echo -n "period $p: "
  SHRAY_DISPATCH_PERIOD=$p bash profiles/r05/r05_quick_ab.sh || exit 1
done; done
