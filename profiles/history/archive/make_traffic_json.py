#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE rows of a profiles/run_profile.sh summary into
profiles/latest_traffic.json, which bench.py reports as roofline.traffic.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both
counters in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for 16-byte-per-lane
loads, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import json
import re
import sys

summary = open(sys.argv[1]).read()
kernel = sys.argv[2] if len(sys.argv) > 2 else "trace_stack_batch_kernel<false, true, true>"
block = summary[summary.index(kernel, summary.index("PMC counters")):]
fetch = float(re.search(r"FETCH_SIZE\s+n=\s*\d+\s+avg\s+([\d.]+)", block).group(1))
write = float(re.search(r"WRITE_SIZE\s+n=\s*\d+\s+avg\s+([\d.]+)", block).group(1))
hbm = int((2 * fetch + write) * 1024)
out = {"kernel": kernel, "fetch_size_kib_raw": fetch, "write_size_kib": write,
       "hbm_bytes_per_launch": hbm,
       "note": "2 x FETCH_SIZE (gfx950 half-count correction for 16 B/lane loads) + WRITE_SIZE, KiB -> bytes; "
               "the scene + environment working set is L2 / Infinity-Cache resident, so this is far below the "
               "algorithmic byte count", "source": sys.argv[1]}
json.dump(out, open("profiles/latest_traffic.json", "w"), indent=1)
print(json.dumps(out))
