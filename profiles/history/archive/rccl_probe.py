"""Does a one-rank RCCL communicator come up on this box?  (NCCL's bootstrap needs a socket interface.)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
print(open("/proc/net/dev").read())
if len(sys.argv) > 1:
    os.environ["NCCL_SOCKET_IFNAME"] = sys.argv[1]
os.environ["NCCL_DEBUG"] = "INFO"
from __graft_entry__ import load_package
pkg = load_package()
from shader_ray_amd import multigpu
world = pkg.World(os.path.join(ROOT, "tests", "golden", "lobed_528.trisrc"))
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(128), device=0)
t0 = time.time()
uid = multigpu.unique_id()
cfg = multigpu.make_config(0, 1, 64, 64, 1, 1, multigpu.ROOT0, multigpu.RCCL)
try:
    me = multigpu.Rank(scene, cfg, uid)
    print("communicator up in %.1f s" % (time.time() - t0))
except Exception as exc:
    print("FAILED after %.1f s: %r" % (time.time() - t0, exc))
