"""Throwaway (round 4): distinct nodes among the walking lanes of a wave-visit.  The -DSHRAY_KHIST build adds its histogram
to the timed form's tallies; this script renders the same frames with the shipped library and with that build (one process
each: SHRAY_HIP_LIB) and prints the difference."""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from __graft_entry__ import load_package
    import helpers
    pkg = load_package()
    world = pkg.World(helpers.bunny_trisrc())
    scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(2048), device=0)
    view = world.default_view()
    tot = {}
    for i in range(5):
        pkg.host.trackball_motion(view.object_rotation, 0.1, 0.04)
        p = world.frame_params(1920, 1080, view)
        _, c = scene.render_counters_timed(p, 1920, 1080, 1, 4, want_image=False)
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
    print(json.dumps(tot))
    sys.exit(0)
res = []
for lib in ("", os.path.join(ROOT, "shader-ray_amd", "_variants", "libshray_hip_khist.so")):
    env = dict(os.environ, SHRAY_HIP_LIB=lib)
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, check=True).stdout
    res.append(json.loads(out.strip().splitlines()[-1]))
d = {k: res[1][k] - res[0][k] for k in res[0]}
wv = d["traversals"]
print("wave-visits", wv, "avg walking lanes %.2f" % (d["shaded_hits"] / wv), "avg distinct nodes %.2f" % (d["env_lookups"] / wv))
print("k=1: %.3f  k=2: %.3f  k=3..4: %.3f  k>4: %.3f" % (d["leaf_visits"] / wv, d["triangle_tests"] / wv, d["bad_hits"] / wv,
                                                       1 - (d["leaf_visits"] + d["triangle_tests"] + d["bad_hits"]) / wv))
print("lane-visits", res[0]["node_visits"])
