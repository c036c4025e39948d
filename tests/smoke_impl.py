"""__graft_entry__.smoke(): one small frame of the hot path on cuda:0, checked against
the oracle (pixels within 1e-4 relative, traversal counters exactly equal)."""
import numpy as np


def run_smoke():
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import helpers
    import oracle
    from __graft_entry__ import load_package

    pkg = load_package()
    world = pkg.World(helpers.small_trisrc())
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(256)
    W = H = 96
    for material in (0, 6):
        params = world.frame_params(W, H, material=material)
        scene = pkg.Scene(desc, env, device=0)
        got, gpu_counters = scene.render_counters(params, W, H, 1)
        plain = scene.render(params, W, H, 1)
        want, cpu_counters = oracle.render(desc, env, params, W, H, 1)
        helpers.assert_images_match(got, want, f"smoke material {material}")
        assert np.array_equal(got, plain), "counting and plain kernels disagree"
        assert gpu_counters == cpu_counters, (gpu_counters, cpu_counters)
        scene.close()
    print("smoke ok: %dx%d, gold + plaster, pixels within 1e-4 of the oracle, counters equal" % (W, H))
