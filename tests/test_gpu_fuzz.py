"""Seeded random scenes through every kernel, against the CPU oracle: frames bit-identical, work counters equal.

The other GPU tests render smooth closed meshes and hand-built corner cases; these are triangle SOUPS, loaded through
the real OBJ reader and BVH builder: overlapping and nested boxes, slivers, triangles five orders of magnitude apart in size,
exact duplicates and coplanar overlaps (candidates at EQUAL distance: the later triangle wins inside a leaf, fs:327, the
earlier visited leaf across leaves, fs:400), degenerate triangles, geometry behind and around the camera -- with random views,
frame shapes, sample counts, bounce counts, iteration caps and leaf caps.  What the reference does with such input is defined
by its text (raytracer.es.fs:297-443); the oracle restates it and the kernels must agree with the oracle exactly."""
import math
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
KERNELS = [0, 1, 2, 3, 4]


def soup(rng, kind):
    """(positions [V,3], triangles [T,3]) of one random scene"""
    def tris_around(centers, size):
        n = len(centers)
        a = centers + rng.normal(0, 1, (n, 3)) * size
        b = centers + rng.normal(0, 1, (n, 3)) * size
        c = centers + rng.normal(0, 1, (n, 3)) * size
        return np.stack([a, b, c], axis=1)

    if kind == "uniform":
        n = int(rng.integers(300, 2500))
        t = tris_around(rng.uniform(-1, 1, (n, 3)), np.exp(rng.uniform(math.log(0.01), math.log(0.4), (n, 1))))
    elif kind == "dense":       # a cloud deep enough for rays to run into the 400-visit cap on their own
        n = int(rng.integers(15000, 40000))
        t = tris_around(rng.uniform(-1, 1, (n, 3)), np.exp(rng.uniform(math.log(0.005), math.log(0.04), (n, 1))))
    elif kind == "clusters":
        k = int(rng.integers(3, 9))
        centers = rng.uniform(-1, 1, (k, 3))
        n = int(rng.integers(500, 4000))
        which = rng.integers(0, k, n)
        t = tris_around(centers[which] + rng.normal(0, 0.08, (n, 3)), np.exp(rng.uniform(math.log(0.002), math.log(0.1), (n, 1))))
    elif kind == "sizes":       # a few huge triangles among thousands of tiny ones
        n = int(rng.integers(800, 3000))
        size = np.exp(rng.uniform(math.log(1e-4), math.log(0.05), (n, 1)))
        size[rng.integers(0, n, 12)] = rng.uniform(1.0, 3.0, (12, 1))
        t = tris_around(rng.uniform(-1, 1, (n, 3)), size)
    elif kind == "duplicates":  # every triangle twice or three times, some copies shifted inside their own plane
        n = int(rng.integers(200, 900))
        base = tris_around(rng.uniform(-1, 1, (n, 3)), np.exp(rng.uniform(math.log(0.05), math.log(0.5), (n, 1))))
        e = base[:, 1] - base[:, 0]
        shifted = base + (e * rng.uniform(-0.3, 0.3, (n, 1)))[:, None, :]
        t = np.concatenate([base, base[rng.permutation(n)], shifted, base[: n // 3]])
        t = t[rng.permutation(len(t))]
    elif kind == "planes":      # axis-aligned sheets: rays in their plane, boxes of zero thickness
        n = int(rng.integers(300, 1500))
        t = tris_around(rng.uniform(-1, 1, (n, 3)), 0.15)
        axis = rng.integers(0, 3, n)
        level = np.round(rng.uniform(-1, 1, n) * 4) / 4
        for k in range(3):
            sel = axis == k
            t[sel, :, k] = level[sel, None]
    elif kind == "degenerate":  # zero-area and needle triangles among ordinary ones
        n = int(rng.integers(300, 1200))
        t = tris_around(rng.uniform(-1, 1, (n, 3)), 0.1)
        z = rng.integers(0, n, n // 6)
        t[z, 2] = t[z, 1]                              # two corners coincide
        w = rng.integers(0, n, n // 6)
        t[w, 2] = t[w, 0] + (t[w, 1] - t[w, 0]) * 0.5  # collinear
        q = rng.integers(0, n, n // 8)
        t[q, 1] = t[q, 0] + rng.normal(0, 1e-7, (len(q), 3))
    else:
        raise ValueError(kind)
    t = t.astype(np.float32)
    pos = t.reshape(-1, 3)
    tri = np.arange(len(pos), dtype=np.int64).reshape(-1, 3)
    return pos, tri


def random_view(rng, world):
    view = world.default_view()
    axis = rng.normal(0, 1, 3)
    axis /= np.linalg.norm(axis)
    # one view in four leaves the object un-rotated: the rays of a frame of odd width / height then include directions with
    # components that are exactly zero (the centre column and row; true division, and the children's order for D[axis] == 0)
    angle = 0.0 if rng.integers(0, 4) == 0 else float(rng.uniform(0, 2 * math.pi))
    view.object_rotation[:] = [angle, *axis.astype(float)]
    axis = rng.normal(0, 1, 3)
    axis /= np.linalg.norm(axis)
    view.light_rotation[:] = [float(rng.uniform(0, 2 * math.pi)), *axis.astype(float)]
    # from well outside to the middle of the soup (geometry behind and around the eye)
    view.zoom = view.zoom * float(rng.choice([1.0, 0.6, 0.3, 0.05]))
    return view


# 35 scenes x 4 shots in the suite; SHRAY_FUZZ_ROUNDS=n runs n times as many (seeds go on counting)
CASES = [(seed, kind) for seed, kind in enumerate(["uniform", "clusters", "sizes", "duplicates", "planes", "degenerate", "dense"]
                                                  * (5 * max(1, int(os.environ.get("SHRAY_FUZZ_ROUNDS", "1")))))]


@pytest.mark.parametrize("seed,kind", CASES)
def test_random_soup(pkg, gpu, oracle_mod, tmp_path, seed, kind):
    rng = np.random.default_rng(1000 + seed)
    pos, tri = soup(rng, kind)
    path = os.path.join(tmp_path, f"soup{seed}.obj")
    pkg.scenes.write_obj(path, pos, tri)
    world = pkg.World(path)
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(256, seed=seed) if seed % 2 else pkg.scenes.environment_grid(256)
    scene = pkg.Scene(desc, env, device=0)
    try:
        for shot in range(4):
            W, H = [(96, 64), (61, 47), (128, 40), (33, 97)][int(rng.integers(0, 4))]
            spp = int(rng.choice([1, 1, 2, 3, 5]))
            material = int(rng.choice([0, 6, 5, 2]))
            params = world.frame_params(W, H, random_view(rng, world), material=material, diffuse=int(rng.integers(0, 3)))
            params.bounce_count = int(rng.choice([1, 2, 3, 4]))
            params.max_bvh_iterations = int(rng.choice([400, 400, 400, 150, 60, 17, 1000]))
            params.max_leaf_tests = int(rng.choice([10, 10, 3, 1]))
            params.normals_fp16 = int(rng.integers(0, 2))
            params.cast_shadows = int(rng.integers(0, 4) > 0)
            params.tonemap = int(rng.integers(0, 4) > 0)
            what = f"seed {seed} {kind} shot {shot}: {W}x{H} spp {spp} material {material} bounces {params.bounce_count} " \
                   f"cap {params.max_bvh_iterations} leaf cap {params.max_leaf_tests}"
            want, cpu = oracle_mod.render(desc, env, params, W, H, spp)
            for kernel in KERNELS:
                scene.set_kernel(kernel)
                got, counters = scene.render_counters(params, W, H, spp)
                plain = scene.render(params, W, H, spp)
                differing = int((got.view(np.uint32) != want.view(np.uint32)).sum())
                assert differing == 0, f"{what}, kernel {kernel}: {differing} floats differ from the oracle's"
                assert np.array_equal(plain.view(np.uint32), got.view(np.uint32)), f"{what}, kernel {kernel}: timed and counting instances differ"
                assert counters == cpu, f"{what}, kernel {kernel}: counters {counters} != oracle {cpu}"
    finally:
        scene.close()


@pytest.mark.parametrize("seed,kind", CASES[:len(CASES) * 3 // 5])
def test_random_soup_in_frame_batches(pkg, gpu, oracle_mod, tmp_path, seed, kind):
    """The same soups through shray_render_batch_device (the instances the bench's frame loop runs: several frames per launch,
    whole frames and tile sets), every frame against the oracle."""
    import torch
    N = pkg._native
    rng = np.random.default_rng(5000 + seed)
    pos, tri = soup(rng, kind)
    path = os.path.join(tmp_path, f"soup{seed}.obj")
    pkg.scenes.write_obj(path, pos, tri)
    world = pkg.World(path)
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(256, seed=seed)
    scene = pkg.Scene(desc, env, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for spp, all_metal in ((1, True), (1, False), (3, True), (4, False)):
            W, H = [(96, 64), (61, 47), (128, 40)][int(rng.integers(0, 3))]
            cap = int(rng.choice([400, 400, 60, 17]))
            frames = []
            for k in range(int(rng.integers(2, 6))):
                material = 0 if all_metal else int(rng.choice([0, 6, 5]))
                p = world.frame_params(W, H, random_view(rng, world), material=material)
                p.max_bvh_iterations = cap
                p.bounce_count = int(rng.choice([2, 3]))
                frames.append(p)
            want = [oracle_mod.render(desc, env, p, W, H, spp)[0] for p in frames]
            for tiles in (None, N.TileSet(32, 32, 3, int(rng.integers(0, 3)))):
                nbytes = pkg.tracer.tile_buffer_bytes(W, H, tiles)
                got = torch.zeros(len(frames), nbytes // 4, dtype=torch.float32, device="cuda:0")
                scene.render_batch_into(frames, W, H, spp, got.data_ptr(), nbytes, stream, tiles)
                torch.cuda.synchronize()
                for k, p in enumerate(frames):
                    if tiles is None:
                        frame = got[k].cpu().numpy().reshape(H, W, 4)
                        differing = int((frame.view(np.uint32) != want[k].view(np.uint32)).sum())
                        assert differing == 0, f"seed {seed} {kind} spp {spp} cap {cap} frame {k} of {len(frames)}: {differing} floats differ"
                    else:
                        one = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda:0")
                        scene.render_into(p, W, H, spp, one.data_ptr(), stream, tiles)
                        torch.cuda.synchronize()
                        assert torch.equal(got[k], one), f"seed {seed} {kind} spp {spp} tile set, frame {k}"
    finally:
        scene.close()
