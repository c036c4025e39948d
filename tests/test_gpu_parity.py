"""Parity of the HIP path against the CPU oracle, through the C ABI, on a real MI355X.

Bar (BASELINE.json north_star): pixels within 1e-4 relative of the CPU evaluation of the
shader.  Because every traversal decision is made with single-rounded IEEE fp32
arithmetic on both sides, the tests also demand that the per-ray work counters (node
visits, leaf visits, triangle tests, shaded hits, ...) are EXACTLY equal -- i.e. the
kernel visits the same nodes and tests the same triangles as the reference's threaded
traversal would -- and allow zero out-of-tolerance pixels."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
from helpers import assert_images_match, default_params, single_leaf_scene

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# 0 = packed stack kernel, 1 = literal threaded kernel, 2 = pool kernel (waves merge mid-traversal),
# 3 = stack kernel with both children of a node per turn (its counters come from its own counting twin),
# 4 = wavefront form: one launch per bounce, live paths compacted in between (counters: kernel 0's counting twin)
KERNELS = [0, 1, 2, 3, 4]


@pytest.fixture(scope="module")
def env_sky(pkg):
    return pkg.scenes.environment_hdr_sky(512)


@pytest.fixture(scope="module")
def bunny(pkg, gpu, env_sky):
    world = pkg.World(helpers.bunny_trisrc())
    desc = world.flatten()
    scene = pkg.Scene(desc, env_sky, device=0)
    yield world, desc, scene
    scene.close()


def check_against_oracle(oracle_mod, scene, desc, env, params, W, H, spp, what):
    want, cpu = oracle_mod.render(desc, env, params, W, H, spp)
    for kernel in KERNELS:
        scene.set_kernel(kernel)
        got, gpu_counters = scene.render_counters(params, W, H, spp)
        plain = scene.render(params, W, H, spp)
        assert_images_match(got, want, f"{what} kernel {kernel}")
        # stronger than the 1e-4 bar: every operation on the path is a specified IEEE fp32
        # operation on both sides, so the frames are expected to be bit-identical
        differing = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        assert differing == 0, f"{what} kernel {kernel}: {differing} floats are not bit-identical to the oracle"
        assert np.array_equal(plain, got), f"{what} kernel {kernel}: plain and counting kernels differ"
        assert gpu_counters == cpu, f"{what} kernel {kernel}: counters {gpu_counters} != oracle {cpu}"
    scene.set_kernel(0)
    return want


@pytest.mark.parametrize("material", [0, 6])
def test_golden_frames(pkg, gpu, material):
    """Committed oracle frames of the committed scene file (tests/golden)."""
    frames = np.load(os.path.join(GOLDEN, "lobed_528.oracle.npz"))
    world = pkg.World(os.path.join(GOLDEN, "lobed_528.trisrc"))
    scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(128), device=0)
    name = "gold" if material == 0 else "plaster"
    for kernel in KERNELS:
        scene.set_kernel(kernel)
        got, counters = scene.render_counters(world.frame_params(64, 64, material=material), 64, 64, 1)
        assert_images_match(got, frames[name], f"golden {name} kernel {kernel}")
        assert [counters[k] for k in sorted(counters)] == frames[name + "_counters"].tolist()
    if material == 6:
        got = scene.render(world.frame_params(64, 64, material=6), 64, 64, 4)
        assert_images_match(got, frames["plaster_4spp"], "golden plaster 4spp")
    scene.close()


def test_config1_primary_rays_256(pkg, gpu, oracle_mod, bunny, env_sky):
    """BASELINE config 1: bunny-class mesh, 256x256, 1 spp, primary rays only."""
    world, desc, scene = bunny
    params = world.frame_params(256, 256, material=0)
    params.bounce_count = 1
    check_against_oracle(oracle_mod, scene, desc, env_sky, params, 256, 256, 1, "config 1")


@pytest.mark.parametrize("material,spp", [(0, 1), (6, 1), (6, 3), (5, 1)])
def test_bunny_full_path(pkg, gpu, oracle_mod, bunny, env_sky, material, spp):
    """3 bounces, Fresnel modulation, shadow rays (plaster), fp16 normals; 160x120."""
    world, desc, scene = bunny
    params = world.frame_params(160, 120, material=material)
    check_against_oracle(oracle_mod, scene, desc, env_sky, params, 160, 120, spp, f"bunny material {material} spp {spp}")


def test_rotated_view_and_fp32_normals(pkg, gpu, oracle_mod, bunny, env_sky):
    world, desc, scene = bunny
    view = world.default_view()
    view.object_rotation[:] = [0.9, 0.26726124, 0.53452248, 0.80178373]
    view.light_rotation[:] = [1.1, 0.0, 0.6, 0.8]
    view.zoom = view.zoom * 0.6
    params = world.frame_params(128, 96, view, material=6, diffuse=2)
    params.normals_fp16 = 0
    check_against_oracle(oracle_mod, scene, desc, env_sky, params, 128, 96, 1, "rotated, fp32 normals")
    params.cast_shadows = 0
    params.tonemap = 0
    check_against_oracle(oracle_mod, scene, desc, env_sky, params, 128, 96, 1, "no shadows, no tonemap")


def test_obj_scene_off_origin(pkg, gpu, oracle_mod, env_sky):
    world = pkg.World(helpers.small_obj_no_normals())
    desc = world.flatten()
    scene = pkg.Scene(desc, pkg.scenes.environment_grid(256), device=0)
    params = world.frame_params(96, 96, material=1)
    check_against_oracle(oracle_mod, scene, desc, pkg.scenes.environment_grid(256), params, 96, 96, 1, "obj + grid env")
    scene.close()


def test_hand_built_edge_cases(pkg, gpu, oracle_mod):
    import test_oracle_kat as kat
    env = pkg.scenes.environment_constant((0.5, 0.25, 2.0))
    cases = {
        "far triangle (env only)": (kat.far_away_triangle(), default_params(pkg, 40, 24)),
        "mirror quad": (single_leaf_scene(kat.mirror_quad()), default_params(pkg, 40, 24, zoom=3.0)),
        "plaster quad": (single_leaf_scene(kat.mirror_quad()), default_params(pkg, 40, 24, zoom=3.0, material=6)),
        "iteration cap 401": (kat.chain_scene(401), default_params(pkg, 8, 8)),
        "iteration cap 400": (kat.chain_scene(400), default_params(pkg, 8, 8)),
    }
    tris = [[[-5, -5, -float(k)], [5, -5, -float(k)], [0, 5, -float(k)]] for k in range(10)] + [[[-5, -5, 1.0], [5, -5, 1.0], [0, 5, 1.0]]]
    cases["11-triangle leaf"] = (single_leaf_scene(tris), default_params(pkg, 16, 16))
    cases["empty leaf"] = (single_leaf_scene([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], count=0), default_params(pkg, 16, 16))
    for what, (hand, params) in cases.items():
        scene = pkg.Scene(hand.desc, env, device=0)
        W = int(round(1.0 / 1.0)) and None
        w, h = (40, 24) if "quad" in what or "far" in what else ((8, 8) if "cap" in what else (16, 16))
        want, cpu = oracle_mod.render(hand.desc, env, params, w, h, 1)
        kernels = [1] if "cap" in what else KERNELS   # a chain of leaves is not a binary tree: threaded kernel only
        for kernel in kernels:
            scene.set_kernel(kernel)
            got, counters = scene.render_counters(params, w, h, 1)
            assert_images_match(got, want, f"{what} kernel {kernel}")
            assert counters == cpu, (what, kernel, counters, cpu)
        if "cap" in what:
            with pytest.raises(pkg._native.ShrayError):
                scene.set_kernel(0)     # refused: tables are not a canonical threaded tree
            got = scene.render(params, w, h, 1)   # default selection falls back to the threaded kernel
            assert_images_match(got, want, f"{what} default kernel")
        scene.close()


def test_iteration_cap_phases_on_a_binary_tree(pkg, gpu, oracle_mod):
    """Kernel 0's TIMED instances compare the visit counter with zero once per four visits (wave_traversal.h: lane_apply_cap),
    the counting twins and the oracle at every visit; a chain of leaves is not a binary tree and runs on kernel 1 only (above).
    Here a hand-built BINARY tree (test_oracle_kat.comb_scene: every ray visits all 2 b + 1 nodes of it, per traversal) is
    rendered with caps of exactly visits - 5 ... visits + 1 -- the ray needs cap + 5 ... cap - 1 visits: every phase of the
    four-visit block, the exact fit (no marker) and the one-too-many -- for two tree sizes whose visit counts differ modulo 4,
    through every form of the timed instances: a lone frame (the ordered dealing instance), two and four frames per launch (the
    throughput form), zero-diffuse and diffuse (the shadow traversal is capped too: an unlit hit), one and four samples (sample
    lanes), one to three bounces.  Frames bit-identical to the oracle's; the marker exactly where the oracle has it."""
    import torch
    import test_oracle_kat as kat
    env = pkg.scenes.environment_constant((0.5, 0.25, 2.0))
    W, H = 24, 16
    stream = torch.cuda.current_stream().cuda_stream
    marker_seen = clean_seen = 0
    for branches in (20, 21):                    # 41 and 43 visits per traversal
        hand = kat.comb_scene(branches)
        visits = 2 * branches + 1
        scene = pkg.Scene(hand.desc, env, device=0)
        scene.set_kernel(0)
        for material in (0, 6):
            for spp in (1, 4):
                for cap in range(visits - 5, visits + 2):
                    params = default_params(pkg, W, H, material=material)
                    params.max_bvh_iterations = cap
                    params.bounce_count = 1 + (cap + branches) % 3
                    want, cpu = oracle_mod.render(hand.desc, env, params, W, H, spp)
                    assert (cpu["bad_hits"] > 0) == (cap < visits), (cap, visits, cpu)
                    marker_seen += cap < visits
                    clean_seen += cap >= visits
                    what = f"comb of {branches} branches, material {material}, {spp} spp, cap {cap} of {visits} visits"
                    got = scene.render(params, W, H, spp)                       # one frame per launch
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what
                    for count in (2, 4):                                        # several frames per launch
                        out = torch.zeros(count, H, W, 4, dtype=torch.float32, device="cuda")
                        scene.render_batch_into([params] * count, W, H, spp, out.data_ptr(), H * W * 16, stream)
                        torch.cuda.synchronize()
                        for k in range(count):
                            assert np.array_equal(out[k].cpu().numpy().view(np.uint32), want.view(np.uint32)), (what, count, k)
                    counted, counters = scene.render_counters(params, W, H, spp)  # the counting twin: the oracle's tallies
                    assert np.array_equal(counted.view(np.uint32), want.view(np.uint32)) and counters == cpu, what
        scene.close()
    assert marker_seen >= 40 and clean_seen >= 16


def test_triangles_at_the_ends_of_a_leaf_range(pkg, gpu, oracle_mod):
    """fs:327-331 rejects a candidate outside the range the leaf's box test left.  The kernels park BOUNDS of that range
    (round 4, wave_traversal.h: visit_decision) and hold a candidate that is about to be accepted within 2^-19 of an end
    against the exact range: triangles a few 1e-7 in front of, on and behind the near and the far face of a hand-built leaf
    box -- no inflation, the faces are the triangles' planes -- must be hit or missed exactly as the oracle's divisions say, in
    the dealt stage (gold, one sample), the plain loop (plaster, two samples) and every other kernel."""
    from helpers import END, HandScene
    env = pkg.scenes.environment_constant((0.5, 0.25, 2.0))

    def leaf(zs, box_z):
        tris = [[[-5, -5, z], [5, -5, z], [0, 5, z]] for z in zs]
        pts = np.asarray(tris, dtype=np.float32).reshape(-1, 3)
        normals = np.tile(np.asarray([0, 0, 1], np.float32), (len(pts), 1))
        hm = np.full((8, 1, 2), END, dtype=np.float32)
        return HandScene(pts, normals, [[-6, -6, box_z[0]]], [[6, 6, box_z[1]]], hm, [[0, len(zs)]], 0)

    near = leaf([1e-5, 3e-6, 1e-6, 3e-7, 1e-7, 0.0, -1e-7, -0.5], (-1.0, 0.0))        # the box's near face is z = 0
    far = leaf([-1.0 - 1e-6, -1.0 - 3e-7, -1.0 - 1e-7, -1.0, -1.0 + 1e-7], (-1.0, 0.0))  # its far face z = -1
    only_outside = leaf([1e-4, 2e-5, -1.0 - 2e-5, -1.0 - 1e-4], (-1.0, 0.0))        # beyond the bounds' 2^-20 band: never hit
    hits = {}
    for what, hand in (("near face", near), ("far face", far), ("outside only", only_outside)):
        scene = pkg.Scene(hand.desc, env, device=0)
        for material, spp in ((0, 1), (6, 2)):
            for zoom in (3.0, 2.9999998, 7.25):
                params = default_params(pkg, 48, 32, zoom=zoom, material=material)
                want = check_against_oracle(oracle_mod, scene, hand.desc, env, params, 48, 32, spp, f"{what} {material} {zoom}")
                hits[(what, material, zoom)] = float((np.abs(want[..., :3] - want[0, 0, :3]).max(axis=-1) > 1e-6).mean())
        scene.close()
    # (the cases are not vacuous: the near- and far-face scenes are hit over most of the frame -- triangles a few 1e-7 outside a
    # face included, where the ray's own rounding decides --, the scene whose triangles are 2e-5 and more outside never)
    assert min(v for k, v in hits.items() if k[0] != "outside only" and k[1] == 0) > 0.5
    assert max(v for k, v in hits.items() if k[0] == "outside only") == 0.0


def test_nan_candidates_in_a_dealt_leaf(pkg, gpu, oracle_mod):
    """A triangle whose determinant overflows gives d = u = v = NaN, which fail none of the shader's comparisons
    (raytracer.es.fs:312-340): the sequential loop accepts it and then accepts the next candidate whatever its
    distance.  The dealt leaf stage ranks candidates by (d, index), which cannot express that; it has to notice the
    unordered candidate and fall back to the sequential loop.  The leaf's box covers a few pixels of each 8x8 wave
    tile, so the parked rays are few and the stage really deals (G = 4..16 workers per ray); the huge triangle sits
    before, between and after finite hits, in the same worker's share and in another's."""
    import test_oracle_kat as kat
    env = pkg.scenes.environment_constant((0.5, 0.25, 2.0))
    for material in (0, 6):
        for bounces in (1, 3):
            params = default_params(pkg, 16, 16, zoom=3.0, material=material)
            params.bounce_count = bounces
            for order in kat.NAN_ORDERS:
                hand = kat.nan_leaf_scene(order)
                scene = pkg.Scene(hand.desc, env, device=0)
                want, cpu = oracle_mod.render(hand.desc, env, params, 16, 16, 1)
                assert not np.isnan(want).any()       # the tone map turns a NaN radiance into 0 (max(0, NaN) = 0)
                for kernel in KERNELS:
                    scene.set_kernel(kernel)
                    got, counters = scene.render_counters(params, 16, 16, 1)     # 256-thread twin: dealt leaf stage
                    plain = scene.render(params, 16, 16, 1)                      # one-wave instance: dealt (one frame per launch)
                    what = f"order {order}, material {material}, {bounces} bounce(s), kernel {kernel}"
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what
                    assert np.array_equal(plain.view(np.uint32), want.view(np.uint32)), what + " (timed instance)"
                    assert counters == cpu, (what, counters, cpu)
                scene.close()


@pytest.mark.parametrize("spp", [1, 4])
def test_counters_of_the_timed_instances(pkg, gpu, oracle_mod, bunny, env_sky, spp):
    """shray_render_counters_timed: the tallies of the instance the timed launches run (one-wave workgroups, sample
    lanes, shadow rays that stop at their first hit).  For a metal there are no shadow rays: every tally equals the
    oracle's.  For the diffuse material the shadow rays do less work than the reference's full traversals -- fewer node
    visits and triangle tests, the same traversals, hits, lookups -- and the image is the same bit for bit."""
    world, desc, scene = bunny
    scene.set_kernel(0)
    W, H = 192, 112
    for material in (0, 6):
        params = world.frame_params(W, H, material=material)
        want, cpu = oracle_mod.render(desc, env_sky, params, W, H, spp)
        for frames_per_launch in (1, 2):           # the dealt and the plain leaf stage (capi.hip: leaf_stage_policy)
            got, timed = scene.render_counters_timed(params, W, H, spp, frames_per_launch)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (material, spp, frames_per_launch)
            if material == 0:
                assert timed == cpu, (timed, cpu)
            else:
                same = ("shaded_hits", "env_lookups", "traversals", "bad_hits", "samples")
                assert all(timed[k] == cpu[k] for k in same), (timed, cpu)
                assert timed["node_visits"] < cpu["node_visits"] and timed["triangle_tests"] < cpu["triangle_tests"]
                assert timed["leaf_visits"] <= cpu["leaf_visits"]
                assert timed["node_visits"] > cpu["node_visits"] // 2      # the primary rays' walks are all there
    # kernels without a timed form of their own report their ordinary counters
    params = world.frame_params(W, H, material=6)
    scene.set_kernel(1)
    assert scene.render_counters_timed(params, W, H, 1)[1] == scene.render_counters(params, W, H, 1)[1]
    scene.set_kernel(0)


@pytest.mark.parametrize("which", [0, 1])
def test_environment_storage_float32_and_unorm8(pkg, gpu, oracle_mod, which):
    """shray_scene_set_environment_storage: the floats as given, or what the reference's unsized GL_RGB upload
    (ray.cpp:508) becomes on most drivers -- 8 bits, clamped to [0, 1], mip levels included.  160x120, both ways,
    the plain view and the filtered (which == 1, mip pyramid) view: bit-identical to the oracle under the same rule."""
    N = pkg._native
    world = pkg.World(os.path.join(GOLDEN, "lobed_528.trisrc"))
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(256)
    W, H = 160, 120
    view = world.default_view()
    view.which = which
    params = world.frame_params(W, H, view, material=0)
    scene = pkg.Scene(desc, None, device=0)
    frames = {}
    for storage in (N.ENV_FLOAT32, N.ENV_UNORM8):
        scene.set_environment(env, storage)
        try:
            oracle_mod.set_env_storage(storage)
            want, cpu = oracle_mod.render(desc, env, params, W, H, 1)
        finally:
            oracle_mod.set_env_storage(0)
        for kernel in KERNELS:
            scene.set_kernel(kernel)
            got, counters = scene.render_counters(params, W, H, 1)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (storage, kernel)
            assert np.array_equal(scene.render(params, W, H, 1), got) and counters == cpu
        frames[storage] = want
    assert not np.array_equal(frames[N.ENV_FLOAT32], frames[N.ENV_UNORM8])     # the sun above 1 is gone
    with pytest.raises(N.ShrayError):
        scene.set_environment(env, 7)
    scene.close()


def test_kernels_match_the_reference_shaders(pkg, gpu, oracle_mod):
    """The HIP kernels against frames rendered by the REFERENCE'S OWN GLSL (tests/golden/glsl_reference/*.npz: raytracer.vs +
    raytracer.es.fs, unmodified, on Mesa's llvmpipe).  tests/test_reference_shader.py holds the ORACLE to those frames -- every
    pixel outside 1e-4 classified, none unexplained --; here every kernel, through the C ABI, renders every one of those
    cases bit-identical to the oracle, so that the same statement holds for the kernels' frames pixel by pixel (and the
    count of pixels outside 1e-4 of the reference's frame is the oracle's)."""
    import glsl_cases
    N = pkg._native
    checked = 0
    for name, case in glsl_cases.cases(pkg).items():
        want = np.load(os.path.join(glsl_cases.FIXTURES, name + ".npz"))["frame"]
        try:
            oracle_mod.set_env_storage(case["env_storage"])
            stated, _ = oracle_mod.render(case["scene"][0], case["env"], case["params"], case["width"], case["height"], 1)
        finally:
            oracle_mod.set_env_storage(0)
        scene = pkg.Scene(case["scene"][0], None, device=0)
        scene.set_environment(case["env"], N.ENV_UNORM8 if case["env_storage"] else N.ENV_FLOAT32)
        for kernel in KERNELS:
            try:
                scene.set_kernel(kernel)
            except N.ShrayError:
                continue            # a chain of leaves is not a binary tree: the literal kernel only
            got = scene.render(case["params"], case["width"], case["height"], 1)
            assert np.array_equal(got.view(np.uint32), stated.view(np.uint32)), (name, kernel)
            assert glsl_cases.out_of_tolerance(got, want).sum() == glsl_cases.out_of_tolerance(stated, want).sum()
            checked += 1
        scene.close()
    assert checked >= 150


def test_empty_world_renders_environment(pkg, gpu, oracle_mod, tmp_path):
    path = tmp_path / "empty.trisrc"
    path.write_text("")
    world = pkg.World(str(path))
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(64)
    scene = pkg.Scene(desc, env, device=0)
    params = default_params(pkg, 32, 32)
    check_against_oracle(oracle_mod, scene, desc, env, params, 32, 32, 1, "empty world")
    scene.close()


def test_tile_sets_reassemble_to_the_full_frame(pkg, gpu, bunny):
    """Interleaved tile ownership (the multi-GPU split) is bit-identical to one full render."""
    import torch
    from shader_ray_amd.multigpu import assemble_tiles
    world, desc, scene = bunny
    W, H = 200, 136           # not a multiple of the tile size: edge tiles are padded
    params = world.frame_params(W, H, material=0)
    full = scene.render(params, W, H, 1)
    N = pkg._native
    for kernel, (tw, th, stride) in [(k, g) for k in (0, 2, 3) for g in ((32, 32, 3), (16, 48, 2), (64, 32, 8))]:
        scene.set_kernel(kernel)
        parts = []
        for phase in range(stride):
            tiles = N.TileSet(tw, th, stride, phase)
            nbytes = pkg.tracer.tile_buffer_bytes(W, H, tiles)
            buf = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda:0")
            scene.render_into(params, W, H, 1, buf.data_ptr(), torch.cuda.current_stream().cuda_stream, tiles)
            torch.cuda.synchronize()
            parts.append(buf.cpu().numpy())
        frame = assemble_tiles(parts, W, H, tw, th)
        assert np.array_equal(frame, full), (kernel, tw, th, stride)
    scene.set_kernel(0)


def test_dispatch_order_leaves_the_frames_alone(pkg, gpu, bunny):
    """Heaviest patches first (shray_scene_dispatch_order): after a few launches of one shape the launches read a
    learnt permutation of the patches -- a permutation it must be, the silhouette's patches in front, and every frame
    bit-identical to the first launch's (which still ran in row-major order).  Lone whole frames and batches of a rank's
    tile set, on two streams at once."""
    import torch
    world, desc, scene = bunny
    N = pkg._native
    W, H = 640, 360
    frames = [world.frame_params(W, H, material=0)]
    view = world.default_view()
    for _ in range(3):
        pkg.host.trackball_motion(view.object_rotation, 0.05, 0.02)
        frames.append(world.frame_params(W, H, view, material=0))
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    # the library re-orders launches of one frame and launches of a tile set (capi.hip: launch_stack_views)
    for tiles in (None, N.TileSet(32, 32, 3, 1, 2)):
        nbytes = pkg.tracer.tile_buffer_bytes(W, H, tiles)

        def launch(out, stream):
            if tiles is None:
                for k, frame in enumerate(frames):
                    scene.render_into(frame, W, H, 1, out.data_ptr() + k * nbytes, stream.cuda_stream, None)
            else:
                scene.render_batch_into(frames, W, H, 1, out.data_ptr(), nbytes, stream.cuda_stream, tiles)

        first = torch.empty(len(frames) * nbytes // 4, dtype=torch.float32, device="cuda:0")
        launch(first, streams[0])
        torch.cuda.synchronize()
        again = [torch.empty_like(first) for _ in range(12)]
        for j, out in enumerate(again):
            launch(out, streams[j % 2])
        torch.cuda.synchronize()
        for out in again:
            assert torch.equal(out, first), "a re-ordered launch changed a frame"
        if os.environ.get("SHRAY_DISPATCH_ORDER") == "0":
            continue
        order = scene.dispatch_order()
        patches = order.size
        assert patches > 0 and np.array_equal(np.sort(order), np.arange(patches)), "not a permutation"
        assert not np.array_equal(order, np.arange(patches)), "twelve launches and still the identity"
    # a lone frame through shray_render (the same machinery, one frame per launch) against the counting twin's image
    params = frames[1]
    want, _ = scene.render_counters(params, W, H, 1)
    for _ in range(5):
        assert np.array_equal(scene.render(params, W, H, 1), want)
    # shapes side by side: a scene keeps the orders of its four most recent shapes (alternating between frame sizes must
    # neither start over every time nor mix the permutations up), a fifth and sixth evict the oldest
    sizes = [(320, 200), (200, 320), (256, 256), (640, 360), (128, 72), (96, 160)]
    wants = {}
    for w, h in sizes:
        p = world.frame_params(w, h, view, material=0)
        wants[(w, h)] = (p, scene.render_counters(p, w, h, 1)[0])
    for _ in range(6):
        for w, h in sizes:
            p, want_frame = wants[(w, h)]
            assert np.array_equal(scene.render(p, w, h, 1), want_frame), (w, h)
            order = scene.dispatch_order()
            patches = ((w + 15) // 16) * ((h + 15) // 16)
            assert order.size in (0, patches), (w, h, order.size)
            if order.size:
                assert np.array_equal(np.sort(order), np.arange(patches))


def test_dispatch_order_soak_without_host_synchronisation(pkg, gpu, bunny):
    """A long run that never synchronises with the host (what shray_dist_step's callers do): 480 one-frame launches
    rotated over four streams, the host ahead of the GPU all the way.  The dispatch-order ring is rewritten ~60 times
    meanwhile -- more often than it has entries -- and no permutation may be rewritten under a launch that still reads it
    (a torn permutation renders some patches twice and leaves others stale): every one of the 480 frames
    must equal its view's first render."""
    import torch
    world, desc, scene = bunny
    W, H, LAUNCHES = 1280, 720, 480
    view = world.default_view()
    views, wants = [], []
    for _ in range(4):
        pkg.host.trackball_motion(view.object_rotation, 0.04, 0.015)
        p = world.frame_params(W, H, view, material=0)
        views.append(p)
        wants.append(torch.from_numpy(scene.render_counters(p, W, H, 1)[0]).reshape(-1).to("cuda:0"))
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = torch.full((LAUNCHES, W * H * 4), -1.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for j in range(LAUNCHES):
        scene.render_into(views[j % 3], W, H, 1, outs[j].data_ptr(), streams[j % 4].cuda_stream, None)
    torch.cuda.synchronize()
    for j in range(LAUNCHES):
        assert torch.equal(outs[j], wants[j % 3]), f"launch {j} of the soak differs from its view's frame"
    if os.environ.get("SHRAY_DISPATCH_ORDER") != "0":
        order = scene.dispatch_order()
        assert np.array_equal(np.sort(order), np.arange(order.size)) and order.size == (W // 16) * (H // 16)


def test_full_size_properties_1080p(pkg, gpu, oracle_mod, bunny, env_sky):
    """BASELINE config 2 size (1920x1080, gold): size-independent properties instead of a
    full CPU render -- kernel 0 == kernel 1 bit for bit, run-to-run determinism, alpha = 1,
    counters identical between kernels, and a band of rows checked against the oracle."""
    world, desc, scene = bunny
    W, H = 1920, 1080
    params = world.frame_params(W, H, material=0)
    scene.set_kernel(0)
    a, ca = scene.render_counters(params, W, H, 1)
    a2 = scene.render(params, W, H, 1)
    scene.set_kernel(1)
    b, cb = scene.render_counters(params, W, H, 1)
    scene.set_kernel(0)      # (the pool kernel, not a product path, stays in the small-frame parity matrix only)
    assert np.array_equal(a, a2) and np.array_equal(a, b) and ca == cb
    # the timed instances' own tallies: a metal has no shadow rays, so they equal the reference's
    for frames_per_launch in (1, 2):
        t_img, t_counters = scene.render_counters_timed(params, W, H, 1, frames_per_launch)
        assert np.array_equal(t_img, a) and t_counters == ca
    # the stack kernel's two gold instances (capi.hip: leaf_stage_policy): one frame per launch runs the dealt
    # leaf stage, two frames per launch the plain leaf loop -- the same frame either way, on two streams at once
    import torch
    pair = torch.empty(2, H * W * 4, dtype=torch.float32, device="cuda:0")
    lone = torch.empty(H * W * 4, dtype=torch.float32, device="cuda:0")
    side = torch.cuda.Stream()
    scene.render_batch_into([params, params], W, H, 1, pair.data_ptr(), H * W * 16, torch.cuda.current_stream().cuda_stream)
    scene.render_into(params, W, H, 1, lone.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    for frame in (pair[0], pair[1], lone):
        assert np.array_equal(frame.cpu().numpy().reshape(H, W, 4), a)
    assert np.all(a[..., 3] == 1.0) and not np.isnan(a).any()
    assert ca["samples"] == W * H and ca["bad_hits"] == 0
    rows = (520, 560)    # through the middle of the object
    want, _ = oracle_mod.render(desc, env_sky, params, W, H, 1, rows=rows)
    assert_images_match(a[rows[0]:rows[1]], want[rows[0]:rows[1]], "1080p rows 520..560")


def test_million_triangle_scene(pkg, gpu, oracle_mod, env_sky):
    """BASELINE config 4 scene (1M-triangle OBJ, deep BVH) on a reduced frame: parity
    including the iteration-cap pixels (max_bvh_iterations = 400 is 'a little too few',
    raytracer.es.fs:381)."""
    world = pkg.World(helpers.million_obj())
    assert world.triangle_count == 1_000_000
    desc = world.flatten()
    scene = pkg.Scene(desc, env_sky, device=0)
    params = world.frame_params(192, 108, material=0)
    check_against_oracle(oracle_mod, scene, desc, env_sky, params, 192, 108, 1, "1M triangles")
    # a tighter cap makes the red marker appear: same pixels on both sides
    params.max_bvh_iterations = 60
    want = check_against_oracle(oracle_mod, scene, desc, env_sky, params, 192, 108, 1, "1M triangles, cap 60")
    red = np.array([oracle_mod.filmic(1.0), 0.0, 0.0], np.float32)
    assert (np.abs(want[..., :3] - red).max(axis=-1) < 1e-6).sum() > 0
    scene.close()


@pytest.mark.parametrize("channels", [3, 4])
def test_deinterleave_kernel_matches_the_host_mapping(pkg, gpu, channels):
    """shray_assemble_tiles_device (rank 0's de-interleave) against the numpy statement of the tile mapping."""
    import ctypes as C
    import torch
    from shader_ray_amd import multigpu
    N = pkg._native
    rng = np.random.default_rng(11)
    for (w, h, tw, th, world, frames) in ((100, 70, 16, 32, 3, 2), (200, 136, 32, 32, 8, 3), (33, 17, 16, 16, 1, 1), (64, 64, 32, 32, 5, 1)):
        per = multigpu.max_tiles_per_rank(w, h, tw, th, world)
        pad = 5                                            # frames need not be densely packed
        gathered = rng.random((world, frames, per * th * tw * channels + pad), dtype=np.float32)
        dev = torch.from_numpy(gathered).cuda()
        out = torch.full((frames, h, w, 4), -3.0, dtype=torch.float32, device="cuda:0")
        N.check(N.load_hip().shray_assemble_tiles_device(
            C.c_void_p(dev.data_ptr()), world, frames, channels, dev.stride(0) * 4, dev.stride(1) * 4, w, h, tw, th,
            C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        for f in range(frames):
            parts = []
            for r in range(world):
                px = gathered[r, f, : per * th * tw * channels].reshape(-1, channels)
                rgba = np.ones((px.shape[0], 4), dtype=np.float32)
                rgba[:, :channels] = px
                parts.append(rgba.reshape(-1))
            want = multigpu.assemble_tiles(parts, w, h, tw, th)
            assert np.array_equal(out[f].cpu().numpy(), want), (w, h, tw, th, world, f)
    with pytest.raises(N.ShrayError):      # two channels is not a wire format
        N.check(N.load_hip().shray_assemble_tiles_device(C.c_void_p(dev.data_ptr()), 5, 1, 2, 64, 64, 64, 64, 32, 32,
                                                         C.c_void_p(out.data_ptr()), None))
    with pytest.raises(N.ShrayError):      # strides smaller than a rank's frame
        N.check(N.load_hip().shray_assemble_tiles_device(C.c_void_p(dev.data_ptr()), 5, 1, 3, 64, 64, 64, 64, 32, 32,
                                                         C.c_void_p(out.data_ptr()), None))


def test_frame_batches_equal_single_launches(pkg, gpu, bunny):
    """shray_render_batch_device: every frame of a batch has its own parameters and lands where a
    single launch would have put it; bytes between frames are left alone."""
    import torch
    world, desc, scene = bunny
    N = pkg._native
    W, H = 200, 136
    frames = []
    for k, (material, rot) in enumerate(((0, 0.0), (6, 0.9), (3, 2.1), (0, 4.0), (5, 5.5))):
        view = world.default_view()
        if rot:
            view.object_rotation[:] = [rot, 0.26726124, 0.53452248, 0.80178373]
        frames.append(world.frame_params(W, H, view, material=material))
    frames[3].which = 5     # a non-differential debug view may share a batch with normal frames
    stream = torch.cuda.current_stream().cuda_stream
    for kernel in KERNELS:  # kernels 1 and 2 take the batch as consecutive launches
        scene.set_kernel(kernel)
        for tiles in (None, N.TileSet(32, 32, 3, 1), N.TileSet(16, 48, 2, 0)):
            nbytes = pkg.tracer.tile_buffer_bytes(W, H, tiles)
            stride = nbytes + 64
            want = []
            for p in frames:
                one = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda:0")
                scene.render_into(p, W, H, 1, one.data_ptr(), stream, tiles)
                want.append(one)
            got = torch.full((len(frames), stride // 4), -7.0, dtype=torch.float32, device="cuda:0")
            scene.render_batch_into(frames, W, H, 1, got.data_ptr(), stride, stream, tiles)
            torch.cuda.synchronize()
            for k in range(len(frames)):
                assert torch.equal(got[k, : nbytes // 4], want[k]), (kernel, tiles and tiles.tile_w, k)
                assert bool((got[k, nbytes // 4:] == -7.0).all())
            assert not torch.equal(want[0], want[1])
    scene.set_kernel(0)
    # more batches in flight than the library has parameter slots
    nbytes = pkg.tracer.tile_buffer_bytes(W, H, None)
    ring = torch.zeros(40, 2, nbytes // 4, dtype=torch.float32, device="cuda:0")
    for j in range(40):
        scene.render_batch_into([frames[j % 3], frames[(j + 1) % 3]], W, H, 1, ring[j].data_ptr(), nbytes, stream, None)
    torch.cuda.synchronize()
    singles = [scene.render(frames[k], W, H, 1) for k in range(3)]
    for j in range(40):
        assert np.array_equal(ring[j, 0].cpu().numpy().reshape(H, W, 4), singles[j % 3])
        assert np.array_equal(ring[j, 1].cpu().numpy().reshape(H, W, 4), singles[(j + 1) % 3])

    # the kernel instances picked by the launcher: all frames metallic (no diffuse branch), spp == 1 or not
    metals = []
    for rot in (0.4, 1.7):
        view = world.default_view()
        view.object_rotation[:] = [rot, 0.26726124, 0.53452248, 0.80178373]
        metals.append(world.frame_params(W, H, view, material=0))
    for spp, batch in ((1, metals), (3, metals), (3, [frames[1], frames[0]])):
        got = torch.zeros(2, nbytes // 4, dtype=torch.float32, device="cuda:0")
        scene.render_batch_into(batch, W, H, spp, got.data_ptr(), nbytes, stream, None)
        torch.cuda.synchronize()
        for k in range(2):
            assert np.array_equal(got[k].cpu().numpy().reshape(H, W, 4), scene.render(batch[k], W, H, spp)), (spp, k)

    one = torch.empty(nbytes // 4 * 2, dtype=torch.float32, device="cuda:0")
    with pytest.raises(N.ShrayError):
        scene.render_batch_into([], W, H, 1, one.data_ptr(), nbytes, stream, None)
    with pytest.raises(N.ShrayError):
        scene.render_batch_into([frames[0]] * 65, W, H, 1, one.data_ptr(), nbytes, stream, None)
    with pytest.raises(N.ShrayError):
        scene.render_batch_into([frames[0]] * 2, W, H, 1, one.data_ptr(), nbytes - 16, stream, None)
    mixed = frames[0].copy()
    mixed.which = 1         # carries ray differentials: a different kernel instance
    with pytest.raises(N.ShrayError) as e:
        scene.render_batch_into([frames[0], mixed], W, H, 1, one.data_ptr(), nbytes, stream, None)
    assert "differential" in str(e.value)
    scene.render_batch_into([mixed, mixed], W, H, 1, one.data_ptr(), nbytes, stream, None)
    torch.cuda.synchronize()
    assert np.array_equal(one[: nbytes // 4].cpu().numpy().reshape(H, W, 4), scene.render(mixed, W, H, 1))


def test_batch_too_large_for_an_interleaved_grid(pkg, gpu, bunny):
    """HIP rejects launches whose gridDim.x * blockDim.x reaches 2^32.  2048x1088 at 32 spp is 8,704 patches x 128
    one-wave workgroups; 64 such frames interleaved along grid.x would be 4.6e9 threads: the launcher goes back to
    grid.y = frame, and the frames are those of single launches."""
    import torch
    world, desc, scene = bunny
    W, H, spp, count = 2048, 1088, 32, 64
    frames = [world.frame_params(W, H, material=(0, 3)[k % 2]) for k in range(count)]
    out = torch.empty(count, H * W * 4, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_batch_into(frames, W, H, spp, out.data_ptr(), H * W * 16, stream, None)
    torch.cuda.synchronize()
    for k in (0, 1, 63):
        assert np.array_equal(out[k].cpu().numpy().reshape(H, W, 4), scene.render(frames[k], W, H, spp)), k
    assert torch.equal(out[0], out[62]) and torch.equal(out[1], out[63])
    del out


def test_error_paths(pkg, gpu, bunny, env_sky):
    N = pkg._native
    world, desc, scene = bunny
    params = world.frame_params(32, 32)
    fresh = pkg.Scene(desc, None, device=0)
    with pytest.raises(N.ShrayError) as e:
        fresh.render(params, 32, 32, 1)
    assert e.value.code == -7          # SHRAY_ERR_NO_ENVIRONMENT
    fresh.close()
    bad = params.copy()
    bad.which = 5
    with pytest.raises(N.ShrayError) as e:
        scene.render(bad, 32, 32, 4)     # the 5x5 reference view is per pixel: spp must be 1
    assert e.value.code == -1 and "which" in str(e.value)
    bad = params.copy()
    bad.struct_size = 12
    with pytest.raises(N.ShrayError):
        scene.render(bad, 32, 32, 1)
    with pytest.raises(N.ShrayError):
        scene.render(params, 0, 32, 1)
    # a link that points outside the node array is rejected at creation, never traversed
    hand = single_leaf_scene([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]])
    hand.keep["hm"][3, 0, 0] = 7.0
    with pytest.raises(N.ShrayError) as e:
        pkg.Scene(hand.desc, env_sky, device=0)
    assert e.value.code == -6          # SHRAY_ERR_BAD_TREE
    hand = single_leaf_scene([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], count=5)   # names triangles that do not exist
    with pytest.raises(N.ShrayError) as e:
        pkg.Scene(hand.desc, env_sky, device=0)
    assert e.value.code == -6


def test_command_line_harness_writes_the_same_frame(pkg, gpu, tmp_path):
    """tools/shray_render (the headless `./ray model background`, ray.cpp:954-1092 + the 's' key's
    color.ppm, :730-787) produces the frame the library renders, quantised to 8 bits, top row first."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "shader-ray_amd", "tools", "shray_render")
    exe = os.path.abspath(exe)
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(exe)), "tools"], check=True, stdout=subprocess.DEVNULL)
    model = os.path.join(GOLDEN, "lobed_528.trisrc")
    out = str(tmp_path / "color.ppm")
    subprocess.run([exe, model, "0.2, 0.4, 0.8", "-o", out, "-w", "96", "-h", "64", "-m", "6"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    blob = open(out, "rb").read()
    header, rest = blob.split(b"\n", 1)
    assert header == b"P6 96 64 255" and len(rest) == 96 * 64 * 3
    got = np.frombuffer(rest, np.uint8).reshape(64, 96, 3)
    world = pkg.World(model)
    scene = pkg.Scene(world.flatten(), pkg.load_background("0.2, 0.4, 0.8"), device=0)
    frame = scene.render(world.frame_params(96, 64, material=6), 96, 64, 1)
    want = (np.clip(frame[::-1, :, :3], 0, 1) * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(got, want)
    scene.close()


def test_axis_aligned_rays_take_the_true_division_path(pkg, gpu, oracle_mod):
    """Odd frame sizes put pixel centres exactly on the view axis: D.x = 0 and/or D.y = 0, so the
    slab test divides by zero (+-inf, or NaN where the origin lies on a slab plane).  Those lanes
    must leave the hoisted-reciprocal path (csrc/exact_div.h) and still match the oracle bit for bit."""
    import test_oracle_kat as kat
    env = pkg.scenes.environment_hdr_sky(64)
    # mirror quad whose box planes pass through x = 0 and y = 0: (lo - P) / D = 0 / 0 on the axis
    quad = [[[0, 0, 0], [4, 0, 0], [4, 4, 0]], [[0, 0, 0], [4, 4, 0], [0, 4, 0]]]
    for hand, W, H in ((single_leaf_scene(kat.mirror_quad()), 33, 17), (single_leaf_scene(quad), 17, 33)):
        params = default_params(pkg, W, H, zoom=3.0, material=6)
        scene = pkg.Scene(hand.desc, env, device=0)
        check_against_oracle(oracle_mod, scene, hand.desc, env, params, W, H, 1, f"axis-aligned rays {W}x{H}")
        scene.close()
    # and on a real tree: bunny at 65 x 65 (centre column and row have a zero direction component)
    world = pkg.World(helpers.small_trisrc())
    desc = world.flatten()
    scene = pkg.Scene(desc, env, device=0)
    view = world.default_view()
    view.object_position[:] = [-world.info.scene_center[0], -world.info.scene_center[1], -world.info.scene_center[2]]
    params = world.frame_params(65, 65, view, material=0)
    check_against_oracle(oracle_mod, scene, desc, env, params, 65, 65, 1, "axis-aligned rays through a BVH")
    scene.close()


@pytest.mark.parametrize("scale", [1e-24, 2e18])
def test_coordinates_outside_the_fast_division_range(pkg, gpu, oracle_mod, tmp_path, scale):
    """Scenes scaled far down / beyond 2^60: for the large one scene creation must turn the
    hoisted-reciprocal slab test off (box coordinates outside [2^-70, 2^60): every lane divides);
    results stay bit-identical either way.  (Neither object is actually hit: at 1e-24 every
    determinant is below the shader's absolute 1e-7 epsilon, raytracer.es.fs:312-315, at 2e18
    the triangle test's products overflow -- on both sides alike.)"""
    pos, tri = pkg.scenes.lobed_sphere_mesh(16, 32, bumpiness=0.2, ears=False, scale=scale)
    path = str(tmp_path / "scaled.obj")
    pkg.scenes.write_obj(path, pos, tri)
    world = pkg.World(path)
    assert np.isfinite(world.info.scene_extent)
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(64)
    scene = pkg.Scene(desc, env, device=0)
    params = world.frame_params(64, 48, material=0)
    check_against_oracle(oracle_mod, scene, desc, env, params, 64, 48, 1, f"scene scaled by {scale}")
    scene.close()


def test_box_plane_at_a_denormal_scale_coordinate(pkg, gpu, oracle_mod):
    """A hand-built leaf whose box has a plane at 1e-30 with the camera at the origin: the slab
    numerator is 1e-30, outside the ranges for which csrc/exact_div.h is proven, so the scene must
    fall back to true division; the quad at z = -2 is hit and frames match bit for bit."""
    from helpers import HandScene, END
    quad = np.array([[[-3, -3, -2], [3, -3, -2], [3, 3, -2]], [[-3, -3, -2], [3, 3, -2], [-3, 3, -2]]], np.float32)
    normals = np.tile(np.array([0, 0, 1], np.float32), (6, 1))
    hm = np.full((8, 1, 2), END, dtype=np.float32)
    hand = HandScene(quad.reshape(-1, 3), normals, [[-3.00001, -3.00001, -2.00001]], [[3.00001, 3.00001, 1e-30]], hm, [[0, 2]], 0)
    env = pkg.scenes.environment_hdr_sky(64)
    params = default_params(pkg, 48, 32, zoom=0.0, material=6)
    scene = pkg.Scene(hand.desc, env, device=0)
    want = check_against_oracle(oracle_mod, scene, hand.desc, env, params, 48, 32, 1, "box plane at 1e-30")
    empty, _ = oracle_mod.render(kat_far_scene().desc, env, params, 48, 32, 1)
    assert not np.array_equal(want, empty)      # the quad is visible
    scene.close()


def kat_far_scene():
    import test_oracle_kat as kat
    return kat.far_away_triangle()


def test_depth_capped_tree_with_large_leaves(pkg, gpu, oracle_mod, tmp_path):
    """BVH_MAX_DEPTH (bvh.cpp:60-79) forces leaves far above 10 triangles; the shader only ever
    tests the first 10 of a leaf (raytracer.es.fs:412-417).  Built in a child process because
    the build parameters are read once per process."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
from __graft_entry__ import load_package
import helpers, oracle
pkg = load_package()
world = pkg.World(helpers.small_trisrc())
assert world.info.max_level <= 4, world.info.max_level
a = world.arrays()
counts = a["group_objects"].reshape(-1, 2)[:, 1]
assert counts.max() > 10
desc = world.flatten()
env = pkg.scenes.environment_hdr_sky(64)
params = world.frame_params(96, 72, material=6)
want, cpu = oracle.render(desc, env, params, 96, 72, 1)
scene = pkg.Scene(desc, env, device=0)
for kernel in (0, 1, 2, 3, 4):
    scene.set_kernel(kernel)
    got, gpu = scene.render_counters(params, 96, 72, 1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), kernel
    assert gpu == cpu, (kernel, gpu, cpu)
params.max_leaf_tests = 1000   # lifting the cap changes the image: the cap was active
uncapped, _ = oracle.render(desc, env, params, 96, 72, 1)
assert not np.array_equal(uncapped, want)
print("ok")
''' % (os.path.dirname(os.path.dirname(GOLDEN)), os.path.dirname(GOLDEN))
    env = dict(os.environ, BVH_MAX_DEPTH="4")
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("which,material", [(1, 0), (1, 6), (2, 0), (2, 6), (3, 0), (5, 0), (5, 6), (4, 0)])
def test_which_views(pkg, gpu, oracle_mod, bunny, env_sky, which, material):
    """The shader's `which` views (raytracer.es.fs:27): 1 = environment through textureGrad (mip
    pyramid, trilinear, 4x anisotropy) with the ray differentials, 2 = differential of the lookup
    carried through the bounces, 3 = the pixel's own differentials, 5 = 5x5 supersampled
    reference image; other values render like 0.  Bit-identical to the oracle, counters equal."""
    world, desc, scene = bunny
    params = world.frame_params(112, 80, material=material)
    params.which = which
    got = check_against_oracle(oracle_mod, scene, desc, env_sky, params, 112, 80, 1, f"which {which} material {material}")
    if which == 4:
        params.which = 0
        base, _ = oracle_mod.render(desc, env_sky, params, 112, 80, 1)
        assert np.array_equal(got, base)
    if which == 3:
        assert np.all(got[..., 2] == 0) and np.all(got[..., 3] == 1)
    with pytest.raises(pkg._native.ShrayError):
        if which in (3, 5):
            scene.render(params, 112, 80, 2)      # per-pixel views: spp must be 1
        else:
            raise pkg._native.ShrayError(-1, "n/a")


def test_filtered_environment_view_on_a_small_frame(pkg, gpu, oracle_mod, env_sky):
    """which == 1 with footprints of several texels (a 40 x 24 frame over a 512-wide environment,
    and the sharp `grid` environment): exercises trilinear blending, the 2/3/4-probe anisotropic
    paths and the poles' non-finite derivatives; bit-identical to the oracle."""
    import test_oracle_kat as kat
    for env in (env_sky, pkg.scenes.environment_grid(256)):
        for hand, zoom in ((kat.far_away_triangle(), 4.0), (single_leaf_scene(kat.mirror_quad()), 3.0)):
            params = default_params(pkg, 40, 24, zoom=zoom, material=0)
            params.which = 1
            scene = pkg.Scene(hand.desc, env, device=0)
            check_against_oracle(oracle_mod, scene, hand.desc, env, params, 40, 24, 1, "filtered environment view")
            # straight up: camera rotated so that the centre of an odd frame looks along +y (D.x = D.z = 0)
            params = default_params(pkg, 33, 33, zoom=zoom, material=0)
            params.which = 1
            params.camera_normal_matrix[:] = [1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1]
            check_against_oracle(oracle_mod, scene, hand.desc, env, params, 33, 33, 1, "filtered view at the pole")
            scene.close()


def test_config3_64spp_plaster_band(pkg, gpu, oracle_mod, bunny, env_sky):
    """BASELINE config 3 at full size (1920x1080, 64 spp, glazed plaster: shadow rays, up to six
    traversals per sample): kernel 0 == kernel 1 bit for bit, and a band of rows equals the oracle."""
    world, desc, scene = bunny
    W, H, spp = 1920, 1080, 64
    params = world.frame_params(W, H, material=6)
    scene.set_kernel(0)
    a = scene.render(params, W, H, spp)
    rows = (300, 304)
    want, _ = oracle_mod.render(desc, env_sky, params, W, H, spp, rows=rows)
    assert np.array_equal(a[rows[0]:rows[1]].view(np.uint32), want[rows[0]:rows[1]].view(np.uint32))
    scene.set_kernel(1)
    b = scene.render(params, W, H, 4)      # the literal kernel at 4 spp (it is ~1.4x slower)
    scene.set_kernel(0)
    assert np.array_equal(b, scene.render(params, W, H, 4))
    assert np.all(a[..., 3] == 1.0) and not np.isnan(a).any()


def test_config5_4k_16spp_eight_way_tile_split(pkg, gpu, bunny):
    """BASELINE config 5's frame (3840x2160, 16 spp) split into the eight interleaved tile sets the
    eight GPUs would own: the packed tile buffers reassemble to the single-GPU frame bit for bit."""
    import torch
    from shader_ray_amd.multigpu import max_tiles_per_rank
    world, desc, scene = bunny
    W, H, spp, tile, ranks = 3840, 2160, 16, 32, 8
    params = world.frame_params(W, H, material=0)
    full = torch.empty(H * W * 4, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_into(params, W, H, spp, full.data_ptr(), stream, None)
    per_rank = max_tiles_per_rank(W, H, tile, tile, ranks)
    gathered = torch.zeros(ranks, per_rank * tile * tile * 4, dtype=torch.float32, device="cuda:0")
    N = pkg._native
    for r in range(ranks):
        scene.render_into(params, W, H, spp, gathered[r].data_ptr(), stream, N.TileSet(tile, tile, ranks, r))
    frame = torch.empty(H * W * 4, dtype=torch.float32, device="cuda:0")
    N.check(N.load_hip().shray_assemble_tiles_device(
        C.c_void_p(gathered.data_ptr()), ranks, 1, 4, gathered.stride(0) * 4, gathered.stride(0) * 4, W, H, tile, tile,
        C.c_void_p(frame.data_ptr()), C.c_void_p(stream)))
    torch.cuda.synchronize()
    assert torch.equal(frame, full)


@pytest.mark.parametrize("material", [0, 6])
def test_sample_lanes_every_group_size(pkg, gpu, oracle_mod, bunny, env_sky, material):
    """Multi-sample frames run a pixel's samples in G = 2, 4, ... 32 neighbouring lanes of a wave (in rounds beyond that) and add them in sample
    order (uniform_driver.h).  Every group size, sample counts that do not fill the last round, a frame whose edges cut
    through patches, whole frames and tile sets: bit-identical to the oracle (raytracer.es.fs:622-640)."""
    world, desc, scene = bunny
    scene.set_kernel(0)
    W, H = 45, 27
    params = world.frame_params(W, H, material=material)
    for spp in (2, 3, 4, 5, 8, 13, 16, 32, 64, 100):
        want, _ = oracle_mod.render(desc, env_sky, params, W, H, spp)
        got = scene.render(params, W, H, spp)
        differing = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        assert differing == 0, f"{spp} spp, material {material}: {differing} floats differ from the oracle"
    # tile sets at 4 and 64 spp: the owned tiles of three uneven sets reassemble to the whole frame
    import torch
    W, H = 96, 64
    params = world.frame_params(W, H, material=material)
    for spp in (4, 64):
        whole = scene.render(params, W, H, spp)
        frame = np.zeros_like(whole)
        for phase, count in ((0, 1), (1, 2), (3, 1)):
            tiles = pkg._native.TileSet(32, 32, 4, phase, count)
            nbytes = pkg.tracer.tile_buffer_bytes(W, H, tiles)
            buf = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
            scene.render_into(params, W, H, spp, buf.data_ptr(), torch.cuda.current_stream().cuda_stream, tiles)
            torch.cuda.synchronize()
            packed = buf.cpu().numpy().reshape(-1, 32, 32, 4)
            owned = [t for t in range((W // 32) * (H // 32)) if phase <= t % 4 < phase + count]
            assert len(owned) == packed.shape[0]
            for k, t in enumerate(owned):
                ty, tx = divmod(t, W // 32)
                frame[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32] = packed[k]
        assert np.array_equal(frame.view(np.uint32), whole.view(np.uint32)), f"tile sets at {spp} spp"
