"""Analytic known-answer tests for the CPU oracle (oracle/shader_oracle.cpp).

The reference has no tests or golden vectors for its per-pixel path (it exists only as
GLSL), so these tests are what keeps an oracle bug from being silently mirrored by the
kernel: each expected value is derived here, independently, from the shader text
(file:line cited) -- closed forms in numpy float64, compared at float32 precision."""
import ctypes as C

import numpy as np
import pytest

import helpers
from helpers import END, HandScene, default_params, single_leaf_scene


def filmic64(c):   # raytracer.es.fs:527-531
    x = np.maximum(0.0, np.asarray(c, dtype=np.float64) - 0.004)
    return (x * (6.2 * x + 0.5)) / (x * (6.2 * x + 1.7) + 0.06)


def env_bilinear64(env, d):
    """raytracer.es.fs:130 + GL bilinear/REPEAT, in float64."""
    d = np.asarray(d, np.float64)
    h, w, _ = env.shape
    s = 1.0 + np.arctan2(-d[2], d[0]) / (2 * np.pi)
    t = 1.0 - np.arccos(np.clip(d[1], -1, 1)) / np.pi
    u, v = s * w - 0.5, t * h - 0.5
    i0, j0 = int(np.floor(u)), int(np.floor(v))
    a, b = u - i0, v - j0
    e = env.astype(np.float64)
    px = lambda i, j: e[j % h, i % w]
    return (1 - a) * (1 - b) * px(i0, j0) + a * (1 - b) * px(i0 + 1, j0) + (1 - a) * b * px(i0, j0 + 1) + a * b * px(i0 + 1, j0 + 1)


def pixel_dir64(p, px, py, W, H):
    """raytracer.vs:39-49 at the pixel centre, camera looking down -z."""
    u, v = (px + 0.5) / W, (py + 0.5) / H
    d = np.array([p.image_plane_width * (u - 0.5), p.image_plane_width * (v - 0.5) * p.aspect, -1.0])
    return d / np.linalg.norm(d)


def test_filmic_spot_values(oracle_mod):
    for c in (0.0, 0.004, 0.01, 0.18, 0.5, 1.0, 4.0, 60.0):
        assert oracle_mod.filmic(c) == pytest.approx(float(filmic64(c)), rel=2e-6, abs=1e-9)
    assert oracle_mod.filmic(0.0) == 0.0 and oracle_mod.filmic(-3.0) == 0.0


def test_half_rounding(oracle_mod):
    h = oracle_mod.half
    assert h(1.0) == 1.0 and h(-0.5) == -0.5 and h(0.0) == 0.0
    assert h(1.0 + 2.0 ** -11) == 1.0                 # tie -> even
    assert h(1.0 + 3 * 2.0 ** -11) == 1.0 + 2.0 ** -9  # tie -> even (up)
    assert h(1.0 + 2.0 ** -11 + 2.0 ** -20) == 1.0 + 2.0 ** -10
    assert h(0.333333343267) == np.float32(np.float16(np.float32(0.333333343267)))
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-1, 1, 2000), rng.uniform(-1e-4, 1e-4, 500), rng.uniform(-7e-8, 7e-8, 200)]).astype(np.float32)
    for x in xs:
        assert h(float(x)) == float(np.float32(np.float16(x))), x


def test_schlick_endpoints(oracle_mod):
    c = (1.0, 0.71, 0.29)
    v = (0.0, 0.0, -1.0)
    # dot(v, r) = 1 -> pow(1, 5) = 1 -> white; dot = -1 -> pow(0, 5) = 0 -> cspec (raytracer.es.fs:479-482)
    assert np.array_equal(oracle_mod.schlick(c, v, v), np.ones(3, np.float32))
    assert np.array_equal(oracle_mod.schlick(c, v, (0.0, 0.0, 1.0)), np.array(c, np.float32))


def test_primary_ray_matches_vertex_shader(pkg, oracle_mod):
    p = default_params(pkg, 1920, 1080, zoom=2.5)
    for (px, py) in ((0, 0), (959, 540), (1919, 1079), (100, 900)):
        o, d = oracle_mod.primary_ray(p, (px + 0.5) / 1920, (py + 0.5) / 1080)
        assert np.array_equal(o, np.array([0, 0, 2.5], np.float32))
        assert np.allclose(d, pixel_dir64(p, px, py, 1920, 1080), rtol=0, atol=2e-7)
    # v = 0 is the BOTTOM of the image (raytracer.vs:43-44, :56)
    assert oracle_mod.primary_ray(p, 0.5, 0.0)[1][1] < 0 < oracle_mod.primary_ray(p, 0.5, 1.0)[1][1]


def far_away_triangle():
    return single_leaf_scene([[[100, 100, 100], [101, 100, 100], [100, 101, 100]]])


def test_env_only_pixels_are_tonemapped_bilinear_samples(pkg, oracle_mod):
    scene = far_away_triangle()   # nothing in view
    env = pkg.scenes.environment_hdr_sky(64)
    W, H = 48, 32
    p = default_params(pkg, W, H)
    img, c = oracle_mod.render(scene.desc, env, p, W, H)
    assert c["env_lookups"] == W * H and c["shaded_hits"] == 0 and c["traversals"] == W * H
    assert c["node_visits"] == W * H and c["triangle_tests"] == 0     # one box test, missed
    for (px, py) in ((0, 0), (47, 31), (10, 20), (24, 16), (33, 3)):
        want = filmic64(env_bilinear64(env, pixel_dir64(p, px, py, W, H)))
        assert np.allclose(img[py, px, :3], want, rtol=2e-5, atol=1e-6), (px, py)
    assert np.all(img[..., 3] == 1.0)


def test_env_wraps_in_s_and_t(pkg, oracle_mod):
    """A 2x2 environment: looking straight up / down blends the rows across the t wrap
    (GL REPEAT is the default for both axes; ray.cpp:499-510 sets no wrap mode)."""
    scene = far_away_triangle()
    env = np.array([[[1, 0, 0], [0, 1, 0]], [[0, 0, 1], [1, 1, 1]]], np.float32)
    p = default_params(pkg, 4, 4)
    p.tonemap = 0
    # aim the camera straight up: rotate eye -z onto +y via camera_normal_matrix
    rot = np.array([1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1], np.float32)   # column-major: world = (x, -z, y)
    p.camera_normal_matrix[:] = rot.tolist()
    img, _ = oracle_mod.render(scene.desc, env, p, 4, 4)
    for py in range(4):
        for px in range(4):
            e = pixel_dir64(p, px, py, 4, 4)
            d = np.array([e[0], -e[2], e[1]])
            assert np.allclose(img[py, px, :3], env_bilinear64(env, d), rtol=1e-4, atol=1e-5)


def mirror_quad(z=0.0, half=10.0):
    a, b, c, d = [-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]
    return [[a, b, c], [a, c, d]]


def test_single_mirror_bounce_closed_form(pkg, oracle_mod):
    """Camera on +z looking at a big mirror in the z = 0 plane: every pixel hits once at
    t = zoom / -D.z, reflects to (D.x, D.y, -D.z), and returns F * env(R) with Schlick's F
    (raytracer.es.fs:484-522, :552-582); the flat normal is exact in fp16."""
    scene = single_leaf_scene(mirror_quad())
    env = pkg.scenes.environment_hdr_sky(64)
    W, H = 40, 24
    zoom = 3.0
    p = default_params(pkg, W, H, zoom=zoom, material=0)
    img, c = oracle_mod.render(scene.desc, env, p, W, H)
    assert c["shaded_hits"] == W * H and c["env_lookups"] == W * H
    assert c["traversals"] == 2 * W * H            # second bounce leaves the scene
    assert c["triangle_tests"] == 2 * W * H        # both triangles once; the bounced ray starts past the box
    assert c["node_visits"] == 2 * W * H
    spec = np.array([1, .71, .29])
    for (px, py) in ((0, 0), (39, 23), (20, 12), (7, 19)):
        D = pixel_dir64(p, px, py, W, H)
        R = D * np.array([1, 1, -1])
        F = spec + (1 - spec) * (np.dot(D, R) * .5 + .5) ** 5
        want = filmic64(F * env_bilinear64(env, R))
        assert np.allclose(img[py, px, :3], want, rtol=3e-5, atol=1e-6), (px, py)


def test_diffuse_shading_and_shadow(pkg, oracle_mod):
    """Glazed plaster on the same mirror quad: accumulated = diffuse * max(0, n.l) (unshadowed,
    raytracer.es.fs:447-472) plus F * env(R); then a blocker between surface and light
    removes the diffuse term."""
    env = pkg.scenes.environment_constant((0.2, 0.3, 0.4))
    W, H = 16, 16
    p = default_params(pkg, W, H, zoom=3.0, material=6)
    light = np.array(p.light_dir[:], np.float64)
    spec = np.array([.05, .05, .05])

    def expected(px, py, lit):
        D = pixel_dir64(p, px, py, W, H)
        R = D * np.array([1, 1, -1])
        F = spec + (1 - spec) * (np.dot(D, R) * .5 + .5) ** 5
        diffuse = max(0.0, light[2]) if lit else 0.0
        return filmic64(diffuse + F * np.array([0.2, 0.3, 0.4]))

    scene = single_leaf_scene(mirror_quad())
    img, c = oracle_mod.render(scene.desc, env, p, W, H)
    assert c["traversals"] == 3 * W * H   # closest hit, shadow ray, second bounce (which misses)
    for (px, py) in ((0, 0), (8, 8), (15, 3)):
        assert np.allclose(img[py, px, :3], expected(px, py, True), rtol=3e-5, atol=1e-6)

    # blocker: a huge quad high above, facing down, hides the light from everything
    blocker = [[[-500, -500, 50], [500, 500, 50], [500, -500, 50]], [[-500, -500, 50], [-500, 500, 50], [500, 500, 50]]]
    scene2 = single_leaf_scene(mirror_quad() + blocker)
    p2 = default_params(pkg, W, H, zoom=3.0, material=6)
    p2.bounce_count = 1
    img2, _ = oracle_mod.render(scene2.desc, env, p2, W, H)
    for (px, py) in ((0, 0), (8, 8), (15, 3)):
        assert np.allclose(img2[py, px, :3], expected(px, py, False), rtol=3e-5, atol=1e-6)


def test_leaf_tests_only_its_first_ten_triangles(pkg, oracle_mod):
    """max_leaf_tests = 10 (raytracer.es.fs:382, :412-417): the 11th triangle of a leaf is
    never tested, even when it is the nearest."""
    quads = []
    for k in range(10):
        z = -float(k)
        quads.append([[-5, -5, z], [5, -5, z], [0, 5, z]])
    quads.append([[-5, -5, 1.0], [5, -5, 1.0], [0, 5, 1.0]])   # nearest to the camera, index 10
    scene = single_leaf_scene(quads)
    env = pkg.scenes.environment_constant((1, 1, 1))
    p = default_params(pkg, 8, 8, zoom=4.0)
    p.bounce_count = 1
    p.tonemap = 0
    _, c = oracle_mod.render(scene.desc, env, p, 8, 8)
    assert c["triangle_tests"] == 10 * c["leaf_visits"] == 10 * 64
    # with the cap lifted the 11th is tested too
    p.max_leaf_tests = 11
    _, c = oracle_mod.render(scene.desc, env, p, 8, 8)
    assert c["triangle_tests"] == 11 * 64


def chain_scene(n):
    """n leaf nodes threaded one after another, each an empty leaf whose box covers the view."""
    hm = np.zeros((8, n, 2), np.float32)
    for g in range(n):
        hm[:, g, :] = (g + 1) if g + 1 < n else END
    lo = np.tile([-50, -50, -50], (n, 1))
    hi = np.tile([50, 50, 50], (n, 1))
    objs = np.zeros((n, 2), np.float32)
    tri = [[100, 100, 100], [101, 100, 100], [100, 101, 100]]
    return HandScene(tri, [[0, 0, 1]] * 3, lo, hi, hm, objs, 0)


def thread_tree(neg, pos, axis, root):
    """The eight (hit, miss) tables of a binary tree given by its branches' children (leaves: -1) and split axes, as
    world.cpp:231-288 threads them: for direction code c a branch's hit link is its near child (the NEGATIVE child when bit
    `axis` of c is set, i.e. D[axis] > 0), its miss link the next subtree on the stack; a leaf's two links are that next subtree."""
    n = len(neg)
    hm = np.zeros((8, n, 2), np.float32)
    for code in range(8):
        def walk(g, after):
            if neg[g] < 0:
                hm[code, g] = (after, after)
                return
            near, far = (neg[g], pos[g]) if (code >> axis[g]) & 1 else (pos[g], neg[g])
            hm[code, g] = (near, after)
            walk(near, far)
            walk(far, after)
        walk(root, END)
    return hm


def comb_scene(branches, hit_triangle=True):
    """A BINARY tree a ray visits node by node: `branches` branch nodes in a row (node 2i), each with a leaf (node 2i + 1) as
    its negative child and the next branch -- the last one: a final leaf -- as its positive child, every box the same box around
    the view, so that every ray visits all 2 * branches + 1 nodes.  The last leaf holds one big triangle in front of the camera (the
    others are empty): a ray that is not capped is shaded, one that needs more than max_bvh_iterations visits is the red
    marker.  Unlike chain_scene this IS a canonical threaded tree: kernel 0 takes it."""
    n = 2 * branches + 1
    neg, pos, axis = [-1] * n, [-1] * n, [2] * n
    for i in range(branches):
        neg[2 * i] = 2 * i + 1
        pos[2 * i] = 2 * i + 2
    hm = thread_tree(neg, pos, axis, 0)
    lo = np.tile([-50, -50, -50], (n, 1))
    hi = np.tile([50, 50, 50], (n, 1))
    objs = np.zeros((n, 2), np.float32)
    tri = [[-40, -40, -1.0], [40, -40, -1.0], [0, 40, -1.0]] if hit_triangle else [[100, 100, 100], [101, 100, 100], [100, 101, 100]]
    objs[n - 1] = (0, 1)
    return HandScene(tri, [[0, 0, 1]] * 3, lo, hi, hm, objs, 0)


def test_comb_tree_visits_and_cap(pkg, oracle_mod):
    """comb_scene(b): 2 b + 1 visits per traversal; the cap strikes exactly when a traversal needs more than
    max_bvh_iterations visits (fs:426-438): cap = visits ends normally, cap = visits - 1 is the marker."""
    env = pkg.scenes.environment_constant((0.5, 0.5, 0.5))
    hand = comb_scene(20)
    p = default_params(pkg, 4, 4)
    p.bounce_count = 1
    img, c = oracle_mod.render(hand.desc, env, p, 4, 4)
    assert c["node_visits"] == 16 * 41 and c["bad_hits"] == 0 and c["shaded_hits"] == 16 and c["leaf_visits"] == 16 * 21
    p.max_bvh_iterations = 41
    same, c = oracle_mod.render(hand.desc, env, p, 4, 4)
    assert np.array_equal(same, img) and c["bad_hits"] == 0
    p.max_bvh_iterations = 40
    capped, c = oracle_mod.render(hand.desc, env, p, 4, 4)
    assert c["bad_hits"] == 16 and c["node_visits"] == 16 * 40
    assert np.allclose(capped[..., :3], filmic64([1.0, 0.0, 0.0]), rtol=1e-6)


def test_iteration_cap_gives_the_red_marker(pkg, oracle_mod):
    """400 iterations without reaching a terminator -> (1,0,0), unmodulated, then tone-mapped
    (raytracer.es.fs:436-438, :497-501, :566-568); exactly 400 nodes still terminate."""
    env = pkg.scenes.environment_constant((0.5, 0.5, 0.5))
    p = default_params(pkg, 4, 4)
    img, c = oracle_mod.render(chain_scene(401).desc, env, p, 4, 4)
    red = filmic64([1.0, 0.0, 0.0])
    assert np.allclose(img[..., :3], red, rtol=1e-6) and c["bad_hits"] == 16 and c["node_visits"] == 16 * 400
    assert c["env_lookups"] == 0
    img, c = oracle_mod.render(chain_scene(400).desc, env, p, 4, 4)
    assert c["bad_hits"] == 0 and c["node_visits"] == 16 * 400
    assert np.allclose(img[..., :3], filmic64([0.5, 0.5, 0.5]), rtol=1e-6)


def test_sphere_hit_distance_and_normal(pkg, oracle_mod, tmp_path):
    """Tessellated unit sphere, white constant environment, gold, one bounce: every pixel
    that hits returns F * 1; with smooth normals F depends on the normal only, so the
    central pixel (normal ~ +z, head-on) must give F = cspec exactly-ish."""
    pos, tri = pkg.scenes.lobed_sphere_mesh(48, 96, bumpiness=0.0, ears=False)
    path = str(tmp_path / "sphere.trisrc")
    pkg.scenes.write_trisrc(path, pos, tri)
    world = pkg.World(path)
    desc = world.flatten()
    env = pkg.scenes.environment_constant((1, 1, 1))
    W = H = 33
    p = world.frame_params(W, H, material=0)
    p.bounce_count = 1
    p.tonemap = 0
    img, c = oracle_mod.render(desc, env, p, W, H)
    centre = img[16, 16, :3]
    assert np.allclose(centre, [1, .71, .29], atol=2e-3)          # head-on: pow(~0, 5) ~ 0
    corner = img[0, 0, :3]
    assert np.all(corner > 0.99999)                               # missed: environment only (weights sum to 1 - ulp)
    # silhouette of a unit sphere seen from `zoom` under fov 40: hit iff the ray passes within r = 1
    zoom = world.default_view().zoom
    hits = 0
    for py in range(H):
        for px in range(W):
            D = pixel_dir64(p, px, py, W, H)
            closest = np.linalg.norm(np.cross(np.array([0, 0, zoom]) - np.array(world.info.scene_center[:]), D))
            if closest < 0.995:
                assert img[py, px, 2] < 0.99, (px, py)       # hit: Fresnel-weighted, blue < 1
                hits += 1
            elif closest > 1.0 + 1e-3:
                assert np.all(img[py, px, :3] > 0.99999), (px, py)
    assert hits > 200 and c["shaded_hits"] >= hits


def test_spp_average_of_constant_is_constant(pkg, oracle_mod):
    scene = far_away_triangle()
    env = pkg.scenes.environment_constant((0.25, 0.5, 2.0))
    p = default_params(pkg, 6, 5)
    one, _ = oracle_mod.render(scene.desc, env, p, 6, 5, spp=1)
    many, c = oracle_mod.render(scene.desc, env, p, 6, 5, spp=7)
    assert c["samples"] == 6 * 5 * 7 and c["env_lookups"] == 6 * 5 * 7
    assert np.allclose(one, many, rtol=1e-6)


def test_golden_frames_still_reproduce(pkg, oracle_mod):
    """The committed oracle frames (tests/golden/lobed_528.oracle.npz) pin today's oracle."""
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    frames = np.load(os.path.join(golden, "lobed_528.oracle.npz"))
    world = pkg.World(os.path.join(golden, "lobed_528.trisrc"))
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(128)
    for name, material in (("gold", 0), ("plaster", 6)):
        img, counters = oracle_mod.render(desc, env, world.frame_params(64, 64, material=material), 64, 64)
        helpers.assert_images_match(img, frames[name], name)
        assert [counters[k] for k in sorted(counters)] == frames[name + "_counters"].tolist()


def test_specified_transcendentals_are_accurate(oracle_mod):
    """sr_atan2 / sr_acos / sr_pow5 replace the GLSL built-ins atan, acos, pow(x, 5.0)
    (raytracer.es.fs:130, :481) with explicit fp32 sequences; they must still BE those
    functions: within 4 ulp of the float64 value (GLSL itself allows far more)."""
    rng = np.random.default_rng(5)
    for _ in range(4000):
        y, x = (float(np.float32(v)) for v in rng.uniform(-1, 1, 2))
        want = np.arctan2(np.float64(y), np.float64(x))
        ulp = float(np.spacing(np.float32(abs(want)))) or 1e-45
        assert abs(oracle_mod.atan2(y, x) - want) <= 4 * ulp, (y, x)
    for c in np.concatenate([rng.uniform(-1, 1, 3000), [1.0, -1.0, 0.0, 0.99999994, -0.99999994]]):
        c = float(np.float32(c))
        want = np.arccos(np.float64(c))
        assert abs(oracle_mod.acos(c) - want) <= 4 * float(np.spacing(np.float32(max(want, 1e-3)))), c
    assert oracle_mod.acos(1.0) == 0.0 and oracle_mod.atan2(0.0, 1.0) == 0.0
    assert oracle_mod.atan2(0.0, -1.0) == float(np.float32(np.pi)) and oracle_mod.atan2(1.0, 0.0) == float(np.float32(np.pi / 2))
    for b in (0.0, 1.0, 0.5, 0.999, 1e-3, 0.37):
        assert oracle_mod.pow5(b) == pytest.approx(b ** 5, rel=4e-7, abs=1e-45)
    # a negative base: undefined in GLSL, NaN on the implementations that run the reference (exp2(5 log2 x)), NaN here
    assert np.isnan(oracle_mod.pow5(-1e-8)) and np.isnan(oracle_mod.pow5(-1.0)) and oracle_mod.pow5(-0.0) == 0.0


# ---- the shader's `which` views (raytracer.es.fs:27, :144-154, :642-673)

def coords64(d):
    d = np.asarray(d, np.float64)
    return np.array([1.0 + np.arctan2(-d[2], d[0]) / (2 * np.pi), 1.0 - np.arccos(np.clip(d[1], -1, 1)) / np.pi])


def differentials64(p, d):
    right, up = np.array(p.right[:], np.float64), np.array(p.up[:], np.float64)
    dd = d @ d
    return (dd * right - (d @ right) * d) / dd ** 1.5, (dd * up - (d @ up) * d) / dd ** 1.5


def test_view_3_draws_the_pixel_differentials(pkg, oracle_mod):
    """which == 3 (fs:642-650): |coords(d + dDdy/2) - coords(d - dDdy/2)| * 100, alpha 1, no
    trace and no tone map."""
    W, H = 64, 36
    p = default_params(pkg, W, H)
    p.which = 3
    img, c = oracle_mod.render(far_away_triangle().desc, pkg.scenes.environment_constant((1, 1, 1)), p, W, H)
    assert c["traversals"] == 0 and c["env_lookups"] == 0
    for (px, py) in ((0, 0), (63, 35), (20, 30), (40, 5)):
        d = pixel_dir64(p, px, py, W, H)
        _, ddy = differentials64(p, d)
        want = np.abs(coords64(d + ddy / 2) - coords64(d - ddy / 2)) * 100
        assert np.allclose(img[py, px, :2], want, rtol=2e-3, atol=1e-6), (px, py)
        assert img[py, px, 2] == 0.0 and img[py, px, 3] == 1.0
    # one pixel step in y moves the lookup by about that much: compare with neighbouring pixel centres
    d0, d1 = pixel_dir64(p, 30, 10, W, H), pixel_dir64(p, 30, 11, W, H)
    assert np.allclose(img[10, 30, :2], np.abs(coords64(d1) - coords64(d0)) * 100, rtol=0.05)


def test_view_2_draws_the_environment_derivative(pkg, oracle_mod):
    """which == 2 (fs:135-149): the analytic d(s, t)/dy of the lookup, times 100, goes through
    trace() and the tone map in place of a texel; checked against central differences."""
    W, H = 64, 36
    p = default_params(pkg, W, H)
    p.which = 2
    p.tonemap = 0
    img, c = oracle_mod.render(far_away_triangle().desc, pkg.scenes.environment_constant((1, 1, 1)), p, W, H)
    assert c["env_lookups"] == W * H
    for (px, py) in ((5, 5), (50, 30), (32, 18)):
        d = pixel_dir64(p, px, py, W, H)
        _, ddy = differentials64(p, d)
        eps = 1e-3
        fd = (coords64(d + eps * ddy) - coords64(d - eps * ddy)) / (2 * eps)
        assert np.allclose(img[py, px, :2], np.abs(fd) * 100, rtol=2e-3, atol=1e-6), (px, py)
    # after a mirror bounce the differentials are transferred and reflected as the shader writes it
    scene = single_leaf_scene(mirror_quad())
    p2 = default_params(pkg, 16, 16, zoom=3.0)
    p2.which = 2
    img2, c2 = oracle_mod.render(scene.desc, pkg.scenes.environment_constant((1, 1, 1)), p2, 16, 16)
    assert c2["shaded_hits"] == 256 and np.isfinite(img2).all() and (img2[..., :2] > 0).any()


def test_view_5_is_the_mean_of_25_rays(pkg, oracle_mod):
    """which == 5 (fs:654-673): 5 x 5 rays around the interpolated varying direction, averaged
    in linear radiance, then tone mapped once."""
    W, H = 40, 24
    env = pkg.scenes.environment_hdr_sky(64)
    p = default_params(pkg, W, H)
    p.which = 5
    img, c = oracle_mod.render(far_away_triangle().desc, env, p, W, H)
    assert c["env_lookups"] == 25 * W * H and c["samples"] == W * H
    right, up = np.array(p.right[:], np.float64), np.array(p.up[:], np.float64)
    hx, hy = p.image_plane_width * 0.5, p.image_plane_width * 0.5 * p.aspect
    corner_len = np.sqrt(hx * hx + hy * hy + 1)
    for (px, py) in ((0, 0), (39, 23), (17, 9)):
        u, v = (px + 0.5) / W, (py + 0.5) / H
        direction = np.array([p.image_plane_width * (u - 0.5), p.image_plane_width * (v - 0.5) * p.aspect, -1.0]) / corner_len
        acc = np.zeros(3)
        for i in range(5):
            for j in range(5):
                d = direction + (i / 5 - 0.5) * 0.2 * right + (j / 5 - 0.5) * 0.2 * up
                acc += env_bilinear64(env, d / np.linalg.norm(d))
        assert np.allclose(img[py, px, :3], filmic64(acc / 25), rtol=3e-5, atol=1e-6), (px, py)
    # a constant environment makes it indistinguishable from the normal view
    cenv = pkg.scenes.environment_constant((0.3, 0.6, 0.9))
    a, _ = oracle_mod.render(far_away_triangle().desc, cenv, p, W, H)
    p.which = 0
    b, _ = oracle_mod.render(far_away_triangle().desc, cenv, p, W, H)
    assert np.allclose(a, b, rtol=1e-6)


def test_other_which_values_render_like_0(pkg, oracle_mod):
    scene = single_leaf_scene(mirror_quad())
    env = pkg.scenes.environment_hdr_sky(64)
    p = default_params(pkg, 12, 12, zoom=3.0)
    base, _ = oracle_mod.render(scene.desc, env, p, 12, 12)
    for w in (4, 6, -1):   # fs:627-628 (empty branch), and the else-branch of fs:150-154
        p.which = w
        img, _ = oracle_mod.render(scene.desc, env, p, 12, 12)
        assert np.array_equal(img, base)


# ---- which == 1: textureGrad on the mip-mapped environment (fs:144-146, ray.cpp:503-509)

def mips64(img):
    levels = [img.astype(np.float64)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        a = levels[-1]
        h, w = max(1, a.shape[0] // 2), max(1, a.shape[1] // 2)
        j0 = np.minimum(2 * np.arange(h), a.shape[0] - 1); j1 = np.minimum(2 * np.arange(h) + 1, a.shape[0] - 1)
        i0 = np.minimum(2 * np.arange(w), a.shape[1] - 1); i1 = np.minimum(2 * np.arange(w) + 1, a.shape[1] - 1)
        levels.append((a[j0][:, i0] + a[j0][:, i1] + a[j1][:, i0] + a[j1][:, i1]) * 0.25)
    return levels


def bilinear64(level, s, t):
    h, w, _ = level.shape
    u, v = s * w - 0.5, t * h - 0.5
    i0, j0 = int(np.floor(u)), int(np.floor(v))
    a, b = u - i0, v - j0
    px = lambda i, j: level[j % h, i % w]
    return (1 - a) * (1 - b) * px(i0, j0) + a * (1 - b) * px(i0 + 1, j0) + (1 - a) * b * px(i0, j0 + 1) + a * b * px(i0 + 1, j0 + 1)


def texture_grad64(img, s, t, dudx, dvdx, dudy, dvdy, max_aniso=4):
    """GL 3.1 section 3.8.9 + EXT_texture_filter_anisotropic reference formulas, in float64."""
    levels = mips64(img)
    h, w, _ = img.shape
    px, py = np.hypot(dudx * w, dvdx * h), np.hypot(dudy * w, dvdy * h)
    pmax, pmin = max(px, py), min(px, py)
    if not np.isfinite(pmax):
        return bilinear64(levels[-1], s, t)
    if pmax == 0:
        return bilinear64(levels[0], s, t)
    n = max_aniso if pmin == 0 else min(np.ceil(pmax / pmin), max_aniso)
    lam = np.log2(pmax / n)
    mu, mv = (dudx, dvdx) if px >= py else (dudy, dvdy)
    acc = np.zeros(3)
    for i in range(1, int(n) + 1):
        o = i / (n + 1) - 0.5
        ss, tt = s + o * mu, t + o * mv
        if lam <= 0:
            acc += bilinear64(levels[0], ss, tt)
        elif lam >= len(levels) - 1:
            acc += bilinear64(levels[-1], ss, tt)
        else:
            d = int(np.floor(lam)); f = lam - d
            acc += (1 - f) * bilinear64(levels[d], ss, tt) + f * bilinear64(levels[d + 1], ss, tt)
    return acc / n


def test_log2_sequence(oracle_mod):
    rng = np.random.default_rng(2)
    xs = np.concatenate([10.0 ** rng.uniform(-6, 6, 3000), [1.0, 2.0, 0.5, 1024.0, 1.9999999, 1.0000001]])
    for x in xs:
        x = float(np.float32(x))
        assert abs(oracle_mod.log2(x) - np.log2(np.float64(x))) < 4e-6, x
    assert oracle_mod.log2(1.0) == 0.0 and oracle_mod.log2(8.0) == 3.0 and oracle_mod.log2(0.25) == -2.0


def test_texture_grad_cases(oracle_mod):
    rng = np.random.default_rng(4)
    img = rng.uniform(0, 3, (16, 32, 3)).astype(np.float32)
    W, H = 32, 16
    cases = {
        "zero gradients = level 0, LINEAR": (0.37, 0.61, 0, 0, 0, 0),
        "sub-texel footprint magnifies": (0.37, 0.61, 0.5 / W, 0, 0, 0.25 / H),
        "isotropic 2-texel footprint = level 1": (0.21, 0.43, 2.0 / W, 0, 0, 2.0 / H),
        "lambda 1.5 blends levels 1 and 2": (0.71, 0.13, 2 ** 1.5 / W, 0, 0, 2 ** 1.5 / H),
        "8:1 anisotropy: 4 probes at level 1": (0.55, 0.52, 8.0 / W, 0, 0, 1.0 / H),
        "anisotropic along y, oblique": (0.30, 0.80, 0.3 / W, 0.2 / H, 2.0 / W, 5.0 / H),
        "3:1 anisotropy: 3 probes": (0.10, 0.90, 6.0 / W, 0, 0, 2.0 / H),
        "huge footprint = 1x1 level": (0.4, 0.4, 1000.0, 0, 0, 1000.0),
        "degenerate (one zero axis)": (0.4, 0.4, 4.0 / W, 0, 0, 0),
    }
    for what, (s, t, dudx, dvdx, dudy, dvdy) in cases.items():
        got = oracle_mod.texture_grad(img, s, t, dudx, dvdx, dudy, dvdy)
        want = texture_grad64(img, s, t, dudx, dvdx, dudy, dvdy)
        assert np.allclose(got, want, rtol=3e-5, atol=1e-5), (what, got, want)
    # non-finite derivatives (straight up / down): the coarsest level = the mean of the image
    got = oracle_mod.texture_grad(img, 0.5, 0.5, np.inf, 0, 0, 1.0)
    assert np.allclose(got, img.reshape(-1, 3).mean(0), rtol=1e-5)
    got = oracle_mod.texture_grad(img, 0.5, 0.5, np.nan, 0, 0, 1.0)
    assert np.allclose(got, img.reshape(-1, 3).mean(0), rtol=1e-5)
    flat = np.full((8, 8, 3), 0.75, np.float32)
    assert np.allclose(oracle_mod.texture_grad(flat, 0.3, 0.3, 0.4, 0.1, -0.2, 0.3), 0.75, rtol=1e-6)


def test_view_1_filters_the_environment_with_the_ray_differentials(pkg, oracle_mod):
    """which == 1 end to end on environment-only pixels: derivatives of fs:135-139 from the
    differentials of fs:621-625 into the textureGrad rule above; one pixel step is about one
    texel at these sizes, so the view must stay close to the unfiltered one."""
    W, H = 48, 27
    env = pkg.scenes.environment_hdr_sky(128)
    p = default_params(pkg, W, H)
    p.which = 1
    img, c = oracle_mod.render(far_away_triangle().desc, env, p, W, H)
    assert c["env_lookups"] == W * H
    for (px, py) in ((3, 4), (40, 20), (24, 13), (10, 25)):
        d = pixel_dir64(p, px, py, W, H)
        ddx, ddy = differentials64(p, d)
        rxz = 2 * np.pi * (d[0] ** 2 + d[2] ** 2)
        dudx, dudy = (d[0] * ddx[2] - d[2] * ddx[0]) / rxz, (d[0] * ddy[2] - d[2] * ddy[0]) / rxz
        ryy = np.pi * np.sqrt(1 - d[1] ** 2)
        dvdx, dvdy = ddx[1] / ryy, ddy[1] / ryy
        st = coords64(d)
        want = filmic64(texture_grad64(env, st[0], st[1], dudx, dvdx, dudy, dvdy))
        assert np.allclose(img[py, px, :3], want, rtol=2e-4, atol=2e-5), (px, py)
    # a bigger frame of the same view = sub-texel footprints = level 0 (a few probes when the
    # footprint is anisotropic): practically the unfiltered view
    p2 = default_params(pkg, 960, 540)
    p2.which = 1
    a, _ = oracle_mod.render(far_away_triangle().desc, env, p2, 960, 540, rows=(100, 102))
    p2.which = 0
    b, _ = oracle_mod.render(far_away_triangle().desc, env, p2, 960, 540, rows=(100, 102))
    assert np.allclose(a[100:102], b[100:102], rtol=2e-3, atol=1e-4)


# ---- NaN candidates in a leaf (raytracer.es.fs:312-346): a triangle so large that its determinant overflows
# fails none of the shader's comparisons, so the sequential loop accepts it (hit.t = NaN) and then accepts
# whatever candidate comes next, whatever its distance
HUGE_TRIANGLE = [[-1e30, -1e30, 0.0], [1e30, -1e30, 0.0], [0.0, 1e30, 0.0]]   # det = inf, d = u = v = NaN for every ray


def finite_triangle(z):
    return [[-2.0, -2.0, z], [2.0, -2.0, z], [0.0, 2.0, z]]


def nan_leaf_scene(order, half=0.25):
    """One leaf whose box is the cube [-half, half]^3 (hand-set: it does not enclose the huge triangle) holding,
    in the order given, 'nan' = the huge triangle, or a z value = a finite triangle in the plane of that z."""
    tris = [HUGE_TRIANGLE if k == "nan" else finite_triangle(float(k)) for k in order]
    tp = np.asarray(tris, dtype=np.float32)
    normals = np.tile(np.asarray([0.0, 0.0, 1.0], np.float32), (len(tris) * 3, 1))
    hm = np.full((8, 1, 2), END, dtype=np.float32)
    return HandScene(tp.reshape(-1, 3), normals, [[-half] * 3], [[half] * 3], hm, [[0, len(tris)]], 0)


NAN_ORDERS = [("nan", 0.1), (0.1, "nan"), ("nan", 0.1, -0.1), (0.1, "nan", -0.1), (0.1, -0.1, "nan"), (-0.1, "nan", 0.1),
              ("nan", -0.1, 0.1), (-0.1, 0.1, "nan"), (0.2, 0.1, 0.0, "nan", -0.1, -0.2, 0.15), (0.0, "nan", "nan", 0.1, 0.2),
              (0.05, 0.1, 0.15, "nan"), ("nan",)]


def test_nan_candidate_is_accepted_and_resets_the_comparison(pkg, oracle_mod):
    """After a NaN candidate every comparison with hit.t is false: the next candidate in order is accepted whatever
    its distance, and the loop goes on from there.  So the leaf's outcome is that of the triangles AFTER the last
    huge one, or a NaN hit (tone-mapped to 0) if none follows."""
    env = pkg.scenes.environment_constant((0.5, 0.25, 2.0))
    p = default_params(pkg, 16, 16, zoom=3.0)
    p.bounce_count = 1    # one traversal per pixel: a reflected ray would meet the huge triangle again
    for order in NAN_ORDERS:
        got, _ = oracle_mod.render(nan_leaf_scene(order).desc, env, p, 16, 16, 1)
        last = max(k for k, what in enumerate(order) if what == "nan")
        tail = order[last + 1:]
        centre = got[8, 8]
        if not tail:
            assert centre.tolist() == [0.0, 0.0, 0.0, 1.0], (order, centre)
        else:
            want, _ = oracle_mod.render(nan_leaf_scene(tail).desc, env, p, 16, 16, 1)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), order
            assert not np.isnan(got).any()


def test_unorm8_environment_storage(pkg, oracle_mod):
    """The reference uploads its environment with an unsized GL_RGB format (ray.cpp:508): on most drivers 8 bits,
    clamped to [0, 1].  With that storage an env-only pixel is the tone-mapped bilinear sample of the QUANTISED image:
    values above 1 are gone, everything sits on 1/255 steps before filtering."""
    env = pkg.scenes.environment_hdr_sky(64)
    assert env.max() > 1.5                      # the sun is above 1: float storage keeps it
    hand = far_away_triangle()
    p = default_params(pkg, 24, 16)
    try:
        oracle_mod.set_env_storage(1)
        got, _ = oracle_mod.render(hand.desc, env, p, 24, 16, 1)
    finally:
        oracle_mod.set_env_storage(0)
    plain, _ = oracle_mod.render(hand.desc, env, p, 24, 16, 1)
    q = (np.floor(np.clip(env, 0.0, 1.0).astype(np.float32) * np.float32(255.0) + np.float32(0.5)) / np.float32(255.0)).astype(np.float32)
    assert len(np.unique(q)) <= 256 and q.max() <= 1.0
    want, _ = oracle_mod.render(hand.desc, q, p, 24, 16, 1)       # float storage of the pre-quantised image
    assert np.array_equal(got, want) and not np.array_equal(got, plain)
    for (px, py) in ((3, 2), (12, 8), (20, 13)):
        d = pixel_dir64(p, px, py, 24, 16)
        ref = filmic64(env_bilinear64(q.astype(np.float64), d))
        assert np.allclose(got[py, px, :3], ref, rtol=2e-5, atol=2e-6), (px, py)
