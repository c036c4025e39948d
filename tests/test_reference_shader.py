"""The oracle against the REFERENCE ITSELF: frames rendered by the reference's own shaders.

tests/golden/glsl_reference/*.npz were written by tests/golden/make_glsl_reference.py in the container that holds the
reference tree: raytracer.vs + raytracer.es.fs, unmodified, compiled as "#version 140" by Mesa's llvmpipe (CPU) behind a
harness that restates ray.cpp's GL calls (oracle/glsl_ref/glsl_ref.cpp).  Here the inputs of every case are rebuilt
(tests/glsl_cases.py), checked against the fixture's hash, and rendered by the CPU oracle: north_star's bar -- every
pixel within 1e-4 relative -- must hold against the reference's frame.  The GLSL compiler's arithmetic is not the
oracle's bit for bit (llvmpipe's pow / atan / acos / rsqrt are its own), so this is a tolerance test, unlike the
bit-exact kernel-vs-oracle tests; what it pins is that the oracle IS the shader.

Two cases are recorded rather than asserted, because they run through a texture filter OpenGL does not define:
textureGrad with zero derivatives under 4x anisotropy (fs:153 with ray.cpp:506) and the which == 1 view.  The same
lookups with the anisotropy off agree to 1e-4 everywhere, which is what the oracle's rule (level-0 bilinear) states."""
import os

import numpy as np
import pytest

import glsl_cases


def load_fixture(name):
    path = os.path.join(glsl_cases.FIXTURES, name + ".npz")
    assert os.path.exists(path), f"{path} is missing: run tests/golden/make_glsl_reference.py where the reference tree is"
    return np.load(path)


def oracle_frame(oracle_mod, case):
    try:
        oracle_mod.set_env_storage(case["env_storage"])
        frame, _ = oracle_mod.render(case["scene"][0], case["env"], case["params"], case["width"], case["height"], 1)
    finally:
        oracle_mod.set_env_storage(0)
    return frame


@pytest.fixture(scope="module")
def all_cases(pkg):
    return glsl_cases.cases(pkg)


def test_every_case_has_a_fixture_of_these_inputs(all_cases):
    assert len(all_cases) >= 45
    for name, case in all_cases.items():
        fx = load_fixture(name)
        assert str(fx["input_hash"]) == glsl_cases.input_hash(case), f"{name}: the fixture was made from other inputs"
        assert fx["frame"].shape == (case["height"], case["width"], 4) and "Mesa" in str(fx["gl"])
    stray = {f[:-4] for f in os.listdir(glsl_cases.FIXTURES) if f.endswith(".npz")} - set(all_cases)
    assert not stray, stray


def agreement(got, want, max_rel=None):
    """(pixels outside 1e-4, pixels, median relative difference, largest -- or, with max_rel, how many pixels exceed it)"""
    bad = glsl_cases.out_of_tolerance(got, want)
    rel = (np.abs(got - want)[..., :3] / np.maximum(np.abs(want[..., :3]), 1e-2)).max(axis=-1)
    return int(bad.sum()), bad.size, float(np.median(rel)), (float(rel.max()) if max_rel is None else int((rel > max_rel).sum()))


def test_oracle_matches_the_reference_shaders(all_cases, oracle_mod):
    report = []
    asserted = 0
    for name, case in all_cases.items():
        want = load_fixture(name)["frame"]
        got = oracle_frame(oracle_mod, case)
        bad, pixels, median, worst = agreement(got, want)
        report.append(f"{name}: {bad} of {pixels} pixels outside 1e-4; median {median:.1e} max {worst:.1e}"
                      + (f"   [recorded only: {case['why']}]" if case["recorded"] else ""))
        if case["recorded"]:
            continue
        asserted += 1
        assert agreement(got, want, case["max_rel"])[3] <= case["flips"], report[-1]
        assert bad <= case["bad_fraction"] * pixels, report[-1]
        assert np.all(want[..., 3] == 1.0) and np.all(got[..., 3] == 1.0)
    print("\n".join(report))
    assert asserted >= 39
    # the cases no texture filter and no transcendental of the compiler's touches are exact to float rounding
    for name in ("kat_mirror_quad", "kat_plaster_quad", "kat_iteration_cap_401", "kat_iteration_cap_400", "kat_eleven_triangle_leaf"):
        assert agreement(oracle_frame(oracle_mod, all_cases[name]), load_fixture(name)["frame"])[3] < 1e-6, name
    # and the traversal / shading cases (constant environment) hold north_star's 1e-4 on all but a handful of pixels
    for name in ("lobed_gold_constant", "lobed_plaster_constant", "lobed_plaster_constant_rotated", "quads_obj_chrome_constant"):
        bad, pixels, _, worst = agreement(oracle_frame(oracle_mod, all_cases[name]), load_fixture(name)["frame"])
        assert bad <= 3 and worst < 2.5e-4, (name, bad, worst)


def test_unsized_background_is_eight_bit_on_this_driver(all_cases, oracle_mod):
    """ray.cpp:508 uploads the float background with the unsized GL_RGB: Mesa keeps 8 bits, clamped to [0, 1] -- the sun of
    the HDR sky is gone.  The oracle told to store 8 bits agrees with that frame to the driver's 8-bit filter weights; the
    oracle with float storage is ten times further off where the sun is reflected."""
    case = all_cases["lobed_gold_sky_unsized_rgb"]
    want = load_fixture("lobed_gold_sky_unsized_rgb")["frame"]
    eight_bit = agreement(oracle_frame(oracle_mod, case), want)
    floats = agreement(oracle_frame(oracle_mod, dict(case, env_storage=0)), want)
    assert eight_bit[3] < 2e-2 and floats[3] > 5 * eight_bit[3] and floats[2] > 1.5 * eight_bit[2], (eight_bit, floats)


def test_the_marker_and_cap_cases_say_what_the_kats_say(all_cases):
    """The analytic expectations of tests/test_oracle_kat.py, read off the REFERENCE's frames: 401 chained nodes give the
    tone-mapped red marker (raytracer.es.fs:436-438, :566-568), 400 do not; the 11th triangle of a leaf is never tested."""
    from test_oracle_kat import filmic64
    capped = load_fixture("kat_iteration_cap_401")["frame"]
    assert np.allclose(capped[..., :3], filmic64([1.0, 0.0, 0.0]), rtol=1e-5, atol=1e-6)
    free = load_fixture("kat_iteration_cap_400")["frame"]
    assert np.allclose(free[..., :3], filmic64([0.5, 0.25, 2.0]), rtol=1e-5)
    assert not np.allclose(capped, free)


def test_the_capped_pixel_of_the_million_triangle_scene_is_the_same_pixel(all_cases, oracle_mod):
    """BASELINE configs[3]'s deep tree runs a few rays into the 400-iteration cap: in the reference's frame and in the
    oracle's the red marker sits on the same pixels, and the oracle's bad-hit counter counts exactly them."""
    from test_oracle_kat import filmic64
    case = all_cases["million_gold_constant_192"]
    want = load_fixture("million_gold_constant_192")["frame"]
    try:
        got, counters = oracle_mod.render(case["scene"][0], case["env"], case["params"], case["width"], case["height"], 1)
    finally:
        oracle_mod.set_env_storage(0)
    red = np.asarray(filmic64([1.0, 0.0, 0.0]), dtype=np.float32)
    marked = lambda frame: np.all(np.abs(frame[..., :3] - red) < 1e-5, axis=-1)   # noqa: E731
    assert marked(want).sum() >= 1 and np.array_equal(marked(want), marked(got)) and counters["bad_hits"] == marked(got).sum()


def test_live_reference_shader_reproduces_a_fixture(all_cases, oracle_mod):
    """Where the reference tree is present, the harness is run again: the fixture is what it renders."""
    if not oracle_mod.reference_shader_available():
        pytest.skip("the reference tree / Mesa software driver / harness is not on this box")
    case = all_cases["lobed_plaster_constant_rotated"]
    frame, log = oracle_mod.render_reference_shader(case["scene"][0], case["env"], case["params"], case["width"], case["height"],
                                                    case["background_mode"], case["anisotropy"])
    assert "llvmpipe" in log
    # (llvmpipe compiles the shaders for the host's vector width: allow the last bits to differ between machines)
    assert agreement(frame, load_fixture("lobed_plaster_constant_rotated")["frame"])[3] < 1e-5
