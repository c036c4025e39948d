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


def own_tolerance(case):
    """the shader's debug views (100 x differences of lookup coordinates; the 5 x 5 supersampled view) and the 8-bit
    background, whose texels this driver filters with 8-bit weights: held to a tolerance of their own, not classified"""
    return case["params"].which != 0 or case["bad_fraction"] >= 1.0


def well_floor(name):
    """The share of a frame that must be WELL-CONDITIONED by the classifier's a-priori criteria (off every change of path, off every
    shared edge, its 3x3 neighbourhood moving by less than 1e-4 under the perturbations) -- every such pixel is thereby within 1e-4
    of the reference's.  Set a little below what the cases reach (round 5; VERDICT round 4 found 0.2 / 0.35 far below them): the
    bunny-class frames 0.84-0.97, the hand-built scenes 0.88-1, the small lobed mesh 0.75-0.96 (64 x 48 frames are one third
    silhouette and shadow edge), its random views 0.50-0.86, the env-only frame 0.66 (acos x the sky's gradient), the 1M-facet
    sphere 0.23-0.36: its surface is displaced vertex by vertex, so that even its matte frames (no mirror bounce) change by more
    than 1e-4 under a turn of 3e-6 rad over most of the frame -- what IS held on that scene is the measured agreement, below."""
    for prefix, floor in (("bunny", 0.8), ("million", 0.2), ("lobed_random", 0.45), ("kat_env_only", 0.6), ("kat", 0.85), ("lobed", 0.7), ("quads", 0.85)):
        if name.startswith(prefix):
            return floor
    return 0.7


def within_floor(name):
    """The share of a frame's pixels that must BE within 1e-4 of the reference shaders' frame, whatever the classifier says of
    them -- the direct statement of north_star's bar; a little below what the cases reach."""
    for prefix, floor in (("million_gold", 0.94), ("million_plaster", 0.985), ("million_matte", 0.98), ("bunny_plaster_sky", 0.97),
                          ("kat_env_only", 0.97), ("lobed_random", 0.99), ("bunny", 0.9999), ("kat", 1.0), ("lobed", 0.998), ("quads", 0.999)):
        if name.startswith(prefix):
            return floor
    return 0.99


def test_oracle_matches_the_reference_shaders(all_cases, oracle_mod):
    """Every plain-view case: a pixel of the oracle's frame further than 1e-4 from the reference shaders' must be
    EXPLAINED (tests/pixel_classifier.py): within one pixel of a change of path -- which bounces hit, which hits were lit,
    the iteration-cap marker -- or where the oracle's own pixel moves as far when its inputs are perturbed by what the
    driver's acos / atan / arithmetic are measured to be off by.  No unexplained pixel is allowed, in any asserted case.
    Round 5 adds what bounds the classifier itself: the share of pixels that ARE within 1e-4 (94 % on the 1M-facet sphere in gold,
    98.3-98.8 % in plaster and matte, 97.6-100 % everywhere else), the share the classifier calls well-conditioned (well_floor),
    and how many bad pixels pass through its one unbounded escape -- a neighbourhood that moves by >= 1e-3 under the
    perturbations --: at most one pixel in a thousand of a fixture frame, reported with the worst of them."""
    import pixel_classifier
    report = []
    asserted = 0
    for name, case in all_cases.items():
        want = load_fixture(name)["frame"]
        if own_tolerance(case):             # the shader's debug views, the 8-bit background: their own tolerances, below
            continue
        verdict = pixel_classifier.classify(oracle_mod, case, want)
        well = verdict["pixels"] - verdict["ill_conditioned_pixels"]
        report.append(f"{name}: {verdict['bad']} of {verdict['pixels']} pixels outside 1e-4 ({100.0 - 100.0 * verdict['bad'] / verdict['pixels']:.2f} % within) -- "
                      f"{verdict['discontinuity']} at a change of path, "
                      f"{verdict['sensitivity']} within the perturbation budget ({verdict['chaotic_admitted']} of them through the chaotic escape, worst "
                      f"{verdict['chaotic_admitted_worst']:.1e}), {verdict['edge']} at a shared edge, {verdict['unexplained']} UNEXPLAINED "
                      f"(worst {verdict['worst_unexplained']:.1e}); {well} well-conditioned pixels ({100.0 * well / verdict['pixels']:.1f} %), all within 1e-4"
                      + (f"   [recorded only: {case['why']}]" if case["recorded"] else ""))
        if case["recorded"]:
            continue
        asserted += 1
        assert verdict["unexplained"] == 0, report[-1]
        assert well >= well_floor(name) * verdict["pixels"], report[-1]
        assert verdict["pixels"] - verdict["bad"] >= within_floor(name) * verdict["pixels"], report[-1]
        assert verdict["chaotic_admitted"] <= max(1, verdict["pixels"] // 1000), report[-1]
        assert np.all(want[..., 3] == 1.0) and np.all(verdict["frame"][..., 3] == 1.0)
    print("\n".join(report))
    assert asserted >= 40
    # the debug views (100 x differences of lookup coordinates: fs:135-149, :642-673) and the 5 x 5 supersampled view keep
    # a tolerance of their own
    for name, case in all_cases.items():
        if not own_tolerance(case) or case["recorded"]:
            continue
        want = load_fixture(name)["frame"]
        got = oracle_frame(oracle_mod, case)
        bad, pixels, _, _ = agreement(got, want)
        assert agreement(got, want, case["max_rel"])[3] <= case["flips"] and bad <= case["bad_fraction"] * pixels, name
    # the cases no texture filter and no transcendental of the compiler's touches are exact to float rounding
    for name in ("kat_mirror_quad", "kat_plaster_quad", "kat_iteration_cap_401", "kat_iteration_cap_400", "kat_eleven_triangle_leaf"):
        assert agreement(oracle_frame(oracle_mod, all_cases[name]), load_fixture(name)["frame"])[3] < 1e-6, name
    # and the traversal / shading cases (constant environment) hold north_star's 1e-4 on all but a handful of pixels
    for name in ("lobed_gold_constant", "lobed_plaster_constant", "lobed_plaster_constant_rotated", "quads_obj_chrome_constant",
                 "bunny_gold_constant_256", "bunny_plaster_constant_256", "bunny_matte_constant_256"):
        bad, pixels, _, worst = agreement(oracle_frame(oracle_mod, all_cases[name]), load_fixture(name)["frame"])
        assert bad <= 3, (name, bad, worst)


def test_the_classifier_does_not_explain_everything(all_cases, oracle_mod):
    """The classifier must be able to fail: the recorded cases, where the driver does something the oracle does not state
    (a smeared anisotropic lookup at zero derivatives, operations on NaN), leave thousands / all of their bad pixels
    unexplained; and an oracle frame with a wrong material is not explained away either."""
    import pixel_classifier
    smeared = pixel_classifier.classify(oracle_mod, all_cases["lobed_gold_sky_anisotropic4"], load_fixture("lobed_gold_sky_anisotropic4")["frame"])
    assert smeared["unexplained"] > 1000, smeared["unexplained"]
    wrong = dict(all_cases["bunny_plaster_constant_256"])
    wrong["params"] = wrong["params"].copy()
    wrong["params"].diffuse_color[0] *= 1.001          # a tenth of a per cent more diffuse red
    off = pixel_classifier.classify(oracle_mod, wrong, load_fixture("bunny_plaster_constant_256")["frame"])
    assert off["unexplained"] > 1000, off["unexplained"]


def test_driver_function_deviations_are_the_committed_ones():
    """tests/golden/measure_glsl_functions.py's measurements of the driver behind the fixtures: its acos is the one
    built-in that is far off (1.6e-4 rad; the oracle's explicit sequence: 2.8e-7) -- what the sky cases' pixels outside 1e-4
    are (|grad env| x that deviation) --; normalize, division and the square roots are correctly rounded."""
    import pixel_classifier
    d = pixel_classifier.driver_functions()
    assert 1e-4 < d["acos_abs_dev_driver"] < 2e-4 and d["acos_abs_dev_oracle"] < 4e-7
    assert d["atan_abs_dev_driver"] < 5e-6 and d["atan_abs_dev_oracle"] < 4e-7
    assert d["pow5_rel_dev_driver"] < 1e-5 and d["pow5_rel_dev_oracle"] < 4e-7
    assert max(d["normalize_abs_dev_driver"], d["division_rel_dev_driver"], d["sqrt_rel_dev_driver"], d["inversesqrt_rel_dev_driver"]) < 1.2e-7


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


def test_pow_of_a_negative_base_is_what_the_reference_frame_shows(all_cases, oracle_mod):
    """fs:481's pow(x, 5.0) with x < 0 is undefined in GLSL; the oracle and the kernels return NaN (a black pixel), because the
    driver behind the fixtures does.  Pinned here against the REFERENCE'S frame, not against the oracle: normals
    1 + 2^-10 long make the base negative within 2.5 degrees of normal incidence -- the reference's frame has a disc of exactly
    black pixels in the middle of the mirror, and the oracle's frame is black on the same pixels (the disc's rim, where the base
    is within 1e-4 of zero, may fall either way)."""
    case = all_cases["kat_long_normal_mirror"]
    want = load_fixture("kat_long_normal_mirror")["frame"]
    got = oracle_frame(oracle_mod, case)
    black = lambda frame: np.all(frame[..., :3] == 0.0, axis=-1)   # noqa: E731
    h, w = case["height"], case["width"]
    assert black(want)[h // 2, w // 2] and black(want)[h // 2 - 1, w // 2 - 1] and 4 <= black(want).sum() <= 40, black(want).sum()
    assert not black(want)[0, 0] and not black(want)[h // 2, 2]
    assert (black(want) ^ black(got)).sum() <= 2, (black(want).sum(), black(got).sum())
    both = ~black(want) & ~black(got)
    assert np.all(np.abs(got - want)[both][:, :3] <= 1e-4 * np.abs(want[both][:, :3]) + 1e-6)


def test_a_nan_in_the_tail_of_a_path_swallows_the_pixel(all_cases, oracle_mod):
    """The classifier's newest arm (`swallowed_by_nan`, tests/pixel_classifier.py) pinned by a frame of the reference's own shaders
    (VERDICT round 5, item 5).  In the matte 1M-facet sphere one triangle in forty has normals twice as long: fs:481's pow base is
    negative within 60 degrees of normal incidence on it, its result NaN, and the pixel black (tonemap_and_gamma's max(0, c - .004)
    of a NaN) WHATEVER the weight of the bounce that hit it -- here the first hit's Fresnel term, a few per cent.  Where the primary
    ray hits a trap both frames are black.  Where a LATER bounce does, and those bounces are chaotic (sub-pixel facets), the
    reference's arithmetic and the oracle's land on different facets: the reference's frame has exactly black pixels where the
    oracle's is lit.  Some of them are explained by nothing else -- their value does not move under the perturbations (the tail
    weighs nothing unless it is NaN), their first hit and its shadow ray are their neighbours', no triangle test of the oracle's
    path is near its margin -- and `swallowed_by_nan` is what explains them; none is left unexplained."""
    import pixel_classifier
    name = "million_matte_traps_constant_384"
    case, want = all_cases[name], load_fixture(name)["frame"]
    verdict = pixel_classifier.classify(oracle_mod, case, want)
    got = verdict["frame"]
    black = lambda frame: np.all(frame[..., :3] == 0.0, axis=-1)   # noqa: E731
    # the traps a primary ray hits: black in both frames, thousands of pixels (2.5 % of the sphere's)
    assert (black(want) & black(got)).sum() > 1000
    # the reference's frame black where the oracle's is lit, and the other way round: the chaotic tails
    assert 3 <= (black(want) & ~black(got)).sum() <= 40 and (black(got) & ~black(want)).sum() <= 40
    assert verdict["unexplained"] == 0, verdict["worst_unexplained"]
    assert verdict["swallowed_by_nan"] >= 1, {k: v for k, v in verdict.items() if isinstance(v, (int, float))}
    # the arm claims nothing that is not exactly black in the reference and lit in the oracle
    assert verdict["swallowed_by_nan"] <= (black(want) & ~black(got)).sum()


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


@pytest.mark.parametrize("which", ["configs[1]: gold under the HDR sky", "glazed plaster under the HDR sky", "gold, constant environment",
                                   "configs[3]'s 1M-triangle OBJ, gold, constant environment",
                                   "configs[3]'s 1M-triangle OBJ, matte (zero specular, diffuse white), constant environment"])
def test_live_reference_shader_at_full_size(pkg, oracle_mod, which):
    """BASELINE's 1920x1080 frames, where the reference's shaders can run (this container: 1.7 s a frame on llvmpipe): the
    same classifier, no unexplained pixel; the figures are printed (pytest -s) and quoted in DESIGN.md section 2."""
    import helpers
    import pixel_classifier
    if not oracle_mod.reference_shader_available():
        pytest.skip("the reference tree / Mesa software driver / harness is not on this box")
    W, H = 1920, 1080
    million = which.startswith("configs[3]")
    world = pkg.World(helpers.million_obj() if million else helpers.bunny_trisrc())
    env = pkg.scenes.environment_hdr_sky(2048) if "sky" in which else pkg.scenes.environment_constant((0.5, 0.25, 2.0))
    matte = "matte" in which
    params = world.frame_params(W, H, material=6 if ("plaster" in which or matte) else 0)
    if matte:    # the first hit's diffuse term alone (glsl_cases.py: the deep tree on well-conditioned pixels)
        params.specular_color[:] = (0.0, 0.0, 0.0)
    case = dict(scene=(world.flatten(), world), env=np.ascontiguousarray(env, dtype=np.float32), params=params, width=W, height=H,
                env_storage=0, path_bits=glsl_cases.MATTE_PATH_BITS if matte else 0xffffffff)
    want, _ = oracle_mod.render_reference_shader(case["scene"][0], case["env"], params, W, H, 0, 1.0)
    verdict = pixel_classifier.classify(oracle_mod, case, want)
    got = verdict["frame"]
    rel = (np.abs(got - want)[..., :3] / np.maximum(np.abs(want[..., :3]), 1e-2)).max(axis=-1)
    print(f"\n{which}, {W}x{H}: {verdict['bad']} of {verdict['pixels']} pixels outside 1e-4 ({100.0 * verdict['bad'] / verdict['pixels']:.3f} %), "
          f"{int((rel > 1e-3).sum())} beyond 1e-3, {int((rel > 1e-2).sum())} beyond 1e-2; {verdict['discontinuity']} at a change of path, "
          f"{verdict['sensitivity']} within the perturbation budget, {verdict['edge']} at a shared edge (of {verdict['edge_candidates']} such pixels), "
          f"{verdict['swallowed_by_nan']} black in the reference among chaotic later bounces, {verdict['unexplained']} unexplained; {verdict['chaotic_admitted']} through the chaotic escape (worst {verdict['chaotic_admitted_worst']:.1e}); "
          f"{verdict['pixels'] - verdict['ill_conditioned_pixels']} well-conditioned pixels")
    assert verdict["unexplained"] == 0 and verdict["swallowed_by_nan"] <= 2
    # what bounds the classifier (round 5): the pixels that ARE within 1e-4, and the share admitted by the unbounded escape alone
    assert verdict["pixels"] - verdict["bad"] >= (0.94 if million and not matte else 0.98) * verdict["pixels"]
    assert verdict["chaotic_admitted"] <= (0.02 if million and not matte else 0.001) * verdict["pixels"]
    if million:
        # the iteration cap's red marker (fs:436-438): the reference's pixels and the oracle's are not all the same ones at
        # this size -- a capped ray sits at the end of 400 visits, any of whose box tests a last bit can turn -- but every
        # one of them, on either side, is an ill-conditioned pixel by the classifier, and their numbers are close
        from test_oracle_kat import filmic64
        red = np.asarray(filmic64([1.0, 0.0, 0.0]), dtype=np.float32)
        marked = lambda frame: np.all(np.abs(frame[..., :3] - red) < 1e-5, axis=-1)   # noqa: E731
        theirs, ours = marked(want), marked(got)
        print(f"marker pixels: the reference's {int(theirs.sum())}, the oracle's {int(ours.sum())}, in common {int((theirs & ours).sum())}")
        assert theirs.sum() >= 1 and ours.sum() >= 1 and abs(int(theirs.sum()) - int(ours.sum())) <= 0.25 * theirs.sum()
        assert not (verdict["unexplained_mask"] & (theirs ^ ours)).any()
