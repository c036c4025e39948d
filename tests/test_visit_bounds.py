"""The node visit's decision from bounds (csrc/wave_traversal.h: visit_decision, round 4), restated in numpy float32 and
checked against the exact quotients on millions of random rays and boxes -- the argument of the header comment, tested
without a GPU:

  * the one-multiplication quotient RN(a * RN(1 / b)) is within 3 * 2^-24 of RN(a / b) and has its sign;
  * whenever r0~ (1 + 2^-20) < min(r1~, hit.t) the exact test `!(r0 >= r1) && r0 < hit.t` (raytracer.es.fs:400) passes, whenever
    r0~ (1 - 2^-20) >= min(r1~, hit.t) it fails -- so a decided visit is decided as the reference decides it; only about one
    visit in 10^5 is left undecided (those evaluate the exact quotients in the kernel);
  * the bounds a leaf parks contain its exact range, and a candidate distance the bounds let through but the exact range
    rejects is always one that near_range_end flags (those are re-checked against the exact range in the kernel).
Every operation below is one correctly rounded binary32 operation, as in the kernels (-ffp-contract=off)."""
import numpy as np

F = np.float32
UP, DOWN = F(1.0) + F(2.0 ** -20), F(1.0) - F(2.0 ** -20)
CHECK_UP, CHECK_DOWN = F(1.0) + F(2.0 ** -19), F(1.0) - F(2.0 ** -19)
RANGE_MAX = F(100000000.0)


def rays_and_boxes(n, seed):
    rng = np.random.default_rng(seed)
    P = rng.uniform(-4, 4, (n, 3)).astype(F)
    centre = rng.uniform(-1, 1, (n, 3))
    half = rng.choice([0.004, 0.02, 0.1, 0.5, 1.5], (n, 1)) * rng.uniform(0.3, 1.0, (n, 3))
    # half of the rays aim at (or just past) their box, the others anywhere
    aimed = centre + rng.uniform(-1.3, 1.3, (n, 3)) * half - P
    D = np.where(rng.random((n, 1)) < 0.5, aimed, rng.normal(size=(n, 3)))
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    # some components tiny (a ray nearly parallel to an axis), still inside exact_div.h's divisor range
    tiny = rng.random((n, 3)) < 0.05
    D = np.where(tiny, D * rng.choice([1e-3, 1e-6, 1e-9], (n, 3)), D).astype(F)
    D[D == 0] = F(1e-9)
    lo, hi = (centre - half).astype(F), (centre + half).astype(F)
    return P, D, lo, hi


def exact_range(P, D, lo, hi):
    near = np.where(D >= 0, lo, hi) - P
    far = np.where(D >= 0, hi, lo) - P
    r0 = np.maximum(F(0), (near / D).max(axis=1)).astype(F)
    r1 = np.minimum(RANGE_MAX, (far / D).min(axis=1)).astype(F)
    return r0, r1, near, far


def approximate_range(D, near, far):
    Y = (F(1.0) / D).astype(F)            # RN(1 / D): what reciprocal_in_range returns on its whole domain
    r0 = np.maximum(F(0), (near * Y).astype(F).max(axis=1)).astype(F)
    r1 = np.minimum(RANGE_MAX, (far * Y).astype(F).min(axis=1)).astype(F)
    return r0, r1, Y


def test_one_multiplication_quotients_are_within_three_half_ulps():
    P, D, lo, hi = rays_and_boxes(1_000_000, 1)
    _, _, near, far = exact_range(P, D, lo, hi)
    _, _, Y = approximate_range(D, near, far)
    for a in (near, far):
        q = (a / D).astype(F)
        approx = (a * Y).astype(F)
        rel = np.abs(approx.astype(np.float64) - q.astype(np.float64)) / np.maximum(np.abs(q.astype(np.float64)), 1e-300)
        assert rel[q != 0].max() <= 3.0 * 2.0 ** -24 * 1.0001
        assert np.array_equal(np.sign(approx), np.sign(q))


def test_decided_visits_are_decided_as_the_exact_quotients_decide():
    decided_hit = decided_miss = undecided = 0
    for seed in range(4):
        P, D, lo, hi = rays_and_boxes(1_000_000, 10 + seed)
        r0, r1, near, far = exact_range(P, D, lo, hi)
        a0, a1, _ = approximate_range(D, near, far)
        rng = np.random.default_rng(100 + seed)
        # the closest hit so far: far away, anywhere, or within a few ulps of the box's entry distance
        T = np.where(rng.random(len(r0)) < 0.4, F(10000000.0), rng.uniform(0, 12, len(r0))).astype(F)
        close = rng.random(len(r0)) < 0.3
        T = np.where(close, r0 * (F(1) + rng.integers(-6, 7, len(r0)).astype(F) * F(2.0 ** -23)), T).astype(F)
        exact = ~(r0 >= r1) & (r0 < T)
        below = np.minimum(a1, T)
        enter = (a0 * UP).astype(F) < below
        miss = (a0 * DOWN).astype(F) >= below
        assert not (enter & miss).any()
        assert exact[enter].all(), "a visit the bounds call entered is not entered by the exact quotients"
        assert not exact[miss].any(), "a visit the bounds call missed is entered by the exact quotients"
        decided_hit += int(enter.sum())
        decided_miss += int(miss.sum())
        undecided += int((~enter & ~miss & ~close).sum())
        # what a leaf parks: lo0 <= r0, hi1 >= r1 wherever the box is entered
        lo0, hi1 = (a0 * DOWN).astype(F), (a1 * UP).astype(F)
        assert (lo0[exact] <= r0[exact]).all() and (hi1[exact] >= r1[exact]).all()
        # a candidate distance within a few ulps of an end: if the bounds let it through and the exact range does not,
        # near_range_end must flag it
        for end in (r0, r1):
            d = (end * (F(1) + rng.integers(-40, 41, len(end)).astype(F) * F(2.0 ** -23))).astype(F)
            passes_bounds = ~((d < lo0) | (d > hi1))
            rejected_exactly = (d < r0) | (d > r1)
            flagged = ((d * CHECK_DOWN).astype(F) < lo0) | ((d * CHECK_UP).astype(F) > hi1)
            assert flagged[exact & passes_bounds & rejected_exactly].all()
    assert decided_hit > 500_000 and decided_miss > 1_000_000
    assert undecided < 1e-4 * 4_000_000, undecided      # (random T: the T ~ r0 cases above are undecided by design)
