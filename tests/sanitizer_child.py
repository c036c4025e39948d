"""Runs INSIDE the child process of tests/test_sanitizers.py, with the AddressSanitizer + UndefinedBehaviorSanitizer runtimes
preloaded and SHRAY_HOST_LIB / SHRAY_ORACLE_LIB naming the sanitizer builds (`make -C shader-ray_amd sanitize`, `make -C oracle
sanitize`): first the loader / flattener / background / host-vs-reference / oracle KAT tests, in-process; then a seeded byte-mutation
fuzz of the three loaders (trisrc: trisrc-support.cpp:43-84 of the reference; OBJ: obj-support.cpp:226-320; Radiance .hdr) -- every
mutant either loads or is refused with a message; a sanitizer report aborts the process, which is what the parent looks for.

    python tests/sanitizer_child.py <fuzz cases per loader> <seed>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def write_hdr(path, rng):
    """A small Radiance picture with new-style run-length-coded scanlines (what host/background.cpp decodes)."""
    import numpy as np
    w, h = 24, 6
    rows = []
    for _ in range(h):
        row = bytearray([2, 2, w >> 8, w & 255])
        for _channel in range(4):
            x = 0
            while x < w:
                if rng.random() < 0.5:
                    n = int(min(w - x, rng.integers(1, 9)))
                    row += bytes([128 + n, int(rng.integers(0, 256))])
                else:
                    n = int(min(w - x, rng.integers(1, 9)))
                    row += bytes([n]) + bytes(int(v) for v in rng.integers(0, 256, n))
                x += n
        rows.append(bytes(row))
    open(path, "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + b"".join(rows))
    return np


def fuzz(cases, seed, tmp):
    import numpy as np
    from __graft_entry__ import load_package
    pkg = load_package()
    rng = np.random.default_rng(seed)
    golden = os.path.join(ROOT, "tests", "golden")
    hdr = os.path.join(tmp, "seed.hdr")
    write_hdr(hdr, rng)
    assert pkg.host.load_background(hdr).shape == (6, 24, 3)
    seeds = [(os.path.join(golden, "lobed_528.trisrc"), ".trisrc", lambda p: pkg.World(p).close()),
             (os.path.join(golden, "quads_mixed.obj"), ".obj", lambda p: pkg.World(p).close()),
             (os.path.join(golden, "quads_nonormals.obj"), ".obj", lambda p: pkg.World(p).close()),
             (hdr, ".hdr", lambda p: pkg.host.load_background(p))]
    tally = {}
    for source, suffix, load in seeds:
        data = open(source, "rb").read()
        # (the trisrc scene is 21 KB of text per 64 triangles; cut it so that a case costs milliseconds)
        if suffix == ".trisrc":
            data = data[:data.index(b'"', 40000)] if len(data) > 40000 and b'"' in data[40000:] else data
        loaded = refused = 0
        for _case in range(cases):
            mutant = bytearray(data)
            kind = int(rng.integers(0, 4))
            if kind == 0:      # a few bytes replaced
                for _ in range(int(rng.integers(1, 6))):
                    mutant[int(rng.integers(0, len(mutant)))] = int(rng.integers(0, 256))
            elif kind == 1:    # truncated
                del mutant[int(rng.integers(1, len(mutant))):]
            elif kind == 2:    # a run of bytes removed
                at = int(rng.integers(0, len(mutant) - 1))
                del mutant[at:at + int(rng.integers(1, 64))]
            else:              # a digit, sign or separator dropped in where a number stands
                for _ in range(int(rng.integers(1, 4))):
                    mutant[int(rng.integers(0, len(mutant)))] = int(rng.choice(list(b"0123456789-+.e /\n\"")))
            path = os.path.join(tmp, "mutant" + suffix)
            open(path, "wb").write(bytes(mutant))
            try:
                load(path)
                loaded += 1
            except (RuntimeError, ValueError, OSError):
                refused += 1
        tally[os.path.basename(source)] = (loaded, refused)
    return tally


def main():
    import tempfile
    import pytest
    cases, seed = int(sys.argv[1]), int(sys.argv[2])
    assert "_san" in os.environ.get("SHRAY_HOST_LIB", "") and "_san" in os.environ.get("SHRAY_ORACLE_LIB", ""), "not the sanitizer builds"
    tests = [os.path.join(ROOT, "tests", name) for name in
             ("test_loaders.py", "test_background.py", "test_flatten_tree.py", "test_host_vs_reference.py", "test_oracle_kat.py")]
    rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                      "--deselect", os.path.join(ROOT, "tests", "test_loaders.py") + "::test_threaded_loaders_equal_the_serial_loaders", *tests])
    if rc != 0:
        sys.exit(int(rc) or 1)
    # whole files through triangle_set::add_bulk (4,096 triangles and more: hashed shards, open-addressing tables) and, for the
    # OBJ, the mapped file and the pieces' parallel append -- the fuzz seeds below are too small to reach them
    import helpers
    from __graft_entry__ import load_package
    pkg = load_package()
    for path in (helpers.bunny_trisrc(), helpers.million_obj()):
        world = pkg.World(path)
        assert world.info.triangle_count >= 4096, path
        world.close()
    with tempfile.TemporaryDirectory() as tmp:
        tally = fuzz(cases, seed, tmp)
    print("SANITIZER_CHILD_OK", tally, flush=True)


if __name__ == "__main__":
    main()
