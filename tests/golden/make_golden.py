"""Regenerates the committed fixtures.  Run from the repository root, in the build
container (it needs /root/reference for oracle/_ref/ref_host):

    python tests/golden/make_golden.py

Writes, next to this script:
  lobed_528.trisrc, quads_mixed.obj, quads_nonormals.obj   small scene files (inputs)
  <scene>.ref.npz     the REFERENCE's flattened arrays + frame parameters for each, dumped
                      by oracle/_ref/ref_host (the reference's own loader / BVH / flattener /
                      view-parameter code compiled where it lies)
  lobed_528.oracle.npz  64x64 frames of the small trisrc scene rendered by the CPU oracle
                      (gold and glazed plaster) -- regression vectors for the GPU path;
                      they pin the oracle's output at the time of commit, not the reference
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from __graft_entry__ import load_package  # noqa: E402
from refdump import read_dump  # noqa: E402
import oracle  # noqa: E402


def write_quads_mixed(path):
    # a cube-ish blob: 8 corners, 6 quads, mixed corner syntaxes, comments, an `o` line
    v = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1.2, 1.1, 1.3), (-1, 1, 1)]
    vn = [(-0.577, -0.577, -0.577), (0.577, -0.577, -0.577), (0.577, 0.577, -0.577), (-0.577, 0.577, -0.577),
          (-0.577, -0.577, 0.577), (0.577, -0.577, 0.577), (0.577, 0.577, 0.577), (-0.577, 0.577, 0.577)]
    quads = [(1, 4, 3, 2), (5, 6, 7, 8), (1, 2, 6, 5), (2, 3, 7, 6), (3, 4, 8, 7), (4, 1, 5, 8)]
    with open(path, "w") as f:
        f.write("# mixed-syntax fixture\n\no blob\n")
        for p in v:
            f.write("v %g %g %g\n" % p)
        for i in range(4):
            f.write("vt %g %g\n" % (i % 2, i // 2))
        for n in vn:
            f.write("vn %g %g %g\n" % n)
        f.write("# faces\n")
        for k, q in enumerate(quads):
            if k % 3 == 0:
                f.write("f " + " ".join("%d//%d" % (i, i) for i in q) + "\n")
            elif k % 3 == 1:
                f.write("f " + " ".join("%d/%d/%d" % (i, (i % 4) + 1, i) for i in q) + "\n")
            else:
                f.write("f  " + "   ".join("%d/%d/%d" % (i, 1, i) for i in q) + "\n")
        f.write("f 1//1 3//3 6//6\n")


def write_quads_nonormals(pkg, path):
    pos, tri = pkg.scenes.lobed_sphere_mesh(10, 16, bumpiness=0.2, ears=False, scale=2.0, center=(0.5, 0.25, -1.0))
    # ring quads between the first two interior rings, then the remaining triangles
    n_lon = 16
    quads = np.array([[1 + j, 1 + (j + 1) % n_lon, 1 + n_lon + (j + 1) % n_lon, 1 + n_lon + j] for j in range(n_lon)])
    keep = tri[n_lon + 2 * n_lon:]   # drop the top fan's successor band (replaced by the quads)
    top = tri[:n_lon]
    pkg.scenes.write_obj(path, pos, np.concatenate([top, keep]), quads=quads)


def dump_reference(scene_path, out_npz):
    tmp = out_npz + ".bin"
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_host"), scene_path, tmp, "1920", "1080"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = read_dump(tmp)
    os.remove(tmp)
    np.savez_compressed(out_npz, **d)


def main():
    pkg = load_package()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)

    tri_path = os.path.join(HERE, "lobed_528.trisrc")
    pos, tri = pkg.scenes.lobed_sphere_mesh(12, 24, bumpiness=0.22, ears=True)
    pkg.scenes.write_trisrc(tri_path, pos, tri)
    mixed = os.path.join(HERE, "quads_mixed.obj")
    write_quads_mixed(mixed)
    nonorm = os.path.join(HERE, "quads_nonormals.obj")
    write_quads_nonormals(pkg, nonorm)

    for path in (tri_path, mixed, nonorm):
        dump_reference(path, os.path.splitext(path)[0] + ".ref.npz")

    world = pkg.World(tri_path)
    desc = world.flatten()
    env = pkg.scenes.environment_hdr_sky(128)
    frames = {}
    for name, material in (("gold", 0), ("plaster", 6)):
        params = world.frame_params(64, 64, material=material)
        img, counters = oracle.render(desc, env, params, 64, 64, 1)
        frames[name] = img
        frames[name + "_counters"] = np.array([counters[k] for k in sorted(counters)], dtype=np.uint64)
    params = world.frame_params(64, 64, material=6)
    img, counters = oracle.render(desc, env, params, 64, 64, 4)
    frames["plaster_4spp"] = img
    np.savez_compressed(os.path.join(HERE, "lobed_528.oracle.npz"), **frames)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
