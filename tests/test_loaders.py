"""Loader behaviour and error handling of the host layer, mirroring what the reference's
loaders do with the same input (trisrc-support.cpp:43-105, obj-support.cpp:226-360,
world.cpp:46-134)."""
import os

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write(path, text):
    with open(path, "w") as f:
        f.write(text)
    return str(path)


TRI = ('"*" t 0.5 0.5 0.5 1 20\n'
       '0 0 0 0 0 1 1 1 1 1 0 0\n'
       '1 0 0 0 0 1 1 1 1 1 0 0\n'
       '0 1 0 0 0 2 1 1 1 1 0 0\n')


def test_trisrc_single_triangle(pkg, tmp_path):
    w = pkg.World(write(tmp_path / "one.trisrc", TRI))
    a = w.arrays()
    assert w.triangle_count == 1 and a["group_count"] == 1 and a["tree_root"] == 0
    assert np.array_equal(a["vertex_positions"], np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32))
    assert np.array_equal(a["vertex_normals"].reshape(3, 3)[2], np.array([0, 0, 1], np.float32))   # normalised
    assert np.array_equal(a["group_objects"], np.array([0, 1], np.float32))
    # a single leaf terminates in every direction table; 0x7fffffff is stored as 2^31
    for c in range(8):
        assert np.array_equal(a[f"group_hitmiss_{c}"], np.array([2147483648.0] * 2, np.float32))
    assert np.float32(2147483648.0) >= np.float32(16777215.0)   # terminator test of raytracer.es.fs:432
    # boxes are inflated by 1e-5 (vectormath.h:189-195)
    assert np.allclose(a["group_boxmin"], [-1e-5, -1e-5, -1e-5]) and np.allclose(a["group_boxmax"], [1 + 1e-5, 1 + 1e-5, 1e-5])


def test_trisrc_truncated_record_fails(pkg, tmp_path):
    with pytest.raises(RuntimeError):
        pkg.World(write(tmp_path / "cut.trisrc", TRI + TRI[:60]))


def test_trisrc_stops_quietly_at_foreign_text(pkg, tmp_path):
    # the reference's scan loop ends (successfully) when the next record does not open with a quote
    w = pkg.World(write(tmp_path / "tail.trisrc", TRI + "end of data\n"))
    assert w.triangle_count == 1


def test_empty_trisrc_gives_empty_world(pkg, tmp_path):
    w = pkg.World(write(tmp_path / "empty.trisrc", ""))
    a = w.arrays()
    assert w.triangle_count == 0 and a["group_count"] == 1 and a["vertex_count"] == 0
    assert np.array_equal(a["group_objects"], np.array([0, 0], np.float32))


def test_unknown_extension_and_missing_file(pkg, tmp_path):
    with pytest.raises(RuntimeError):
        pkg.World(write(tmp_path / "scene.ply", "ply\n"))
    with pytest.raises(RuntimeError):
        pkg.World(str(tmp_path / "nope.obj"))


def test_vertex_sharing_and_leaf_split(pkg, tmp_path):
    # 12 triangles in a strip: more than bvh_leaf_max (10) -> one split, 3 nodes
    rows = []
    for k in range(12):
        x = float(k)
        rows.append('"*" t 0.5 0.5 0.5 1 20\n'
                    f'{x} 0 0 0 0 1 1 1 1 1 0 0\n{x + 1} 0 0 0 0 1 1 1 1 1 0 0\n{x} 1 0 0 0 1 1 1 1 1 0 0\n')
    w = pkg.World(write(tmp_path / "strip.trisrc", "".join(rows)))
    a = w.arrays()
    assert w.triangle_count == 12 and w.info.independent_vertex_count == 25   # (k,0) shared with (k-1)+1
    assert a["group_count"] == 3 and a["tree_root"] == 1            # in-order numbering: root mid-array
    objs = a["group_objects"].reshape(3, 2)
    assert objs[1].tolist() == [0, 0] and objs[0, 0] == 0 and objs[0, 1] + objs[2, 1] == 12 and objs[2, 0] == objs[0, 1]
    assert a["group_directions"].reshape(3, 3)[1].tolist() == [1, 0, 0]   # split along x
    # code 1 (+x): negative child first; code 0 (-x): positive child first
    hm1 = a["group_hitmiss_1"].reshape(3, 2)
    hm0 = a["group_hitmiss_0"].reshape(3, 2)
    assert hm1[1].tolist() == [0, 2147483648.0] and hm1[0].tolist() == [2, 2] and hm1[2].tolist() == [2147483648.0] * 2
    assert hm0[1].tolist() == [2, 2147483648.0] and hm0[2].tolist() == [0, 0] and hm0[0].tolist() == [2147483648.0] * 2


def test_obj_fan_triangulation_and_generated_normals(pkg, tmp_path):
    text = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"
    w = pkg.World(write(tmp_path / "quad.obj", text))
    a = w.arrays()
    assert w.triangle_count == 2
    assert np.array_equal(a["vertex_positions"].reshape(2, 3, 3)[1], np.array([[0, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32))
    assert np.array_equal(a["vertex_normals"].reshape(6, 3), np.tile(np.array([0, 0, 1], np.float32), (6, 1)))
    assert np.array_equal(a["vertex_colors"], np.ones(18, np.float32))


def test_obj_trailing_blank_quirks(pkg, tmp_path):
    # upstream's field splitter yields an extra empty field for a trailing blank: an attribute
    # line then has 4 fields and stays (0,0,0)   (obj-support.cpp:61-82, :148-169)
    text = "v 5 5 5 \nv 1 0 0\nv 0 1 0\nf 1 2 3\n"
    w = pkg.World(write(tmp_path / "trail.obj", text))
    assert np.array_equal(w.arrays()["vertex_positions"][:3], np.zeros(3, np.float32))


def test_obj_bad_index_fails(pkg, tmp_path):
    with pytest.raises(RuntimeError):
        pkg.World(write(tmp_path / "bad.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n"))


def test_generators_are_deterministic(pkg, tmp_path):
    import hashlib
    p1, p2 = str(tmp_path / "a.trisrc"), str(tmp_path / "b.trisrc")
    pos, tri = pkg.scenes.lobed_sphere_mesh(12, 24)
    pkg.scenes.write_trisrc(p1, pos, tri)
    pos, tri = pkg.scenes.lobed_sphere_mesh(12, 24)
    pkg.scenes.write_trisrc(p2, pos, tri)
    h = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in (p1, p2)]
    assert h[0] == h[1]
    assert len(tri) == 2 * 24 + 2 * 24 * 10
    e1, e2 = pkg.scenes.environment_hdr_sky(64), pkg.scenes.environment_hdr_sky(64)
    assert np.array_equal(e1, e2) and e1.max() > 10.0 and e1.dtype == np.float32
    g = pkg.scenes.environment_grid(64)
    assert g.shape == (32, 64, 3) and g[0].min() == 1.0 and g[1, 1].max() == 0.0 and g[3, 8].min() == 1.0


def test_trackball_rotation_matches_the_reference_formulas(pkg):
    """drag_to_rotation / trackball_motion (ray.cpp:76-98) through the frame parameters: a drag
    turns the object about (dy, dx, 0) by pi * |drag|; compared with numpy's axis-angle matrix."""
    import ctypes as C
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lobed_528.trisrc")
    world = pkg.World(golden)
    view = world.default_view()
    dx, dy = 0.03, -0.02
    dist = np.hypot(dx, dy)
    angle, axis = np.pi * dist, np.array([dy / dist, dx / dist, 0.0])
    view.object_rotation[:] = [float(angle), float(axis[0]), float(axis[1]), 0.0]
    fp = world.frame_params(64, 64, view)
    m = np.array(fp.object_matrix[:], dtype=np.float64).reshape(4, 4).T     # column-major -> rows
    c, s_, t = np.cos(angle), np.sin(angle), 1 - np.cos(angle)
    x, y, z = axis
    rot = np.array([[t * x * x + c, t * x * y - s_ * z, t * x * z + s_ * y],
                    [t * x * y + s_ * z, t * y * y + c, t * y * z - s_ * x],
                    [t * x * z - s_ * y, t * y * z + s_ * x, t * z * z + c]])
    assert np.allclose(m[:3, :3], rot, atol=1e-6)
    # the object matrix maps world to object space about the scene centre (ray.cpp:118-123)
    assert np.allclose(np.array(fp.object_normal_inverse[:]).reshape(4, 4).T[:3, :3], rot.T, atol=1e-6)


def test_threaded_bvh_build_equals_the_serial_build(pkg):
    """Large nodes build their two sub-trees concurrently (host/bvh.cpp); the flattened arrays --
    tree, boxes, triangle order -- must be those of the serial build (SHRAY_BVH_THREADS=0)."""
    import hashlib
    import subprocess
    import sys

    def digest(arrays):
        h = hashlib.sha256()
        for key in sorted(arrays):
            v = arrays[key]
            h.update(key.encode())
            h.update(v.tobytes() if isinstance(v, np.ndarray) else str(v).encode())
        return h.hexdigest()

    path = helpers.bunny_trisrc()           # 69,168 triangles: the root and both its children fork
    threaded = digest(pkg.World(path).arrays())
    code = ("import sys, hashlib, numpy as np; sys.path[:0] = [%r, %r]\n"
            "from __graft_entry__ import load_package\n"
            "a = load_package().World(%r).arrays()\n"
            "h = hashlib.sha256()\n"
            "for k in sorted(a):\n"
            "    v = a[k]; h.update(k.encode()); h.update(v.tobytes() if isinstance(v, np.ndarray) else str(v).encode())\n"
            "print(h.hexdigest())\n") % (ROOT, os.path.join(ROOT, "tests"), path)
    env = dict(os.environ, SHRAY_BVH_THREADS="0")
    serial = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()[-1]
    assert threaded == serial


def _digest_in_a_process(paths, threads):
    """sha256 of World(path).arrays() for each path, in a fresh process with SHRAY_LOAD_THREADS = threads (the loaders read
    it once); '!' where the load fails"""
    import subprocess
    import sys
    code = ("import sys, hashlib, numpy as np; sys.path[:0] = [%r, %r]\n"
            "from __graft_entry__ import load_package\n"
            "pkg = load_package()\n"
            "for path in %r:\n"
            "    try:\n"
            "        a = pkg.World(path).arrays()\n"
            "    except Exception:\n"
            "        print('!'); continue\n"
            "    h = hashlib.sha256()\n"
            "    for k in sorted(a):\n"
            "        v = a[k]; h.update(k.encode()); h.update(v.tobytes() if isinstance(v, np.ndarray) else str(v).encode())\n"
            "    print(h.hexdigest())\n") % (ROOT, os.path.join(ROOT, "tests"), list(paths))
    env = dict(os.environ, SHRAY_LOAD_THREADS=str(threads))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    return out[-len(paths):]


def test_threaded_loaders_equal_the_serial_loaders(pkg, tmp_path):
    """Round 4: files are parsed in pieces on several threads, vertices are de-duplicated in shards (triangle_set::add_bulk)
    and synthesized normals are summed per vertex by the thread that owns it -- the flattened arrays must be those of the
    one-thread load (SHRAY_LOAD_THREADS=1, the reference's order of everything): same vertex numbering, same triangle
    order, same float sums.  The BASELINE scenes, plus trisrc texts that defeat the piece-wise parse and must fall back:
    a tag that begins with a quote after a line end (a cut lands on it), a truncated last record, foreign text at the end."""
    rng = np.random.default_rng(7)

    def records(n, tag="t"):
        rows = []
        for _ in range(n):
            v = rng.uniform(-1, 1, (3, 3)).round(3)
            rows.append(f'"*" {tag} 0.5 0.5 0.5 1 20\n' + "".join(
                f"{p[0]} {p[1]} {p[2]} 0 0 1 0.8 0.8 0.8 1 0 0\n" for p in v))
        return "".join(rows)

    body = records(4000)                      # ~700 KB: several pieces
    tricky = write(tmp_path / "quoted_tag.trisrc", records(2000) + records(2000, tag='\n"q'))
    truncated = write(tmp_path / "truncated.trisrc", body + '"*" t 0.5 0.5 0.5 1 20\n0 0 0 0 0 1 1 1 1 1 0 0\n')
    foreign = write(tmp_path / "foreign.trisrc", body + "end of file\n")
    plain = write(tmp_path / "plain.trisrc", body)
    paths = [helpers.bunny_trisrc(), helpers.million_obj(), os.path.join(ROOT, "tests", "golden", "quads_nonormals.obj"),
             plain, tricky, truncated, foreign]
    serial = _digest_in_a_process(paths, 1)
    for threads in (3, 8):
        assert _digest_in_a_process(paths, threads) == serial, threads
    assert serial[5] == "!" and "!" not in serial[:5] + serial[6:]      # the truncated record fails the load either way
    assert len(set(serial[3:5])) == 2                                    # (the tricky text is another scene than the plain one)


def test_load_triangles_and_adopt_tree(pkg):
    """load_world in two steps, for a BVH built elsewhere (the GPU build: include/shader_ray_host.h).  Without a GPU: the tree the HOST
    builder made for a scene, exported (shray_host_export_tree), is adopted by a second world that only loaded the triangles -- with
    the build's triangle order recovered from the two worlds' vertex indices (the lobed mesh has no two triangles with the same three
    vertices) -- and flattens to the same arrays; trees that are not pre-order binary trees over exactly these triangles are refused
    and leave the world as it was."""
    import ctypes as C
    N = pkg._native
    lib = N.load_host()
    path = os.path.join(helpers.GOLDEN, "lobed_528.trisrc") if hasattr(helpers, "GOLDEN") else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "golden", "lobed_528.trisrc")
    built = pkg.World(path)
    tree = built.export_tree()
    n, t = tree.node_count, tree.triangle_count
    after = np.ctypeslib.as_array(tree.triangle_vertices, shape=(t, 3)).copy()

    def fresh():
        handle = C.c_void_p()
        assert lib.shray_host_load_triangles(path.encode(), C.byref(handle)) == 0 and handle
        tv, vd, nt, nv = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int32(), C.c_int32()
        assert lib.shray_host_triangles(handle, C.byref(tv), C.byref(nt), C.byref(vd), C.byref(nv)) == 0 and nt.value == t
        return handle, np.ctypeslib.as_array(tv, shape=(t, 3)).copy()

    handle, before = fresh()
    where = {tuple(row): k for k, row in enumerate(before.tolist())}
    assert len(where) == t
    order = np.asarray([where[tuple(row)] for row in after.tolist()], dtype=np.int32)
    order_p = order.ctypes.data_as(C.POINTER(C.c_int32))
    # refused: a triangle twice; a tree whose negative child is not the next node
    twice = order.copy()
    twice[1] = twice[0]
    assert lib.shray_host_adopt_tree(handle, C.byref(tree), twice.ctypes.data_as(C.POINTER(C.c_int32)), 0.0) == -1
    negative = np.ctypeslib.as_array(tree.node_negative, shape=(n,)).copy()
    branch = int(np.nonzero(negative >= 0)[0][0])
    saved = tree.node_negative[branch]
    tree.node_negative[branch] = saved + 1
    assert lib.shray_host_adopt_tree(handle, C.byref(tree), order_p, 0.0) == -1
    tree.node_negative[branch] = saved
    # refused: arrays that pass every local check (negative == g + 1 < positive < node_count, leaves in triangle order) and are
    # still no tree -- a branch's positive child must start right behind its negative child's subtree; moved by one, a node would
    # have two parents (and be deleted twice)
    positive = np.ctypeslib.as_array(tree.node_positive, shape=(n,)).copy()
    for branch in np.nonzero(negative >= 0)[0].tolist():
        was = int(positive[branch])
        for moved in (was - 1, was + 1):
            if int(negative[branch]) < moved < n:
                tree.node_positive[branch] = moved
                assert lib.shray_host_adopt_tree(handle, C.byref(tree), order_p, 0.0) == -1, (branch, moved)
        tree.node_positive[branch] = was
    # adopted: the same flattened arrays and statistics as the world make_bvh built
    assert lib.shray_host_adopt_tree(handle, C.byref(tree), order_p, 0.125) == 0
    info = N.HostWorldInfo()
    lib.shray_host_get_world_info(handle, C.byref(info))
    for field in ("node_count", "leaf_count", "max_level", "large_leaves", "triangle_count"):
        assert getattr(info, field) == getattr(built.info, field), field
    assert info.build_seconds == 0.125
    desc = N.SceneDesc()
    assert lib.shray_host_flatten(handle, 2048, C.byref(desc)) == 0
    mine, want = pkg.host.desc_arrays(desc), built.arrays()
    for key, value in want.items():
        if isinstance(value, np.ndarray):
            assert np.array_equal(value.view(np.uint32), mine[key].view(np.uint32)), key
        else:
            assert value == mine[key], key
    assert lib.shray_host_adopt_tree(handle, C.byref(tree), order_p, 0.0) == -1      # a world that has its tree takes no other
    lib.shray_host_free_world(handle)
    built.close()
