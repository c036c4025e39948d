"""The GPU flattener's input and algorithm, checked on the CPU: shray_host_export_tree delivers the BVH in
pre-order, and the ancestor-walk formulas of shader-ray_amd/csrc/flatten.hip (in-order number, threaded links;
reference world.cpp:145-177, :231-288), restated in numpy, reproduce the host flattener's arrays -- which
tests/test_host_vs_reference.py pins to the reference bit for bit."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STOP = np.float32(2147483648.0)


def tree_arrays(tree):
    n, t, v = tree.node_count, tree.triangle_count, tree.vertex_count
    take = lambda p, k, dt: np.ctypeslib.as_array(p, shape=(k,)).astype(dt).copy()   # noqa: E731
    return {"parent": take(tree.node_parent, n, np.int64), "negative": take(tree.node_negative, n, np.int64),
            "positive": take(tree.node_positive, n, np.int64), "box": take(tree.node_box, 6 * n, np.float32).reshape(n, 6),
            "direction": take(tree.node_direction, 3 * n, np.float32).reshape(n, 3),
            "start": take(tree.node_start, n, np.int64), "triangles": take(tree.node_triangles, n, np.int64),
            "triangle_vertices": take(tree.triangle_vertices, 3 * t, np.int64),
            "vertex_data": take(tree.vertex_data, 9 * v, np.float32).reshape(v, 9)}


def flatten_by_ancestor_walks(a):
    """flatten.hip in numpy (loops over nodes; small trees only)."""
    n = len(a["parent"])
    par, neg, pos = a["parent"], a["negative"], a["positive"]
    index_of = np.zeros(n, np.int64)
    for g in range(n):
        idx = pos[g] - g - 1 if neg[g] >= 0 else 0
        child, p = g, par[g]
        while p >= 0:
            if child == pos[p]:
                idx += pos[p] - p
            child, p = p, par[p]
        index_of[g] = idx
    out = {"tree_root": int(index_of[0])}
    boxmin, boxmax = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    directions, children, objects = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32)
    for g in range(n):
        me = index_of[g]
        boxmin[me], boxmax[me] = a["box"][g, :3], a["box"][g, 3:]
        if neg[g] < 0:
            children[me] = STOP
            objects[me] = (a["start"][g], a["triangles"][g])
        else:
            directions[me] = a["direction"][g]
            children[me] = (index_of[neg[g]], index_of[pos[g]])
    out.update(group_boxmin=boxmin.reshape(-1), group_boxmax=boxmax.reshape(-1), group_directions=directions.reshape(-1),
               group_children=children.reshape(-1), group_objects=objects.reshape(-1))
    for code in range(8):
        sign = np.array([1 if code & 1 else -1, 1 if code & 2 else -1, 1 if code & 4 else -1], np.float32)

        def pos_near(g):
            d = a["direction"][g]
            return np.float32(np.float32(sign[0] * d[0] + sign[1] * d[1]) + sign[2] * d[2]) < 0
        table = np.zeros((n, 2), np.float32)
        for g in range(n):
            nxt, child, p = -1, g, par[g]
            while p >= 0:
                near = pos[p] if pos_near(p) else neg[p]
                if child == near:
                    nxt = neg[p] if pos_near(p) else pos[p]
                    break
                child, p = p, par[p]
            miss = STOP if nxt < 0 else np.float32(index_of[nxt])
            hit = miss if neg[g] < 0 else np.float32(index_of[pos[g] if pos_near(g) else neg[g]])
            table[index_of[g]] = (hit, miss)
        out[f"group_hitmiss_{code}"] = table.reshape(-1)
    corners = a["vertex_data"][a["triangle_vertices"]]
    out.update(vertex_positions=corners[:, 0:3].reshape(-1), vertex_colors=corners[:, 3:6].reshape(-1),
               vertex_normals=corners[:, 6:9].reshape(-1))
    return out


@pytest.mark.parametrize("name", ["lobed_528.trisrc", "quads_mixed.obj", "quads_nonormals.obj"])
def test_exported_tree_is_preorder_and_flattens_to_the_host_arrays(pkg, name):
    world = pkg.World(os.path.join(GOLDEN, name))
    a = tree_arrays(world.export_tree())
    n = len(a["parent"])
    assert n == world.info.node_count and a["parent"][0] == -1
    branch = a["negative"] >= 0
    assert np.array_equal(a["negative"][branch], np.nonzero(branch)[0] + 1)          # negative child = node + 1
    assert np.all(a["positive"][branch] > a["negative"][branch]) and np.all(a["positive"][~branch] == -1)
    assert np.array_equal(a["parent"][a["negative"][branch]], np.nonzero(branch)[0])
    assert np.array_equal(a["parent"][a["positive"][branch]], np.nonzero(branch)[0])
    assert a["triangles"][~branch].sum() == world.info.triangle_count                # leaves partition the triangles
    mine = flatten_by_ancestor_walks(a)
    host = world.arrays()
    assert mine["tree_root"] == host["tree_root"]
    for key, got in mine.items():
        if key == "tree_root":
            continue
        assert got.shape == host[key].shape, key
        assert np.array_equal(np.ascontiguousarray(got, np.float32).view(np.uint32), host[key].view(np.uint32)), key
    world.close()
