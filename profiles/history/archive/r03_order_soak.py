#!/usr/bin/env python3
"""Soak of the dispatch order: many launches of a tile set and of lone frames on four streams, every frame compared with a
row-major render of the same view (SHRAY_DISPATCH_ORDER=0 cannot be switched inside a process, so the reference is the
counting twin's image, which never reads an order); every permutation the library hands out must be one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
import helpers  # noqa: E402
import bench  # noqa: E402

pkg = load_package()
W, H = 960, 540
world = pkg.World(helpers.bunny_trisrc())
scene = pkg.Scene(world.flatten(), pkg.scenes.environment_hdr_sky(1024), device=0)
orbit = bench.orbit_params(pkg, world, W, H, 0)
N = pkg._native
want = [torch.from_numpy(scene.render_counters(v, W, H, 1)[0].reshape(-1)).cuda() for v in orbit]
streams = [torch.cuda.Stream() for _ in range(4)]
bad = launches = 0
for tiles in (N.TileSet(32, 32, 8, 7, 1), None, N.TileSet(32, 32, 4, 1, 2)):
    nbytes = pkg.tracer.tile_buffer_bytes(W, H, tiles)
    count = 1 if tiles is None else 8
    outs = [torch.empty(count * nbytes // 4, dtype=torch.float32, device="cuda") for _ in range(8)]
    for j in range(400):
        st = streams[j % 4]
        views = [orbit[(j * count + k) % 20] for k in range(count)]
        out = outs[j % 8]
        if count == 1:
            scene.render_into(views[0], W, H, 1, out.data_ptr(), st.cuda_stream, tiles)
        else:
            scene.render_batch_into(views, W, H, 1, out.data_ptr(), nbytes, st.cuda_stream, tiles)
        launches += 1
        if j % 8 == 7:
            torch.cuda.synchronize()
            order = scene.dispatch_order()
            assert order.size == 0 or np.array_equal(np.sort(order), np.arange(order.size)), "not a permutation"
            # check the eight buffers just written
            for b in range(8):
                jj = j - 7 + b
                for k in range(count):
                    v = (jj * count + k) % 20
                    got = outs[jj % 8][k * nbytes // 4:(k + 1) * nbytes // 4]
                    if tiles is None:
                        same = torch.equal(got, want[v])
                    else:
                        # the packed buffer holds the set's tiles in order: compare tile by tile with the whole frame
                        from shader_ray_amd.multigpu import owned_tiles
                        full = want[v].reshape(H, W, 4)
                        packed = got.reshape(-1, tiles.tile_h, tiles.tile_w, 4)
                        same = True
                        tx_n = -(-W // tiles.tile_w)
                        for i, t in enumerate(owned_tiles(W, H, tiles.tile_w, tiles.tile_h, tiles.tile_stride, tiles.tile_phase,
                                                          max(1, tiles.tile_phase_count))[0]):
                            ty, tx = divmod(int(t), tx_n)
                            y0, x0 = ty * tiles.tile_h, tx * tiles.tile_w
                            h, w = min(tiles.tile_h, H - y0), min(tiles.tile_w, W - x0)
                            same = same and torch.equal(packed[i, :h, :w], full[y0:y0 + h, x0:x0 + w])
                    bad += 0 if same else 1
print(f"{launches} launches, {bad} frames that differ")
sys.exit(1 if bad else 0)
