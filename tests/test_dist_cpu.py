"""World-size-2 (and 3) check of the tile-sharded frame path on CPU with the gloo backend.

Each rank renders its interleaved tiles with the CPU oracle (the GPU is not available here), then runs the product's
gather + de-interleave (rust-raytracer_amd/dist_tiles.py), the same code bench.py uses with the nccl (RCCL) backend.
Rank 0 must end up with exactly the full-frame image.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TW, TH = 100, 72, 32, 24


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as entry
    import oracle_lib as ol
    entry.load_package()
    from rust_raytracer_amd import dist_tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ol.OracleScene.main(sky=ol.procedural_sky(), cam=(W, H))
    tiles_x, tiles_y = dist_tiles.tile_grid(W, H, TW, TH)
    ids = list(range(rank, tiles_x * tiles_y, world))
    tiles = np.zeros((len(ids), TH, TW, 3), np.float32)
    for q, t in enumerate(ids):  # what rtamd_render writes for this rank's region: tile-major, clipped tiles zero-padded
        ty, tx = divmod(t, tiles_x)
        w, h = min(TW, W - tx * TW), min(TH, H - ty * TH)
        img, _, _ = scene.render(max_depth=3, antialiasing=1, rect=(tx * TW, ty * TH, w, h), threads=1)
        tiles[q, :h, :w] = img
    frame = dist_tiles.gather_frame(torch.from_numpy(tiles), W, H, TW, TH, rank, world)
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_gather_gloo(world, tmp_path, ol):
    out = str(tmp_path / "frame.npy")
    port = 29500 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    full, _, _ = ol.OracleScene.main(sky=ol.procedural_sky(), cam=(W, H)).render(max_depth=3, antialiasing=1, threads=2)
    got = np.load(out)
    assert got.shape == (H, W, 3)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_owned_tile_count():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    entry.load_package()
    from rust_raytracer_amd import dist_tiles
    for n in (1, 7, 255):
        for world in (1, 2, 3, 8):
            assert sum(dist_tiles.owned_tile_count(n, r, world) for r in range(world)) == n
