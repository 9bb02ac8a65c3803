"""Image-tile sharding of one frame over the ranks of a torch.distributed job, and the framebuffer gather.

The frame rectangle is cut into tile_w x tile_h tiles in row-major order; rank r of N owns tiles
r, r+N, r+2N, ... (interleaved, so cheap sky tiles and expensive geometry tiles are spread evenly).
Every pixel has exactly one owner, so the only exchange step of a frame is one gather of the owned tiles
to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on CPU tensors in the tests), followed by a
de-interleave into the row-major image.  No reduction is needed.
"""
import torch
import torch.distributed as dist


def tile_grid(w, h, tile_w, tile_h):
    return (w + tile_w - 1) // tile_w, (h + tile_h - 1) // tile_h


def owned_tile_count(n_tiles, rank, world):
    return 0 if rank >= n_tiles else (n_tiles - rank + world - 1) // world


def gather_frame(tiles, w, h, tile_w, tile_h, rank, world, dst=0):
    """tiles: this rank's [n_owned, tile_h, tile_w, 3] float tensor (tile-major, as rtamd_render writes it).

    Returns the [h, w, 3] image on rank `dst`, None elsewhere.  With world == 1 no collective runs."""
    tiles_x, tiles_y = tile_grid(w, h, tile_w, tile_h)
    n_tiles = tiles_x * tiles_y
    per_rank = (n_tiles + world - 1) // world  # ranks are padded to the same tile count for the gather
    tiles = tiles.reshape(-1, tile_h, tile_w, 3)
    assert tiles.shape[0] == owned_tile_count(n_tiles, rank, world)
    if world > 1:
        if tiles.is_cuda and dist.get_backend() == "gloo":
            tiles = tiles.cpu()  # gloo gathers host tensors only (rehearsal of the multi-rank path without one GPU per rank)
        if tiles.shape[0] < per_rank:
            pad = torch.zeros((per_rank - tiles.shape[0], tile_h, tile_w, 3), dtype=tiles.dtype, device=tiles.device)
            tiles = torch.cat([tiles, pad], dim=0)
        tiles = tiles.contiguous()
        if rank == dst:
            parts = [torch.empty_like(tiles) for _ in range(world)]
            dist.gather(tiles, gather_list=parts, dst=dst)
        else:
            dist.gather(tiles, gather_list=None, dst=dst)
            return None
    else:
        parts = [tiles]
    # de-interleave: tile t = r + q * world lives at parts[r][q]
    grid = torch.empty((tiles_y * tiles_x, tile_h, tile_w, 3), dtype=tiles.dtype, device=tiles.device)
    for r in range(world):
        cnt = owned_tile_count(n_tiles, r, world)
        if cnt:
            grid[r::world] = parts[r][:cnt]
    img = grid.reshape(tiles_y, tiles_x, tile_h, tile_w, 3).permute(0, 2, 1, 3, 4).reshape(tiles_y * tile_h, tiles_x * tile_w, 3)
    return img[:h, :w].contiguous()
