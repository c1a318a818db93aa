"""Row sharding of the image over ranks and the final FrameBuffer gather (one process per GPU).

Pixels are independent (the Halton index is a function of pixel and sample number only,
samplers/HaltonSampler.cpp:63-83, and Render box-averages per pixel, core/Integrator.cpp:293), so rows are
dealt round-robin in blocks of `shard_rows` with no exchange during rendering; one gather of the row
shards to rank 0 assembles the FrameBuffer.  torch.distributed is plumbing here: backend "nccl" (RCCL over
xGMI) on GPUs, "gloo" in the CPU tests.
"""


def shard_row_index(height, rank, world, shard_rows=1):
    """Rows owned by `rank`: (y // shard_rows) % world == rank  (gnxr_render_params)."""
    return [y for y in range(height) if (y // shard_rows) % world == rank]


def gather_framebuffer(local_full, rank, world, shard_rows=1, dst=0):
    """local_full: [H, W, C] tensor where only this rank's rows are valid.  Returns the assembled image on
    `dst` (None elsewhere).  One dist.gather of the compacted row shards."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local_full
    H = local_full.shape[0]
    rows = [torch.tensor(shard_row_index(H, r, world, shard_rows), dtype=torch.long, device=local_full.device) for r in range(world)]
    mine = local_full.index_select(0, rows[rank]).contiguous()
    n_max = max(len(r) for r in rows)
    if mine.shape[0] < n_max:  # equal-sized gather buffers
        pad = torch.zeros((n_max - mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        mine = torch.cat([mine, pad], 0)
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, parts, dst=dst)
    if rank != dst:
        return None
    out = torch.zeros_like(local_full)
    for r in range(world):
        out.index_copy_(0, rows[r], parts[r][: len(rows[r])])
    return out
