"""N > 1 path on CPU: two gloo ranks render their row shards (with the oracle standing in for the kernels,
which need a GPU) and rank 0 assembles the FrameBuffer with the same gather bench.py uses over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shard_rows, out_path, H=37):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import gnxraytracer_amd as gx
    import oracle_lib as ol
    import scenes
    from gnxraytracer_amd.distributed import gather_framebuffer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = scenes.cornell()
    integ = gx.PathIntegrator(8, 1.0, "spatial")
    W, spp = 40, 4                 # H: odd heights give unequal shard sizes
    img, st = ol.OracleScene(b).render(integ, W, H, spp, threads=2, shard_index=rank, shard_count=world, shard_rows=shard_rows)
    full = gather_framebuffer(torch.from_numpy(img), rank, world, shard_rows, dst=0)
    rays = torch.tensor([st["rays_closest"] + st["rays_any"]], dtype=torch.int64)
    dist.all_reduce(rays)
    if rank == 0:
        ref, rst = ol.OracleScene(b).render(integ, W, H, spp, threads=2)
        np.savez(out_path, full=full.numpy(), ref=ref, rays=rays.numpy(), ref_rays=rst["rays_closest"] + rst["rays_any"])
    else:
        assert full is None
    dist.destroy_process_group()


@pytest.mark.parametrize("shard_rows", [1, 4])
def test_two_rank_gather_reassembles_the_framebuffer(tmp_path, shard_rows):
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), shard_rows, out), nprocs=2, join=True)
    r = np.load(out)
    assert (r["full"].view(np.uint32) == r["ref"].view(np.uint32)).all()
    assert int(r["rays"][0]) == int(r["ref_rays"])


def test_four_rank_gather_with_a_height_not_divisible_by_four(tmp_path):
    """world 4, 38 rows: ranks 0 and 1 own ten rows, ranks 2 and 3 nine -- the padded equal-size gather must drop the padding of the
    short shards and the ray totals of the four shards must add up to the unsharded render's."""
    out = str(tmp_path / "out4.npz")
    mp.spawn(_worker, args=(4, _free_port(), 1, out, 38), nprocs=4, join=True)
    r = np.load(out)
    assert r["full"].shape[0] == 38
    assert (r["full"].view(np.uint32) == r["ref"].view(np.uint32)).all()
    assert int(r["rays"][0]) == int(r["ref_rays"])
