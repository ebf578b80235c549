"""world_size-2 gloo test (CPU) of the multi-GPU shard + gather logic (llamagen_amd/dist.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from llamagen_amd.dist import gather_to_root, init_from_env, rank_seed
    r, _, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank_seed(3, rank, world))
    n = 4
    local = torch.full((n, 2, 3), float(rank)) + torch.arange(n).view(n, 1, 1) * 10  # value = 10*i + rank
    out = gather_to_root(local)
    u8 = gather_to_root((local[:, :1] % 251).to(torch.uint8))  # the bench gathers uint8 HWC images
    dist.barrier()
    if rank == 0:
        assert u8.dtype == torch.uint8 and u8.shape == (n * world, 1, 3) and out.is_contiguous()
        assert all(int(u8[g, 0, 0]) == (10 * (g // world) + g % world) % 251 for g in range(n * world))
        q.put(out.clone())
    else:
        assert out is None and u8 is None
    dist.destroy_process_group()


def test_gather_orders_by_reference_global_index():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    out = q.get()
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert out.shape == (8, 2, 3)
    # global index g = i*world + rank  (sample_c2i_ddp.py:146)  -> value 10*i + rank
    for g in range(8):
        i, r = divmod(g, world)
        assert (out[g] == 10 * i + r).all()


def test_rank_seed_rule():
    from llamagen_amd.dist import rank_seed
    assert [rank_seed(0, r, 8) for r in range(8)] == list(range(8))
    assert rank_seed(2, 3, 8) == 19
