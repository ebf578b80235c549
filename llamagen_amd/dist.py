"""Data-parallel sampling across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference shards FID sampling as fully independent replicas (autoregressive/sample/
sample_c2i_ddp.py:44-50, 114-157): per-rank seed `global_seed * world_size + rank`, own label draw,
no tensor collective -- results meet on a shared filesystem as PNGs and the global image index is
`i * world_size + rank + total` (:146).  Here the shards meet in memory instead: ONE `gather`
(RCCL, root receives over its 7 xGMI links concurrently) of the decoded batch at the end of each
sampling pass, re-ordered to the reference's interleaved global index.  No collective runs during
generation (images are independent units).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def rank_seed(global_seed: int, rank: int, world: int) -> int:
    """sample_c2i_ddp.py:47"""
    return global_seed * world + rank


def gather_to_root(local: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """One collective per sampling pass: every rank contributes its [n, ...] shard; rank `dst` returns
    [world * n, ...] ordered by the reference's global index i * world + rank (sample_c2i_ddp.py:146); others None.

    The receive side is ONE [world, n, ...] allocation whose rows are the gather targets (no per-rank buffers, no
    torch.stack); the interleaved order is a transposed view of it, made contiguous once.  Send uint8 HWC images
    (postprocess.to_uint8_hwc, what the reference writes to disk) rather than fp32: 14.2 MB instead of 56.6 MB per
    rank at 32 x 384 px.  The collective is enqueued behind the CURRENT stream (the lane that produced `local`), so a
    pipelined caller can issue it as soon as that lane's batch is enqueued; all ranks must call in the same order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    local = local.contiguous()
    if rank == dst:
        buf = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=list(buf.unbind(0)), dst=dst)
        return buf.transpose(0, 1).reshape(world * local.shape[0], *local.shape[1:])  # [n, world, ...] -> i * world + r
    dist.gather(local, gather_list=None, dst=dst)
    return None


@torch.no_grad()
def sample_shard(gpt, vq, n: int, latent: int, num_classes: int, embed_dim: int = 8, **gen_kw):
    """One per-rank sampling pass of sample_c2i_ddp.py:128-140: labels -> generate -> decode_code."""
    from .generate import generate
    dev = next(gpt.parameters()).device
    c_indices = torch.randint(0, num_classes, (n,), device=dev)
    idx = generate(gpt, c_indices, latent * latent, **gen_kw)
    return vq.decode_code(idx, [n, embed_dim, latent, latent]), idx
