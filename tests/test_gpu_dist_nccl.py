"""N-rank RCCL test of the sharded sampling pass (llamagen_amd/dist.py): one process per GPU, per-rank seed, no
collective during generation, ONE uint8 gather at the end in the reference's interleaved order
(sample_c2i_ddp.py:44-50, 128-149).  Needs >= 2 GPUs on the node: skipped on the 1-GPU development box, exercised by
the driver's multi-GPU run.  (The same logic runs on CPU/gloo in tests/test_dist_gloo.py.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from llamagen_amd import VQ_models
    from llamagen_amd.dist import gather_to_root, init_from_env, rank_seed, sample_shard
    from llamagen_amd.gpt import ModelArgs, Transformer
    from llamagen_amd.postprocess import to_uint8_hwc
    from llamagen_amd.testing import synth_for_module
    r, local, w = init_from_env("nccl")
    dev = torch.device(f"cuda:{local}")
    kw = dict(n_layer=2, n_head=4, dim=256, vocab_size=1024, block_size=16, num_classes=10, cls_token_num=1, model_type="c2i")
    m = Transformer(ModelArgs(**kw))
    m.load_state_dict(synth_for_module(m, seed=1, lin_std=0.05), strict=False)  # same weights on every rank
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    vq = VQ_models["VQ-16"](codebook_size=1024, codebook_embed_dim=8)
    vq.load_state_dict(synth_for_module(vq, seed=3))
    vq = vq.to(dev).eval()
    torch.manual_seed(rank_seed(5, r, w))
    img, idx = sample_shard(m, vq, 3, 4, 10, cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=100, top_p=1.0, sample_logits=True)
    u8 = to_uint8_hwc(img)
    out = gather_to_root(u8)
    ids = gather_to_root(idx)
    dist.barrier()
    if r == 0:
        q.put((out.cpu(), ids.cpu()))
    else:
        assert out is None
        q.put((u8.cpu(), idx.cpu()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_sampling_gathers_in_reference_order(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs on this node")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    qs = [ctx.SimpleQueue() for _ in range(world)]
    procs = [ctx.Process(target=_worker, args=(r, world, port, qs[r])) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get() for q in qs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    out, ids = res[0]
    assert out.dtype == torch.uint8 and tuple(out.shape) == (3 * world, 64, 64, 3) and tuple(ids.shape) == (3 * world, 16)
    for r in range(1, world):  # rank r's i-th image sits at global index i * world + r
        u8, idx = res[r]
        for i in range(3):
            assert torch.equal(out[i * world + r], u8[i]) and torch.equal(ids[i * world + r], idx[i])
    assert not torch.equal(ids[0], ids[1])  # per-rank seeds: different samples
