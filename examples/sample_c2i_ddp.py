#!/usr/bin/env python3
"""FID-sample generation across the GPUs of a node on the MI355X-native engine: the flag surface, sharding rule, seeding and
output of the reference's autoregressive/sample/sample_c2i_ddp.py (:38-157 main, :160-187 flags).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/sample_c2i_ddp.py \\
        --vq-ckpt vq_ds16_c2i.pt --gpt-ckpt c2i_L_384.pt --gpt-model GPT-L --image-size 384 --cfg-scale 2.0

Same as the reference: one process per GPU, per-rank seed `global_seed * world + rank` (:47), every iteration draws its labels
with `torch.randint` right before its generate() call (:128-133, so labels and sampling noise come out of the device generator
in the reference's order), bicubic resize to `--image-size-eval`, uint8 HWC, global image index `i * world + rank + total`
(:146), `<sample-dir>/<folder>.npz` with `arr_0` = the first `--num-fid-samples` images (:21-35).
Different on purpose: `--batches-per-chain` iterations share one decode chain and `--lanes` chains are kept in flight per GPU
(llamagen_amd.pipeline), the shards meet through ONE RCCL gather of uint8 images per iteration instead of PNG files on a shared
filesystem (`--png` still writes them from rank 0), every gathered batch is streamed to pinned host memory at once, and
`--compile` is accepted and ignored.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, VQ_models  # noqa: E402
from llamagen_amd import dist as ldist  # noqa: E402
from llamagen_amd.checkpoint import load_gpt_checkpoint, load_vq_checkpoint  # noqa: E402
from llamagen_amd.pipeline import SamplingPipeline  # noqa: E402
from llamagen_amd.postprocess import to_uint8_hwc  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--gpt-model", type=str, choices=list(GPT_models.keys()), default="GPT-B")
    p.add_argument("--gpt-ckpt", type=str, default=None)
    p.add_argument("--gpt-type", type=str, choices=["c2i", "t2i"], default="c2i", help="class-conditional or text-conditional")
    p.add_argument("--from-fsdp", action="store_true")
    p.add_argument("--cls-token-num", type=int, default=1, help="max token number of condition input")
    p.add_argument("--precision", type=str, default="bf16", choices=["none", "fp16", "bf16"])
    p.add_argument("--compile", action="store_true", default=True)
    p.add_argument("--vq-model", type=str, choices=list(VQ_models.keys()), default="VQ-16")
    p.add_argument("--vq-ckpt", type=str, default=None, help="ckpt path for vq model")
    p.add_argument("--codebook-size", type=int, default=16384, help="codebook size for vector quantization")
    p.add_argument("--codebook-embed-dim", type=int, default=8, help="codebook dimension for vector quantization")
    p.add_argument("--image-size", type=int, choices=[256, 384, 512], default=384)
    p.add_argument("--image-size-eval", type=int, choices=[256, 384, 512], default=256)
    p.add_argument("--downsample-size", type=int, choices=[8, 16], default=16)
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--cfg-scale", type=float, default=1.5)
    p.add_argument("--cfg-interval", type=float, default=-1)
    p.add_argument("--sample-dir", type=str, default="samples")
    p.add_argument("--per-proc-batch-size", type=int, default=32)
    p.add_argument("--num-fid-samples", type=int, default=50000)
    p.add_argument("--global-seed", type=int, default=0)
    p.add_argument("--top-k", type=int, default=0, help="top-k value to sample with")
    p.add_argument("--temperature", type=float, default=1.0, help="temperature value to sample with")
    p.add_argument("--top-p", type=float, default=1.0, help="top-p value to sample with")
    p.add_argument("--lanes", type=int, default=2, help="decode chains kept in flight per GPU")
    p.add_argument("--batches-per-chain", type=int, default=2, help="consecutive iterations (batches) that share one decode chain "
                                                                    "(llamagen_amd.pipeline; every image is what its own generate() call "
                                                                    "would produce from the same noise)")
    p.add_argument("--png", action="store_true", help="also write the individual .png files of the reference (rank 0)")
    return p


def main(args):
    if not torch.cuda.is_available():
        raise RuntimeError("sampling needs an AMD GPU: llamagen_amd has no CPU path")
    torch.set_grad_enabled(False)
    rank, local, world = ldist.init_from_env()
    dev = torch.device(f"cuda:{local % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    seed = ldist.rank_seed(args.global_seed, rank, world)
    torch.manual_seed(seed)
    print(f"Starting rank={rank}, seed={seed}, world_size={world}.")

    vq_model = VQ_models[args.vq_model](codebook_size=args.codebook_size, codebook_embed_dim=args.codebook_embed_dim).to(dev).eval()
    load_vq_checkpoint(vq_model, args.vq_ckpt)
    precision = {"none": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.precision]
    latent = args.image_size // args.downsample_size
    gpt_model = GPT_models[args.gpt_model](vocab_size=args.codebook_size, block_size=latent ** 2, num_classes=args.num_classes,
                                           cls_token_num=args.cls_token_num, model_type=args.gpt_type).to(device=dev, dtype=precision)
    load_gpt_checkpoint(gpt_model, args.gpt_ckpt, from_fsdp=args.from_fsdp)
    gpt_model.eval()

    ckpt_name = args.gpt_ckpt.split("/")[-2] if args.from_fsdp else os.path.basename(args.gpt_ckpt).replace(".pth", "").replace(".pt", "")
    folder = (f"{args.gpt_model.replace('/', '-')}-{ckpt_name}-size-{args.image_size}-size-{args.image_size_eval}-{args.vq_model}-"
              f"topk-{args.top_k}-topp-{args.top_p}-temperature-{args.temperature}-cfg-{args.cfg_scale}-seed-{args.global_seed}")
    out_dir = os.path.join(args.sample_dir, folder)
    if rank == 0:
        os.makedirs(out_dir if args.png else args.sample_dir, exist_ok=True)

    n = args.per_proc_batch_size
    global_batch = n * world
    total_samples = int(math.ceil(args.num_fid_samples / global_batch) * global_batch)
    iterations = total_samples // world // n
    if rank == 0:
        print(f"Total number of images that will be sampled: {total_samples} ({iterations} iterations of {n} per GPU)")

    def draw_labels():
        return torch.randint(0, args.num_classes, (n,), device=dev)

    # rank 0 keeps the result on the HOST: one pinned [total, H, W, 3] uint8 array that every iteration's gathered batch is
    # copied into as soon as it exists (non-blocking, on that lane's stream), so the GPU never holds more than the batches in
    # flight (50 000 x 256 x 256 x 3 = 9.8 GB would otherwise sit in HBM until the end)
    E = args.image_size_eval
    host = torch.empty((total_samples, E, E, 3), dtype=torch.uint8, pin_memory=True) if rank == 0 else None

    def finish(job_id, idx, img):  # on the lane's stream, as soon as decode_code() of that batch is enqueued
        g = ldist.gather_to_root(to_uint8_hwc(img, E))  # [world * n, H, W, 3] in global index order on rank 0, None elsewhere
        if g is not None:
            host[job_id * global_batch:(job_id + 1) * global_batch].copy_(g, non_blocking=True)  # iteration t = indices t * global_batch ..
        return None

    bpc = max(1, args.batches_per_chain) if args.gpt_type == "c2i" else 1
    chains = (iterations + bpc - 1) // bpc
    pipe = SamplingPipeline(gpt_model, vq_model, lanes=max(1, min(args.lanes, chains)), batches_per_chain=bpc)
    pipe.run([draw_labels] * iterations, latent ** 2, on_done=finish, cfg_scale=args.cfg_scale, cfg_interval=int(args.cfg_interval),
             temperature=args.temperature, top_k=args.top_k, top_p=args.top_p, sample_logits=True)
    torch.cuda.synchronize(dev)
    npz_path = None
    if rank == 0:
        arr = host.numpy()[: args.num_fid_samples]
        assert arr.shape == (min(args.num_fid_samples, total_samples), E, E, 3) and arr.dtype == np.uint8
        if args.png:
            from PIL import Image
            for i, a in enumerate(arr):
                Image.fromarray(a).save(f"{out_dir}/{i:06d}.png")
        npz_path = f"{out_dir}.npz"
        np.savez(npz_path, arr_0=arr)
        print(f"Saved .npz file to {npz_path} [shape={arr.shape}].")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return npz_path


if __name__ == "__main__":
    main(build_parser().parse_args())
