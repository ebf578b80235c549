#!/usr/bin/env python3
"""Class-conditional sampling demo on the MI355X-native engine: the flag surface and call sequence of the reference's
autoregressive/sample/sample_c2i.py (:18-98 main, :101-123 flags), with the three model imports swapped for
`llamagen_amd` -- the "drop-in" claim of INTEGRATION.md, executed.

    python examples/sample_c2i.py --vq-ckpt vq_ds16_c2i.pt --gpt-ckpt c2i_L_384.pt --gpt-model GPT-L --image-size 384

Differences that are deliberate: `--compile` is accepted and ignored (the decode step is always one captured hipGraph),
all three `--precision` choices run on the HIP kernels (bf16, fp16, none = fp32), the image grid is written with PIL (torchvision is
not a dependency) and `main()` returns the tensors so that a test can hold them against the oracle.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, VQ_models, generate  # noqa: E402
from llamagen_amd.checkpoint import load_gpt_checkpoint, load_vq_checkpoint  # noqa: E402
from llamagen_amd.postprocess import save_image_grid  # noqa: E402

DEMO_LABELS = [207, 360, 387, 974, 88, 979, 417, 279]  # the reference demo's ImageNet classes (sample_c2i.py:77)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--gpt-model", type=str, choices=list(GPT_models.keys()), default="GPT-B")
    p.add_argument("--gpt-ckpt", type=str, default=None)
    p.add_argument("--gpt-type", type=str, choices=["c2i", "t2i"], default="c2i", help="class-conditional or text-conditional")
    p.add_argument("--from-fsdp", action="store_true")
    p.add_argument("--cls-token-num", type=int, default=1, help="max token number of condition input")
    p.add_argument("--precision", type=str, default="bf16", choices=["none", "fp16", "bf16"])
    p.add_argument("--compile", action="store_true", default=False)
    p.add_argument("--vq-model", type=str, choices=list(VQ_models.keys()), default="VQ-16")
    p.add_argument("--vq-ckpt", type=str, default=None, help="ckpt path for vq model")
    p.add_argument("--codebook-size", type=int, default=16384, help="codebook size for vector quantization")
    p.add_argument("--codebook-embed-dim", type=int, default=8, help="codebook dimension for vector quantization")
    p.add_argument("--image-size", type=int, choices=[256, 384, 512], default=384)
    p.add_argument("--downsample-size", type=int, choices=[8, 16], default=16)
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--cfg-scale", type=float, default=4.0)
    p.add_argument("--cfg-interval", type=float, default=-1)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--top-k", type=int, default=2000, help="top-k value to sample with")
    p.add_argument("--temperature", type=float, default=1.0, help="temperature value to sample with")
    p.add_argument("--top-p", type=float, default=1.0, help="top-p value to sample with")
    p.add_argument("--out", type=str, default=None, help="image grid path (default sample_<gpt-type>.png)")
    return p


def main(args, class_labels=None):
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        raise RuntimeError("examples/sample_c2i.py needs an AMD GPU: llamagen_amd has no CPU path")
    device = "cuda"

    vq_model = VQ_models[args.vq_model](codebook_size=args.codebook_size, codebook_embed_dim=args.codebook_embed_dim)
    vq_model.to(device).eval()
    load_vq_checkpoint(vq_model, args.vq_ckpt)
    print("image tokenizer is loaded")

    precision = {"none": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.precision]
    latent_size = args.image_size // args.downsample_size
    gpt_model = GPT_models[args.gpt_model](vocab_size=args.codebook_size, block_size=latent_size ** 2, num_classes=args.num_classes,
                                           cls_token_num=args.cls_token_num, model_type=args.gpt_type).to(device=device, dtype=precision)
    missing, unexpected = load_gpt_checkpoint(gpt_model, args.gpt_ckpt, from_fsdp=args.from_fsdp)
    gpt_model.eval()
    print(f"gpt model is loaded ({len(missing)} keys missing, {len(unexpected)} unexpected)")
    print("--compile ignored: the decode step is a captured hipGraph" if args.compile else "no need to compile model in demo")

    c_indices = torch.tensor(DEMO_LABELS if class_labels is None else class_labels, device=device)
    qzshape = [len(c_indices), args.codebook_embed_dim, latent_size, latent_size]

    t1 = time.time()
    index_sample = generate(gpt_model, c_indices, latent_size ** 2, cfg_scale=args.cfg_scale, cfg_interval=int(args.cfg_interval),
                            temperature=args.temperature, top_k=args.top_k, top_p=args.top_p, sample_logits=True)
    torch.cuda.synchronize()  # the reference's bracket has no device sync; this one does
    sampling_time = time.time() - t1
    print(f"gpt sampling takes about {sampling_time:.2f} seconds.")

    t2 = time.time()
    samples = vq_model.decode_code(index_sample, qzshape)  # output value is between [-1, 1]
    torch.cuda.synchronize()
    decoder_time = time.time() - t2
    print(f"decoder takes about {decoder_time:.2f} seconds.")

    out = args.out or f"sample_{args.gpt_type}.png"
    save_image_grid(samples, out, nrow=4)
    print(f"image is saved to {out}")
    return dict(index_sample=index_sample, samples=samples, sampling_time=sampling_time, decoder_time=decoder_time, path=out,
                missing=missing, unexpected=unexpected)


if __name__ == "__main__":
    main(build_parser().parse_args())
