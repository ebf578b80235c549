#!/usr/bin/env python3
"""Text-conditional sampling demo on the MI355X-native engine: the flag surface and call sequence of the reference's
autoregressive/sample/sample_t2i.py (:19-127 main, :131-153 flags) from the T5 features onward.

    python examples/sample_t2i.py --vq-ckpt vq_ds16_t2i.pt --gpt-ckpt t2i_XL_stage2_512.pt --gpt-model GPT-XL --image-size 512 \\
        --t5-feature-path prompts_t5.npz

The T5 text encoder is outside the sampling hot path (out of scope) and not part of this package, so the
caption features come from a file instead of `T5Embedder.get_text_embeddings` (sample_t2i.py:77-93): an .npz / .pt with
`caption_embs` [B, L, 2048] (`--t5-feature-dim`) and `emb_masks` [B, L] (1 = valid token, RIGHT-padded as the tokenizer emits
them; L <= --t5-feature-max-len, shorter inputs are padded).  Everything after that is the reference's sequence: the naive
left-padding rotation (:95-107, `--no-left-padding` keeps the right-padded layout), `c_indices = embs * masks[:, :, None]`
(:108), `generate(gpt, c_indices, latent^2, c_emb_masks, cfg_scale, temperature, top_k, top_p)` (:112-118), `decode_code`
(:123) and the image grid (:127).  `--compile` is accepted and ignored (the decode step is a captured hipGraph).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, VQ_models, generate  # noqa: E402
from llamagen_amd.checkpoint import load_gpt_checkpoint, load_vq_checkpoint  # noqa: E402
from llamagen_amd.postprocess import save_image_grid  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--t5-feature-path", type=str, required=True, help=".npz / .pt holding caption_embs [B, L, D] and emb_masks [B, L]")
    p.add_argument("--t5-feature-max-len", type=int, default=120)
    p.add_argument("--t5-feature-dim", type=int, default=2048)
    p.add_argument("--no-left-padding", action="store_true", default=False)
    p.add_argument("--gpt-model", type=str, choices=list(GPT_models.keys()), default="GPT-XL")
    p.add_argument("--gpt-ckpt", type=str, default=None)
    p.add_argument("--gpt-type", type=str, choices=["c2i", "t2i"], default="t2i", help="class->image or text->image")
    p.add_argument("--cls-token-num", type=int, default=120, help="max token number of condition input")
    p.add_argument("--precision", type=str, default="bf16", choices=["none", "fp16", "bf16"])
    p.add_argument("--compile", action="store_true", default=False)
    p.add_argument("--vq-model", type=str, choices=list(VQ_models.keys()), default="VQ-16")
    p.add_argument("--vq-ckpt", type=str, default=None, help="ckpt path for vq model")
    p.add_argument("--codebook-size", type=int, default=16384, help="codebook size for vector quantization")
    p.add_argument("--codebook-embed-dim", type=int, default=8, help="codebook dimension for vector quantization")
    p.add_argument("--image-size", type=int, choices=[256, 384, 512], default=512)
    p.add_argument("--downsample-size", type=int, choices=[8, 16], default=16)
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--cfg-scale", type=float, default=7.5)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--top-k", type=int, default=1000, help="top-k value to sample with")
    p.add_argument("--temperature", type=float, default=1.0, help="temperature value to sample with")
    p.add_argument("--top-p", type=float, default=1.0, help="top-p value to sample with")
    p.add_argument("--out", type=str, default=None, help="image grid path (default sample_<gpt-type>.png)")
    return p


def load_t5_features(path, max_len, dim):
    """(caption_embs fp32 [B, max_len, dim], emb_masks int64 [B, max_len]) as T5Embedder.get_text_embeddings returns them
    (language/t5.py:96-113: padding='max_length', right-padded, attention_mask 1 on valid tokens)."""
    obj = np.load(path) if str(path).endswith(".npz") else torch.load(path, map_location="cpu")
    embs = torch.as_tensor(np.asarray(obj["caption_embs"])).float()
    masks = torch.as_tensor(np.asarray(obj["emb_masks"])).long()
    if embs.dim() != 3 or embs.shape[2] != dim or masks.shape != embs.shape[:2] or embs.shape[1] > max_len:
        raise ValueError(f"expected caption_embs [B, L <= {max_len}, {dim}] and emb_masks [B, L], got {tuple(embs.shape)} / {tuple(masks.shape)}")
    pad = max_len - embs.shape[1]
    if pad:
        embs = torch.cat([embs, torch.zeros(embs.shape[0], pad, dim)], dim=1)
        masks = torch.cat([masks, torch.zeros(masks.shape[0], pad, dtype=masks.dtype)], dim=1)
    return embs, masks


def left_pad(caption_embs, emb_masks):
    """sample_t2i.py:95-107, "a naive way to implement left-padding": the mask is flipped, every caption is rotated so that its
    valid tokens end at position L - 1."""
    new_masks = torch.flip(emb_masks, dims=[-1])
    out = []
    for emb, mask in zip(caption_embs, emb_masks):
        valid = int(mask.sum().item())
        out.append(torch.cat([emb[valid:], emb[:valid]]))
    return torch.stack(out), new_masks


def main(args):
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        raise RuntimeError("examples/sample_t2i.py needs an AMD GPU: llamagen_amd has no CPU path")
    device = "cuda"

    vq_model = VQ_models[args.vq_model](codebook_size=args.codebook_size, codebook_embed_dim=args.codebook_embed_dim)
    vq_model.to(device).eval()
    load_vq_checkpoint(vq_model, args.vq_ckpt)
    print("image tokenizer is loaded")

    precision = {"none": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.precision]
    latent_size = args.image_size // args.downsample_size
    gpt_model = GPT_models[args.gpt_model](block_size=latent_size ** 2, cls_token_num=args.cls_token_num,
                                           model_type=args.gpt_type).to(device=device, dtype=precision)
    missing, unexpected = load_gpt_checkpoint(gpt_model, args.gpt_ckpt)
    gpt_model.eval()
    print(f"gpt model is loaded ({len(missing)} keys missing, {len(unexpected)} unexpected)")
    print("--compile ignored: the decode step is a captured hipGraph" if args.compile else "no need to compile model in demo")

    caption_embs, emb_masks = load_t5_features(args.t5_feature_path, args.t5_feature_max_len, args.t5_feature_dim)
    if not args.no_left_padding:
        print("processing left-padding...")
        for idx, m in enumerate(emb_masks):
            print(f"  prompt {idx} token len: {int(m.sum().item())}")
        new_caption_embs, new_emb_masks = left_pad(caption_embs, emb_masks)
    else:
        new_caption_embs, new_emb_masks = caption_embs, emb_masks
    c_indices = (new_caption_embs * new_emb_masks[:, :, None]).to(device=device, dtype=precision)
    c_emb_masks = new_emb_masks.to(device)

    qzshape = [len(c_indices), args.codebook_embed_dim, latent_size, latent_size]
    t1 = time.time()
    index_sample = generate(gpt_model, c_indices, latent_size ** 2, c_emb_masks, cfg_scale=args.cfg_scale,
                            temperature=args.temperature, top_k=args.top_k, top_p=args.top_p, sample_logits=True)
    torch.cuda.synchronize()
    sampling_time = time.time() - t1
    print(f"Full sampling takes about {sampling_time:.2f} seconds.")

    t2 = time.time()
    samples = vq_model.decode_code(index_sample, qzshape)  # output value is between [-1, 1]
    torch.cuda.synchronize()
    decoder_time = time.time() - t2
    print(f"decoder takes about {decoder_time:.2f} seconds.")

    out = args.out or f"sample_{args.gpt_type}.png"
    save_image_grid(samples, out, nrow=4)
    print(f"image is saved to {out}")
    return dict(index_sample=index_sample, samples=samples, c_indices=c_indices, c_emb_masks=c_emb_masks, sampling_time=sampling_time,
                decoder_time=decoder_time, path=out, missing=missing, unexpected=unexpected)


if __name__ == "__main__":
    main(build_parser().parse_args())
