"""What the reference does with the decoded batch right after the hot path (sample_c2i_ddp.py:141-148,
21-35): optional bicubic resize to the evaluation size, [-1, 1] float -> uint8 HWC, and the `.npz` pack of
all samples that the FID evaluator reads.  The arithmetic runs in liblgen_hip.so (post_ops.hip); the PNG
round trip of the reference (PIL, one file per image on a shared filesystem) is replaced by keeping the
uint8 arrays in memory -- after the single RCCL gather they are already on rank 0."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib as L


@torch.no_grad()
def resize_bicubic(samples: torch.Tensor, size: int) -> torch.Tensor:
    """F.interpolate(samples, size=(size, size), mode='bicubic') -- sample_c2i_ddp.py:141-142."""
    if samples.device.type != "cuda":
        raise RuntimeError("llamagen_amd.postprocess runs only on an AMD GPU through the HIP library (no CPU fallback)")
    B, C, H, W = samples.shape
    x = samples.float().contiguous()
    out = torch.empty(B, C, size, size, dtype=torch.float32, device=x.device)
    L.check(L.lib().lgen_resize_bicubic(L.ptr(x), L.ptr(out), B * C, H, W, size, size, L.stream()), "resize_bicubic")
    return out


@torch.no_grad()
def to_uint8_hwc(samples: torch.Tensor, image_size_eval: Optional[int] = None) -> torch.Tensor:
    """sample_c2i_ddp.py:141-143: [resize ->] clamp(127.5 x + 128, 0, 255) -> uint8 [B, H, W, C] (on the device)."""
    if samples.device.type != "cuda":
        raise RuntimeError("llamagen_amd.postprocess runs only on an AMD GPU through the HIP library (no CPU fallback)")
    if image_size_eval is not None and image_size_eval != samples.shape[-1]:
        samples = resize_bicubic(samples, image_size_eval)
    B, C, H, W = samples.shape
    x = samples.float().contiguous()
    out = torch.empty(B, H, W, C, dtype=torch.uint8, device=x.device)
    L.check(L.lib().lgen_to_uint8_hwc(L.ptr(x), L.ptr(out), B, C, H, W, L.stream()), "to_uint8_hwc")
    return out


def save_image_grid(samples: torch.Tensor, path: str, nrow: int = 4, padding: int = 2) -> str:
    """What sample_c2i.py:95 does with torchvision's `save_image(samples, path, nrow=4, normalize=True, value_range=(-1, 1))`:
    [-1, 1] -> [0, 1] clamp, a grid `nrow` images wide with `padding` black pixels between them, PNG via PIL.  The
    clamp / scale / uint8 conversion of every image runs in the HIP library (lgen_to_uint8_hwc computes clamp(127.5 x + 128),
    within half a grey level of torchvision's round(255 (x + 1) / 2)); only the paste and the file write are host work."""
    from PIL import Image
    u8 = to_uint8_hwc(samples).cpu().numpy()
    n, h, w, c = u8.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = np.zeros((rows * (h + padding) + padding, cols * (w + padding) + padding, c), dtype=np.uint8)
    for i in range(n):
        y, x = (i // cols) * (h + padding) + padding, (i % cols) * (w + padding) + padding
        grid[y:y + h, x:x + w] = u8[i]
    Image.fromarray(grid).save(path)
    return path


def save_npz(samples_uint8: torch.Tensor, npz_path: str, num: Optional[int] = None) -> str:
    """create_npz_from_sample_folder (sample_c2i_ddp.py:21-35) without the PNG detour: [N, H, W, 3] uint8 ->
    `arr_0` of an .npz, exactly what evaluations/c2i/evaluator.py reads."""
    arr = samples_uint8.cpu().numpy() if isinstance(samples_uint8, torch.Tensor) else np.asarray(samples_uint8)
    if num is not None:
        arr = arr[:num]
    assert arr.dtype == np.uint8 and arr.ndim == 4 and arr.shape[3] == 3, arr.shape
    np.savez(npz_path, arr_0=arr)
    return npz_path


@torch.no_grad()
def extract_codes(vq_model, x: torch.Tensor, ten_crop: bool = False) -> torch.Tensor:
    """The per-batch body of autoregressive/train/extract_codes_c2i.py:92-103: augment (ten-crop input
    [B, 10, 3, H, W], or horizontal flip of [B, 3, H, W]), VQModel.encode on the HIP path, ids reshaped to
    [B, num_aug, h*w] int64 (on the device)."""
    if ten_crop:
        x_all, num_aug = x.flatten(0, 1), 10
    else:
        x_all, num_aug = torch.cat([x, torch.flip(x, dims=[-1])]), 2
    _, _, (_, _, indices) = vq_model.encode(x_all)
    if ten_crop:
        return indices.reshape(x.shape[0], num_aug, -1)
    # extract_codes_c2i.py:103 reshapes the [orig..., flipped...] batch as (B, num_aug, -1); with its batch size of 1
    # that is (original, flipped) -- kept for B = 1 and made explicit (per image) for larger B
    return indices.reshape(num_aug, x.shape[0], -1).transpose(0, 1).contiguous()


def save_codes(codes: torch.Tensor, labels: torch.Tensor, code_path: str, dataset: str, image_size: int, index: int):
    """On-disk code format of extract_codes_c2i.py:105-111, read back by dataset/imagenet.py:34-50:
    `{code_path}/{dataset}{image_size}_codes/{index}.npy` int64 [1, num_aug, h*w] and
    `{code_path}/{dataset}{image_size}_labels/{index}.npy` int64 [1]."""
    import os
    cdir = os.path.join(code_path, f"{dataset}{image_size}_codes")
    ldir = os.path.join(code_path, f"{dataset}{image_size}_labels")
    os.makedirs(cdir, exist_ok=True)
    os.makedirs(ldir, exist_ok=True)
    c = codes.detach().cpu().numpy().astype(np.int64)
    y = labels.detach().cpu().numpy().astype(np.int64)
    assert c.ndim == 3 and c.shape[0] == 1 and y.shape == (1,), (c.shape, y.shape)
    np.save(os.path.join(cdir, f"{index}.npy"), c)
    np.save(os.path.join(ldir, f"{index}.npy"), y)
    return os.path.join(cdir, f"{index}.npy"), os.path.join(ldir, f"{index}.npy")
