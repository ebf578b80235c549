"""Host-side mirror of the reference VQ-VAE tokenizer (tokenizer/tokenizer_image/vq_model.py).

Keeps `ModelArgs`, `VQModel.decode_code / decode / encode`, the `VQ_models` registry and the
state_dict key names; the arithmetic -- codebook gather + post_quant_conv,
GroupNorm statistics, implicit-GEMM 3x3/1x1 convolutions on MFMA with fused
GroupNorm-apply + swish prologue / nearest-2x upsample indexing / residual epilogue, the
single-head AttnBlock and the 16384-entry codebook argmin -- runs in the HIP library via
`VQEngine` (llamagen_amd/vq_engine.py).  There is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class ModelArgs:
    """vq_model.py:12-24"""
    codebook_size: int = 16384
    codebook_embed_dim: int = 8
    codebook_l2_norm: bool = True
    codebook_show_usage: bool = True
    commit_loss_beta: float = 0.25
    entropy_loss_ratio: float = 0.0
    encoder_ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    decoder_ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    z_channels: int = 256
    dropout_p: float = 0.0


class _Conv(nn.Module):
    """nn.Conv2d parameter holder ([Cout, Cin, kh, kw] + bias), default-PyTorch-like init."""

    def __init__(self, cin: int, cout: int, k: int, stride: int = 1, padding: int = 0):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, k, stride, padding
        bound = 1.0 / (cin * k * k) ** 0.5
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).uniform_(-bound, bound), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound), requires_grad=False)


class _GN(nn.Module):
    """nn.GroupNorm(32, C, eps=1e-6, affine=True) holder, vq_model.py:359-362."""

    def __init__(self, c: int):
        super().__init__()
        self.num_groups, self.eps = 32, 1e-6
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)


class _Res(nn.Module):
    """ResnetBlock parameters, vq_model.py:279-297."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = _GN(cin)
        self.conv1 = _Conv(cin, cout, 3, 1, 1)
        self.norm2 = _GN(cout)
        self.conv2 = _Conv(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = _Conv(cin, cout, 1)


class _Attn(nn.Module):
    """AttnBlock parameters, vq_model.py:317-324."""

    def __init__(self, c: int):
        super().__init__()
        self.norm = _GN(c)
        self.q = _Conv(c, c, 1)
        self.k = _Conv(c, c, 1)
        self.v = _Conv(c, c, 1)
        self.proj_out = _Conv(c, c, 1)


class _Resample(nn.Module):
    def __init__(self, c: int, stride: int):
        super().__init__()
        self.conv = _Conv(c, c, 3, stride, 1 if stride == 1 else 0)


class Encoder(nn.Module):
    """Parameter layout of vq_model.py:64-103."""

    def __init__(self, in_channels=3, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, z_channels=256):
        super().__init__()
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.conv_in = _Conv(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.conv_blocks = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            blk = nn.Module()
            res, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                res.append(_Res(block_in, block_out))
                block_in = block_out
                if i_level == self.num_resolutions - 1:
                    attn.append(_Attn(block_in))
            blk.res, blk.attn = res, attn
            if i_level != self.num_resolutions - 1:
                blk.downsample = _Resample(block_in, 2)
            self.conv_blocks.append(blk)
        self.mid = nn.ModuleList([_Res(block_in, block_in), _Attn(block_in), _Res(block_in, block_in)])
        self.norm_out = _GN(block_in)
        self.conv_out = _Conv(block_in, z_channels, 3, 1, 1)


class Decoder(nn.Module):
    """Parameter layout of vq_model.py:128-167 (conv_blocks stored in execution order)."""

    def __init__(self, z_channels=256, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, out_channels=3):
        super().__init__()
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = _Conv(z_channels, block_in, 3, 1, 1)
        self.mid = nn.ModuleList([_Res(block_in, block_in), _Attn(block_in), _Res(block_in, block_in)])
        self.conv_blocks = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            blk = nn.Module()
            res, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                res.append(_Res(block_in, block_out))
                block_in = block_out
                if i_level == self.num_resolutions - 1:
                    attn.append(_Attn(block_in))
            blk.res, blk.attn = res, attn
            if i_level != 0:
                blk.upsample = _Resample(block_in, 1)
            self.conv_blocks.append(blk)
        self.norm_out = _GN(block_in)
        self.conv_out = _Conv(block_in, out_channels, 3, 1, 1)

    @property
    def last_layer(self):
        return self.conv_out.weight


class VectorQuantizer(nn.Module):
    """vq_model.py:197-213 parameter layout; lookups/argmin run in HIP (VQEngine)."""

    def __init__(self, n_e, e_dim, beta, entropy_loss_ratio, l2_norm, show_usage):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.entropy_loss_ratio, self.l2_norm, self.show_usage = entropy_loss_ratio, l2_norm, show_usage
        self.embedding = nn.Module()
        w = torch.empty(n_e, e_dim).uniform_(-1.0 / n_e, 1.0 / n_e)
        if l2_norm:
            w = F.normalize(w, p=2, dim=-1)
        self.embedding.weight = nn.Parameter(w, requires_grad=False)
        if show_usage:
            self.register_buffer("codebook_used", torch.zeros(65536))


class VQModel(nn.Module):
    """Drop-in for vq_model.py:28-60 on the inference path."""

    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.encoder = Encoder(ch_mult=config.encoder_ch_mult, z_channels=config.z_channels)
        self.decoder = Decoder(ch_mult=config.decoder_ch_mult, z_channels=config.z_channels)
        self.quantize = VectorQuantizer(config.codebook_size, config.codebook_embed_dim, config.commit_loss_beta,
                                        config.entropy_loss_ratio, config.codebook_l2_norm, config.codebook_show_usage)
        self.quant_conv = _Conv(config.z_channels, config.codebook_embed_dim, 1)
        self.post_quant_conv = _Conv(config.codebook_embed_dim, config.z_channels, 1)
        self._engine = None

    def _eng(self):
        if self.post_quant_conv.weight.device.type != "cuda":
            raise RuntimeError("llamagen_amd.VQModel runs only on an AMD GPU through the HIP library (no CPU fallback)")
        from .vq_engine import VQEngine
        if self._engine is None or not self._engine.compatible(self):
            self._engine = VQEngine(self)
        return self._engine

    @torch.no_grad()
    def decode_code(self, code_b, shape=None, channel_first=True):
        """vq_model.py:52-55.  code_b: integer ids [B, h*w] (any int dtype); shape =
        [B, C, h, w] (channel_first) -> fp32 [B, 3, 16h/8h, ...] in ~[-1, 1]."""
        return self._eng().decode_code(code_b, shape, channel_first)

    @torch.no_grad()
    def decode(self, quant):
        """vq_model.py:47-50 (quant: fp32 [B, C_e, h, w])."""
        return self._eng().decode(quant)

    @torch.no_grad()
    def quantize_indices(self, z):
        """Nearest-codebook-entry ids of latents z [B, C_e, h, w] (eval path of
        VectorQuantizer.forward, vq_model.py:215-232) -> int64 [B*h*w]."""
        return self._eng().argmin(z)

    @torch.no_grad()
    def encode(self, x):
        """vq_model.py:41-45 (eval): Encoder -> quant_conv -> VectorQuantizer.forward, all on the HIP kernels
        -> (quant [B, e_dim, h, w], (None, None, None, 0), (None, None, indices int64 [B*h*w]))."""
        return self._eng().encode(x)

    @torch.no_grad()
    def forward(self, input):
        """vq_model.py:57-60 in eval mode (the form reconstruction_vq_ddp.py / vq_demo.py call): encode -> decode,
        returns (reconstruction [B, 3, H, W], emb_loss tuple).  Train mode (losses, straight-through gradient) is
        outside the sampling hot path."""
        if self.training:
            raise NotImplementedError("train-mode VQModel.forward (losses + backward) is outside the sampling hot path")
        quant, diff, _ = self.encode(input)
        return self.decode(quant), diff


def VQ_8(**kwargs):
    return VQModel(ModelArgs(encoder_ch_mult=[1, 2, 2, 4], decoder_ch_mult=[1, 2, 2, 4], **kwargs))


def VQ_16(**kwargs):
    return VQModel(ModelArgs(encoder_ch_mult=[1, 1, 2, 2, 4], decoder_ch_mult=[1, 1, 2, 2, 4], **kwargs))


VQ_models = {"VQ-16": VQ_16, "VQ-8": VQ_8}
