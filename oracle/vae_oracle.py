"""CPU ORACLE (test infrastructure, NOT product code) for the CogVideoX 3D causal VAE ENCODER -- the step before the DiT
(videotuna/models/cogvideo_hf/cogvideo_pl.py:792-813: ``encode_video`` -> ``first_stage_model.encode(x).latent_dist.sample() *
scaling_factor``).  Only ``tests/`` may import this file.

What it follows: the reference's in-tree twin of diffusers' ``AutoencoderKLCogVideoX`` encoder,
``ContextParallelEncoder3D`` (videotuna/models/cogvideo_sat/vae_modules/cp_enc_dec.py:779-907) with
  * causal convolution .... ContextParallelCausalConv3d :356-433 + _fake_cp_pass_from_previous_rank :228-273 (context-parallel
                            world 1: the first frame is replicated kernel_t - 1 times in front of t; zeros around h / w)
  * ResNet block .......... ContextParallelResnetBlock3D :681-777 (GroupNorm(32, eps 1e-6) -> swish -> conv, twice; 1x1x1
                            `nin_shortcut` when the channel count changes; no time embedding in the encoder, dropout 0)
  * downsample ............ DownSample3D :625-678 (compress_time: frame 0 kept, frames 1.. avg-pooled in pairs; then zero line /
                            column at the bottom / right and Conv2d(3, stride 2) per frame)
  * encoder walk .......... conv_in, levels x num_res_blocks (+ downsample except at the last level; temporal compression at the
                            first log2(temporal_compress_times) levels), mid.block_1, mid.block_2, norm_out, swish, conv_out
CogVideoX-2B/5B: ch 128, ch_mult (1,2,2,4), 3 ResNet blocks per level, z_channels 16 (double_z: 32 output channels = mean | logvar),
temporal_compress_times 4.

PARITY STATUS: pinned.  tests/golden/vae_encoder_tiny.npz holds seeded weights, an input clip and the output of the reference's own
``ContextParallelEncoder3D`` imported in the build container (tests/golden/make_golden_vae.py; beartype / sgm.util / SafeConv3d
stubbed, a 1-rank gloo group for the context-parallel helpers); tests/test_oracle_golden.py checks this restatement against it.
Parameter names are the SAT twin's (``down.0.block.0.conv1.conv.weight`` ...); the diffusers checkpoint names
(``encoder.down_blocks.0.resnets.0.conv1.conv.weight`` ...) could not be checked offline.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VaeEncConfig:
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 2, 4)
    num_res_blocks: int = 3
    in_channels: int = 3
    z_channels: int = 16
    double_z: bool = True
    temporal_compress_times: int = 4


def tiny_config(**kw) -> VaeEncConfig:
    base = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, z_channels=4, temporal_compress_times=4)
    base.update(kw)
    return VaeEncConfig(**base)


def swish(x):
    return x * torch.sigmoid(x)


def causal_conv3d(x, w, b):
    """x [B, C, T, H, W]; kernel (kt, 3, 3): kt - 1 copies of the first frame in front of t, zeros around h / w"""
    kt = w.shape[2]
    if kt > 1:
        x = torch.cat([x[:, :, :1]] * (kt - 1) + [x], dim=2)
    x = F.pad(x, (1, 1, 1, 1))
    return F.conv3d(x, w, b)


def group_norm(x, w, b, frame_batch=None, t_in=0):
    """GroupNorm(32, eps 1e-6) over the whole clip (the in-tree twin; frame_batch None), or with the statistics of every frame batch of
    diffusers 0.32.2 AutoencoderKLCogVideoX._encode (frame_batch 8: first batch 8 + 1 frames, then 8 at a time; restated from the
    published source of a dependency that is not in this image -- parity unpinned for this mode)"""
    T = x.shape[2]
    stride = max(1, (t_in - 1) // max(T - 1, 1)) if T > 1 else 1
    if frame_batch is None or t_in <= frame_batch + 1 or (t_in - 1) % frame_batch or frame_batch % stride:
        return F.group_norm(x, 32, w, b, 1e-6)
    first, rest = frame_batch // stride + 1, frame_batch // stride
    parts = [x[:, :, :first]] + list(torch.split(x[:, :, first:], rest, dim=2))
    return torch.cat([F.group_norm(p_, 32, w, b, 1e-6) for p_ in parts], dim=2)


def resnet_block(x, P, pre, cin, cout, frame_batch=None, t_in=0):
    h = group_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], frame_batch, t_in)
    h = causal_conv3d(swish(h), P[pre + "conv1.conv.weight"], P[pre + "conv1.conv.bias"])
    h = group_norm(h, P[pre + "norm2.weight"], P[pre + "norm2.bias"], frame_batch, t_in)
    h = causal_conv3d(swish(h), P[pre + "conv2.conv.weight"], P[pre + "conv2.conv.bias"])
    if cin != cout:
        x = F.conv3d(x, P[pre + "nin_shortcut.weight"], P[pre + "nin_shortcut.bias"])
    return x + h


def downsample(x, P, pre, compress_time, keep_first=None):
    """DownSample3D.forward (cp_enc_dec.py:640-676).  keep_first None: diffusers' rule (CogVideoXDownsample3D) -- odd frame count:
    the twin's rank-0 / fake_cp branch (:645-657, first frame kept, pairs after it); even: its other branch (:658-667, plain pairs)."""
    B, C, T, H, W = x.shape
    if compress_time and T > 1:
        xt = x.permute(0, 3, 4, 1, 2).reshape(B * H * W, C, T)
        if keep_first is None:
            keep_first = (T % 2 == 1)
        if keep_first:
            first, rest = xt[..., :1], xt[..., 1:]
            if rest.shape[-1] > 1:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            else:
                rest = rest[..., :0]
            xt = torch.cat([first, rest], dim=-1)
        else:
            xt = F.avg_pool1d(xt, kernel_size=2, stride=2)
        T = xt.shape[-1]
        x = xt.reshape(B, H, W, C, T).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1))
    x2 = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H + 1, W + 1)
    x2 = F.conv2d(x2, P[pre + "conv.weight"], P[pre + "conv.bias"], stride=2)
    return x2.reshape(B, T, -1, x2.shape[-2], x2.shape[-1]).permute(0, 2, 1, 3, 4)


def level_channels(cfg: VaeEncConfig):
    in_mult = (1,) + tuple(cfg.ch_mult)
    return [(cfg.ch * in_mult[i], cfg.ch * cfg.ch_mult[i]) for i in range(len(cfg.ch_mult))]


def init_params(cfg: VaeEncConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    P = {}

    def conv(name, cout, cin, *k):
        fan = cin
        for kk in k:
            fan *= kk
        P[name + ".weight"] = torch.randn(cout, cin, *k, generator=g) / fan ** 0.5
        P[name + ".bias"] = 0.1 * torch.randn(cout, generator=g)

    def norm(name, c):
        P[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        P[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    def block(pre, cin, cout):
        norm(pre + "norm1", cin); conv(pre + "conv1.conv", cout, cin, 3, 3, 3)
        norm(pre + "norm2", cout); conv(pre + "conv2.conv", cout, cout, 3, 3, 3)
        if cin != cout:
            conv(pre + "nin_shortcut", cout, cin, 1, 1, 1)

    conv("conv_in.conv", cfg.ch, cfg.in_channels, 3, 3, 3)
    chans = level_channels(cfg)
    for i, (cin, cout) in enumerate(chans):
        for j in range(cfg.num_res_blocks):
            block(f"down.{i}.block.{j}.", cin if j == 0 else cout, cout)
        if i != len(chans) - 1:
            conv(f"down.{i}.downsample.conv", cout, cout, 3, 3)
    top = chans[-1][1]
    block("mid.block_1.", top, top); block("mid.block_2.", top, top)
    norm("norm_out", top)
    conv("conv_out.conv", 2 * cfg.z_channels if cfg.double_z else cfg.z_channels, top, 3, 3, 3)
    return {k: v.to(dtype) for k, v in P.items()}


def encoder_forward(P: Dict[str, torch.Tensor], cfg: VaeEncConfig, x: torch.Tensor, frame_batch=None) -> torch.Tensor:
    """x [B, 3, T, H, W] -> moments [B, 2 * z_channels, T', H / 2^(L-1), W / 2^(L-1)]"""
    import math
    t_in = x.shape[2]
    tlevels = int(math.log2(cfg.temporal_compress_times))
    h = causal_conv3d(x, P["conv_in.conv.weight"], P["conv_in.conv.bias"])
    chans = level_channels(cfg)
    for i, (cin, cout) in enumerate(chans):
        for j in range(cfg.num_res_blocks):
            h = resnet_block(h, P, f"down.{i}.block.{j}.", cin if j == 0 else cout, cout, frame_batch, t_in)
        if i != len(chans) - 1:
            h = downsample(h, P, f"down.{i}.downsample.", i < tlevels)
    top = chans[-1][1]
    h = resnet_block(h, P, "mid.block_1.", top, top, frame_batch, t_in)
    h = resnet_block(h, P, "mid.block_2.", top, top, frame_batch, t_in)
    h = group_norm(h, P["norm_out.weight"], P["norm_out.bias"], frame_batch, t_in)
    return causal_conv3d(swish(h), P["conv_out.conv.weight"], P["conv_out.conv.bias"])


def sample_latent(moments: torch.Tensor, noise: torch.Tensor, scaling_factor: float) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample() * scaling_factor (cogvideo_pl.py:792-813): mean + exp(0.5 clamp(logvar, -30, 20)) * eps"""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return (mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise) * scaling_factor
