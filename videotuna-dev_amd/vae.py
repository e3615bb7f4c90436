"""CogVideoX 3D causal VAE -- the ENCODER, on the vt355 kernels (the step before the DiT).

What the reference runs per sample before the denoiser (videotuna/models/cogvideo_hf/cogvideo_pl.py:792-806):
``self.vae.encode(video).latent_dist.sample() * self.vae.config.scaling_factor`` with diffusers' ``AutoencoderKLCogVideoX``.
This module mirrors that surface (``.encode(x).latent_dist.sample()``, ``.config.scaling_factor``) for the encoder half; the
network is the one the reference holds in-tree as ``ContextParallelEncoder3D``
(videotuna/models/cogvideo_sat/vae_modules/cp_enc_dec.py:779-907), same module tree and parameter names, so that twin's
state dict loads directly.  The diffusers checkpoint uses other names (``encoder.down_blocks.N.resnets.M...``) that could not be
checked offline: no mapping is shipped yet.  Forward only (the VAE is frozen); the decoder is not built.

Layout: activations are channels-last ``[B, T, H, W, C]`` bf16 from the first convolution to the last -- the reference's
``b c t h w <-> (b t) c h w <-> (b h w) c t`` rearranges disappear.  Kernels: ``vt_causal_conv3d_cl`` (implicit-GEMM causal 3x3x3
convolution, optional residual add), ``vt_groupnorm_silu_cl``, ``vt_gemm_bf16`` (the 1x1x1 shortcut), ``vt_temporal_pool_cl``,
``vt_downsample_conv2d_cl``.  The RGB input is stored with 8 channels per position (one 16-byte chunk) and the first convolution
(``vt_causal_conv3d_in8_cl``) packs 8 taps into a K-tile: 4 K-tiles instead of 27.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import BF16


class _Conv(nn.Module):
    """holder with torch's parameter layout: weight [Cout, Cin, *k], bias [Cout]"""

    def __init__(self, cin, cout, *k):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, *k, dtype=BF16), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout, dtype=BF16), requires_grad=False)


class _CausalConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(cin, cout, 3, 3, 3)


class _Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c, dtype=BF16), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c, dtype=BF16), requires_grad=False)


class _ResBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1, self.conv1 = _Norm(cin), _CausalConv(cin, cout)
        self.norm2, self.conv2 = _Norm(cout), _CausalConv(cout, cout)
        if cin != cout:
            self.nin_shortcut = _Conv(cin, cout, 1, 1, 1)


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv(c, c, 3, 3)


class _Level(nn.Module):
    pass


class DiagonalGaussianDistribution:
    """moments [B, 2z, T, H, W] -> mean | logvar (clamped to [-30, 20]); sample() = mean + std * eps"""

    def __init__(self, moments: torch.Tensor):
        self.mean, logvar = torch.chunk(moments.float(), 2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self) -> torch.Tensor:
        return self.mean


class CogVideoXVaeEncoder(nn.Module):
    def __init__(self, ch: int = 128, ch_mult: Tuple[int, ...] = (1, 2, 2, 4), num_res_blocks: int = 3, in_channels: int = 3,
                 z_channels: int = 16, double_z: bool = True, temporal_compress_times: int = 4, scaling_factor: float = 1.15258426,
                 num_sample_frames_batch_size: Optional[int] = None, **unused):
        super().__init__()
        # None: GroupNorm statistics over the whole clip -- the in-tree twin (ContextParallelGroupNorm gathers every rank's frames), what
        # the oracle is pinned to.  8: the frame batching of diffusers 0.32.2 ``AutoencoderKLCogVideoX._encode`` (a dependency that is
        # not in this image: restated from its published source, parity unpinned) -- a 8n+1-frame clip is encoded as 9 frames, then 8 at a
        # time with the causal-convolution cache carried over, so only the GroupNorm statistics see the boundaries: the first 9 frames
        # encode exactly as a 9-frame clip (tests/test_model_gpu.py, causality property).
        self.frame_batch = num_sample_frames_batch_size
        self._t_in = 0
        if ch % 64 or in_channels > 8:
            raise ValueError("ch must be a multiple of 64 (one K-tile of the convolution kernel) and in_channels <= 8")
        self.config = SimpleNamespace(ch=ch, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks, in_channels=in_channels,
                                      z_channels=z_channels, double_z=double_z, temporal_compress_times=temporal_compress_times,
                                      scaling_factor=scaling_factor)
        self.temporal_levels = int(math.log2(temporal_compress_times))
        self.conv_in = _CausalConv(in_channels, ch)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i in range(len(ch_mult)):
            lvl = _Level()
            cin, cout = ch * in_mult[i], ch * ch_mult[i]
            lvl.block = nn.ModuleList([_ResBlock(cin if j == 0 else cout, cout) for j in range(num_res_blocks)])
            if i != len(ch_mult) - 1:
                lvl.downsample = _Down(cout)
            self.down.append(lvl)
        top = ch * ch_mult[-1]
        self.mid = nn.Module()
        self.mid.block_1, self.mid.block_2 = _ResBlock(top, top), _ResBlock(top, top)
        self.norm_out = _Norm(top)
        self.conv_out = _CausalConv(top, 2 * z_channels if double_z else z_channels)
        self._packed = None

    # ------------------------------------------------------------------ weights
    _CTOR_KEYS = ("in_channels", "ch", "ch_mult", "num_res_blocks", "z_channels", "double_z", "temporal_compress_times", "scaling_factor")
    _PREFIXES = ("first_stage_model.encoder.", "encoder.")      # full SAT / autoencoder checkpoints carry the encoder under these

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """accepts the encoder's own keys or a whole autoencoder checkpoint of the in-tree twin (its ``encoder.`` sub-tree is taken,
        decoder / loss keys are ignored)"""
        self._packed = None
        sd = dict(state_dict)
        for pre in self._PREFIXES:
            if any(k.startswith(pre) for k in sd):
                sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
                break
        if any(k.startswith(("down_blocks.", "mid_block.")) or ".resnets." in k for k in sd):
            raise KeyError("this looks like a diffusers AutoencoderKLCogVideoX checkpoint (keys like encoder.down_blocks.N.resnets.M...): "
                           "that key layout is not supported (diffusers is absent offline, the map cannot be verified); convert to the "
                           "SAT / in-tree twin names (down.N.block.M..., cogvideo_sat/vae_modules/cp_enc_dec.py) -- see INTEGRATION.md")
        return super().load_state_dict({k: v.to(BF16) for k, v in sd.items()}, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path: str = None, pretrained_model_name_or_path: str = None, subfolder: str = None, **config):
        """``path`` (or the reference YAML's ``pretrained_model_name_or_path`` + ``subfolder``, configs/004_cogvideox/*.yaml:6-10): a
        .safetensors file (or a directory holding ``model.safetensors`` / ``diffusion_pytorch_model.safetensors``) with the in-tree
        twin's parameter names; ``config`` overrides the CogVideoX defaults.  Nothing is fetched."""
        import os
        from safetensors.torch import load_file
        if path is None:
            path = os.path.join(pretrained_model_name_or_path, subfolder or "")
        config = {k: v for k, v in config.items() if k in cls._CTOR_KEYS}
        if os.path.isdir(path):
            cj = os.path.join(path, "config.json")
            if os.path.exists(cj):              # scaling_factor (2B: 1.15258426, 5B: 0.7) and any constructor key the file states
                import json
                with open(cj) as f:
                    cfg_file = json.load(f)
                for k in cls._CTOR_KEYS:
                    if k in cfg_file and k not in config:
                        config[k] = cfg_file[k]
            for fn in ("model.safetensors", "diffusion_pytorch_model.safetensors"):
                if os.path.exists(os.path.join(path, fn)):
                    path = os.path.join(path, fn)
                    break
        model = cls(**config)
        model.load_state_dict(load_file(path))
        return model

    def init_weights(self, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() > 1:
                    fan = p[0].numel()
                    p.copy_((torch.randn(p.shape, generator=g) / fan ** 0.5).to(p.dtype))
                elif name.endswith("weight"):
                    p.copy_((1.0 + 0.1 * torch.randn(p.shape, generator=g)).to(p.dtype))
                else:
                    p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(p.dtype))
        self._packed = None
        return self

    @property
    def device(self):
        return self.norm_out.weight.device

    @property
    def dtype(self):
        return self.norm_out.weight.dtype

    def _pack(self):
        """tap-major convolution weights ([Cout, taps * Cin], input channel innermost), built once (the VAE is frozen)"""
        if self._packed is None or next(iter(self._packed.values())).device != self.device:
            pk = {}
            for name, m in self.named_modules():
                if isinstance(m, _Conv) and m.weight.dim() >= 4:
                    w = m.weight
                    if name == "conv_in.conv":                       # RGB: 8 channels per position, 8 taps per K-tile
                        pk[name] = ops.pack_conv_in8_weight(w)
                        continue
                    if w.dim() == 5 and tuple(w.shape[2:]) == (1, 1, 1):
                        pk[name] = w.reshape(w.shape[0], w.shape[1]).contiguous()
                    else:
                        pk[name] = ops.pack_conv_weight(w)
            self._packed = pk
        return self._packed

    # ------------------------------------------------------------------ forward
    def _gn(self, x: torch.Tensor, norm: "_Norm", y: torch.Tensor):
        """GroupNorm(32, eps 1e-6) + SiLU on a channels-last clip [B, T, H, W, C]; with frame batching the statistics are per frame batch"""
        B, T, H, W, C = x.shape
        fb = self.frame_batch
        stride = max(1, (self._t_in - 1) // max(T - 1, 1)) if T > 1 else 1            # temporal compression at this level
        if fb is None or self._t_in <= fb + 1 or (self._t_in - 1) % fb or fb % stride:
            ops.groupnorm_silu(x.view(B, -1, C), norm.weight, norm.bias, y.view(B, -1, C), 32, 1e-6, True)
            return
        first, rest = fb // stride + 1, fb // stride
        n = (T - first) // rest
        for b in range(B):
            ops.groupnorm_silu(x[b:b + 1, :first].reshape(1, -1, C), norm.weight, norm.bias, y[b:b + 1, :first].view(1, -1, C), 32, 1e-6, True)
            ops.groupnorm_silu(x[b, first:].view(n, rest * H * W, C), norm.weight, norm.bias, y[b, first:].view(n, rest * H * W, C), 32, 1e-6, True)

    def _res(self, blk: _ResBlock, pre: str, x: torch.Tensor, pk) -> torch.Tensor:
        B, T, H, W, cin = x.shape
        cout = blk.out_channels
        E = lambda c: torch.empty(B, T, H, W, c, dtype=BF16, device=x.device)
        a = E(cin)
        self._gn(x, blk.norm1, a)
        h = E(cout)
        ops.causal_conv3d(a, pk[pre + "conv1.conv"], blk.conv1.conv.bias, h)
        a2 = E(cout)
        self._gn(h, blk.norm2, a2)
        if cin != cout:
            skip = E(cout)
            ops.gemm(x.view(-1, cin), pk[pre + "nin_shortcut"], skip.view(-1, cout), blk.nin_shortcut.bias)
        else:
            skip = x
        out = E(cout)
        ops.causal_conv3d(a2, pk[pre + "conv2.conv"], blk.conv2.conv.bias, out, residual=skip)
        return out

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, 3, T, H, W] (any float dtype, values in [-1, 1]) -> moments [B, 2 z, T', H', W'] bf16"""
        c = self.config
        if x.dim() != 5 or x.shape[1] != c.in_channels:
            raise ValueError(f"expected [B, {c.in_channels}, T, H, W], got {tuple(x.shape)}")
        dev = self.device
        pk = self._pack()
        B, _, T, H, W = x.shape
        self._t_in = T
        nlev = len(c.ch_mult)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise ValueError(f"H, W must be multiples of {1 << (nlev - 1)}")
        xin = torch.zeros(B, T, H, W, 8, dtype=BF16, device=dev)                   # channels-last, RGB in the first 3 of 8 channels
        xin[..., :c.in_channels] = x.to(dev).permute(0, 2, 3, 4, 1)
        h = torch.empty(B, T, H, W, c.ch, dtype=BF16, device=dev)
        ops.causal_conv3d_in8(xin, pk["conv_in.conv"], self.conv_in.conv.bias, h)
        del xin
        for i, lvl in enumerate(self.down):
            for j, blk in enumerate(lvl.block):
                h = self._res(blk, f"down.{i}.block.{j}.", h, pk)
            if i != nlev - 1:
                Bq, Tq, Hq, Wq, Cq = h.shape
                if i < self.temporal_levels and Tq > 1:
                    # odd frame count: first frame kept, pairs after it (the in-tree twin's rank-0 branch, cp_enc_dec.py:645-657);
                    # even: plain pairs over all frames (its other branch, :658-667) -- the rule of diffusers' CogVideoXDownsample3D,
                    # whose chunked encode feeds 9 frames first and 8-frame chunks after that
                    keep_first = (Tq % 2 == 1)
                    hp = torch.empty(Bq, ops.temporal_pool_frames(Tq, keep_first), Hq, Wq, Cq, dtype=BF16, device=dev)
                    ops.temporal_pool(h, hp, keep_first)
                    h = hp
                hd = torch.empty(Bq, h.shape[1], Hq // 2, Wq // 2, Cq, dtype=BF16, device=dev)
                ops.downsample_conv2d(h, pk[f"down.{i}.downsample.conv"], lvl.downsample.conv.bias, hd)
                h = hd
        h = self._res(self.mid.block_1, "mid.block_1.", h, pk)
        h = self._res(self.mid.block_2, "mid.block_2.", h, pk)
        Bq, Tq, Hq, Wq, Cq = h.shape
        a = torch.empty_like(h)
        self._gn(h, self.norm_out, a)
        zc = self.conv_out.conv.weight.shape[0]
        m = torch.empty(Bq, Tq, Hq, Wq, zc, dtype=BF16, device=dev)
        ops.causal_conv3d(a, pk["conv_out.conv"], self.conv_out.conv.bias, m)
        return m.permute(0, 4, 1, 2, 3)                                             # [B, 2z, T', H', W'] view

    def encode(self, x: torch.Tensor):
        """``.encode(x).latent_dist.sample() * .config.scaling_factor`` as the reference calls it (cogvideo_pl.py:792-806)"""
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.forward(x)))

    def latents(self, video: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """one clip [3, T, H, W] (or a batch [B, 3, T, H, W]) -> sampled latents * scaling_factor, [B, z, T', H', W'] fp32: what
        ``encode_video`` + ``get_batch_input`` compute per sample (cogvideo_pl.py:792-806); the callable to pass as the workflow's
        ``first_stage``"""
        if video.dim() == 4:
            video = video.unsqueeze(0)
        return self.encode(video).latent_dist.sample(generator) * self.config.scaling_factor
