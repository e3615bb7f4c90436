"""CogVideoX 3D-full-attention DiT on the vt355 HIP kernels -- drop-in for the module the reference selects with
``denoiser_config.target: diffusers.CogVideoXTransformer3DModel`` (configs/004_cogvideox/cogvideo2b.yaml:22-27) and
calls at videotuna/models/cogvideo_hf/cogvideo_pl.py:865-871:

    model(hidden_states=[B,F,C,H,W], encoder_hidden_states=[B,226,4096], timestep=int64[B],
          image_rotary_emb=None, return_dict=False)[0] -> [B,F,C,H,W]

Same attribute surface the workflow touches (``.config.*``, ``.dtype``, ``.enable_gradient_checkpointing()``,
``.requires_grad_()``, ``named_parameters()`` with the HF ``diffusion_pytorch_model.safetensors`` key names,
``from_pretrained(path, subfolder=...)``).  The arithmetic runs entirely in libvt355.so;
PyTorch only owns memory, streams and the autograd hook.  There is no eager fallback.

Engine layout (per sample S = St + Sv rows, text rows first; M = B*S):
  h   [M, d]        residual stream, bf16
  x   [M, d+64]     LayerNorm-modulated GEMM input; columns d..d+15 carry the LoRA down-projection T = x A^T,
                    the packed weights carry (alpha/r) B in the matching K-extension, so LoRA costs no extra GEMM
  qkv [M, 3d]       fused projection, consumed in place by the attention kernels (no head transposes)
"""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops

BF16 = torch.bfloat16
EXT = 64          # K-extension width (GEMM K must be a multiple of 64); LoRA uses the first 16 columns

_DEFAULT_CONFIG = dict(
    num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16, flip_sin_to_cos=True,
    freq_shift=0, time_embed_dim=512, text_embed_dim=4096, num_layers=30, dropout=0.0, attention_bias=True,
    sample_width=90, sample_height=60, sample_frames=49, patch_size=2, temporal_compression_ratio=4,
    max_text_seq_length=226, activation_fn="gelu-approximate", timestep_activation_fn="silu",
    norm_elementwise_affine=True, norm_eps=1e-5, spatial_interpolation_scale=1.875,
    temporal_interpolation_scale=1.0, use_rotary_positional_embeddings=False,
    use_learned_positional_embeddings=False,
)


# ---------------------------------------------------------------------------------------------------
# parameter containers (naming only -- the math lives in the engine below)
# ---------------------------------------------------------------------------------------------------
class Linear(nn.Module):
    """Parameter holder with nn.Linear's names/shapes; ``forward`` runs the HIP GEMM (inference use)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=BF16))
        self.bias = nn.Parameter(torch.empty(out_features, dtype=BF16)) if bias else None

    def forward(self, x):
        x2 = x.reshape(-1, self.in_features).contiguous()
        out = torch.empty(x2.shape[0], self.out_features, dtype=BF16, device=x.device)
        ops.gemm(x2, self.weight, out, self.bias)
        return out.view(*x.shape[:-1], self.out_features)


class LayerNormP(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(dim, dtype=BF16))
        self.bias = nn.Parameter(torch.empty(dim, dtype=BF16))


class PatchProj(nn.Module):          # Conv2d(k=p, s=p) parameter holder
    def __init__(self, cin, cout, p):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, p, p, dtype=BF16))
        self.bias = nn.Parameter(torch.empty(cout, dtype=BF16))


class _Container(nn.Module):
    pass


def _block(d, te, hd, ff_mult):
    b = _Container()
    for n in ("norm1", "norm2"):
        m = _Container()
        m.linear = Linear(te, 6 * d)
        m.norm = LayerNormP(d, 1e-5)
        setattr(b, n, m)
    a = _Container()
    a.to_q, a.to_k, a.to_v = Linear(d, d), Linear(d, d), Linear(d, d)
    a.to_out = nn.ModuleList([Linear(d, d), nn.Identity()])
    a.norm_q, a.norm_k = LayerNormP(hd, 1e-6), LayerNormP(hd, 1e-6)
    b.attn1 = a
    ff = _Container()
    g = _Container()
    g.proj = Linear(d, ff_mult * d)
    ff.net = nn.ModuleList([g, nn.Identity(), Linear(ff_mult * d, d)])
    b.ff = ff
    return b


def sincos_pos_embed_3d(embed_dim, grid_h, grid_w, t_size, spatial_scale, temporal_scale) -> np.ndarray:
    """Fixed 3D sincos table [T*H*W, D]: D/4 temporal then 3D/4 spatial; the first spatial half encodes the w
    coordinate (meshgrid(w, h)); each 1-D table is [sin | cos]  (dit_video_concat.py:59-160)."""
    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    d_sp, d_t = embed_dim // 4 * 3, embed_dim // 4
    gh = np.arange(grid_h, dtype=np.float32) / spatial_scale
    gw = np.arange(grid_w, dtype=np.float32) / spatial_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_h, grid_w)
    sp = np.concatenate([one_d(d_sp // 2, grid[0]), one_d(d_sp // 2, grid[1])], axis=1)
    tt = one_d(d_t, np.arange(t_size, dtype=np.float32) / temporal_scale)
    tt = np.repeat(tt[:, None, :], grid_h * grid_w, axis=1)
    sp = np.repeat(sp[None], t_size, axis=0)
    return np.concatenate([tt, sp], axis=-1).reshape(t_size * grid_h * grid_w, embed_dim)


# ---------------------------------------------------------------------------------------------------
class CogVideoXTransformer3DModel(nn.Module):
    _supports_gradient_checkpointing = True

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(_DEFAULT_CONFIG)
        unknown = set(kwargs) - set(cfg) - {"ff_mult"}
        if unknown:
            raise TypeError(f"unknown config keys {sorted(unknown)}")
        cfg.update(kwargs)
        cfg.setdefault("ff_mult", 4)
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if c.attention_head_dim != 64:
            raise NotImplementedError("the HIP attention kernels are specialised for head_dim 64")
        if c.use_learned_positional_embeddings and not c.use_rotary_positional_embeddings:
            raise ValueError("learned positional embeddings come with RoPE (CogVideoX-5B-I2V); 2B uses the fixed sincos table")
        d = c.num_attention_heads * c.attention_head_dim
        if d % 64 or c.time_embed_dim % 64 or c.text_embed_dim % 64 or (c.in_channels * c.patch_size ** 2) % 64:
            raise NotImplementedError("inner dims must be multiples of 64 (GEMM K-tile)")
        self.inner_dim = d
        pe = _Container()
        pe.proj = PatchProj(c.in_channels, d, c.patch_size)
        pe.text_proj = Linear(c.text_embed_dim, d)
        if c.use_learned_positional_embeddings:
            # CogVideoX-5B-I2V: a persistent (checkpointed, never trained) [1, max_text + patches, d] buffer, initialised
            # with the sincos table, added to the joint sequence; fixed to the configured sample size
            frames = (c.sample_frames - 1) // c.temporal_compression_ratio + 1
            tab = sincos_pos_embed_3d(d, c.sample_height // c.patch_size, c.sample_width // c.patch_size, frames,
                                      c.spatial_interpolation_scale, c.temporal_interpolation_scale)
            full = torch.zeros(1, c.max_text_seq_length + tab.shape[0], d, dtype=BF16)
            full[0, c.max_text_seq_length:] = torch.from_numpy(tab).to(torch.float32).to(BF16)
            pe.register_buffer("pos_embedding", full, persistent=True)
        self.patch_embed = pe
        te = _Container()
        te.linear_1, te.linear_2 = Linear(d, c.time_embed_dim), Linear(c.time_embed_dim, c.time_embed_dim)
        self.time_embedding = te
        self.transformer_blocks = nn.ModuleList([_block(d, c.time_embed_dim, c.attention_head_dim, c.ff_mult)
                                                 for _ in range(c.num_layers)])
        self.norm_final = LayerNormP(d, c.norm_eps)
        no = _Container()
        no.linear = Linear(c.time_embed_dim, 2 * d)
        no.norm = LayerNormP(d, c.norm_eps)
        self.norm_out = no
        self.proj_out = Linear(d, c.patch_size ** 2 * c.out_channels)
        self.gradient_checkpointing = False
        self.recompute = "never"          # set by enable_gradient_checkpointing()
        self._packed = None           # engine operands, built lazily from the parameters
        self._pos_cache: Dict[tuple, torch.Tensor] = {}
        self.lora = None              # set by vt355.lora.inject
        self.fullft = None            # set by vt355.fullft.enable_full_finetune

    # ----- HF-like surface -----
    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    @property
    def device(self):
        return self.proj_out.weight.device

    def enable_gradient_checkpointing(self, mode: str = "auto"):
        """API parity with diffusers (cogvideo_pl.py:141): per-block activation recompute.  ``mode`` "auto" (default) keeps
        the ~27 GB/sample of block activations while they fit comfortably in the free HBM of a 288 GB MI355X (-25 % executed
        FLOPs) and recomputes each block from its input otherwise; "always" / "never" force either (engine._use_recompute)."""
        if mode not in ("auto", "always", "never"):
            raise ValueError(f"mode must be auto | always | never, got {mode!r}")
        self.gradient_checkpointing = True
        self.recompute = mode

    def init_weights(self, seed: int = 0, std: float = 0.02):
        """Seeded random init (no checkpoints offline): W,b ~ N(0, std), LayerNorm gamma ~ 1 + N(0, std)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if "lora" in name:
                    continue
                w = torch.randn(p.shape, generator=g, dtype=torch.float32) * std
                if name.endswith("norm.weight") or name.endswith("norm_final.weight") or name.endswith("norm_q.weight") \
                        or name.endswith("norm_k.weight"):
                    w = w + 1.0
                p.copy_(w.to(p.dtype))
        self._packed = None
        return self

    @classmethod
    def from_config(cls, config: dict):
        return cls(**{k: v for k, v in config.items() if k in _DEFAULT_CONFIG or k == "ff_mult"})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, **kw):
        """Reads HF ``config.json`` + ``diffusion_pytorch_model*.safetensors`` from a LOCAL directory."""
        root = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        with open(os.path.join(root, "config.json")) as f:
            cfg = json.load(f)
        model = cls.from_config(cfg)
        from safetensors.torch import load_file
        files = sorted(f for f in os.listdir(root) if f.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors under {root}")
        sd = {}
        for f in files:
            sd.update(load_file(os.path.join(root, f)))
        if not model.config.use_learned_positional_embeddings:
            sd.pop("patch_embed.pos_embedding", None)   # fixed buffer, regenerated
        model.load_state_dict({k: v.to(BF16) for k, v in sd.items()}, strict=True)
        return model

    def _apply(self, fn, *a, **k):
        if getattr(self, "fullft", None) is not None:
            raise RuntimeError("move / convert the model BEFORE enable_full_finetune(): its parameters are views of one flat buffer")
        out = super()._apply(fn, *a, **k)
        self._packed = None
        self._pos_cache = {}
        if self.lora is not None:
            self.lora.on_module_moved()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._packed = None
        ft = getattr(self, "fullft", None)
        if ft is not None:              # the parameters are bf16 views: the fp32 master and the transposed operand copies follow
            with torch.no_grad():
                ft.flat.copy_(ft.flat_bf16)
            ft.mark_changed()
        if self.lora is not None:
            self.lora.mark_changed()
        return out

    # ----- forward -----
    def forward(self, hidden_states, encoder_hidden_states, timestep, timestep_cond=None, image_rotary_emb=None,
                return_dict: bool = False, **unused):
        if (image_rotary_emb is not None) != bool(self.config.use_rotary_positional_embeddings):
            raise ValueError("image_rotary_emb=(cos, sin) must be given exactly when config.use_rotary_positional_embeddings "
                             "is set (CogVideoX-5B recipes; cogvideo_pl.py:846-859)")
        if not hidden_states.is_cuda:
            raise RuntimeError("vt355 CogVideoXTransformer3DModel runs only on an MI355X device (no CPU fallback)")
        from .engine import dit_apply
        out = dit_apply(self, hidden_states, encoder_hidden_states, timestep, image_rotary_emb)
        if return_dict:
            return SimpleNamespace(sample=out)
        return (out,)

    # ----- positional table -----
    def pos_table(self, frames: int, height: int, width: int, device):
        """bf16 [Sv, d] table added to the video tokens (None for the RoPE models without a learned table)"""
        c = self.config
        if c.use_learned_positional_embeddings:
            tab = self.patch_embed.pos_embedding
            Sv = frames * (height // c.patch_size) * (width // c.patch_size)
            if tab.shape[1] != c.max_text_seq_length + Sv:
                raise ValueError(f"learned positional embeddings are fixed to the configured sample size "
                                 f"({tab.shape[1] - c.max_text_seq_length} patches), got {Sv}")
            return tab[0, c.max_text_seq_length:]
        if c.use_rotary_positional_embeddings:
            return None
        key = (frames, height, width, str(device))
        if key not in self._pos_cache:
            c = self.config
            tab = sincos_pos_embed_3d(self.inner_dim, height // c.patch_size, width // c.patch_size, frames,
                                      c.spatial_interpolation_scale, c.temporal_interpolation_scale)
            self._pos_cache[key] = torch.from_numpy(tab).to(torch.float32).to(BF16).to(device).contiguous()
        return self._pos_cache[key]
