"""CPU ORACLE (test infrastructure, NOT product code) for the HunyuanVideo transformer blocks (BASELINE configs[4]):
MMDoubleStreamBlock and MMSingleStreamBlock of videotuna/models/hunyuan/hyvideo_t2v/modules/models.py (:21-252, :255-393) with
  * ModulateDiT / modulate / apply_gate ..... modules/modulate_layers.py:7-66 (SiLU -> Linear -> chunk; x (1 + scale) + shift; x * gate)
  * RMSNorm on q and k per head ............. modules/norm_layers.py:5-58 (eps 1e-6, weight per head dim; fp32 statistics)
  * apply_rotary_emb on the IMAGE tokens .... modules/posemb_layers.py:133-188 (pairs (2i, 2i+1): x cos + rotate_half(x) sin)
  * joint [image; text] attention ........... modules/attenion.py:60-156 with get_cu_seqlens (:34-57): per sample one segment of the image
                                              tokens + the VALID text tokens and one segment of the padding text tokens
  * MLP (GELU tanh) ......................... modules/mlp_layers.py:13-60
and the flow-matching loss of HunyuanVideoWorkFlow.training_step (hyvideo_t2v/hunyuanvideo.py:923-971): x_t = (1 - sigma) x0 + sigma eps,
target = eps - x0, mean squared error.
PARITY STATUS: pinned for the blocks -- tests/golden/hunyuan_blocks.npz holds outputs and gradients of the reference's own block classes
imported in the build container (tests/golden/make_golden_hunyuan.py; diffusers' ModelMixin / ConfigMixin stubbed, flash_attn_varlen_func
-- an absent binary -- re-expressed with per-segment SDPA, so the varlen attention itself is pinned by restatement).  Only ``tests/`` may
import this file.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def double_block_shapes(D: int, H: int, ratio: float = 4.0, pre: str = "") -> Dict[str, tuple]:
    hd, M4 = D // H, int(D * ratio)
    sh = {}
    for s in ("img", "txt"):
        sh[f"{pre}{s}_mod.linear.weight"] = (6 * D, D); sh[f"{pre}{s}_mod.linear.bias"] = (6 * D,)
        sh[f"{pre}{s}_attn_qkv.weight"] = (3 * D, D); sh[f"{pre}{s}_attn_qkv.bias"] = (3 * D,)
        sh[f"{pre}{s}_attn_q_norm.weight"] = (hd,); sh[f"{pre}{s}_attn_k_norm.weight"] = (hd,)
        sh[f"{pre}{s}_attn_proj.weight"] = (D, D); sh[f"{pre}{s}_attn_proj.bias"] = (D,)
        sh[f"{pre}{s}_mlp.fc1.weight"] = (M4, D); sh[f"{pre}{s}_mlp.fc1.bias"] = (M4,)
        sh[f"{pre}{s}_mlp.fc2.weight"] = (D, M4); sh[f"{pre}{s}_mlp.fc2.bias"] = (D,)
    return sh


def single_block_shapes(D: int, H: int, ratio: float = 4.0, pre: str = "") -> Dict[str, tuple]:
    hd, M4 = D // H, int(D * ratio)
    return {pre + "linear1.weight": (3 * D + M4, D), pre + "linear1.bias": (3 * D + M4,),
            pre + "linear2.weight": (D, D + M4), pre + "linear2.bias": (D,),
            pre + "q_norm.weight": (hd,), pre + "k_norm.weight": (hd,),
            pre + "modulation.linear.weight": (3 * D, D), pre + "modulation.linear.bias": (3 * D,)}


def init(shapes: Dict[str, tuple], seed: int):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, s in shapes.items():
        if len(s) == 1:
            out[k] = torch.randn(s, generator=g) * 0.05 + (1.0 if "norm.weight" in k else 0.0)
        else:
            out[k] = torch.randn(s, generator=g) * (0.7 / s[1] ** 0.5)
    return out


def rms_norm(x, w, eps=1e-6):
    return (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype) * w


def rotate_half(x):
    re, im = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    return torch.stack([-im, re], dim=-1).flatten(-2)


def rope(x, cos, sin):
    """x [B, L, H, D]; cos / sin [L, D]"""
    return x * cos[None, :, None] + rotate_half(x) * sin[None, :, None]


def varlen_attention(q, k, v, valid_len):
    """q, k, v [B, S, H, D]; sample b: rows [0, valid_len[b]) attend among themselves, rows [valid_len[b], S) (padding) among
    themselves -- the two segments per sample of get_cu_seqlens (attenion.py:34-57)"""
    B, S, H, Dh = q.shape
    out = torch.zeros_like(q)
    for b in range(B):
        n = int(valid_len[b])
        for lo, hi in ((0, n), (n, S)):
            if hi > lo:
                o = F.scaled_dot_product_attention(q[b, lo:hi].transpose(0, 1), k[b, lo:hi].transpose(0, 1), v[b, lo:hi].transpose(0, 1))
                out[b, lo:hi] = o.transpose(0, 1)
    return out.reshape(B, S, H * Dh)


def _mod(vec, P, name, n):
    return F.linear(F.silu(vec), P[name + ".linear.weight"], P[name + ".linear.bias"]).chunk(n, dim=-1)


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def double_block(img, txt, vec, P, pre, H, txt_valid, cos=None, sin=None):
    """MMDoubleStreamBlock.forward (models.py:132-252).  img [B, Li, D], txt [B, Lt, D], vec [B, D], txt_valid [B] valid text tokens"""
    B, Li, D = img.shape
    i1s, i1c, i1g, i2s, i2c, i2g = _mod(vec, P, pre + "img_mod", 6)
    t1s, t1c, t1g, t2s, t2c, t2g = _mod(vec, P, pre + "txt_mod", 6)

    def qkv(x, s, shift, scale):
        xm = _ln(x) * (1 + scale[:, None]) + shift[:, None]
        y = F.linear(xm, P[f"{pre}{s}_attn_qkv.weight"], P[f"{pre}{s}_attn_qkv.bias"]).view(B, x.shape[1], 3, H, D // H)
        q, k, v = y[:, :, 0], y[:, :, 1], y[:, :, 2]
        return rms_norm(q, P[f"{pre}{s}_attn_q_norm.weight"]), rms_norm(k, P[f"{pre}{s}_attn_k_norm.weight"]), v
    iq, ik, iv = qkv(img, "img", i1s, i1c)
    if cos is not None:
        iq, ik = rope(iq, cos, sin), rope(ik, cos, sin)
    tq, tk, tv = qkv(txt, "txt", t1s, t1c)
    a = varlen_attention(torch.cat([iq, tq], 1), torch.cat([ik, tk], 1), torch.cat([iv, tv], 1), [Li + int(n) for n in txt_valid])
    ia, ta = a[:, :Li], a[:, Li:]

    def tail(x, s, att, g1, s2, c2, g2):
        x = x + F.linear(att, P[f"{pre}{s}_attn_proj.weight"], P[f"{pre}{s}_attn_proj.bias"]) * g1[:, None]
        h = _ln(x) * (1 + c2[:, None]) + s2[:, None]
        h = F.linear(F.gelu(F.linear(h, P[f"{pre}{s}_mlp.fc1.weight"], P[f"{pre}{s}_mlp.fc1.bias"]), approximate="tanh"),
                     P[f"{pre}{s}_mlp.fc2.weight"], P[f"{pre}{s}_mlp.fc2.bias"])
        return x + h * g2[:, None]
    return tail(img, "img", ia, i1g, i2s, i2c, i2g), tail(txt, "txt", ta, t1g, t2s, t2c, t2g)


def single_block(x, vec, P, pre, H, txt_len, txt_valid, cos=None, sin=None):
    """MMSingleStreamBlock.forward (models.py:319-393).  x [B, Li + Lt, D] = [image; text]"""
    B, S, D = x.shape
    Li = S - txt_len
    ms, mc, mg = _mod(vec, P, pre + "modulation", 3)
    xm = _ln(x) * (1 + mc[:, None]) + ms[:, None]
    y = F.linear(xm, P[pre + "linear1.weight"], P[pre + "linear1.bias"])
    qkv, mlp = y[..., :3 * D], y[..., 3 * D:]
    qkv = qkv.view(B, S, 3, H, D // H)
    q, k, v = rms_norm(qkv[:, :, 0], P[pre + "q_norm.weight"]), rms_norm(qkv[:, :, 1], P[pre + "k_norm.weight"]), qkv[:, :, 2]
    if cos is not None:
        q = torch.cat([rope(q[:, :Li], cos, sin), q[:, Li:]], 1)
        k = torch.cat([rope(k[:, :Li], cos, sin), k[:, Li:]], 1)
    a = varlen_attention(q, k, v, [Li + int(n) for n in txt_valid])
    out = F.linear(torch.cat([a, F.gelu(mlp, approximate="tanh")], 2), P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return x + out * mg[:, None]


def flow_matching(x0, noise, sigma):
    """hunyuanvideo.py:931-966: noisy latents and the regression target of the flow-matching loss"""
    s = sigma.view(-1, *([1] * (x0.dim() - 1)))
    return (1.0 - s) * x0 + s * noise, noise - x0
