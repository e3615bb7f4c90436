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


# ---------------------------------------------------------------------------------------------------------------------------------------
# The rest of HYVideoDiffusionTransformer (models.py:396-720): embedders, token refiner, final layer, unpatchify.  Pinned by
# tests/golden/hunyuan_model.npz (the reference's own class imported at a tiny size: tests/golden/make_golden_hunyuan.py).
#   PatchEmbed (Conv3d kernel = stride = patch) ......... modules/embed_layers.py:9-55
#   TimestepEmbedder / timestep_embedding ............... modules/embed_layers.py:86-157 (cos | sin, max_period 1e4, Linear-SiLU-Linear)
#   TextProjection, MLPEmbedder ......................... modules/embed_layers.py:58-84, modules/mlp_layers.py:63-75
#   SingleTokenRefiner / IndividualTokenRefiner[Block] .. modules/token_refiner.py:16-236 (LayerNorm affine, self-attention over the text
#       tokens with mask (valid x valid) | key 0, gates from SiLU-Linear(c), MLP with SiLU)
#   FinalLayer .......................................... modules/mlp_layers.py:78-118
def model_shapes(D, H, n_double, n_single, in_ch, out_ch, patch, text_dim, text_dim_2, ratio=4.0, refiner_depth=2, guidance=False):
    M4 = int(D * ratio)
    pt, ph, pw = patch
    sh = {"img_in.proj.weight": (D, in_ch, pt, ph, pw), "img_in.proj.bias": (D,),
          "txt_in.input_embedder.weight": (D, text_dim), "txt_in.input_embedder.bias": (D,),
          "txt_in.t_embedder.mlp.0.weight": (D, 256), "txt_in.t_embedder.mlp.0.bias": (D,),
          "txt_in.t_embedder.mlp.2.weight": (D, D), "txt_in.t_embedder.mlp.2.bias": (D,),
          "txt_in.c_embedder.linear_1.weight": (D, text_dim), "txt_in.c_embedder.linear_1.bias": (D,),
          "txt_in.c_embedder.linear_2.weight": (D, D), "txt_in.c_embedder.linear_2.bias": (D,)}
    for i in range(refiner_depth):
        p = f"txt_in.individual_token_refiner.blocks.{i}."
        sh.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "self_attn_qkv.weight": (3 * D, D), p + "self_attn_qkv.bias": (3 * D,),
                   p + "self_attn_proj.weight": (D, D), p + "self_attn_proj.bias": (D,), p + "norm2.weight": (D,), p + "norm2.bias": (D,),
                   p + "mlp.fc1.weight": (M4, D), p + "mlp.fc1.bias": (M4,), p + "mlp.fc2.weight": (D, M4), p + "mlp.fc2.bias": (D,),
                   p + "adaLN_modulation.1.weight": (2 * D, D), p + "adaLN_modulation.1.bias": (2 * D,)})
    sh.update({"time_in.mlp.0.weight": (D, 256), "time_in.mlp.0.bias": (D,), "time_in.mlp.2.weight": (D, D), "time_in.mlp.2.bias": (D,),
               "vector_in.in_layer.weight": (D, text_dim_2), "vector_in.in_layer.bias": (D,),
               "vector_in.out_layer.weight": (D, D), "vector_in.out_layer.bias": (D,)})
    if guidance:
        sh.update({"guidance_in.mlp.0.weight": (D, 256), "guidance_in.mlp.0.bias": (D,), "guidance_in.mlp.2.weight": (D, D), "guidance_in.mlp.2.bias": (D,)})
    for i in range(n_double):
        sh.update(double_block_shapes(D, H, ratio, f"double_blocks.{i}."))
    for i in range(n_single):
        sh.update(single_block_shapes(D, H, ratio, f"single_blocks.{i}."))
    sh.update({"final_layer.linear.weight": (pt * ph * pw * out_ch, D), "final_layer.linear.bias": (pt * ph * pw * out_ch,),
               "final_layer.adaLN_modulation.1.weight": (2 * D, D), "final_layer.adaLN_modulation.1.bias": (2 * D,)})
    return sh


def init_model(shapes: Dict[str, tuple], seed: int):
    """like init(), conv weights with their own fan-in, every bias / norm non-trivial (the reference zero-initialises several: a test that
    multiplies by zero pins nothing)"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, s in shapes.items():
        if len(s) == 1:
            out[k] = torch.randn(s, generator=g) * 0.05 + (1.0 if k.endswith("norm.weight") or ".norm1.weight" in k or ".norm2.weight" in k else 0.0)
        else:
            fan = 1
            for d in s[1:]:
                fan *= d
            out[k] = torch.randn(s, generator=g) * (0.7 / fan ** 0.5)
    return out


def timestep_embedding(t, dim=256, max_period=10000):
    import math
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _tembed(t, P, pre, dtype):
    e = timestep_embedding(t).to(dtype)
    return F.linear(F.silu(F.linear(e, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"])), P[pre + "mlp.2.weight"], P[pre + "mlp.2.bias"])


def token_refiner(text_states, t, mask, P, H, depth=2):
    """SingleTokenRefiner.forward (token_refiner.py:199-236) with IndividualTokenRefiner's mask (:136-160)"""
    pre = "txt_in."
    B, L, _ = text_states.shape
    temb = _tembed(t, P, pre + "t_embedder.", text_states.dtype)
    if mask is None:
        ctx = text_states.mean(1)
    else:
        mf = mask.to(text_states.dtype).unsqueeze(-1)
        ctx = (text_states * mf).sum(1) / mf.sum(1)
    ctx = F.linear(F.silu(F.linear(ctx, P[pre + "c_embedder.linear_1.weight"], P[pre + "c_embedder.linear_1.bias"])),
                   P[pre + "c_embedder.linear_2.weight"], P[pre + "c_embedder.linear_2.bias"])
    c = temb + ctx
    x = F.linear(text_states, P[pre + "input_embedder.weight"], P[pre + "input_embedder.bias"])
    D = x.shape[-1]
    am = None
    if mask is not None:
        m1 = mask.bool().view(B, 1, 1, L).repeat(1, 1, L, 1)
        am = m1 & m1.transpose(2, 3)
        am[:, :, :, 0] = True
    for i in range(depth):
        p = f"{pre}individual_token_refiner.blocks.{i}."
        g_msa, g_mlp = F.linear(F.silu(c), P[p + "adaLN_modulation.1.weight"], P[p + "adaLN_modulation.1.bias"]).chunk(2, dim=1)
        nx = F.layer_norm(x, (D,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-6)
        qkv = F.linear(nx, P[p + "self_attn_qkv.weight"], P[p + "self_attn_qkv.bias"]).view(B, L, 3, H, D // H)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=am).transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(a, P[p + "self_attn_proj.weight"], P[p + "self_attn_proj.bias"]) * g_msa[:, None]
        h = F.layer_norm(x, (D,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-6)
        h = F.linear(F.silu(F.linear(h, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"])), P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
        x = x + h * g_mlp[:, None]
    return x


def transformer_forward(P, x, t, text_states, text_mask, text_states_2, cos, sin, H, n_double, n_single, patch, out_ch, guidance=None):
    """HYVideoDiffusionTransformer.forward (models.py:592-700).  x [B, C, T, Hh, Ww]; text_mask [B, L] int (1 = valid)"""
    B, C, T, Hh, Ww = x.shape
    pt, ph, pw = patch
    tt, th, tw = T // pt, Hh // ph, Ww // pw
    vec = _tembed(t, P, "time_in.", x.dtype)
    vec = vec + F.linear(F.silu(F.linear(text_states_2, P["vector_in.in_layer.weight"], P["vector_in.in_layer.bias"])),
                         P["vector_in.out_layer.weight"], P["vector_in.out_layer.bias"])
    if guidance is not None:
        vec = vec + _tembed(guidance, P, "guidance_in.", x.dtype)
    img = F.conv3d(x, P["img_in.proj.weight"], P["img_in.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    txt = token_refiner(text_states, t, text_mask, P, H)
    Lt = txt.shape[1]
    tv = text_mask.sum(1)
    for i in range(n_double):
        img, txt = double_block(img, txt, vec, P, f"double_blocks.{i}.", H, tv, cos, sin)
    xx = torch.cat([img, txt], 1)
    for i in range(n_single):
        xx = single_block(xx, vec, P, f"single_blocks.{i}.", H, Lt, tv, cos, sin)
    img = xx[:, :img.shape[1]]
    shift, scale = F.linear(F.silu(vec), P["final_layer.adaLN_modulation.1.weight"], P["final_layer.adaLN_modulation.1.bias"]).chunk(2, dim=1)
    img = F.linear(_ln(img) * (1 + scale[:, None]) + shift[:, None], P["final_layer.linear.weight"], P["final_layer.linear.bias"])
    img = img.reshape(B, tt, th, tw, out_ch, pt, ph, pw)
    return torch.einsum("nthwcopq->nctohpwq", img).reshape(B, out_ch, tt * pt, th * ph, tw * pw)
