"""CPU ORACLE (test infrastructure, NOT product code) for OpenSora v1.0: the STDiT denoiser and the IDDPM training loss with the
learned-variance VB term (BASELINE configs[0]).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
cpu_baseline may import this file.

What it follows (under /root/reference/videotuna/models/opensora/models/):
  * STDiT.forward ............... stdit/stdit.py:236-311 (PatchEmbed3D -> + spatial sincos -> t_embedder / t_block / y_embedder ->
                                   masked_select of the text tokens -> blocks (tpe on block 0 only) -> T2IFinalLayer -> unpatchify -> fp32)
  * STDiTBlock.forward .......... stdit/stdit.py:102-132 (scale_shift_table + t; spatial attention per frame and temporal attention
                                   per pixel BOTH gated by gate_msa; un-gated text cross-attention; gated MLP)
  * Attention ................... layers/blocks.py:139-225 (fused qkv + bias, 16 heads x 72, softmax(q k^T / sqrt(72)) v, proj)
  * MultiHeadCrossAttention ..... layers/blocks.py:472-505 (q_linear / kv_linear, every sample's tokens against ITS y_len text tokens
                                   -- xformers BlockDiagonalMask.from_seqlens([N]*B, y_lens))
  * PatchEmbed3D, T2IFinalLayer, TimestepEmbedder, CaptionEmbedder, t2i_modulate, sincos tables ... layers/blocks.py:75-136, 585-796, 857-919
  * loss ........................ iddpm3d.py:1332-1413 (mse(eps, eps_hat) + vb), _vb_terms_bpd :1543-1583, OpenSoraScheduler.p_mean_variance
                                   :444-519 -- for the default ModelMeanType.EPSILON its branch at :497-500 takes the RAW eps_hat as
                                   x_recon (the parent class predicts x0 first, :417-431); reproduced here, flagged REFERENCE QUIRK --,
                                   schedule :188-290 (float32 betas of get_named_beta_schedule("linear", 1000), :100-125)
PARITY STATUS: pinned.  tests/golden/stdit_tiny.npz (output, all parameter-gradient checksums, 10 gradients in full, XL/2 positional
tables) and stdit_loss.npz come from the reference's own code run by tests/golden/make_golden_stdit.py.  One op is pinned by
restatement only: the varlen text attention, because xformers is an absent binary (the generator re-expresses it with per-sample SDPA).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class STDiTConfig:
    input_size: Tuple[int, int, int] = (16, 32, 32)
    in_channels: int = 4
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    hidden_size: int = 1152
    depth: int = 28
    num_heads: int = 16
    mlp_ratio: float = 4.0
    caption_channels: int = 4096
    model_max_length: int = 120
    space_scale: float = 0.5
    time_scale: float = 1.0

    @property
    def out_channels(self):
        return 2 * self.in_channels          # pred_sigma

    @property
    def num_temporal(self):
        return self.input_size[0] // self.patch_size[0]

    @property
    def num_spatial(self):
        return (self.input_size[1] // self.patch_size[1]) * (self.input_size[2] // self.patch_size[2])


def tiny_config(**kw) -> STDiTConfig:
    base = dict(input_size=(4, 8, 8), hidden_size=576, depth=2, num_heads=8, caption_channels=64, model_max_length=12)
    base.update(kw)
    return STDiTConfig(**base)


def param_shapes(cfg: STDiTConfig) -> Dict[str, tuple]:
    """named_parameters() of the reference STDiT, in registration order (buffers pos_embed / pos_embed_temporal / y_embedding excluded)"""
    D, H4 = cfg.hidden_size, int(cfg.hidden_size * cfg.mlp_ratio)
    sh = {"x_embedder.proj.weight": (D, cfg.in_channels) + tuple(cfg.patch_size), "x_embedder.proj.bias": (D,),
          "t_embedder.mlp.0.weight": (D, 256), "t_embedder.mlp.0.bias": (D,), "t_embedder.mlp.2.weight": (D, D), "t_embedder.mlp.2.bias": (D,),
          "t_block.1.weight": (6 * D, D), "t_block.1.bias": (6 * D,),
          "y_embedder.y_proj.fc1.weight": (D, cfg.caption_channels), "y_embedder.y_proj.fc1.bias": (D,),
          "y_embedder.y_proj.fc2.weight": (D, D), "y_embedder.y_proj.fc2.bias": (D,)}
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        sh[b + "scale_shift_table"] = (6, D)
        sh[b + "attn.qkv.weight"] = (3 * D, D); sh[b + "attn.qkv.bias"] = (3 * D,)
        sh[b + "attn.proj.weight"] = (D, D); sh[b + "attn.proj.bias"] = (D,)
        sh[b + "cross_attn.q_linear.weight"] = (D, D); sh[b + "cross_attn.q_linear.bias"] = (D,)
        sh[b + "cross_attn.kv_linear.weight"] = (2 * D, D); sh[b + "cross_attn.kv_linear.bias"] = (2 * D,)
        sh[b + "cross_attn.proj.weight"] = (D, D); sh[b + "cross_attn.proj.bias"] = (D,)
        sh[b + "mlp.fc1.weight"] = (H4, D); sh[b + "mlp.fc1.bias"] = (H4,)
        sh[b + "mlp.fc2.weight"] = (D, H4); sh[b + "mlp.fc2.bias"] = (D,)
        sh[b + "attn_temp.qkv.weight"] = (3 * D, D); sh[b + "attn_temp.qkv.bias"] = (3 * D,)
        sh[b + "attn_temp.proj.weight"] = (D, D); sh[b + "attn_temp.proj.bias"] = (D,)
    sh["final_layer.scale_shift_table"] = (2, D)
    sh["final_layer.linear.weight"] = (math.prod(cfg.patch_size) * cfg.out_channels, D)
    sh["final_layer.linear.bias"] = (math.prod(cfg.patch_size) * cfg.out_channels,)
    return sh


def init_params(cfg: STDiTConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """seeded init; nothing zero (the reference zero-initialises attn_temp.proj, cross_attn.proj and final_layer.linear, stdit.py:381-416)"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, s in param_shapes(cfg).items():
        if k.endswith("scale_shift_table"):
            w = torch.randn(s, generator=g) / s[1] ** 0.5
        elif len(s) == 1:
            w = torch.randn(s, generator=g) * 0.05
        else:
            fan_in = math.prod(s[1:])
            w = torch.randn(s, generator=g) * (0.8 / math.sqrt(fan_in))
        out[k] = w.to(dtype)
    return out


# ---- sincos tables (blocks.py:857-919) ----
def _sincos_1d_from_grid(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def spatial_pos_embed(cfg: STDiTConfig):
    gh, gw = cfg.input_size[1] // cfg.patch_size[1], cfg.input_size[2] // cfg.patch_size[2]
    grid_h = np.arange(gh, dtype=np.float32) / cfg.space_scale
    grid_w = np.arange(gw, dtype=np.float32) / cfg.space_scale
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, gw, gh])       # w first, as the reference does
    emb_h = _sincos_1d_from_grid(cfg.hidden_size // 2, grid[0])
    emb_w = _sincos_1d_from_grid(cfg.hidden_size // 2, grid[1])
    return torch.from_numpy(np.concatenate([emb_h, emb_w], axis=1)).float()            # [S, D]


def temporal_pos_embed(cfg: STDiTConfig):
    pos = np.arange(0, cfg.num_temporal)[..., None] / cfg.time_scale
    return torch.from_numpy(_sincos_1d_from_grid(cfg.hidden_size, pos)).float()        # [T, D]


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def gelu_tanh(x):
    return F.gelu(x, approximate="tanh")


def _lin(x, P, n):
    return F.linear(x, P[n + ".weight"], P[n + ".bias"])


def attention(x, P, pre, heads):
    """Attention.forward (blocks.py:176-225), softmax path"""
    B, N, C = x.shape
    hd = C // heads
    qkv = _lin(x, P, pre + ".qkv").view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    return _lin((a @ v).transpose(1, 2).reshape(B, N, C), P, pre + ".proj")


def cross_attention(x, y_packed, y_lens, P, pre, heads):
    """MultiHeadCrossAttention.forward (blocks.py:486-505): x [B, N, C]; y_packed [1, sum(y_lens), C]"""
    B, N, C = x.shape
    hd = C // heads
    q = _lin(x, P, pre + ".q_linear").view(B, N, heads, hd)
    kv = _lin(y_packed, P, pre + ".kv_linear").view(-1, 2, heads, hd)
    outs, o = [], 0
    for b, L in enumerate(y_lens):
        k, v = kv[o:o + L, 0], kv[o:o + L, 1]
        a = (torch.einsum("nhd,lhd->hnl", q[b], k) * hd ** -0.5).softmax(dim=-1)
        outs.append(torch.einsum("hnl,lhd->nhd", a, v).reshape(N, C))
        o += L
    return _lin(torch.stack(outs), P, pre + ".proj")


def stdit_block(x, y_packed, y_lens, t0, P, pre, cfg: STDiTConfig, tpe=None):
    B, N, C = x.shape
    T, S = cfg.num_temporal, cfg.num_spatial
    ln = lambda v: F.layer_norm(v, (C,), None, None, 1e-6)
    sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = (P[pre + "scale_shift_table"][None] + t0.reshape(B, 6, -1)).chunk(6, dim=1)
    x_m = ln(x) * (1 + sc_msa) + sh_msa
    x_s = attention(x_m.reshape(B * T, S, C), P, pre + "attn", cfg.num_heads).reshape(B, T * S, C)
    x = x + g_msa * x_s
    x_t = x.reshape(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
    if tpe is not None:
        x_t = x_t + tpe
    x_t = attention(x_t, P, pre + "attn_temp", cfg.num_heads).reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(B, T * S, C)
    x = x + g_msa * x_t                       # REFERENCE QUIRK: the temporal branch reuses gate_msa (stdit.py:114,122)
    x = x + cross_attention(x, y_packed, y_lens, P, pre + "cross_attn", cfg.num_heads)
    h = ln(x) * (1 + sc_mlp) + sh_mlp
    h = _lin(gelu_tanh(_lin(h, P, pre + "mlp.fc1")), P, pre + "mlp.fc2")
    return x + g_mlp * h


def stdit_forward(P, cfg: STDiTConfig, x, timestep, y, mask=None, y_embedding=None):
    """x [B, C, T, H, W], timestep [B], y [B, 1, L, caption_channels], mask [B, L] (1 = token present) -> fp32 [B, 2C, T, H, W].
    class_dropout_prob is 0 in this restatement (the reference's token_drop calls .cuda())"""
    B = x.shape[0]
    D = cfg.hidden_size
    pt, ph, pw = cfg.patch_size
    w = P["x_embedder.proj.weight"]
    h = F.conv3d(x, w, P["x_embedder.proj.bias"], stride=cfg.patch_size).flatten(2).transpose(1, 2)       # [B, T*S, D], tokens (t h w)
    T, S = cfg.num_temporal, cfg.num_spatial
    h = (h.reshape(B, T, S, D) + spatial_pos_embed(cfg).to(h.dtype)[None, None]).reshape(B, T * S, D)
    t = _lin(F.silu(_lin(timestep_embedding(timestep).to(h.dtype), P, "t_embedder.mlp.0")), P, "t_embedder.mlp.2")
    t0 = _lin(F.silu(t), P, "t_block.1")
    ye = _lin(gelu_tanh(_lin(y, P, "y_embedder.y_proj.fc1")), P, "y_embedder.y_proj.fc2")                 # [B, 1, L, D]
    if mask is not None:
        ysq = ye.squeeze(1)
        y_lens = [int(v) for v in mask.sum(dim=1)]
        y_packed = torch.cat([ysq[b][mask[b] != 0] for b in range(B)], dim=0)[None]
    else:
        y_lens = [ye.shape[2]] * B
        y_packed = ye.squeeze(1).reshape(1, -1, D)
    tpe = temporal_pos_embed(cfg).to(h.dtype)[None]
    for i in range(cfg.depth):
        h = stdit_block(h, y_packed[0], y_lens, t0, P, f"blocks.{i}.", cfg, tpe if i == 0 else None)
    shift, scale = (P["final_layer.scale_shift_table"][None] + t[:, None]).chunk(2, dim=1)
    h = F.layer_norm(h, (D,), None, None, 1e-6) * (1 + scale) + shift
    h = _lin(h, P, "final_layer.linear")
    Nt, Nh, Nw = cfg.input_size[0] // pt, cfg.input_size[1] // ph, cfg.input_size[2] // pw
    out = h.reshape(B, Nt, Nh, Nw, pt, ph, pw, cfg.out_channels).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return out.reshape(B, cfg.out_channels, Nt * pt, Nh * ph, Nw * pw).float()


# ---------------------------------------------------------------------------------------------------------------
# IDDPM loss with the learned-variance VB term
# ---------------------------------------------------------------------------------------------------------------
def schedule(timesteps: int = 1000):
    """get_named_beta_schedule("linear", T) (iddpm3d.py:100-125) through IDDPMScheduler.register_schedule (:188-290): betas are cast to
    float32 FIRST, every table is then float32 numpy arithmetic"""
    scale = 1000 / timesteps
    betas = np.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=np.float64).astype(np.float32)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])              # float64 from here on (python float prepended), as in the reference
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    T = lambda a: torch.tensor(a)
    return dict(betas=T(betas), alphas_cumprod=T(ac), sqrt_ac=T(np.sqrt(ac)), sqrt_1mac=T(np.sqrt(1.0 - ac)),
                posterior_log_variance_clipped=T(np.log(np.append(post_var[1], post_var[1:]))),
                coef1=T(betas * np.sqrt(ac_prev) / (1.0 - ac)), coef2=T((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)))


def _ext(a, t, like):
    """extract_into_tensor: NO dtype cast -- alphas_cumprod_prev = np.append(1.0, ...) is float64 in the reference, so the posterior
    tables are float64 and everything they touch (model mean, KL, decoder NLL) is promoted to float64; betas / sqrt tables stay float32"""
    return a[t].view(-1, *([1] * (like.dim() - 1)))


def normal_kl(mean1, logvar1, mean2, logvar2):
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + (mean1 - mean2) ** 2 * torch.exp(-logvar2))


def _approx_cdf(x):
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * torch.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, means, log_scales):
    """videotuna/utils/diffusion_utils.py:250-276"""
    centered = x - means
    inv = torch.exp(-log_scales)
    cdf_plus = _approx_cdf(inv * (centered + 1.0 / 255.0))
    cdf_min = _approx_cdf(inv * (centered - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    delta = cdf_plus - cdf_min
    return torch.where(x < -0.999, log_cdf_plus, torch.where(x > 0.999, log_one_minus, torch.log(delta.clamp(min=1e-12))))


def q_sample(x0, t, noise, sch):
    return _ext(sch["sqrt_ac"], t, x0) * x0 + _ext(sch["sqrt_1mac"], t, x0) * noise


def opensora_loss(model_output, x0, noise, t, sch):
    """p_losses (iddpm3d.py:1332-1413) for ModelMeanType.EPSILON / ModelVarType.LEARNED_RANGE / LossType.MSE, no frame mask.
    Returns (loss, mse, vb): loss = mean_b(mse_b + vb_b)."""
    C = x0.shape[1]
    x_t = q_sample(x0, t, noise, sch)
    eps_hat, var_values = torch.split(model_output, C, dim=1)
    # --- vb: the mean prediction is detached (:1366) ---
    e = eps_hat.detach()
    min_log = _ext(sch["posterior_log_variance_clipped"], t, x0)
    max_log = _ext(torch.log(sch["betas"]), t, x0)
    frac = (var_values + 1) / 2
    model_log_var = frac * max_log + (1 - frac) * min_log
    x_recon = e               # REFERENCE QUIRK (:497-500): EPSILON models skip predict_start_from_noise here, the raw eps_hat is "x0"
    model_mean = _ext(sch["coef1"], t, x0) * x_recon + _ext(sch["coef2"], t, x0) * x_t
    true_mean = _ext(sch["coef1"], t, x0) * x0 + _ext(sch["coef2"], t, x0) * x_t
    kl = normal_kl(true_mean, min_log, model_mean, model_log_var).flatten(1).mean(1) / math.log(2.0)
    nll = -discretized_gaussian_log_likelihood(x0, model_mean, 0.5 * model_log_var).flatten(1).mean(1) / math.log(2.0)
    vb = torch.where(t == 0, nll, kl)
    mse = ((noise - eps_hat) ** 2).flatten(1).mean(1)
    return (mse + vb).mean(), mse.mean(), vb.mean()
