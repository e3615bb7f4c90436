"""CPU ORACLE (test infrastructure, NOT product code) for the frozen T5 text encoder -- the step before the DiT.

Only ``tests/`` may import this file; nothing under ``videotuna-dev_amd/`` does.

What it follows:

* call sites (reference, read as text) ... videotuna/models/cogvideo_hf/cogvideo_pl.py:254-286
                                           (``self.cond_stage_model.transformer(text_input_ids)[0]`` -- input ids only:
                                           NO attention mask, all 226 padded positions attend to each other) and
                                           videotuna/models/lvdm/modules/encoders/condition.py:61-95 (FrozenT5Embedder)
* arithmetic ............................. THIRD-PARTY, absent from /root/reference: ``transformers==4.46.2``
                                           (poetry.lock:5638) ``T5EncoderModel`` (models/t5/modeling_t5.py), restated here
                                           from its published algorithm (T5 v1.1: pre-RMSNorm, unscaled dot-product attention
                                           plus a bucketed relative position bias shared by all layers, gated-GELU MLP,
                                           final RMSNorm; no biases anywhere).

PARITY STATUS: pinned.  tests/golden/t5_tiny.npz holds seeded weights, input ids and the output of transformers'
own ``T5EncoderModel`` (tests/golden/make_golden_t5.py; generated with the transformers 5.15.0 that is importable in the build
container -- the encoder's arithmetic is unchanged since 4.46.2), and tests/test_oracle_golden.py checks this restatement
against it.  Real T5-XXL weights are not available offline: the full-size check runs on seeded random weights.

Parameter names are the HF ``text_encoder`` safetensors keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch


@dataclass
class T5Config:
    vocab_size: int = 32128
    d_model: int = 4096
    d_kv: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


def tiny_config(**kw) -> T5Config:
    base = dict(vocab_size=97, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)
    base.update(kw)
    return T5Config(**base)


def relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """bidirectional bucketing (encoder): half of the buckets per sign; within a sign, half exact, half log-spaced"""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


def position_bias(table: torch.Tensor, S: int, cfg: T5Config) -> torch.Tensor:
    """table [num_buckets, H] -> bias [H, S(query), S(key)]; relative position = key - query"""
    pos = torch.arange(S)
    rel = pos[None, :] - pos[:, None]
    b = relative_position_bucket(rel, cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    return table[b].permute(2, 0, 1)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def init_params(cfg: T5Config, seed: int = 0, dtype=torch.float32, std: float = 0.05) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    inner = cfg.num_heads * cfg.d_kv
    P = {"shared.weight": torch.randn(cfg.vocab_size, cfg.d_model, generator=g)}
    for i in range(cfg.num_layers):
        a = f"encoder.block.{i}.layer.0."
        for n in ("q", "k", "v"):
            P[a + f"SelfAttention.{n}.weight"] = torch.randn(inner, cfg.d_model, generator=g) * std
        P[a + "SelfAttention.o.weight"] = torch.randn(cfg.d_model, inner, generator=g) * std
        P[a + "layer_norm.weight"] = 1.0 + 0.1 * torch.randn(cfg.d_model, generator=g)
        f = f"encoder.block.{i}.layer.1."
        P[f + "DenseReluDense.wi_0.weight"] = torch.randn(cfg.d_ff, cfg.d_model, generator=g) * std
        P[f + "DenseReluDense.wi_1.weight"] = torch.randn(cfg.d_ff, cfg.d_model, generator=g) * std
        P[f + "DenseReluDense.wo.weight"] = torch.randn(cfg.d_model, cfg.d_ff, generator=g) * std
        P[f + "layer_norm.weight"] = 1.0 + 0.1 * torch.randn(cfg.d_model, generator=g)
    P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = torch.randn(
        cfg.relative_attention_num_buckets, cfg.num_heads, generator=g)
    P["encoder.final_layer_norm.weight"] = 1.0 + 0.1 * torch.randn(cfg.d_model, generator=g)
    return {k: v.to(dtype) for k, v in P.items()}


def encoder_forward(P: Dict[str, torch.Tensor], cfg: T5Config, input_ids: torch.Tensor) -> torch.Tensor:
    """input_ids [B, S] int64 -> last_hidden_state [B, S, d_model] (dtype of the parameters)"""
    B, S = input_ids.shape
    H, dk = cfg.num_heads, cfg.d_kv
    h = P["shared.weight"][input_ids]
    bias = position_bias(P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], S, cfg)
    for i in range(cfg.num_layers):
        a = f"encoder.block.{i}.layer.0."
        x = rmsnorm(h, P[a + "layer_norm.weight"], cfg.layer_norm_epsilon)
        q = (x @ P[a + "SelfAttention.q.weight"].T).view(B, S, H, dk).transpose(1, 2)
        k = (x @ P[a + "SelfAttention.k.weight"].T).view(B, S, H, dk).transpose(1, 2)
        v = (x @ P[a + "SelfAttention.v.weight"].T).view(B, S, H, dk).transpose(1, 2)
        s = q @ k.transpose(-1, -2) + bias[None]                # no 1/sqrt(d_kv): folded into the T5 initialisation
        o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, H * dk)
        h = h + o @ P[a + "SelfAttention.o.weight"].T
        f = f"encoder.block.{i}.layer.1."
        x = rmsnorm(h, P[f + "layer_norm.weight"], cfg.layer_norm_epsilon)
        u = gelu_new(x @ P[f + "DenseReluDense.wi_0.weight"].T) * (x @ P[f + "DenseReluDense.wi_1.weight"].T)
        h = h + u @ P[f + "DenseReluDense.wo.weight"].T
    return rmsnorm(h, P["encoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)
