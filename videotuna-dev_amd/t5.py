"""Frozen T5 text encoder on the vt355 kernels -- the step before the DiT.

Mirrors what the reference calls: ``transformers.T5EncoderModel`` as used by ``_get_t5_prompt_embeds``
(videotuna/models/cogvideo_hf/cogvideo_pl.py:254-286: ``self.cond_stage_model.transformer(text_input_ids)[0]`` -- ids only,
no attention mask) and by ``FrozenT5Embedder`` (videotuna/models/lvdm/modules/encoders/condition.py:61-98:
``self.transformer(input_ids=tokens).last_hidden_state``).  Same module tree and parameter names as the HF checkpoint
(``text_encoder/*.safetensors``: ``shared.weight``, ``encoder.block.N.layer.0.SelfAttention.{q,k,v,o}.weight`` ...), so a
real checkpoint loads with ``load_state_dict`` / ``from_pretrained``; forward only (the encoder is frozen).

Per layer: vt_rmsnorm_bf16 -> one fused QKV GEMM -> vt_attn_fwd_bias_hd64 (unscaled scores + the bucketed relative position
bias, shared by all layers) -> output GEMM with the residual add in its epilogue -> vt_rmsnorm_bf16 -> one fused wi_0|wi_1 GEMM -> vt_gated_gelu_bf16 ->
wo GEMM with the residual add in its epilogue (option use_splitk: the two narrow GEMMs as vt_gemm_splitk_f32 + vt_residual_cast_bf16).  At 226 tokens per prompt the encoder is bound by
streaming its weights once per batch (9.4 GB for T5-XXL).
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ops import BF16, EPI_GATED_RES


class T5Config:
    """the fields of the HF ``config.json`` this encoder reads (defaults: T5 v1.1 XXL, CogVideoX's text encoder)"""

    def __init__(self, vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
                 relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                 feed_forward_proj="gated-gelu", **unused):
        if d_kv != 64:
            raise ValueError(f"the attention kernel is built for d_kv = 64 (T5 v1.1 large / xl / xxl), got {d_kv}")
        if feed_forward_proj != "gated-gelu":
            raise ValueError(f"only the gated-GELU feed-forward of T5 v1.1 is implemented, got {feed_forward_proj!r}")
        self.vocab_size, self.d_model, self.d_kv, self.d_ff = vocab_size, d_model, d_kv, d_ff
        self.num_layers, self.num_heads = num_layers, num_heads
        self.relative_attention_num_buckets = relative_attention_num_buckets
        self.relative_attention_max_distance = relative_attention_max_distance
        self.layer_norm_epsilon = layer_norm_epsilon
        self.feed_forward_proj = feed_forward_proj


class _Norm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d, dtype=BF16), requires_grad=False)


def _linear(n_out, n_in):
    lin = nn.Linear(n_in, n_out, bias=False, dtype=BF16)
    lin.weight.requires_grad_(False)
    return lin


class _SelfAttention(nn.Module):
    def __init__(self, cfg: T5Config, has_bias_table: bool):
        super().__init__()
        inner = cfg.num_heads * cfg.d_kv
        self.q, self.k, self.v = _linear(inner, cfg.d_model), _linear(inner, cfg.d_model), _linear(inner, cfg.d_model)
        self.o = _linear(cfg.d_model, inner)
        if has_bias_table:
            self.relative_attention_bias = nn.Embedding(cfg.relative_attention_num_buckets, cfg.num_heads, dtype=BF16)
            self.relative_attention_bias.weight.requires_grad_(False)


class _Dense(nn.Module):
    def __init__(self, cfg: T5Config):
        super().__init__()
        self.wi_0, self.wi_1 = _linear(cfg.d_ff, cfg.d_model), _linear(cfg.d_ff, cfg.d_model)
        self.wo = _linear(cfg.d_model, cfg.d_ff)


class _AttnLayer(nn.Module):
    def __init__(self, cfg, first):
        super().__init__()
        self.SelfAttention = _SelfAttention(cfg, first)
        self.layer_norm = _Norm(cfg.d_model)


class _FFLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.DenseReluDense = _Dense(cfg)
        self.layer_norm = _Norm(cfg.d_model)


class _Block(nn.Module):
    def __init__(self, cfg, first):
        super().__init__()
        self.layer = nn.ModuleList([_AttnLayer(cfg, first), _FFLayer(cfg)])


class _Stack(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.block = nn.ModuleList([_Block(cfg, i == 0) for i in range(cfg.num_layers)])
        self.final_layer_norm = _Norm(cfg.d_model)


class EncoderOutput(tuple):
    """``out[0]`` (cogvideo_pl.py:276) and ``out.last_hidden_state`` (condition.py:94) both work"""

    @property
    def last_hidden_state(self):
        return self[0]


def relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """bidirectional T5 bucketing of (key position - query position): integer index arithmetic, done once per length"""
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


class T5EncoderModel(nn.Module):
    def __init__(self, config: Optional[T5Config] = None, **kw):
        super().__init__()
        self.config = config if config is not None else T5Config(**kw)
        c = self.config
        self.shared = nn.Embedding(c.vocab_size, c.d_model, dtype=BF16)
        self.shared.weight.requires_grad_(False)
        self.encoder = _Stack(c)
        self._packed = None
        self._bias_cache = {}
        self._graphs = {}
        self.use_graph = True
        self.use_splitk = False        # measured slower than the plain producer / consumer GEMM at 2 x 226 tokens (11.2 vs 10.0 ms)

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: str = "", **unused):
        """local directory in the HF layout: config.json + model.safetensors (or the sharded index); nothing is fetched"""
        from safetensors.torch import load_file
        root = os.path.join(pretrained_model_name_or_path, subfolder)
        with open(os.path.join(root, "config.json")) as f:
            model = cls(T5Config(**json.load(f)))
        idx = os.path.join(root, "model.safetensors.index.json")
        if os.path.exists(idx):
            with open(idx) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        else:
            files = ["model.safetensors"]
        sd = {}
        for fn in files:
            sd.update(load_file(os.path.join(root, fn)))
        model.load_state_dict(sd)
        return model

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        sd.pop("encoder.embed_tokens.weight", None)         # tied to shared.weight in the HF checkpoint
        sd = {k: v.to(BF16) for k, v in sd.items()}
        self._packed = None
        self._bias_cache = {}
        self._graphs = {}
        return super().load_state_dict(sd, strict=strict, **kw)

    def init_weights(self, seed: int = 0, std: float = 0.05):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                v = torch.randn(p.shape, generator=g)
                if name.endswith("layer_norm.weight"):
                    v = 1.0 + 0.1 * v
                elif name != "shared.weight" and "relative_attention_bias" not in name:
                    v = v * std
                p.copy_(v.to(p.dtype))
        self._packed = None
        self._bias_cache = {}
        self._graphs = {}
        return self

    @property
    def dtype(self):
        return self.shared.weight.dtype

    @property
    def device(self):
        return self.shared.weight.device

    def _pack(self):
        """fused [Wq; Wk; Wv] and [Wi0; Wi1] per layer (the encoder is frozen: built once, rebuilt after load_state_dict)"""
        if self._packed is None or self._packed[0].device != self.device:
            qkv, wi = [], []
            for blk in self.encoder.block:
                at, ff = blk.layer[0].SelfAttention, blk.layer[1].DenseReluDense
                qkv.append(torch.cat([at.q.weight, at.k.weight, at.v.weight], 0).contiguous())
                wi.append(torch.cat([ff.wi_0.weight, ff.wi_1.weight], 0).contiguous())
            self._packed = (self.shared.weight, qkv, wi)
        return self._packed[1], self._packed[2]

    def position_bias_t(self, S: int) -> torch.Tensor:
        """fp32 [H, S keys, S queries]: table[bucket(key - query)] transposed for the attention kernel's lane layout"""
        b = self._bias_cache.get(S)
        if b is None or b.device != self.device:
            pos = torch.arange(S)
            c = self.config
            bucket = relative_position_bucket(pos[None, :] - pos[:, None], c.relative_attention_num_buckets,
                                              c.relative_attention_max_distance).to(self.device)        # [query, key]
            table = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.float()
            b = self._bias_cache[S] = table[bucket].permute(2, 1, 0).contiguous()                        # [H, key, query]
        return b

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, **unused) -> EncoderOutput:
        """``use_graph`` (attribute, default True): the ~240 launches of a forward are short (the whole encoder streams its
        weights in a few ms), so a forward of a given [B, S] is captured once into a HIP graph and replayed afterwards."""
        if attention_mask is not None:
            raise NotImplementedError("the reference calls the encoder without a mask (cogvideo_pl.py:276); masks are not built")
        if self.dtype != BF16:
            raise TypeError("the encoder runs in bf16; call .to(torch.bfloat16)")
        ids = input_ids.to(self.device)
        if not (self.use_graph and ids.is_cuda) or torch.cuda.is_current_stream_capturing():
            return self._forward(ids)
        key = (tuple(ids.shape), str(self.device))
        ent = self._graphs.get(key)
        if ent is None:
            self._pack(); self.position_bias_t(ids.shape[1])       # one-time allocations stay outside the capture
            static_ids = ids.clone()
            self._forward(static_ids)                                # warm-up on the side stream torch.cuda.graph requires
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._forward(static_ids)[0]
            ent = self._graphs[key] = (graph, static_ids, static_out)
        graph, static_ids, static_out = ent
        static_ids.copy_(ids)
        graph.replay()
        return EncoderOutput((static_out.clone(),))

    def _forward(self, ids: torch.Tensor) -> EncoderOutput:
        c = self.config
        dev = self.device
        B, S = ids.shape
        M, d, H, F = B * S, c.d_model, c.num_heads, c.d_ff
        inner = H * c.d_kv
        wqkv, wi = self._pack()
        bias_t = self.position_bias_t(S)
        E = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        h = torch.nn.functional.embedding(ids.reshape(-1), self.shared.weight)          # gather (plumbing): [M, d]
        x, qkv, o, u, g = E(M, d), E(M, 3 * inner), E(M, inner), E(M, 2 * F), E(M, F)
        lse = E(B, H, S, dt=torch.float32)
        # option (use_splitk): the two GEMMs with only d / 128 column tiles run split-K over the chip (fp32 partial sums, residual
        # added by the finishing pass).  Off by default: at M = 452 the extra epilogues and atomics cost more than the idle CUs.
        splitk = M <= 1024 and self.use_splitk
        acc = E(M, d, dt=torch.float32) if splitk else None
        qkv3, o3 = qkv.view(B, S, 3 * inner), o.view(B, S, inner)
        for blk, w_qkv, w_i in zip(self.encoder.block, wqkv, wi):
            at, ff = blk.layer[0], blk.layer[1]
            ops.rmsnorm(h, at.layer_norm.weight, x, c.layer_norm_epsilon)
            ops.gemm(x, w_qkv, qkv)
            ops.attn_fwd_bias(qkv3[:, :, :inner], qkv3[:, :, inner:2 * inner], qkv3[:, :, 2 * inner:], bias_t, o3, lse, B, H, S, 1.0)
            h1 = E(M, d)
            if splitk:
                ops.gemm_splitk(o, at.SelfAttention.o.weight, acc)
                ops.residual_cast(acc, h, h1)
            else:
                ops.gemm(o, at.SelfAttention.o.weight, h1, None, epilogue=EPI_GATED_RES, residual=h)
            ops.rmsnorm(h1, ff.layer_norm.weight, x, c.layer_norm_epsilon)
            ops.gemm(x, w_i, u)
            ops.gated_gelu(u, g)
            h = E(M, d)
            if splitk:
                ops.gemm_splitk(g, ff.DenseReluDense.wo.weight, acc)
                ops.residual_cast(acc, h1, h)
            else:
                ops.gemm(g, ff.DenseReluDense.wo.weight, h, None, epilogue=EPI_GATED_RES, residual=h1)
        out = E(M, d)
        ops.rmsnorm(h, self.encoder.final_layer_norm.weight, out, c.layer_norm_epsilon)
        return EncoderOutput((out.view(B, S, d),))


class FrozenT5Embedder(nn.Module):
    """Drop-in for videotuna.models.lvdm.modules.encoders.condition.FrozenT5Embedder (:61-98): ``.tokenizer`` +
    ``.transformer``; ``forward(text) -> last_hidden_state``.  Tokenisation is not arithmetic and stays the HF tokenizer's:
    it is loaded from the local ``version`` directory when ``tokenizer`` is not handed in (nothing is fetched)."""

    def __init__(self, version: str = "google/t5-v1_1-xxl", device: str = "cuda", max_length: int = 77, freeze: bool = True,
                 tokenizer=None, transformer: Optional[T5EncoderModel] = None):
        super().__init__()
        if tokenizer is None:
            from transformers import T5Tokenizer
            tokenizer = T5Tokenizer.from_pretrained(version, local_files_only=True)
        self.tokenizer = tokenizer
        self.transformer = transformer if transformer is not None else T5EncoderModel.from_pretrained(version)
        self.device = device
        self.max_length = max_length
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(self.device)
        return self.transformer(input_ids=tokens).last_hidden_state

    def encode(self, text):
        return self(text)
