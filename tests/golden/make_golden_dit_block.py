#!/usr/bin/env python3
"""Pins the COMPOSITION of a CogVideoX block and of the final layer against the in-tree SAT twin
(videotuna/models/cogvideo_sat/dit_video_concat.py): ``AdaLNMixin.layer_forward`` (577-666: the 12-way adaLN chunk order,
LayerNorm -> modulate on video and text separately, text-first concat, gated residuals), ``AdaLNMixin.attention_fn``
(674-701: per-head q/k LayerNorm placement) and ``FinalLayerMixin.final_forward`` (478-498: video rows only, LN ->
modulate(shift, scale) -> linear -> unpatchify (c o p q)).  The twin's methods are the reference's own code, imported here;
what they call on ``self.transformer.layers[i]`` (``sat``'s BaseTransformerLayer: layernorms, the QKV/dense linears around
``attention_fn``, the MLP) is absent third-party code and is supplied as plain torch stand-ins with seeded weights -- so the
fixture pins the wiring, the oracle's own attention / MLP primitives are pinned by opensora_primitives.npz.

Run in the build container only:  python tests/golden/make_golden_dit_block.py   -> tests/golden/sat_dit_block.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stubs + loader)


class _SatAttention(nn.Module):
    """stand-in for sat's SelfAttention module: fused QKV linear -> heads -> mixin.attention_fn -> merge -> dense"""

    def __init__(self, d, heads, mixin):
        super().__init__()
        self.query_key_value = nn.Linear(d, 3 * d)
        self.dense = nn.Linear(d, d)
        self.heads, self.mixin = heads, mixin

    def forward(self, x, mask, **kw):
        B, S, d = x.shape
        q, k, v = self.query_key_value(x).chunk(3, dim=-1)
        sp = lambda t: t.view(B, S, self.heads, d // self.heads).transpose(1, 2)

        def sdpa(q, k, v, mask, attention_dropout=None, log_attention_weights=None, scaling_attention_score=True, **kwargs):
            return F.scaled_dot_product_attention(q, k, v)

        ctx = self.mixin.attention_fn(sp(q), sp(k), sp(v), mask, old_impl=sdpa, **kw)
        return self.dense(ctx.transpose(1, 2).reshape(B, S, d))


class _SatMlp(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.dense_h_to_4h = nn.Linear(d, 4 * d)
        self.dense_4h_to_h = nn.Linear(4 * d, d)

    def forward(self, x, **kw):
        return self.dense_4h_to_h(F.gelu(self.dense_h_to_4h(x), approximate="tanh"))


def main():
    torch.manual_seed(20230211)
    sys.path.insert(0, MG.REF)
    sat_dir = MG.install_stubs()
    dvc = MG.load_file("ref_dit_video_concat", os.path.join(sat_dir, "dit_video_concat.py"))
    d, heads, te, St, B = 128, 2, 32, 5, 2
    T_, Hp, Wp = 2, 3, 4                       # patch grid -> 24 video tokens
    Sv = T_ * Hp * Wp
    mixin = dvc.AdaLNMixin(hidden_size=d, num_layers=1, time_embed_dim=te, compressed_num_frames=T_, qk_ln=True,
                           hidden_size_head=d // heads, elementwise_affine=True)
    layer = nn.Module()
    layer.input_layernorm = nn.LayerNorm(d, eps=1e-5)
    layer.post_attention_layernorm = nn.LayerNorm(d, eps=1e-5)
    layer.attention = _SatAttention(d, heads, mixin)
    layer.mlp = _SatMlp(d)
    mixin.transformer = SimpleNamespace(layers=[layer], layernorm_order="pre")
    with torch.no_grad():
        for m in (mixin, layer):
            for n, p in m.named_parameters():
                p.normal_(0.0, 0.2 if p.dim() > 1 else 0.3)
                if "layernorm" in n and n.endswith("weight"):
                    p.add_(1.0)
    hidden = torch.randn(B, St + Sv, d)
    emb = torch.randn(B, te)
    with torch.no_grad():
        out = mixin.layer_forward(hidden, None, text_length=St, layer_id=0, emb=emb)
    rec = {"hidden": hidden.numpy(), "emb": emb.numpy(), "out": out.numpy(), "St": np.int64(St), "heads": np.int64(heads)}
    for n, p in list(mixin.named_parameters()) + [("layer." + n, p) for n, p in layer.named_parameters() if not n.startswith("attention.mixin")]:
        if n.startswith("transformer"):
            continue
        rec["w." + n] = p.detach().numpy()

    # ---- final layer ----
    fin = dvc.FinalLayerMixin(hidden_size=d, time_embed_dim=te, patch_size=(1, 2, 2), out_channels=4, latent_width=2 * Wp,
                              latent_height=2 * Hp, elementwise_affine=True)
    with torch.no_grad():
        for n, p in fin.named_parameters():
            p.normal_(0.0, 0.2 if p.dim() > 1 else 0.3)
            if n == "norm_final.weight":
                p.add_(1.0)
        img = fin.final_forward(out, text_length=St, emb=emb, rope_T=T_, rope_H=Hp, rope_W=Wp)
    rec["final_out"] = img.numpy()
    rec["grid"] = np.array([T_, Hp, Wp])
    for n, p in fin.named_parameters():
        rec["f." + n] = p.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "sat_dit_block.npz"), **rec)
    print("wrote sat_dit_block.npz", {k: v.shape for k, v in rec.items()})


if __name__ == "__main__":
    main()
