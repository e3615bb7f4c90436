#!/usr/bin/env python3
"""Generates tests/golden/t5_tiny.npz: seeded weights + input ids + the output of transformers' own T5EncoderModel
(the third-party module the reference calls at cogvideo_pl.py:254-286; pinned 4.46.2 in its poetry.lock, 5.15.0 importable
here).  Run in the build container:  python tests/golden/make_golden_t5.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import t5_oracle as T                                               # only for the seeded weights / config
import transformers
from transformers import T5Config, T5EncoderModel


def main():
    cfg = T.tiny_config()
    P = T.init_params(cfg, seed=3, dtype=torch.float32)
    hf = T5EncoderModel(T5Config(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff,
                                 num_layers=cfg.num_layers, num_heads=cfg.num_heads, feed_forward_proj="gated-gelu",
                                 relative_attention_num_buckets=cfg.relative_attention_num_buckets,
                                 relative_attention_max_distance=cfg.relative_attention_max_distance,
                                 layer_norm_epsilon=cfg.layer_norm_epsilon, dropout_rate=0.0)).eval()
    sd = dict(P)
    sd["encoder.embed_tokens.weight"] = P["shared.weight"]
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and all("embed_tokens" in m or "shared" in m for m in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, cfg.vocab_size, (2, 150), generator=g)     # 150 > max_distance: exercises the log-spaced and clamped buckets
    with torch.no_grad():
        out = hf(input_ids=ids)[0]                                    # no attention mask, as the reference calls it
        bias = hf.encoder.block[0].layer[0].SelfAttention.compute_bias(150, 150)[0]
    np.savez_compressed(os.path.join(HERE, "t5_tiny.npz"), ids=ids.numpy(), out=out.numpy().astype(np.float32),
                        bias=bias.numpy().astype(np.float32), transformers_version=np.array(transformers.__version__),
                        **{"P." + k: v.numpy() for k, v in P.items()})
    print("wrote t5_tiny.npz", out.shape, float(out.abs().max()))


if __name__ == "__main__":
    main()
