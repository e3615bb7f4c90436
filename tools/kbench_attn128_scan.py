#!/usr/bin/env python3
"""attn128 forward / two-pass backward over sequence length and head count (where does the efficiency go at the recipe's 10 456 tokens?)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
hd = 128
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for S, H in ((10456, 24), (10456, 6), (10496, 24), (20912, 24), (20912, 6), (41824, 6), (41824, 3)):
    c = H * hd
    j = torch.randn(1, S, 3 * c, device=dev).to(BF)
    q, k, v = j[:, :, :c], j[:, :, c:2 * c], j[:, :, 2 * c:]
    o = torch.empty(1, S, c, dtype=BF, device=dev); lse = torch.empty(1, H, S, device=dev)
    do = torch.randn(1, S, c, device=dev).to(BF)
    dj = torch.empty(1, S, 3 * c, dtype=BF, device=dev)
    for ragged in (False, True):
        kv = torch.tensor([S - 56], dtype=torch.int32, device=dev) if ragged else None
        f = t(lambda: ops.attn128_fwd(q, k, v, o, lse, H, hd ** -0.5, kv_len=kv))
        ops.profile_reset(True)
        g = t(lambda: ops.attn128_bwd(q, k, v, o, do, lse, dj[:, :, :c], dj[:, :, c:2 * c], dj[:, :, 2 * c:], H, hd ** -0.5, kv_len=kv))
        fl = 4.0 * S * S * c
        print(f"S={S:6d} H={H:2d} ragged={int(ragged)}: fwd {f:7.3f} ms {fl / f / 1e9:5.0f} TF/s | bwd {g:7.3f} ms {2.0 * fl / g / 1e9:5.0f} TF/s alg (8 S^2 d H)", flush=True)
