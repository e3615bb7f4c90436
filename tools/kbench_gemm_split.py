#!/usr/bin/env python3
"""Is one GEMM over 72 008 rows slower than two over 36 004?  (qkv / out / ff1 / ff2 of CogVideoX-2B at B = 4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, N, K in (("qkv", 5760, 1920), ("out", 1920, 1920), ("ff1", 7680, 1920), ("ff2", 1920, 7680)):
    for M in (17776, 35552, 53328, 71104, 72008):
        x = (torch.randn(M, K, device=dev) * 0.5).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF); b = torch.randn(N, device=dev).to(BF)
        y = torch.empty(M, N, dtype=BF, device=dev)
        one = t(lambda: ops.gemm(x, w, y, b))
        h = M // 2
        two = t(lambda: (ops.gemm(x[:h], w, y[:h], b), ops.gemm(x[h:], w, y[h:], b)))
        q = M // 4
        four = t(lambda: [ops.gemm(x[i * q:(i + 1) * q if i < 3 else M], w, y[i * q:(i + 1) * q if i < 3 else M], b) for i in range(4)])
        fl = 2.0 * M * N * K
        print(f"{name} M={M:6d}: one launch {one:6.3f} ms {fl / one / 1e9:5.0f} TF/s | two {two:6.3f} ms {fl / two / 1e9:5.0f} | four {four:6.3f} ms {fl / four / 1e9:5.0f}", flush=True)
