#!/usr/bin/env python3
"""Block GEMM shapes of CogVideoX-2B (B = 4: 72 008 rows; B = 2: 35 552) and HunyuanVideo (10 456 rows, d 3072) through vt_gemm_bf16 of the library
named by VT355_LIB (A/B of two builds: run once per build, interleaved).  Checks the result against torch on a row sample first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
from vt355.ops import EPI_BIAS_GELU
dev = torch.device("cuda:0"); BF = torch.bfloat16
tag = os.path.basename(os.environ.get("VT355_LIB", "libvt355.so"))
def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
torch.manual_seed(0)
tot = 0.0
for name, M, N, K, epi in (("qkv B4", 72008, 5760, 1920, None), ("out B4", 72008, 1920, 1920, None), ("ff1 B4 gelu", 72008, 7680, 1920, "gelu"), ("ff2 B4", 72008, 1920, 7680, None),
                           ("qkv B2", 35552, 5760, 1920, None), ("ff1 B2 gelu", 35552, 7680, 1920, "gelu"), ("ff2 B2", 35552, 1920, 7680, None),
                           ("hy qkv", 10456, 9216, 3072, None), ("hy fc1 gelu", 10456, 12288, 3072, "gelu"), ("hy fc2", 10456, 3072, 12288, None), ("8192^3", 8192, 8192, 8192, None)):
    x = (torch.randn(M, K, device=dev) * 0.5).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF); b = torch.randn(N, device=dev).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    pre = torch.empty(M, N, dtype=BF, device=dev) if epi else None
    fn = (lambda: ops.gemm(x, w, y, b, epilogue=EPI_BIAS_GELU, pre_act_out=pre)) if epi else (lambda: ops.gemm(x, w, y, b))
    fn(); torch.cuda.synchronize()
    rows = torch.randint(0, M, (256,), device=dev)
    ref = x[rows].float() @ w.float().t() + b.float()
    if epi: ref = torch.nn.functional.gelu(ref, approximate="tanh")
    err = ((y[rows].float() - ref).norm() / ref.norm()).item()
    assert err < 1e-2, (name, err)
    ms = t(fn)
    tot += ms
    print(f"[{tag}] {name:12s} M={M:6d} N={N:5d} K={K:5d}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:6.0f} TF/s  (rel err {err:.1e})", flush=True)
print(f"[{tag}] total {tot:.3f} ms")
