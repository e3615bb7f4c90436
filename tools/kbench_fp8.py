"""timing of vt_gemm_fp8 against the bf16 GEMM on HunyuanVideo's block shapes (d = 3072; 10 200 tokens of the shipped 544x960x17f recipe)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vt355 import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
for (M, N, K) in [(10240, 9216, 3072), (10240, 3072, 3072), (10240, 12288, 3072), (10240, 3072, 12288), (32768, 9216, 3072)]:
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    aq, sa = ops.quantize_fp8(a); wq, sw = ops.quantize_fp8(w)
    o = torch.empty(M, N, dtype=BF, device=dev)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    tf = t(lambda: ops.gemm_fp8(aq, wq, o, sa, sw)); tb = t(lambda: ops.gemm(a, w, o)); tq = t(lambda: ops.quantize_fp8(a))
    fl = 2.0 * M * N * K
    print(f"M {M} N {N} K {K}: fp8 {tf*1e3:.3f} ms = {fl/tf/1e12:.0f} TFLOP/s | bf16 {tb*1e3:.3f} ms = {fl/tb/1e12:.0f} TFLOP/s | activation quantise {tq*1e3:.3f} ms", flush=True)
