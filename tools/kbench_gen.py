#!/usr/bin/env python3
"""vt_attn_gen (the generic head-dim attention) at the HunyuanVideo recipe's sequence: 10 200 image + 256 text tokens, 24 heads x 128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops          # VT355_LIB=<path> selects another build of the library (ablations)
dev = torch.device("cuda:0"); BF = torch.bfloat16
B, H, hd = 1, 24, 128
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10456
C = H * hd
qkv = torch.randn(B, S, 3 * C, device=dev).to(BF)
q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
o = torch.empty(B, S, C, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
do = torch.randn(B, S, C, device=dev).to(BF)
dq = torch.empty(B, S, C, dtype=BF, device=dev); dk = torch.empty(B, S, C, device=dev); dv = torch.empty(B, S, C, device=dev)
kv_len = torch.tensor([S - 56], dtype=torch.int32, device=dev)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
f = t(lambda: ops.attn_gen_fwd(q, k, v, o, lse, H, hd, hd, hd ** -0.5, kv_len=kv_len))
g = t(lambda: ops.attn_gen_bwd(q, k, v, o, do, lse, dq, dk, dv, H, hd, hd, hd ** -0.5, kv_len=kv_len))
fl = 4.0 * S * S * C * B
print(f"attn_gen hd128 S={S} H={H}: fwd {f:.2f} ms = {fl / f / 1e9:.0f} TF/s; bwd {g:.2f} ms = {2 * fl / g / 1e9:.0f} TF/s algorithmic")
dq32 = torch.empty(B, S, C, device=dev); dkb = torch.empty(B, S, C, dtype=BF, device=dev); dvb = torch.empty(B, S, C, dtype=BF, device=dev)
f2 = t(lambda: ops.attn128_fwd(q, k, v, o, lse, H, hd ** -0.5, kv_len=kv_len))
g2 = t(lambda: ops.attn128_bwd(q, k, v, o, do, lse, dq32, dkb, dvb, H, hd ** -0.5, kv_len=kv_len))
print(f"attn128  hd128 S={S} H={H}: fwd {f2:.2f} ms = {fl / f2 / 1e9:.0f} TF/s; one-pass bwd (fp32 dQ atomics) {g2:.2f} ms = {2 * fl / g2 / 1e9:.0f} TF/s algorithmic (incl. delta pre-pass and dQ memset)")
dqb = torch.empty(B, S, C, dtype=BF, device=dev)
g3 = t(lambda: ops.attn128_bwd(q, k, v, o, do, lse, dqb, dkb, dvb, H, hd ** -0.5, kv_len=kv_len))
print(f"attn128  hd128 S={S} H={H}: two-pass bwd (dK/dV pass + dQ pass) {g3:.2f} ms = {2 * fl / g3 / 1e9:.0f} TF/s algorithmic (incl. delta pre-pass)")
