#!/usr/bin/env python3
"""One rank's attention of an 8-way Ulysses split of the HunyuanVideo 720p x 129-frame sequence (SURVEY 8(e), configs[4]): 33 x 45 x 80 =
118 800 image tokens + 256 text tokens, 24 heads / 8 ranks = 3 heads on this rank (vt355.sp hands vt_attn128 exactly this operand:
[1, S, 3 x 3 x 128] with q | k | v column blocks).  Times forward and the two-pass backward and checks 64 sampled output rows against a dense
fp32 statement (the whole matrix does not fit a test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Li, Lt, H, hd = 118800, 256, 24, 128
h, S = H // P, Li + Lt
c = h * hd
torch.manual_seed(0)
j2 = torch.randn(1, S, 3 * c, device=dev).to(BF)
q, k, v = j2[:, :, :c], j2[:, :, c:2 * c], j2[:, :, 2 * c:]
o = torch.empty(1, S, c, dtype=BF, device=dev); lse = torch.empty(1, h, S, device=dev)
do = torch.randn(1, S, c, device=dev).to(BF)
dj = torch.empty(1, S, 3 * c, dtype=BF, device=dev)
kv_len = torch.tensor([S - 56], dtype=torch.int32, device=dev)
scale = hd ** -0.5
def t(fn, n=2):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
f = t(lambda: ops.attn128_fwd(q, k, v, o, lse, h, scale, kv_len=kv_len))
print(f"forward done: {f:.1f} ms", flush=True)
g = t(lambda: ops.attn128_bwd(q, k, v, o, do, lse, dj[:, :, :c], dj[:, :, c:2 * c], dj[:, :, 2 * c:], h, scale, kv_len=kv_len))
fl = 4.0 * S * S * c
print(f"attn128 at S = {S} (720p x 129 frames), {h} of {H} heads ({P}-way Ulysses): fwd {f:.1f} ms = {fl / f / 1e9:.0f} TFLOP/s; "
      f"two-pass bwd {g:.1f} ms = {2.0 * fl / g / 1e9:.0f} TFLOP/s algorithmic (8 S^2 d h: dP, dV, dK, dQ; the recomputed S is not counted); x 60 blocks = {(f + g) * 60 / 1e3:.1f} s of attention per step and rank "
      f"(unsharded on one card: {P}x the heads = {(f + g) * 60 * P / 1e3:.0f} s)")
rows = torch.randint(0, S, (64,), device=dev)
worst = 0.0
for hh in range(h):
    qs = q[0, rows, hh * hd:(hh + 1) * hd].float()
    s = (qs @ k[0, :, hh * hd:(hh + 1) * hd].float().t()) * scale
    s[:, S - 56:] = float("-inf")
    ref = s.softmax(-1) @ v[0, :, hh * hd:(hh + 1) * hd].float()
    got = o[0, rows, hh * hd:(hh + 1) * hd].float()
    worst = max(worst, ((got - ref).norm() / ref.norm()).item())
print(f"64 sampled rows x {h} heads vs dense fp32: rel-L2 {worst:.2e}")
assert worst < 2e-2
