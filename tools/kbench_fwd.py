#!/usr/bin/env python3
"""A/B timing of the attention forward from two builds of the library in one process (interleaved rounds):
   usage: kbench_fwd.py [B] libA.so libB.so ...   (paths relative to videotuna-dev_amd/)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355._lib import PROTOTYPES
dev = torch.device("cuda:0"); BF = torch.bfloat16
args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 2
libs = {}
for a in args or ["libvt355.so"]:
    l = C.CDLL(os.path.join(ROOT, "videotuna-dev_amd", a))
    f = l.vt_attn_fwd_hd64; f.argtypes = PROTOTYPES["vt_attn_fwd_hd64"]; f.restype = C.c_int
    libs[a] = f
S, H = 17776, 30
d = H * 64
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * d, device=dev).to(BF)
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
st = torch.cuda.current_stream().cuda_stream
def run(f, o, lse):
    rc = f(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, S, q.stride(1), k.stride(1), v.stride(1), o.stride(1),
           q.stride(0), k.stride(0), v.stride(0), o.stride(0), 0.125, 1, st)
    assert rc == 0, rc
outs = {}
for n, f in libs.items():
    o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
    run(f, o, lse); torch.cuda.synchronize()
    outs[n] = (o, lse)
ref = next(iter(outs.values()))
for n, (o, lse) in outs.items():
    print(f"{n:28s} max |o - first| {(o.float() - ref[0].float()).abs().max().item():.3e}  max |lse - first| {(lse - ref[1]).abs().max().item():.3e}", flush=True)
times = {n: [] for n in libs}
o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
for rnd in range(8):
    for n, f in libs.items():
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); run(f, o, lse); b.record(); torch.cuda.synchronize()
        if rnd > 1: times[n].append(a.elapsed_time(b))
fl = 4.0 * S * S * d * B
for n, ts in times.items():
    ts.sort(); med = ts[len(ts) // 2]
    print(f"{n:28s} median {med:.3f} ms  min {ts[0]:.3f}  -> {fl / med / 1e9:.0f} TF/s", flush=True)
