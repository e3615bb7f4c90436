#!/usr/bin/env python3
"""A/B timing of attention-backward variants (shipped + libvt355_exp.so suffixes), interleaved rounds in one process."""
import ctypes as C, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
from vt355._lib import PROTOTYPES, load_library
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = load_library()
exp = C.CDLL(os.path.join(ROOT, "videotuna-dev_amd", "libvt355_exp.so"))
# name -> (entry point, set_chain entry point, chain length)   usage: kbench_variants.py [B] suffix[:L] ...
def setter(l, suf):
    f = getattr(l, "vt_attn_bwd_set_chain" + suf); f.argtypes = [C.c_int, C.c_int]; f.restype = C.c_int
    return f
args = sys.argv[1:]
B = int(args.pop(0)) if args and args[0].isdigit() else 1
variants = {"shipped:1": (lib.vt_attn_bwd_hd64, setter(lib, ""), 1), "shipped:8": (lib.vt_attn_bwd_hd64, setter(lib, ""), 8)}
for a in args or ["_st0_ld16:8", "_st16_ld16:8", "_st0_ld17:8"]:
    suf, L = (a.split(":") + ["8"])[:2]
    if suf == "shipped":
        variants[a] = (lib.vt_attn_bwd_hd64, setter(lib, ""), int(L))
        continue
    fn = getattr(exp, "vt_attn_bwd_hd64" + suf)
    fn.argtypes = PROTOTYPES["vt_attn_bwd_hd64"]; fn.restype = C.c_int
    variants[a] = (fn, setter(exp, suf), int(L))
S, H = 17776, 30
d = H * 64
qkv = torch.randn(B, S, 3 * d, device=dev).to(BF)
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
ops.attn_fwd(q, k, v, o, lse, B, H, S, q_prescaled=True)
do = torch.randn(B, S, d, device=dev).to(BF)
delta = torch.empty(B * H * S, device=dev)
st = torch.cuda.current_stream().cuda_stream
def wsbytes(l, suf):
    f = getattr(l, "vt_attn_bwd_chain_ws_bytes" + suf); f.argtypes = [C.c_int] * 3; f.restype = C.c_longlong
    return int(f(B, H, S))
need = max([wsbytes(lib, "")] + [wsbytes(exp, a.split(":")[0]) for a in variants if a.startswith("_")])
ws = torch.empty(need, dtype=torch.uint8, device=dev)
def run(var, dq, dk, dv):
    fn, setc, L = var
    assert setc(L, 0) == 0
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, S, q.stride(1), k.stride(1), v.stride(1), o.stride(1), do.stride(1),
            dq.stride(1), dk.stride(1), dv.stride(1), q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0),
            dq.stride(0), dk.stride(0), dv.stride(0), 0.125, 1, ws.data_ptr(), ws.numel(), st)
    assert rc == 0, rc
outs = {}
for name, fn in variants.items():
    dq = torch.zeros(B, S, d, device=dev); dk = torch.empty(B, S, d, dtype=BF, device=dev); dv = torch.empty_like(dk)
    run(fn, dq, dk, dv); torch.cuda.synchronize()
    outs[name] = (dq.clone(), dk.clone(), dv.clone())
ref = outs["shipped:1"]
for name, (a, b_, c) in outs.items():
    e = [((x.float() - y.float()).abs().max() / y.float().abs().max()).item() for x, y in zip((a, b_, c), ref)]
    print(f"{name:16s} rel-max diff vs shipped:1: dq {e[0]:.2e} dk {e[1]:.2e} dv {e[2]:.2e}  err word {ops.attn_bwd_chain_error(ws)}", flush=True)
times = {n: [] for n in variants}
dq = torch.zeros(B, S, d, device=dev); dk = torch.empty(B, S, d, dtype=BF, device=dev); dv = torch.empty_like(dk)
for rnd in range(6):
    for name, fn in variants.items():
        a = torch.cuda.Event(enable_timing=True); b_ = torch.cuda.Event(enable_timing=True)
        a.record(); run(fn, dq, dk, dv); b_.record(); torch.cuda.synchronize()
        if rnd > 0: times[name].append(a.elapsed_time(b_))
        if rnd == 5: print(f"{name:16s} polls waiting for predecessor / successor: {ws[36:44].view(torch.int32).tolist()}", flush=True)
fl = 8.0 * S * S * d * B
for name, ts in times.items():
    ts.sort(); med = ts[len(ts) // 2]
    print(f"{name:16s} median {med:.3f} ms  min {ts[0]:.3f}  -> {fl/med/1e9:.0f} TF/s algorithmic", flush=True)
