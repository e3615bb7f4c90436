#!/usr/bin/env python3
"""where a variant's dQ differs from the shipped kernel's: per head and per 64-row tile (debug aid)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
from vt355._lib import PROTOTYPES, load_library
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = load_library()
exp = C.CDLL(os.path.join(ROOT, "videotuna-dev_amd", "libvt355_exp.so"))
suf, L = sys.argv[1], int(sys.argv[2])
B, S, H = 1, int(sys.argv[3]) if len(sys.argv) > 3 else 17776, int(sys.argv[4]) if len(sys.argv) > 4 else 30
d = H * 64
def entry(l, s_):
    fn = getattr(l, "vt_attn_bwd_hd64" + s_); fn.argtypes = PROTOTYPES["vt_attn_bwd_hd64"]; fn.restype = C.c_int
    sc = getattr(l, "vt_attn_bwd_set_chain" + s_); sc.argtypes = [C.c_int, C.c_int]; sc.restype = C.c_int
    wb = getattr(l, "vt_attn_bwd_chain_ws_bytes" + s_); wb.argtypes = [C.c_int] * 3; wb.restype = C.c_longlong
    return fn, sc, wb
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * d, device=dev).to(BF)
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
ops.attn_fwd(q, k, v, o, lse, B, H, S, q_prescaled=True)
do = torch.randn(B, S, d, device=dev).to(BF)
delta = torch.empty(B * H * S, device=dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for name, (fn, sc, wb), LL in (("ref", entry(lib, ""), 1), ("var", entry(exp, suf), L)):
    assert sc(LL, 0) == 0
    ws = torch.zeros(max(int(wb(B, H, S)), 4096), dtype=torch.uint8, device=dev)
    dq = torch.zeros(B, S, d, device=dev); dk = torch.empty(B, S, d, dtype=BF, device=dev); dv = torch.empty_like(dk)
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, S, q.stride(1), k.stride(1), v.stride(1), o.stride(1), do.stride(1),
            dq.stride(1), dk.stride(1), dv.stride(1), q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0),
            dq.stride(0), dk.stride(0), dv.stride(0), 0.125, 1, ws.data_ptr(), ws.numel(), st)
    assert rc == 0
    torch.cuda.synchronize()
    res[name] = dq
    print(name, "diag", ws[:64].view(torch.int32).tolist())
e = (res["var"] - res["ref"]).abs().view(S, H, 64).amax(-1)          # [S, H]
m = res["ref"].abs().max().item()
print("ref max", m, "err max", e.max().item())
print("per head max err:", [f"{x:.2e}" for x in e.amax(0).tolist()])
hh = int(e.amax(0).argmax())
t = e[:, hh]
nt = (S + 63) // 64
per_tile = [t[i * 64:(i + 1) * 64].max().item() for i in range(nt)]
bad = [i for i, x in enumerate(per_tile) if x > 1e-3 * m]
print(f"head {hh}: bad 64-row tiles ({len(bad)} of {nt}):", bad[:60])
if bad:
    i = bad[0]
    rows = t[i * 64:(i + 1) * 64]
    print("rows in first bad tile:", [f"{x:.1e}" for x in rows.tolist()])
