#!/usr/bin/env python3
"""One step of one workgroup of the attention backward, in shader cycles: runs the VT_STAMP build (libvt355_exp.so, suffix _stamp) at
B=2, S=17776, H=30 and prints the s_memtime stamps of every wave relative to the step's first stamp."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355 import ops
from vt355._lib import PROTOTYPES, load_library
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = load_library()
exp = C.CDLL(os.path.join(ROOT, "videotuna-dev_amd", "libvt355_exp.so"))
suf = sys.argv[1] if len(sys.argv) > 1 else "_stamp"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fn = getattr(exp, "vt_attn_bwd_hd64" + suf); fn.argtypes = PROTOTYPES["vt_attn_bwd_hd64"]; fn.restype = C.c_int
setc = getattr(exp, "vt_attn_bwd_set_chain" + suf); setc.argtypes = [C.c_int, C.c_int]; setc.restype = C.c_int
wsb = getattr(exp, "vt_attn_bwd_chain_ws_bytes" + suf); wsb.argtypes = [C.c_int] * 3; wsb.restype = C.c_longlong
get = getattr(exp, "vt_attn_bwd_stamps" + suf); get.argtypes = [C.c_void_p]; get.restype = C.c_int
B, S, H = 2, 17776, 30
d = H * 64
qkv = torch.randn(B, S, 3 * d, device=dev).to(BF)
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
ops.attn_fwd(q, k, v, o, lse, B, H, S, q_prescaled=True)
do = torch.randn(B, S, d, device=dev).to(BF)
delta = torch.empty(B * H * S, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert setc(L, 0) == 0
ws = torch.empty(max(int(wsb(B, H, S)), 4096), dtype=torch.uint8, device=dev)
dq = torch.zeros(B, S, d, device=dev); dk = torch.empty(B, S, d, dtype=BF, device=dev); dv = torch.empty_like(dk)
def run():
    rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, S, q.stride(1), k.stride(1), v.stride(1), o.stride(1), do.stride(1),
            dq.stride(1), dk.stride(1), dv.stride(1), q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0),
            dq.stride(0), dk.stride(0), dv.stride(0), 0.125, 1, ws.data_ptr(), ws.numel(), st)
    assert rc == 0, rc
for it in range(3):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    buf = (C.c_uint * 256)()
    assert get(buf) == 0
    v_ = list(buf)
    print(f"run {it}: {ms:.3f} ms; role {v_[128]} item {v_[129]}")
    names = ["top", "loads issued", "S/dP MFMAs q0", "softmax q0", "dK/dV+dS q0", "S/dP MFMAs q1", "softmax q1", "dK/dV+dS q1", "tile stored", "barrier",
             "dQ MFMAs", "dQ out"]
    t0 = min(v_[w * 16] for w in range(8))
    print("wave " + " ".join(f"{n[:13]:>13s}" for n in names))
    for w in range(8):
        row = [(v_[w * 16 + i] - t0) & 0xffffffff for i in range(12)]
        print(f"{w:4d} " + " ".join(f"{x:13d}" if x < (1 << 30) else f"{'-':>13s}" for x in row))
