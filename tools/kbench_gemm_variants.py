#!/usr/bin/env python3
"""A/B timing of GEMM variants (shipped vs libvt355_exp.so suffixes) at the CogVideoX block shapes, one process."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vt355._lib import PROTOTYPES, load_library
dev = torch.device("cuda:0"); BF = torch.bfloat16
lib = load_library()
exp = C.CDLL(os.path.join(ROOT, "videotuna-dev_amd", "libvt355_exp.so"))
variants = {"shipped": lib.vt_gemm_bf16}
for suf in sys.argv[1:] or ["_nodma"]:
    fn = getattr(exp, "vt_gemm_bf16" + suf); fn.argtypes = PROTOTYPES["vt_gemm_bf16"]; fn.restype = C.c_int
    variants[suf] = fn
st = torch.cuda.current_stream().cuda_stream
S = 17776
for (M, N, K, name) in [(S, 5760, 1984, "qkv"), (S, 1920, 1984, "out"), (S, 7680, 1920, "ff1"), (S, 1920, 7680, "ff2"),
                        (2 * S, 5760, 1984, "qkv_B2"), (2 * S, 1920, 7680, "ff2_B2"), (8192, 8192, 8192, "sq8k")]:
    a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF); b = torch.randn(N, device=dev).to(BF)
    outs = {}
    times = {n: [] for n in variants}
    for name_v, fn in variants.items():
        out = torch.empty(M, N, dtype=BF, device=dev)
        def run():
            rc = fn(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, 0, None, 0, 0, None, None, 0, 1, 0,
                    None, 0, None, 0, st)
            assert rc == 0, rc
        run(); torch.cuda.synchronize(); outs[name_v] = out.clone()
        variants[name_v]._run = run
    for rnd in range(7):
        for name_v, fn in variants.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn._run(); e1.record(); torch.cuda.synchronize()
            if rnd: times[name_v].append(e0.elapsed_time(e1))
    ref = outs["shipped"].float()
    line = f"{name:7s} M={M} N={N} K={K}: "
    for name_v, ts in times.items():
        ts.sort(); med = ts[len(ts) // 2]
        err = (outs[name_v].float() - ref).abs().max().item()
        line += f"{name_v} {med:.3f} ms {2.0*M*N*K/med/1e9:.0f} TF/s (maxdiff {err:.1e})  "
    print(line, flush=True)
