#!/usr/bin/env python3
"""Kernel micro-benchmarks at the CogVideoX-2B shapes (random data, HIP events on the launch stream)."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vt355 import ops

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2], ts[0]


def main():
    which = sys.argv[1:] or ["gemm", "attn", "mem"]
    S, H, d = 17776, 30, 1920
    res = {}
    if "attn128q" in which:   # HunyuanVideo attention (head_dim 128, 10 456 tokens, 24 heads), few launches: a PMC target
        S1, H1 = 10456, 24
        c1 = H1 * 128
        j = torch.randn(1, S1, 3 * c1, device=dev).to(BF)
        o1 = torch.empty(1, S1, c1, dtype=BF, device=dev); l1 = torch.empty(1, H1, S1, device=dev)
        do1 = torch.randn(1, S1, c1, device=dev).to(BF); dj = torch.empty(1, S1, 3 * c1, dtype=BF, device=dev)
        kv = torch.tensor([S1 - 56], dtype=torch.int32, device=dev)
        for _ in range(3):
            ops.attn128_fwd(j[:, :, :c1], j[:, :, c1:2 * c1], j[:, :, 2 * c1:], o1, l1, H1, 128 ** -0.5, kv_len=kv)
            ops.attn128_bwd(j[:, :, :c1], j[:, :, c1:2 * c1], j[:, :, 2 * c1:], o1, do1, l1, dj[:, :, :c1], dj[:, :, c1:2 * c1], dj[:, :, 2 * c1:], H1, 128 ** -0.5, kv_len=kv)
        torch.cuda.synchronize()
        return
    if "attnq" in which:      # attention forward + backward, few launches: the PMC target (tools/pmc_kbench.sh)
        B = 2
        qkv = torch.randn(B, S, 3 * d, device=dev).to(BF)
        q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
        do = torch.randn(B, S, d, device=dev).to(BF)
        dq = torch.zeros(B, S, d, device=dev); dk = torch.empty(B, S, d, dtype=BF, device=dev); dv = torch.empty_like(dk)
        delta = torch.empty(B * H * S, device=dev)
        ws = ops.attn_bwd_chain_workspace(B, H, S, dev)
        for _ in range(3):
            ops.attn_fwd(q, k, v, o, lse, B, H, S, q_prescaled=True)
            ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, S, q_prescaled=True, chain_ws=ws)
        torch.cuda.synchronize()
    if "gemmq" in which:      # one shape, both tilings, few launches: the PMC target (tools/pmc_kbench.sh)
        M, N, K = 2 * S, 5760, 1984
        a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev); b = torch.zeros(N, dtype=BF, device=dev)
        for mode in (1, 2, 3):
            ops.gemm_set_tile(mode)
            for _ in range(4):
                ops.gemm(a, w, out, b)
        torch.cuda.synchronize()
        ops.gemm_set_tile(0)
    if "gemm" in which:
        for (M, N, K, name) in [(2 * S, 5760, 1984, "qkv2"), (2 * S, 1920, 1984, "out2"), (2 * S, 7680, 1920, "ff1_2"), (2 * S, 1920, 7680, "ff2_2"), (2 * S, 1984, 5760, "dqkv2"), (2 * S, 1984, 1920, "dout2"),
                                (S, 5760, 1984, "qkv"), (S, 1920, 1984, "out"), (S, 7680, 1920, "ff1"), (S, 1920, 7680, "ff2"),
                                (S, 1984, 5760, "dqkv"), (4096, 4096, 4096, "sq4k"), (8192, 8192, 8192, "sq8k")]:
            a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
            out = torch.empty(M, N, dtype=BF, device=dev); b = torch.zeros(N, dtype=BF, device=dev)
            for mode, tag in ((1, "tile128"), (2, "tile256"), (3, "tile_pc ")):
                ops.gemm_set_tile(mode)
                med, mn = timeit(lambda: ops.gemm(a, w, out, b))
                tf = 2.0 * M * N * K / med / 1e9
                res["gemm_" + name + "_" + tag] = dict(ms=med, tflops=tf)
                print(f"gemm {name:5s} {tag} M={M} N={N} K={K}: {med:.3f} ms  {tf:.0f} TF/s (min {mn:.3f})", flush=True)
            ops.gemm_set_tile(0)
            if name == "ff1":
                pre = torch.empty(M, N, dtype=BF, device=dev)
                med, mn = timeit(lambda: ops.gemm(a, w, out, b, epilogue=ops.EPI_BIAS_GELU, pre_act_out=pre))
                print(f"   +gelu epilogue: {med:.3f} ms {2.0*M*N*K/med/1e9:.0f} TF/s", flush=True)
    if "blas" in which:       # calibration only: the vendor libraries on the same shapes (hipBLASLt through torch.matmul, torch SDPA)
        import torch.nn.functional as F
        for (M, N, K, name) in [(2 * S, 5760, 1984, "qkv2"), (2 * S, 7680, 1920, "ff1_2"), (2 * S, 1920, 7680, "ff2_2"), (S, 5760, 1984, "qkv"), (8192, 8192, 8192, "sq8k")]:
            a = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
            med, mn = timeit(lambda: torch.matmul(a, w.t()))
            print(f"torch.matmul {name:5s} M={M} N={N} K={K}: {med:.3f} ms  {2.0*M*N*K/med/1e9:.0f} TF/s (min {mn:.3f})", flush=True)
        for (M, P, Q, name) in [(S, 5760, 1920, "dW_qkv"), (S, 7680, 1920, "dW_ff1")]:
            a = torch.randn(M, P, device=dev).to(BF); b = torch.randn(M, Q, device=dev).to(BF)
            med, mn = timeit(lambda: torch.matmul(a.t(), b))
            print(f"torch.matmul {name} (bf16 out) M={M} P={P} Q={Q}: {med:.3f} ms  {2.0*M*P*Q/med/1e9:.0f} TF/s", flush=True)
        try:
            q = torch.randn(1, 30, S, 64, device=dev).to(BF).requires_grad_(True)
            k = torch.randn(1, 30, S, 64, device=dev).to(BF).requires_grad_(True)
            v = torch.randn(1, 30, S, 64, device=dev).to(BF).requires_grad_(True)
            fl = 4.0 * S * S * 1920
            med, mn = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
            print(f"torch SDPA fwd B=1: {med:.3f} ms  {fl/med/1e9:.0f} TF/s", flush=True)
            o = F.scaled_dot_product_attention(q, k, v); go = torch.randn_like(o)
            med, mn = timeit(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
            print(f"torch SDPA bwd B=1: {med:.3f} ms  {2*fl/med/1e9:.0f} TF/s algorithmic", flush=True)
        except Exception as e:      # no flash backend in this torch build
            print("torch SDPA unavailable:", repr(e)[:200], flush=True)
    if "epi" in which:        # the two FF GEMMs with their fused GELU / dGELU epilogues at the bench micro-batch (B=2)
        M = 2 * S
        a = torch.randn(M, 1920, device=dev).to(BF); w = (torch.randn(7680, 1920, device=dev) * 0.02).to(BF)
        out = torch.empty(M, 7680, dtype=BF, device=dev); pre = torch.empty(M, 7680, dtype=BF, device=dev); b = torch.zeros(7680, dtype=BF, device=dev)
        g = torch.randn(M, 1920, device=dev).to(BF); w2 = (torch.randn(1920, 7680, device=dev) * 0.02).to(BF); w2t = w2.t().contiguous()
        fl = 2.0 * M * 7680 * 1920
        for mode, tag in ((1, "tile128"), (2, "tile256"), (3, "tile_pc "), (0, "auto    ")):
            ops.gemm_set_tile(mode)
            m0, _ = timeit(lambda: ops.gemm(a, w, out, b))
            m1, _ = timeit(lambda: ops.gemm(a, w, out, b, epilogue=ops.EPI_BIAS_GELU, pre_act_out=pre))
            m2, _ = timeit(lambda: ops.gemm(g, w2t, out, None, epilogue=ops.EPI_DGELU, pre_act_in=pre))
            print(f"ff1 {tag}: plain {m0:.3f} ms {fl/m0/1e9:.0f} TF/s | +GELU {m1:.3f} ms {fl/m1/1e9:.0f} | dX*dGELU {m2:.3f} ms {fl/m2/1e9:.0f}", flush=True)
        ops.gemm_set_tile(0)
        # attention / FF output projections with the gated residual epilogue (K = 1920 + 64 LoRA columns, and the FF one)
        res_ = torch.randn(M, 1920, device=dev).to(BF); gt = torch.randn(2, 1920, device=dev); gv = torch.randn(2, 1920, device=dev)
        o2 = torch.empty(M, 1920, dtype=BF, device=dev); b2 = torch.zeros(1920, dtype=BF, device=dev)
        for K_ in (1984, 7680):
            a2 = torch.randn(M, K_, device=dev).to(BF); w3 = (torch.randn(1920, K_, device=dev) * 0.02).to(BF)
            fl2 = 2.0 * M * 1920 * K_
            m0, _ = timeit(lambda: ops.gemm(a2, w3, o2, b2))
            m1, _ = timeit(lambda: ops.gemm(a2, w3, o2, b2, epilogue=ops.EPI_GATED_RES, residual=res_, gate_txt=gt, gate_vid=gv,
                                            gate_bstride=1920, S=S, St=226))
            print(f"proj K={K_}: plain {m0:.3f} ms {fl2/m0/1e9:.0f} TF/s | +gated residual {m1:.3f} ms {fl2/m1/1e9:.0f}", flush=True)
    if "t5" in which:         # the frozen text encoder at T5 v1.1 XXL size (random weights), 2 prompts x 226 tokens
        from vt355.t5 import T5EncoderModel
        with torch.device(dev):
            m = T5EncoderModel()
        gw = torch.Generator(device=dev).manual_seed(1)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                # q / k weights scaled so that the unscaled scores q.k have unit-order spread (trained T5 weights do: the
                # 1/sqrt(d_kv) is folded into its initialisation); at 0.02 the softmax is one-hot and any two bf16
                # implementations disagree chaotically
                p_.normal_(0.0, 0.006 if (".q.weight" in n_ or ".k.weight" in n_) else 0.02, generator=gw)
                if n_.endswith("layer_norm.weight"):
                    p_.add_(1.0)
        ids = torch.randint(0, 32128, (2, 226), device=dev)
        nparam = sum(p_.numel() for n_, p_ in m.named_parameters() if n_ != "shared.weight")
        for mode in (1, 3):     # which GEMM tiling streams cold weights best at M = 452 (the default picks by shape)
            ops.gemm_set_tile(mode)
            m.use_graph = False
            medm, _ = timeit(lambda: m(ids), iters=3, warm=1)
            print(f"T5-XXL encoder fwd with vt_gemm_set_tile({mode}): {medm:.2f} ms", flush=True)
        ops.gemm_set_tile(0)
        m.use_graph = False
        med0, _ = timeit(lambda: m(ids), iters=5, warm=2)
        print(f"T5-XXL encoder fwd, 2 x 226 tokens, launch by launch: {med0:.2f} ms", flush=True)
        m.use_graph = True
        med, mn = timeit(lambda: m(ids), iters=5, warm=2)
        print(f"T5-XXL encoder fwd, 2 x 226 tokens, HIP graph replay: {med:.2f} ms (min {mn:.2f})  weights streamed {nparam*2/1e9:.1f} GB -> {nparam*2/med/1e6:.0f} GB/s, "
              f"{2.0*nparam*452/med/1e9:.0f} TFLOP/s", flush=True)
        try:                      # calibration only: transformers' own module on the same device
            import transformers
            sd = {k: v for k, v in m.state_dict().items()}
            with torch.device(dev):
                hf = transformers.T5EncoderModel(transformers.T5Config(vocab_size=32128, d_model=4096, d_kv=64, d_ff=10240, num_layers=24,
                                                                       num_heads=64, feed_forward_proj="gated-gelu", dropout_rate=0.0)).to(BF).eval()
            hf.load_state_dict(sd, strict=False)
            with torch.no_grad():
                ref = hf(input_ids=ids)[0]
                out = m(ids)[0]
                print(f"   vs transformers {transformers.__version__} bf16 on the same weights: max |diff| {(out.float()-ref.float()).abs().max().item():.3f} "
                      f"of max |ref| {ref.float().abs().max().item():.3f}", flush=True)
                med2, _ = timeit(lambda: hf(input_ids=ids), iters=5, warm=2)
            print(f"   transformers T5EncoderModel bf16 (eager, rocBLAS/hipBLASLt): {med2:.2f} ms", flush=True)
        except Exception as e:
            print("   transformers calibration unavailable:", repr(e)[:200], flush=True)
    if "gn" in which:         # GroupNorm + SiLU, channels-last, at the VAE encoder's block sizes (49x480x720 input): 6 bytes per element
        for (P, C, name) in [(13 * 480 * 720, 128, "block0 (quarter of the 49 frames)"), (25 * 240 * 360, 256, "block1"), (13 * 120 * 180, 256, "block2"),
                             (13 * 60 * 90, 512, "block3 / mid")]:
            x = torch.randn(1, P, C, device=dev).to(BF); y = torch.empty_like(x)
            ga = torch.ones(C, dtype=BF, device=dev); be = torch.zeros(C, dtype=BF, device=dev)
            ws = torch.empty(4 * C, device=dev)
            med, mn = timeit(lambda: ops.groupnorm_silu(x, ga, be, y, 32, 1e-6, True, ws))
            print(f"groupnorm+silu {name}: P={P} C={C}: {med*1e3:.1f} us  {6.0*P*C/med/1e6:.0f} GB/s", flush=True)
    if "conv" in which:       # causal 3x3x3 convolution (implicit GEMM) at the CogVideoX VAE encoder's block sizes
        for (T_, H_, W_, Ci, Co, name) in [(4, 480, 720, 128, 128, "block0 (4 of 49 frames)"), (13, 240, 360, 256, 256, "block1 (13 of 25 frames)"),
                                           (13, 120, 180, 256, 256, "block2"), (13, 60, 90, 512, 512, "block3 / mid")]:
            x = torch.randn(1, T_, H_, W_, Ci, device=dev).to(BF); y = torch.empty(1, T_, H_, W_, Co, dtype=BF, device=dev)
            wk = (torch.randn(Co, 27 * Ci, device=dev) * 0.02).to(BF); b = torch.zeros(Co, dtype=BF, device=dev)
            med, mn = timeit(lambda: ops.causal_conv3d(x, wk, b, y), iters=5, warm=2)
            fl = 2.0 * T_ * H_ * W_ * Co * 27 * Ci
            print(f"causal conv3d {name}: {T_}x{H_}x{W_} {Ci}->{Co}: {med:.3f} ms  {fl/med/1e9:.0f} TF/s", flush=True)
    if "vae" in which:        # the CogVideoX VAE encoder (random weights) on one 49x480x720 clip -- or fewer frames: kbench.py vae 13
        from vt355.vae import CogVideoXVaeEncoder
        frames = int(next((a for a in sys.argv[1:] if a.isdigit()), "49"))
        with torch.device(dev):
            enc = CogVideoXVaeEncoder()
        gw = torch.Generator(device=dev).manual_seed(2)
        with torch.no_grad():
            for n_, p_ in enc.named_parameters():
                if p_.dim() > 1:
                    p_.normal_(0.0, 1.0 / p_[0].numel() ** 0.5, generator=gw)
                elif n_.endswith("weight"):
                    p_.fill_(1.0)
                else:
                    p_.zero_()
        x = torch.rand(1, 3, frames, 480, 720, device=dev).mul_(2).sub_(1).to(BF)
        # FLOPs of the convolutions / shortcut GEMMs at their true channel counts
        fl, T_, H_, W_ = 0.0, frames, 480, 720
        c = enc.config
        cin = c.in_channels
        fl += 2.0 * T_ * H_ * W_ * c.ch * 27 * cin
        prev = c.ch
        for i, mlt in enumerate(c.ch_mult):
            co = c.ch * mlt
            for j in range(c.num_res_blocks):
                ci = prev if j == 0 else co
                fl += 2.0 * T_ * H_ * W_ * (27 * ci * co + 27 * co * co + (ci * co if ci != co else 0))
            prev = co
            if i != len(c.ch_mult) - 1:
                if i < 2 and T_ > 1:
                    T_ = 1 + (T_ - 1) // 2
                H_, W_ = H_ // 2, W_ // 2
                fl += 2.0 * T_ * H_ * W_ * 9 * co * co
        fl += 2 * 2.0 * T_ * H_ * W_ * 2 * 27 * prev * prev + 2.0 * T_ * H_ * W_ * 27 * prev * 32
        torch.cuda.reset_peak_memory_stats()
        med, mn = timeit(lambda: enc(x), iters=3, warm=1)
        print(f"CogVideoX VAE encode, 1 x {frames}x480x720: {med:.1f} ms (min {mn:.1f})  {fl/1e12:.1f} TFLOP -> {fl/med/1e9:.0f} TFLOP/s, "
              f"peak HBM {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
    if "red" in which:        # token-axis reductions: bias / adaLN column sums (full fine-tune) and the LoRA skinny products
        M = 2 * S
        x = torch.randn(M, d, device=dev).to(BF); y = torch.randn(M, d, device=dev).to(BF)
        o1 = torch.zeros(4, d, device=dev); o2 = torch.zeros(4, d, device=dev)
        mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
        med, _ = timeit(lambda: ops.group_colsum(x, o1, D=d))
        print(f"group_colsum plain  M={M} D={d}: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        med, _ = timeit(lambda: ops.group_colsum(x, o1, y=y, out2=o2, mean=mean, rstd=rstd, D=d, S=S, St=226, grouped=True))
        print(f"group_colsum LN-grouped M={M} D={d}: {med*1e3:.1f} us  {2*M*d*2/med/1e6:.0f} GB/s", flush=True)
        t4 = torch.randn(M, 64, device=dev).to(BF)
        ob = torch.zeros(d * 4, device=dev)
        med, _ = timeit(lambda: ops.skinny_tn(x, t4, 4, ob, 4, 1, 0.25, d))
        print(f"skinny_tn R=4  M={M} P={d}: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        oa = torch.zeros(12 * d, device=dev)
        med, _ = timeit(lambda: ops.skinny_tn(x, t4, 12, oa, 1, d, 1.0, d))
        print(f"skinny_tn R=12 M={M} P={d}: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        med, _ = timeit(lambda: ops.skinny_tn(x, t4, 4, ob, 4, 1, 0.25, d, use_workspace=True))
        print(f"skinny_tn R=4  copies M={M} P={d}: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        med, _ = timeit(lambda: ops.skinny_tn(x, t4, 12, oa, 1, d, 1.0, d, use_workspace=True))
        print(f"skinny_tn R=12 copies M={M} P={d}: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
    if "nt" in which:
        for (M, P, Q, name) in [(S, 5760, 1920, "dW_qkv"), (S, 7680, 1920, "dW_ff1"), (S, 1920, 7680, "dW_ff2"), (2 * S, 5760, 1920, "dW_qkv_B2")]:
            a = torch.randn(M, P, device=dev).to(BF); b = torch.randn(M, Q, device=dev).to(BF)
            c = torch.zeros(P, Q, device=dev)
            med, mn = timeit(lambda: ops.gemm_nt(a, b, c))
            print(f"gemm_nt {name:9s} M={M} P={P} Q={Q}: {med:.3f} ms  {2.0*M*P*Q/med/1e9:.0f} TF/s", flush=True)
    if "attn" in which:
        B = 1
        qkv = torch.randn(B, S, 3 * d, device=dev).to(BF)
        q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        o = torch.empty(B, S, d, dtype=BF, device=dev); lse = torch.empty(B, H, S, device=dev)
        med0, _ = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, B, H, S, q_prescaled=False), iters=5, warm=2)
        med, mn = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, B, H, S, q_prescaled=True), iters=5, warm=2)
        fl = 4.0 * S * S * d * B
        print(f"attn fwd (classic online softmax, unscaled q): {med0:.3f} ms  {fl/med0/1e9:.0f} TF/s", flush=True)
        res["attn_fwd"] = dict(ms=med, tflops=fl / med / 1e9)
        print(f"attn fwd: {med:.3f} ms  {fl/med/1e9:.0f} TF/s (min {mn:.3f})", flush=True)
        do = torch.randn(B, S, d, device=dev).to(BF)
        dq = torch.zeros(B, S, d, device=dev); dk = torch.empty(B, S, d, dtype=BF, device=dev); dv = torch.empty_like(dk)
        delta = torch.empty(B * H * S, device=dev)
        ws = ops.attn_bwd_chain_workspace(B, H, S, dev)
        med, mn = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, S, q_prescaled=True, chain_ws=ws), iters=5, warm=2)
        print('chain ws', None if ws is None else (ws.numel(), ops.attn_bwd_chain_error(ws)), 'VT_BWD_CHAIN', os.environ.get('VT_BWD_CHAIN'), flush=True)
        fl2 = 2.5 * fl
        res["attn_bwd"] = dict(ms=med, tflops_exec=fl2 / med / 1e9, tflops_alg=2 * fl / med / 1e9)
        print(f"attn bwd: {med:.3f} ms  {fl2/med/1e9:.0f} TF/s executed (5 products), {2*fl/med/1e9:.0f} algorithmic (min {mn:.3f})", flush=True)
    if "mem" in which:
        M = S
        x = torch.randn(M, d + 64, device=dev).to(BF); y = torch.empty_like(x)
        ga = torch.ones(d, dtype=BF, device=dev); be = torch.zeros(d, dtype=BF, device=dev)
        mod = torch.randn(1, 6 * d, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
        m4 = (mod[:, 3 * d:], mod[:, 4 * d:], mod[:, 0:], mod[:, d:], 6 * d)
        med, _ = timeit(lambda: ops.ln_modulate_fwd(x, y, ga, be, m4, mean, rstd, d, S, 226, 1e-5))
        print(f"ln_modulate fwd: {med*1e3:.1f} us  {2*M*d*2/med/1e6:.0f} GB/s", flush=True)
        dx = torch.empty(M, d, dtype=BF, device=dev)
        med, _ = timeit(lambda: ops.ln_modulate_bwd(y, x, mean, rstd, ga, (mod[:, 4 * d:], mod[:, d:], 6 * d), dx, dx, d, S, 226))
        print(f"ln_modulate bwd: {med*1e3:.1f} us  {4*M*d*2/med/1e6:.0f} GB/s", flush=True)
        qkv = torch.randn(M, 3 * d, device=dev).to(BF); out = torch.empty(M, 2 * d, dtype=BF, device=dev)
        g64 = torch.ones(64, dtype=BF, device=dev); b64 = torch.zeros(64, dtype=BF, device=dev)
        mq = torch.empty(M, 2 * H, device=dev); rq = torch.empty(M, 2 * H, device=dev)
        med, _ = timeit(lambda: ops.qk_layernorm_fwd(qkv, out, g64, b64, g64, b64, mq, rq, H, 1e-6))
        print(f"qk_layernorm fwd: {med*1e3:.1f} us  {4*M*d*2/med/1e6:.0f} GB/s", flush=True)
        A = torch.randn(16, d, device=dev).to(BF)
        med, _ = timeit(lambda: ops.lora_down(x, A, 12, x[:, d:], d))
        print(f"lora_down: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        out16 = torch.zeros(16, d, device=dev)
        med, _ = timeit(lambda: ops.skinny_tn(x, x[:, d:], 12, out16, 1, d, 1.0, d))
        print(f"skinny_tn R=12 P=1920: {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        ob = torch.zeros(d, 4, device=dev)
        med, _ = timeit(lambda: ops.skinny_tn(qkv, x[:, d:], 4, ob, 4, 1, 0.25, d))
        print(f"skinny_tn R=4 P=1920 (ld 5760): {med*1e3:.1f} us  {M*d*2/med/1e6:.0f} GB/s", flush=True)
        med, _ = timeit(lambda: ops.lora_up_add(dx, x[:, d:], A, 12, d))
        print(f"lora_up_add: {med*1e3:.1f} us  {2*M*d*2/med/1e6:.0f} GB/s", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
