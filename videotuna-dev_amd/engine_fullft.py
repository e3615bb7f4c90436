"""Backward schedule for FULL fine-tuning of the CogVideoX DiT (every parameter trainable; BASELINE config 3).

Same block walk as engine.run_backward, plus every parameter gradient that ``loss.backward()`` produces in the reference
(cogvideo_pl.py:865-887 under PL): Linear weights (token-reduction MFMA GEMM ``vt_gemm_nt_bf16``), biases and adaLN
shift/scale/gate sums (``vt_group_colsum``), LayerNorm gamma/beta (``vt_ln_param_combine``), per-head q/k LayerNorm
(``vt_qk_ln_param_grads``), the adaLN / time-embedding MLPs that see one row per sample (``vt_small_linear_bwd``,
``vt_silu_bwd``), patch / text / output projections.  Gradients are ACCUMULATED into the flat fp32 buffer of
``FullFTState`` (zeroed by the optimizer), so micro-batch accumulation and the data-parallel reduction work on one
contiguous buffer; ``on_grads_ready(lo, hi)`` is called as soon as a slice of that buffer is final (bucketed overlap).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import ops
from .engine import _mod, block_forward, packed
from .ops import BF16, EPI_DGELU

F32 = torch.float32


def _dw(dy: torch.Tensor, x: torch.Tensor, gw: torch.Tensor, P: int, Q: int):
    """gw[P,Q] += dy[:, :P]^T @ x[:, :Q]  -- MFMA kernel when the tile shape allows, rank-16 sweeps otherwise (small dims)."""
    if P % 128 == 0 and Q % 128 == 0:
        ops.gemm_nt(dy, x, gw, P=P, Q=Q, alpha=1.0, accumulate=True)
    elif Q <= P:
        for c in range(0, Q, 16):          # out[p*Q + c + r] : Big = dy (P columns), Small = 16 columns of x
            r = min(16, Q - c)
            ops.skinny_tn(dy, x[:, c:], r, gw.view(-1)[c:], Q, 1, 1.0, P)
    else:
        for c in range(0, P, 16):          # out[(c + r)*Q + q] : Big = x (Q columns), Small = 16 columns of dy
            r = min(16, P - c)
            ops.skinny_tn(x, dy[:, c:], r, gw.view(-1)[c * Q:], 1, Q, 1.0, Q)


def run_backward_fullft(model, ctx, dout: torch.Tensor, on_grads_ready: Optional[Callable[[int, int], None]] = None):
    c = model.config
    P = packed(model)
    ft = model.fullft
    d, H, L = model.inner_dim, c.num_attention_heads, c.num_layers
    B, Fr, C, Hh, Ww, S, St, Sv, M = ctx.dims
    p = c.patch_size
    te = c.time_embed_dim
    dev = dout.device
    E = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
    Z = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
    mod = ctx.mod
    nmod = mod.shape[1]
    g = ft.g
    dmod = Z(B, nmod)                      # gradient of the whole modulation table (shift/scale/gate of every adaLN)
    G1, G2 = Z(2 * B, d), Z(2 * B, d)      # grouped-sum scratch

    def ready(first: str, last: str):
        if on_grads_ready is not None:
            o0, _ = ft.offsets[first]
            o1, s1 = ft.offsets[last]
            n1 = 1
            for s in s1:
                n1 *= s
            on_grads_ready(o0, o1 + n1)

    def dm(idx):                           # views of dmod laid out like engine._mod
        return _mod(dmod, idx, d)

    def ln_grads(dy, x, mean, rstd, gamma_name, beta_name, m_scales, dm_slots, grouped_S, grouped_St):
        """LayerNorm gamma/beta (+ adaLN shift/scale) gradients of y = LN(x)*(1+scale)+shift given dy."""
        G1.zero_(); G2.zero_()
        ops.group_colsum(dy, G1, y=x, out2=G2, mean=mean, rstd=rstd, D=d, S=grouped_S, St=grouped_St, grouped=True)
        gam = ft.view(ft.flat_bf16, gamma_name)
        bet = ft.view(ft.flat_bf16, beta_name)
        ops.ln_param_combine(G1, G2, d, gam, bet, m_scales, g(gamma_name), g(beta_name), dm_slots, True)

    # ---------------- final layers ----------------
    Co = c.out_channels                 # I2V: in_channels (video | image latents) = 2 * out_channels
    dtok = E(B * Sv, Co * p * p); ops.patchify(dout, dtok, p)
    _dw(dtok, ctx.y2, g("proj_out.weight"), Co * p * p, d)
    ops.group_colsum(dtok, g("proj_out.bias"), D=Co * p * p)
    dy2 = E(B * Sv, d); ops.gemm(dtok, P.proj_w_t, dy2, None)
    mo = 2 * L * 6 * d
    f_dshift, f_dscale = dmod[:, mo:], dmod[:, mo + d:]
    ln_grads(dy2, ctx.y1, ctx.fm2, ctx.fr2, "norm_out.norm.weight", "norm_out.norm.bias",
             (ctx.f_scale, ctx.f_scale, mod.stride(0)), (f_dshift, f_dshift, f_dscale, f_dscale, nmod), Sv, 0)
    dy1 = E(B * Sv, d)
    ops.ln_modulate_bwd(dy2, ctx.y1, ctx.fm2, ctx.fr2, model.norm_out.norm.weight, (ctx.f_scale, ctx.f_scale, mod.stride(0)),
                        None, dy1, d, Sv, 0)
    dh = torch.zeros(M, d, dtype=BF16, device=dev)
    G1.zero_(); G2.zero_()
    for b in range(B):
        rows = slice(b * Sv, (b + 1) * Sv)
        hv = ctx.h_last[b * S + St:(b + 1) * S]
        ops.group_colsum(dy1[rows], G1[1:2], y=hv, out2=G2[1:2], mean=ctx.fm1[rows], rstd=ctx.fr1[rows], D=d)
        ops.ln_modulate_bwd(dy1[rows], hv, ctx.fm1[rows], ctx.fr1[rows], model.norm_final.weight, None, None,
                            dh[b * S + St:(b + 1) * S], d, Sv, 0)
    ops.ln_param_combine(G1[1:2], G2[1:2], d, model.norm_final.weight, model.norm_final.bias, None,
                         g("norm_final.weight"), g("norm_final.bias"), None, False)
    ready("norm_final.weight", "proj_out.bias")

    # ---------------- blocks, last to first ----------------
    tg = E(M, d); du = E(M, c.ff_mult * d); dx2 = E(M, d); dh1 = E(M, d); dO = E(M, d)
    dq = E(B, S, d, dt=F32); dkh = E(M, d); dqkv = E(M, 3 * d); dx1 = E(M, d)
    delta = E(B * H * S, dt=F32); dh_in = E(M, d)
    # dQ hand-off chains need every persistent workgroup resident: not while RCCL kernels of the overlapped gradient
    # all-reduce share the CUs (on_grads_ready set) -- then the kernel runs persistent with plain atomics
    reducer = getattr(on_grads_ready, "__self__", None)
    overlapped = on_grads_ready is not None and getattr(reducer, "world", 2) > 1
    chain_ws = None if overlapped else ops.attn_bwd_chain_workspace(B, H, S, dev)
    for i in reversed(range(L)):
        Lw, a = P.layers[i], ctx.blocks[i]
        if ctx.recompute:           # rebuild this block's activations from its input
            _, a = block_forward(model, i, a.h_in, mod, ctx.dims, ctx.rope, True, ctx.scratch)
        pre = f"transformer_blocks.{i}."
        m1, m2 = _mod(mod, 2 * i, d), _mod(mod, 2 * i + 1, d)
        dm1, dm2 = dm(2 * i), dm(2 * i + 1)
        # ---- feed-forward branch ----
        # gate gradient: sum over the rows of (sample, segment) of dh * ff_out ; text slot is 3d after the video slot
        ops.group_colsum(dh, None, y=a.fo, out2=dm2.gate_txt, D=d, S=S, St=St, grouped=True, o_bstride=nmod, o_segstride=-3 * d)
        ops.gate_mul(dh, tg, m2.gate_txt, m2.gate_vid, m2.bs, d, S, St)
        _dw(tg, a.g, g(pre + "ff.net.2.weight"), d, c.ff_mult * d)
        ops.group_colsum(tg, g(pre + "ff.net.2.bias"), D=d)
        ops.gemm(tg, Lw.w2_t, du, None, epilogue=EPI_DGELU, pre_act_in=a.u)
        _dw(du, a.x2, g(pre + "ff.net.0.proj.weight"), c.ff_mult * d, d)
        ops.group_colsum(du, g(pre + "ff.net.0.proj.bias"), D=c.ff_mult * d)
        ops.gemm(du, Lw.w1_t, dx2, None)
        ln_grads(dx2, a.h1, a.mean2, a.rstd2, pre + "norm2.norm.weight", pre + "norm2.norm.bias",
                 (m2.scale_txt, m2.scale_vid, m2.bs), (dm2.shift_txt, dm2.shift_vid, dm2.scale_txt, dm2.scale_vid, nmod), S, St)
        ops.ln_modulate_bwd(dx2, a.h1, a.mean2, a.rstd2, Lw.n2g, (m2.scale_txt, m2.scale_vid, m2.bs), dh, dh1, d, S, St)
        # ---- attention branch ----
        ops.group_colsum(dh1, None, y=a.ao, out2=dm1.gate_txt, D=d, S=S, St=St, grouped=True, o_bstride=nmod, o_segstride=-3 * d)
        ops.gate_mul(dh1, tg, m1.gate_txt, m1.gate_vid, m1.bs, d, S, St)
        _dw(tg, a.o, g(pre + "attn1.to_out.0.weight"), d, d)
        ops.group_colsum(tg, g(pre + "attn1.to_out.0.bias"), D=d)
        ops.gemm(tg, Lw.w_o_t, dO, None)
        dq.zero_()
        qk3, qkv3 = a.qkh.view(B, S, 2 * d), a.qkv.view(B, S, 3 * d)
        ops.attn_bwd(qk3[:, :, :d], qk3[:, :, d:], qkv3[:, :, 2 * d:], a.o.view(B, S, a.o.shape[1])[:, :, :d],
                     dO.view(B, S, d), a.lse, delta, dq, dkh.view(B, S, d), dqkv.view(B, S, 3 * d)[:, :, 2 * d:], B, H, S,
                     q_prescaled=True, chain_ws=chain_ws)
        # q/k LayerNorm parameters: the four 64-vectors are adjacent in the flat layout -> written in place
        ops.qk_ln_param_grads(dq.view(M, d), dkh, a.qkv, a.qmean, a.qrstd, g(pre + "attn1.norm_q.weight"), H, rope=ctx.rope)
        ops.qk_layernorm_bwd(dq.view(M, d), dkh, a.qkv, a.qmean, a.qrstd, Lw.gq, Lw.gk, dqkv, H, rope=ctx.rope)
        gw_qkv = ft.span(ft.grad, pre + "attn1.to_q.weight", pre + "attn1.to_v.weight", (3 * d, d))
        gb_qkv = ft.span(ft.grad, pre + "attn1.to_q.bias", pre + "attn1.to_v.bias", (3 * d,))
        _dw(dqkv, a.x1, gw_qkv, 3 * d, d)
        ops.group_colsum(dqkv, gb_qkv, D=3 * d)
        ops.gemm(dqkv, Lw.w_qkv_t, dx1, None)
        ln_grads(dx1, a.h_in, a.mean1, a.rstd1, pre + "norm1.norm.weight", pre + "norm1.norm.bias",
                 (m1.scale_txt, m1.scale_vid, m1.bs), (dm1.shift_txt, dm1.shift_vid, dm1.scale_txt, dm1.scale_vid, nmod), S, St)
        ops.ln_modulate_bwd(dx1, a.h_in, a.mean1, a.rstd1, Lw.n1g, (m1.scale_txt, m1.scale_vid, m1.bs), dh1, dh_in, d, S, St)
        dh, dh_in = dh_in, dh
        ctx.blocks[i] = None
        ready(pre + "attn1.to_q.weight", pre + "attn1.norm_k.bias")

    # ---------------- embeddings ----------------
    for b in range(B):
        dtxt = dh[b * S:b * S + St]
        dvid = dh[b * S + St:(b + 1) * S]
        _dw(dtxt, ctx.text[b], g("patch_embed.text_proj.weight"), d, c.text_embed_dim)
        ops.group_colsum(dtxt, g("patch_embed.text_proj.bias"), D=d)
        _dw(dvid, ctx.patches[b * Sv:(b + 1) * Sv], g("patch_embed.proj.weight").view(d, -1), d, C * p * p)
        ops.group_colsum(dvid, g("patch_embed.proj.bias"), D=d)

    # ---------------- adaLN linears and the time-embedding MLP (one row per sample) ----------------
    gw_ada = ft.span(ft.grad, "transformer_blocks.0.norm1.linear.weight", "norm_out.linear.weight", (nmod, te))
    gb_ada = ft.span(ft.grad, "transformer_blocks.0.norm1.linear.bias", "norm_out.linear.bias", (nmod,))
    dse = Z(B, te)
    ops.small_linear_bwd(dmod, ctx.se, P.w_ada, gw_ada, gb_ada, dse)
    demb = E(B, te, dt=F32); ops.silu_bwd(dse, ctx.emb_pre, demb)
    de1s = Z(B, te)
    ops.small_linear_bwd(demb, ctx.e1, P.t2_w, g("time_embedding.linear_2.weight"), g("time_embedding.linear_2.bias"), de1s)
    de1 = E(B, te, dt=F32); ops.silu_bwd(de1s, ctx.e1_pre, de1)
    ops.small_linear_bwd(de1, ctx.tsin, P.t1_w, g("time_embedding.linear_1.weight"), g("time_embedding.linear_1.bias"), None)
    ready("patch_embed.proj.weight", "time_embedding.linear_2.bias")
    ready("transformer_blocks.0.norm1.linear.weight", "norm_out.linear.bias")
    return None
