"""HunyuanVideo transformer trunk on the vt355 kernels (BASELINE configs[4]): the double-stream and
single-stream blocks of videotuna/models/hunyuan/hyvideo_t2v/modules/models.py (MMDoubleStreamBlock :21-252, MMSingleStreamBlock :255-393)
with the parameter names of ``HYVideoDiffusionTransformer.double_blocks / single_blocks``, forward and backward (full fine-tune tape of
vt355.unet), plus the flow-matching loss of ``HunyuanVideoWorkFlow.training_step`` (hyvideo_t2v/hunyuanvideo.py:923-971).

What runs where: SiLU -> Linear modulation (small GEMM, fp32 out); LayerNorm(no affine) + modulate in vt_ln_modulate; fused qkv GEMM;
per-head RMS q/k norm + rotary embedding of the image tokens + the [image; text] concatenation in ONE pass (vt_qk_rmsnorm_rope128);
joint attention with per-sample valid lengths through vt_attn128_fwd / _bwd (the long-sequence head_dim-128 kernels); projections / MLPs with GELU-tanh and gated
residuals in the GEMM epilogues; the single block's ``linear1`` split into its qkv and MLP row ranges (the GELU lives in the second
GEMM's epilogue, both write into one [attn | gelu(mlp)] buffer that ``linear2`` reads).  ``fp8=True`` runs the block Linears' FORWARD
on the fp8 matrix cores (vt_gemm_fp8: E4M3 weights with per-tensor scale as the reference's fp8_optimization.py stores them, activations
quantised per tensor on the fly); gradients stay bf16.

NOT built (recorded in DESIGN.md): the embedders / token refiner / final layer of HYVideoDiffusionTransformer, the diffusers
``HunyuanVideoTransformer3DModel`` key map and LoRA wrappers of the shipped recipe, (the head_dim-128 attention backward is atomics-only: no dQ hand-off chains yet).  Padding text rows attend to the valid keys here
(the reference gives them their own segment, attenion.py:34-57); they are never read by valid rows or by the loss.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import ops
from .ops import BF16, EPI_BIAS_GELU, EPI_DGELU, EPI_GATED_RES
from .stdit import _STRun
from .unet import F32, FlatParamModule, _Var


def _double_shapes(D, H, ratio, pre):
    hd, M4 = D // H, int(D * ratio)
    sh = {}
    for s in ("img", "txt"):
        sh[f"{pre}{s}_mod.linear.weight"] = (6 * D, D); sh[f"{pre}{s}_mod.linear.bias"] = (6 * D,)
        sh[f"{pre}{s}_attn_qkv.weight"] = (3 * D, D); sh[f"{pre}{s}_attn_qkv.bias"] = (3 * D,)
        sh[f"{pre}{s}_attn_q_norm.weight"] = (hd,); sh[f"{pre}{s}_attn_k_norm.weight"] = (hd,)
        sh[f"{pre}{s}_attn_proj.weight"] = (D, D); sh[f"{pre}{s}_attn_proj.bias"] = (D,)
        sh[f"{pre}{s}_mlp.fc1.weight"] = (M4, D); sh[f"{pre}{s}_mlp.fc1.bias"] = (M4,)
        sh[f"{pre}{s}_mlp.fc2.weight"] = (D, M4); sh[f"{pre}{s}_mlp.fc2.bias"] = (D,)
    return sh


def _single_shapes(D, H, ratio, pre):
    hd, M4 = D // H, int(D * ratio)
    return {pre + "linear1.weight": (3 * D + M4, D), pre + "linear1.bias": (3 * D + M4,), pre + "linear2.weight": (D, D + M4),
            pre + "linear2.bias": (D,), pre + "q_norm.weight": (hd,), pre + "k_norm.weight": (hd,),
            pre + "modulation.linear.weight": (3 * D, D), pre + "modulation.linear.bias": (3 * D,)}


class HunyuanBlocks(FlatParamModule):
    """double_blocks.{i}.* / single_blocks.{i}.* of HYVideoDiffusionTransformer (models.py:396-...): hidden_size 3072, 24 heads x 128,
    20 + 40 blocks in HunyuanVideo-T2V"""

    def __init__(self, hidden_size: int = 3072, heads_num: int = 24, mlp_width_ratio: float = 4.0, mm_double_blocks_depth: int = 20,
                 mm_single_blocks_depth: int = 40, fp8: bool = False):
        super().__init__()
        if hidden_size // heads_num != 128 or hidden_size % 128:
            raise ValueError("HunyuanVideo heads are 128 wide")
        self.hidden_size, self.heads_num, self.ratio = hidden_size, heads_num, mlp_width_ratio
        self.n_double, self.n_single, self.fp8 = mm_double_blocks_depth, mm_single_blocks_depth, fp8
        sh: Dict[str, tuple] = {}
        for i in range(mm_double_blocks_depth):
            sh.update(_double_shapes(hidden_size, heads_num, mlp_width_ratio, f"double_blocks.{i}."))
        for i in range(mm_single_blocks_depth):
            sh.update(_single_shapes(hidden_size, heads_num, mlp_width_ratio, f"single_blocks.{i}."))
        self._setup_flat(sh)

    def init_weights(self, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for n, p in self._plist.items():
                s = self.shapes[n]
                if len(s) == 1:
                    w = torch.randn(s, generator=g) * 0.05 + (1.0 if "norm.weight" in n else 0.0)
                else:
                    w = torch.randn(s, generator=g) * (0.7 / s[1] ** 0.5)
                p.copy_(w.to(p.device, BF16))
        self._packed = None
        return self

    def forward(self, img, txt, vec, txt_valid, freqs_cis=None):
        """img [B, Li, D], txt [B, Lt, D], vec [B, D] bf16; txt_valid int [B] valid text tokens; freqs_cis = (cos, sin) fp32 [Li, 128]
        -> x [B, Li + Lt, D] = [image; text] after every double and single block (models.py: HYVideoDiffusionTransformer.forward trunk)"""
        if not img.is_cuda:
            raise RuntimeError("vt355 HunyuanBlocks runs only on an MI355X device (no CPU fallback)")
        if torch.is_grad_enabled() and self.train_state is not None:
            anchor = torch.zeros(1, device=img.device, requires_grad=True)
            return _HYFn.apply(anchor, self, img, txt, vec, txt_valid, freqs_cis)
        return _HYRun(self, save=False).forward(img, txt, vec, txt_valid, freqs_cis)


class _HYFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, img, txt, vec, txt_valid, freqs):
        run = _HYRun(model, save=True)
        out = run.forward(img, txt, vec, txt_valid, freqs)
        ctx.run = run
        return out

    @staticmethod
    def backward(ctx, dout):
        run = ctx.run
        ctx.run = None
        dimg, dtxt, dvec = run.backward(dout)
        return None, None, dimg, dtxt, dvec, None, None


def _packed_hy(model: HunyuanBlocks) -> SimpleNamespace:
    ver = -1 if model.train_state is None else model.train_state.version
    if model._packed is not None and model._packed_version == ver:
        return model._packed
    P = SimpleNamespace(wt={}, w={}, b={}, q={})
    fb = model.flat_bf16
    with torch.no_grad():
        for n, shp in model.shapes.items():
            if n.endswith(".weight") and len(shp) == 2:
                w = model.flat(fb, n)
                if model.train_state is not None:
                    P.wt[n] = w.t().contiguous()
                if model.fp8 and n.endswith("_attn_qkv.weight"):    # E4M3 copy + per-tensor scale (fp8_optimization.py:55-64)
                    P.q[n] = ops.quantize_fp8(w)
    model._packed, model._packed_version = P, ver
    return P


class _HYRun(_STRun):
    def __init__(self, model: HunyuanBlocks, save: bool):
        self.m, self.save = model, save
        self.P = _packed_hy(model)
        self.fb = model.flat_bf16
        self.ts = model.train_state
        self.tape = []
        self.dev = model.device

    def linear(self, x: _Var, wname: str, bname, residual=None, wspan=None, out=None) -> _Var:
        """block Linear without epilogue: with fp8=True its FORWARD product runs on the fp8 matrix cores (activation quantised per tensor
        here, weight copy + scale from _packed_hy); the backward is the bf16 one of vt355.unet._Run.linear"""
        if not (self.m.fp8 and wname in self.P.q and residual is None and wspan is None):
            return super().linear(x, wname, bname, residual, wspan, out)
        M = x.d.shape[0]
        wq, sw = self.P.q[wname]
        y = out if out is not None else self.E(M, wq.shape[0])
        xq, sa = ops.quantize_fp8(x.d)
        ops.gemm_fp8(xq, wq, y, sa, sw, None if bname is None else self.W(bname))
        yv = _Var(y)
        if self.save:
            def bwd_linear_fp8():
                g = yv.g
                if bname is not None:
                    ops.group_colsum(g, self.G(bname), D=wq.shape[0])
                self.dW(g, x.d, self.G(wname))
                dx = self.E(M, wq.shape[1])
                ops.gemm(g, self.P.wt[wname], dx, None)
                self.acc(x, dx)
            self.tape.append(bwd_linear_fp8)
        return yv

    def modulation(self, sv: _Var, pre: str, n: int):
        """ModulateDiT (modulate_layers.py:7-27): Linear(SiLU(vec)) -> fp32 [B, n*D]; returns (mod, dmod); sv = SiLU(vec) bf16"""
        B, D = sv.d.shape
        mod = self.E(B, n * D, dt=F32)
        ops.gemm(sv.d, self.W(pre + ".linear.weight"), mod, self.W(pre + ".linear.bias"))
        dmod = torch.zeros(B, n * D, dtype=F32, device=self.dev) if self.save else None
        if self.save:
            def bwd_modulation():
                ops.small_linear_bwd(dmod, sv.d, self.W(pre + ".linear.weight"), self.G(pre + ".linear.weight"), self.G(pre + ".linear.bias"), self._dsv)
            self.tape.append(bwd_modulation)
        return mod, dmod

    def glinear(self, x: _Var, wname: str, bname: str, residual: _Var, gate, dgate, rps: int, bs: int) -> _Var:
        """y = residual + gate[b] * (x W^T + b)   (apply_gate, modulate_layers.py:49-66)"""
        w = self.W(wname)
        M = x.d.shape[0]
        y = self.E(M, w.shape[0])
        branch = self.E(M, w.shape[0]) if self.save else None
        ops.gemm(x.d, w, y, self.W(bname), epilogue=EPI_GATED_RES, residual=residual.d, gate_txt=gate, gate_vid=gate, gate_bstride=bs, S=rps, St=0,
                 pre_act_out=branch)
        yv = _Var(y)
        if self.save:
            def bwd_glinear():
                g_ = yv.g
                self.acc(residual, g_)
                ops.group_colsum(g_, None, y=branch, out2=dgate, D=w.shape[0], S=rps, St=0, grouped=True, o_bstride=bs, o_segstride=0)
                gg = self.E(M, w.shape[0])
                ops.gate_mul(g_, gg, gate, gate, bs, w.shape[0], rps, 0)
                ops.group_colsum(gg, self.G(bname), D=w.shape[0])
                self.dW(gg, x.d, self.G(wname))
                dx = self.E(M, w.shape[1])
                ops.gemm(gg, self.P.wt[wname], dx, None)
                self.acc(x, dx)
            self.tape.append(bwd_glinear)
        return yv

    def qkv_to_joint(self, qkv: _Var, pre_q: str, pre_k: str, joint, djoint_ref, L: int, Lj: int, off: int, rope):
        """RMS q/k norm + rope + scatter into the joint [B*Lj, 3C] buffer; backward reads the joint gradient buffer djoint_ref[0]"""
        H = self.m.heads_num
        M = qkv.d.shape[0]
        rstd = self.E(M, 2 * H, dt=F32)
        gq, gk = self.W(pre_q), self.W(pre_k)
        ops.qk_rmsnorm_rope128_fwd(qkv.d, joint, gq, gk, rstd, H, L, Lj, off, rope)
        if self.save:
            def bwd_qkv_to_joint():
                dq = self.E(M, qkv.d.shape[1])
                ops.qk_rmsnorm_rope128_bwd(djoint_ref[0], qkv.d, dq, gq, gk, rstd, self.G(pre_q), self.G(pre_k), H, L, Lj, off, rope)
                qkv.g = dq
            self.tape.append(bwd_qkv_to_joint)

    def joint_attention(self, joint, B: int, Lj: int, kv_len, o_out, djoint_ref):
        """attention over the joint sequence; o_out: [B*Lj, >= C] buffer view (row stride may exceed C); the backward fills djoint_ref[0]"""
        H = self.m.heads_num
        C = H * 128
        j3 = joint.view(B, Lj, 3 * C)
        o3 = o_out.view(B, Lj, o_out.shape[1]) if o_out.is_contiguous() else o_out.as_strided((B, Lj, C), (Lj * o_out.stride(0), o_out.stride(0), 1))
        lse = self.E(B, H, Lj, dt=F32)
        scale = 128 ** -0.5
        ops.attn128_fwd(j3[:, :, :C], j3[:, :, C:2 * C], j3[:, :, 2 * C:], o3[:, :, :C], lse, H, scale, kv_len=kv_len)
        ov = _Var(o_out)
        if self.save:
            def bwd_joint_attention():
                g = ov.g
                g3 = g.view(B, Lj, g.shape[1]) if g.is_contiguous() else g.as_strided((B, Lj, C), (Lj * g.stride(0), g.stride(0), 1))
                dj = self.E(B * Lj, 3 * C)
                d3 = dj.view(B, Lj, 3 * C)
                dq32 = self.E(B, Lj, C, dt=F32)
                ops.attn128_bwd(j3[:, :, :C], j3[:, :, C:2 * C], j3[:, :, 2 * C:], o3[:, :, :C], g3[:, :, :C], lse, dq32, d3[:, :, C:2 * C],
                                d3[:, :, 2 * C:], H, scale, kv_len=kv_len)
                ops.residual_cast(dq32.view(B * Lj, C), None, dj[:, :C])
                djoint_ref[0] = dj
            self.tape.append(bwd_joint_attention)
        return ov

    # ------------------------------------------------------------------------------------------------------------------
    def double_block(self, pre: str, img: _Var, txt: _Var, sv: _Var, B, Li, Lt, kv_len, rope):
        D, H = self.m.hidden_size, self.m.heads_num
        C, Lj = D, Li + Lt
        bs = 6 * D
        joint = torch.zeros(B * Lj, 3 * C, dtype=BF16, device=self.dev)
        dj = [None]
        streams = {}
        for s, x, L, off, rp in (("img", img, Li, 0, rope), ("txt", txt, Lt, Li, None)):
            mod, dmod = self.modulation(sv, pre + s + "_mod", 6)
            sl = lambda k, buf=mod: buf[:, k * D:(k + 1) * D]
            dsl = (lambda k, buf=dmod: buf[:, k * D:(k + 1) * D]) if self.save else (lambda k: None)
            xm = self.ln_mod(x, sl(0), sl(1), bs, L, dsl(0), dsl(1), bs)
            qkv = self.linear(xm, pre + s + "_attn_qkv.weight", pre + s + "_attn_qkv.bias")
            streams[s] = (x, L, off, sl, dsl, qkv, rp)
        # the backward of the scatter must run AFTER the attention's backward has produced the joint gradient: push order = forward order
        for s in ("img", "txt"):
            x, L, off, sl, dsl, qkv, rp = streams[s]
            self.qkv_to_joint(qkv, pre + s + "_attn_q_norm.weight", pre + s + "_attn_k_norm.weight", joint, dj, L, Lj, off, rp)
        o = self.E(B * Lj, C)
        ov = self.joint_attention(joint, B, Lj, kv_len, o, dj)
        outs = {}
        for s in ("img", "txt"):
            x, L, off, sl, dsl, qkv, rp = streams[s]
            a = o.view(B, Lj, C)[:, off:off + L].reshape(B * L, C)                  # this stream's rows of the joint output (a copy)
            av = _Var(a)
            if self.save:
                def bwd_split(av=av, off=off, L=L, ov=ov):
                    if ov.g is None:
                        ov.g = torch.zeros(B * Lj, C, dtype=BF16, device=self.dev)
                    ov.g.view(B, Lj, C)[:, off:off + L].copy_(av.g.view(B, L, C))
                self.tape.append(bwd_split)
            x1 = self.glinear(av, pre + s + "_attn_proj.weight", pre + s + "_attn_proj.bias", x, sl(2), dsl(2), L, bs)
            hm = self.ln_mod(x1, sl(3), sl(4), bs, L, dsl(3), dsl(4), bs)
            outs[s] = self.mlp(hm, pre + s + "_mlp.", residual=x1, gate=(sl(5), bs, L), dgate=dsl(5))
        return outs["img"], outs["txt"]

    def single_block(self, pre: str, x: _Var, sv: _Var, B, Li, Lt, kv_len, rope):
        D, H = self.m.hidden_size, self.m.heads_num
        M4 = int(D * self.m.ratio)
        Lj = Li + Lt
        M = B * Lj
        bs = 3 * D
        mod, dmod = self.modulation(sv, pre + "modulation", 3)
        sl = lambda k: mod[:, k * D:(k + 1) * D]
        dsl = (lambda k: dmod[:, k * D:(k + 1) * D]) if self.save else (lambda k: None)
        xm = self.ln_mod(x, sl(0), sl(1), bs, Lj, dsl(0), dsl(1), bs)
        w1, b1 = self.W(pre + "linear1.weight"), self.W(pre + "linear1.bias")
        qkv = self.E(M, 3 * D)
        ops.gemm(xm.d, w1[:3 * D], qkv, b1[:3 * D])
        cat = self.E(M, D + M4)                                  # [attn | gelu(mlp)], read by linear2
        u = self.E(M, M4)
        ops.gemm(xm.d, w1[3 * D:], cat[:, D:], b1[3 * D:], epilogue=EPI_BIAS_GELU, pre_act_out=u)
        qkvv = _Var(qkv)
        joint = self.E(M, 3 * D)
        dj = [None]
        catv = _Var(cat)
        if self.save:
            def bwd_linear1():
                du = catv.g[:, D:]                               # bwd_linear2 left d u = (g W2[:, D:]) * gelu'(u) in the mlp columns
                dqkv = qkvv.g
                gw, gb = self.G(pre + "linear1.weight"), self.G(pre + "linear1.bias")
                ops.group_colsum(dqkv, gb[:3 * D], D=3 * D); ops.group_colsum(du, gb[3 * D:], D=M4)
                self.dW(dqkv, xm.d, gw[:3 * D]); self.dW(du, xm.d, gw[3 * D:])
                dx = self.E(M, D)
                ops.gemm(dqkv, self._wt_rows(pre + "linear1.weight", 0, 3 * D), dx, None)
                dx2 = self.E(M, D)
                ops.gemm(du, self._wt_rows(pre + "linear1.weight", 3 * D, 3 * D + M4), dx2, None, epilogue=EPI_GATED_RES, residual=dx)
                self.acc(xm, dx2)
            self.tape.append(bwd_linear1)
        self.qkv_to_joint(qkvv, pre + "q_norm.weight", pre + "k_norm.weight", joint, dj, Lj, Lj, 0, rope)
        ov = self.joint_attention(joint, B, Lj, kv_len, cat[:, :D], dj)
        # linear2 with the gated residual
        w2 = self.W(pre + "linear2.weight")
        y = self.E(M, D)
        branch = self.E(M, D) if self.save else None
        ops.gemm(cat, w2, y, self.W(pre + "linear2.bias"), epilogue=EPI_GATED_RES, residual=x.d, gate_txt=sl(2), gate_vid=sl(2), gate_bstride=bs, S=Lj,
                 St=0, pre_act_out=branch)
        yv = _Var(y)
        if self.save:
            def bwd_linear2():
                g_ = yv.g
                self.acc(x, g_)
                ops.group_colsum(g_, None, y=branch, out2=dsl(2), D=D, S=Lj, St=0, grouped=True, o_bstride=bs, o_segstride=0)
                gg = self.E(M, D); ops.gate_mul(g_, gg, sl(2), sl(2), bs, D, Lj, 0)
                ops.group_colsum(gg, self.G(pre + "linear2.bias"), D=D)
                self.dW(gg, cat, self.G(pre + "linear2.weight"))
                dcat = self.E(M, D + M4)
                wt2 = self.P.wt[pre + "linear2.weight"]          # [D + M4, D]
                ops.gemm(gg, wt2[:D], dcat[:, :D], None)                                     # d attn
                ops.gemm(gg, wt2[D:], dcat[:, D:], None, epilogue=EPI_DGELU, pre_act_in=u)   # d u = (g W2[:, D:]) * gelu'(u)
                catv.g = dcat
                ov.g = dcat[:, :D]
            self.tape.append(bwd_linear2)
        return yv

    def _wt_rows(self, wname, lo, hi):
        key = f"{wname}[{lo}:{hi}]^T"
        if key not in self.P.wt:
            self.P.wt[key] = self.m.flat(self.fb, wname)[lo:hi].t().contiguous()
        return self.P.wt[key]

    def forward(self, img, txt, vec, txt_valid, freqs):
        m = self.m
        B, Li, D = img.shape
        Lt = txt.shape[1]
        kv_len = (txt_valid.to(self.dev).to(torch.int32) + Li).contiguous()
        rope = None if freqs is None else (freqs[0].to(self.dev, F32).contiguous(), freqs[1].to(self.dev, F32).contiguous())
        self._vec = vec.to(BF16).contiguous()
        sv = self.E(B, D); ops.silu(self._vec, sv)
        svv = _Var(sv)
        self._dsv = torch.zeros(B, D, dtype=F32, device=self.dev) if self.save else None
        iv, tv = _Var(img.to(BF16).reshape(B * Li, D).contiguous()), _Var(txt.to(BF16).reshape(B * Lt, D).contiguous())
        self._in = (iv, tv)
        for i in range(m.n_double):
            iv, tv = self.double_block(f"double_blocks.{i}.", iv, tv, svv, B, Li, Lt, kv_len, rope)
        Lj = Li + Lt
        x = self.E(B, Lj, D)
        x[:, :Li].copy_(iv.d.view(B, Li, D)); x[:, Li:].copy_(tv.d.view(B, Lt, D))          # torch.cat((img, txt), 1) (models.py trunk)
        xv = _Var(x.view(B * Lj, D))
        if self.save:
            def bwd_cat(xv=xv, iv=iv, tv=tv):
                g = xv.g.view(B, Lj, D)
                self.acc(iv, g[:, :Li].reshape(B * Li, D)); self.acc(tv, g[:, Li:].reshape(B * Lt, D))
            self.tape.append(bwd_cat)
        for i in range(m.n_single):
            xv = self.single_block(f"single_blocks.{i}.", xv, svv, B, Li, Lt, kv_len, rope)
        self._out = xv
        self._dims = (B, Lj, D)
        return xv.d.view(B, Lj, D)

    def backward(self, dout):
        B, Lj, D = self._dims
        self._out.g = dout.to(BF16).reshape(B * Lj, D).contiguous()
        while self.tape:
            self.tape.pop()()
        iv, tv = self._in
        Li, Lt = iv.d.shape[0] // B, tv.d.shape[0] // B
        dvec = torch.empty(B, D, dtype=F32, device=self.dev)
        ops.silu_bwd(self._dsv, self._vec, dvec)
        return iv.g.view(B, Li, D), tv.g.view(B, Lt, D), dvec.to(BF16)


def flow_matching_loss(pred, x0, noise):
    """mean_b mean (pred - (noise - x0))^2 (hunyuanvideo.py:963-970, weights 1); pred bf16, x0 / noise fp32 -> (loss fp32 [1], dpred bf16)"""
    target = torch.empty_like(x0)
    torch.sub(noise, x0, out=target)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred)
    ops.mse_loss(pred.contiguous(), target, loss, dpred)
    return loss, dpred
