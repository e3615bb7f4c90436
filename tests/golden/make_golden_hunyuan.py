#!/usr/bin/env python3
"""Golden vectors for the HunyuanVideo transformer blocks, produced by IMPORTING the reference's own classes
(videotuna/models/hunyuan/hyvideo_t2v/modules/models.py: MMDoubleStreamBlock :21-252, MMSingleStreamBlock :255-393) in the build container.
Stubs (SURVEY 8(c) point 3): diffusers' ModelMixin / ConfigMixin / register_to_config (class plumbing only); `flash_attn_varlen_func` is an
absent binary -- it is re-expressed with per-segment SDPA over cu_seqlens and patched into modules/attenion.py, so the varlen attention is
pinned by restatement; everything else (modulation, RMS q/k norm, rotary embedding, projections, MLPs, gating) is the reference's code.

    python tests/golden/make_golden_hunyuan.py   -> tests/golden/hunyuan_blocks.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import make_golden as MG  # noqa: E402
import hunyuan_oracle as HO  # noqa: E402


def varlen(q, k, v, cu_q, cu_k, max_q, max_k, **kw):
    """flash_attn_varlen_func(q [T, H, D], ...) re-expressed: independent attention inside every [cu[i], cu[i+1]) segment"""
    out = torch.zeros_like(q)
    for i in range(len(cu_q) - 1):
        lo, hi = int(cu_q[i]), int(cu_q[i + 1])
        if hi > lo:
            o = F.scaled_dot_product_attention(q[lo:hi].transpose(0, 1), k[lo:hi].transpose(0, 1), v[lo:hi].transpose(0, 1))
            out[lo:hi] = o.transpose(0, 1)
    return out


def main():
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    class _ModelMixin(torch.nn.Module):
        pass

    class _ConfigMixin:
        pass
    MG.stub("diffusers"); MG.stub("diffusers.models", ModelMixin=_ModelMixin)
    MG.stub("diffusers.configuration_utils", ConfigMixin=_ConfigMixin, register_to_config=lambda f: f)
    import importlib
    att = importlib.import_module("videotuna.models.hunyuan.hyvideo_t2v.modules.attenion")
    att.flash_attn_varlen_func = varlen
    models = importlib.import_module("videotuna.models.hunyuan.hyvideo_t2v.modules.models")
    D, H, Li, Lt, B = 256, 2, 40, 12, 2           # head_dim 128
    g = torch.Generator().manual_seed(17)
    img = torch.randn(B, Li, D, generator=g); txt = torch.randn(B, Lt, D, generator=g); vec = torch.randn(B, D, generator=g)
    ang = torch.rand(Li, D // H // 2, generator=g) * 6.28
    cos, sin = torch.cos(ang).repeat_interleave(2, dim=1), torch.sin(ang).repeat_interleave(2, dim=1)      # [Li, 128], pair-repeated
    txt_valid = torch.tensor([5, 12])
    Lj = Li + Lt
    cu = torch.zeros(2 * B + 1, dtype=torch.int32)
    for i in range(B):                                # get_cu_seqlens (attenion.py:34-57) without its device="cuda"
        cu[2 * i + 1] = i * Lj + Li + txt_valid[i]; cu[2 * i + 2] = (i + 1) * Lj
    rec = dict(img=img.numpy(), txt=txt.numpy(), vec=vec.numpy(), cos=cos.numpy(), sin=sin.numpy(), txt_valid=txt_valid.numpy())

    dbl = models.MMDoubleStreamBlock(D, H, 4.0, qkv_bias=True).eval()
    Pd = HO.init(HO.double_block_shapes(D, H), 1)
    assert set(Pd) == {k for k, _ in dbl.named_parameters()}, set(Pd) ^ {k for k, _ in dbl.named_parameters()}
    dbl.load_state_dict(Pd)
    i_in, t_in, v_in = img.clone().requires_grad_(True), txt.clone().requires_grad_(True), vec.clone().requires_grad_(True)
    io, to = dbl(i_in, t_in, v_in, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=Lj, max_seqlen_kv=Lj, freqs_cis=(cos, sin))
    gi, gt = torch.randn(io.shape, generator=g), torch.randn(to.shape, generator=g)
    valid = torch.zeros(B, Lt, 1); valid[0, :5] = 1; valid[1, :12] = 1         # padding text rows carry no gradient (they are never read)
    (io * gi).sum().add((to * gt * valid).sum()).backward()
    rec.update(d_img=io.detach().numpy(), d_txt=to.detach().numpy(), d_gi=gi.numpy(), d_gt=(gt * valid).numpy(), d_dimg=i_in.grad.numpy(),
               d_dtxt=t_in.grad.numpy(), d_dvec=v_in.grad.numpy())
    def keep(gr):          # large matrices: the first 8 rows + checksums
        g2 = gr.detach().reshape(gr.shape[0], -1)
        return (g2 if g2.numel() <= 20000 else g2[:8]).numpy(), np.array([float(gr.double().sum()), float(gr.double().abs().sum())])
    for n, p in dbl.named_parameters():
        rec["d_g." + n], rec["d_c." + n] = keep(p.grad)

    sgl = models.MMSingleStreamBlock(D, H, 4.0).eval()
    Ps = HO.init(HO.single_block_shapes(D, H), 2)
    assert set(Ps) == {k for k, _ in sgl.named_parameters()}
    sgl.load_state_dict(Ps)
    x = torch.cat([img, txt], 1).clone().requires_grad_(True)
    v2 = vec.clone().requires_grad_(True)
    xo = sgl(x, v2, Lt, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=Lj, max_seqlen_kv=Lj, freqs_cis=(cos, sin))
    gx = torch.randn(xo.shape, generator=g)
    vmask = torch.cat([torch.ones(B, Li, 1), valid], 1)
    (xo * gx * vmask).sum().backward()
    rec.update(s_x=xo.detach().numpy(), s_gx=(gx * vmask).numpy(), s_dx=x.grad.numpy(), s_dvec=v2.grad.numpy())
    for n, p in sgl.named_parameters():
        rec["s_g." + n], rec["s_c." + n] = keep(p.grad)
    np.savez_compressed(os.path.join(HERE, "hunyuan_blocks.npz"), **rec)
    print("hunyuan_blocks:", len(rec), "arrays; double out", tuple(io.shape), "single out", tuple(xo.shape))

    # ---- the whole HYVideoDiffusionTransformer at a tiny size (embedders, token refiner, 1 + 1 blocks, final layer, unpatchify) ----
    from types import SimpleNamespace
    def cu_cpu(text_mask, img_len):               # get_cu_seqlens (attenion.py:34-57) without its device="cuda"
        Bq = text_mask.shape[0]
        text_len = text_mask.sum(dim=1)
        max_len = text_mask.shape[1] + img_len
        c = torch.zeros([2 * Bq + 1], dtype=torch.int32)
        for i in range(Bq):
            c[2 * i + 1] = i * max_len + text_len[i] + img_len
            c[2 * i + 2] = (i + 1) * max_len
        return c
    models.get_cu_seqlens = cu_cpu
    Cin, T, Hh, Ww, Ltx, TD, TD2 = 4, 3, 8, 12, 12, 64, 32
    net = models.HYVideoDiffusionTransformer(SimpleNamespace(text_states_dim=TD, text_states_dim_2=TD2), patch_size=[1, 2, 2], in_channels=Cin,
                                             hidden_size=D, heads_num=H, mm_double_blocks_depth=1, mm_single_blocks_depth=1).eval()
    shapes = HO.model_shapes(D, H, 1, 1, Cin, Cin, (1, 2, 2), TD, TD2)
    assert {k: tuple(v.shape) for k, v in net.named_parameters()} == shapes, set(shapes) ^ {k for k, _ in net.named_parameters()}
    Pm = HO.init_model(shapes, 3)
    net.load_state_dict(Pm)
    g2 = torch.Generator().manual_seed(23)
    x = torch.randn(B, Cin, T, Hh, Ww, generator=g2)
    tstep = torch.tensor([37.0, 912.0])
    ts_ = torch.randn(B, Ltx, TD, generator=g2); ts2 = torch.randn(B, TD2, generator=g2)
    tmask = torch.zeros(B, Ltx, dtype=torch.int64); tmask[0, :5] = 1; tmask[1, :12] = 1
    ntok = T * (Hh // 2) * (Ww // 2)
    ang2 = torch.rand(ntok, D // H // 2, generator=g2) * 6.28
    cos2, sin2 = torch.cos(ang2).repeat_interleave(2, dim=1), torch.sin(ang2).repeat_interleave(2, dim=1)
    with torch.no_grad():
        y = net(x, tstep, text_states=ts_, text_mask=tmask, text_states_2=ts2, freqs_cos=cos2, freqs_sin=sin2, return_dict=False)
        txt_ref = net.txt_in(ts_, tstep, tmask)
    np.savez_compressed(os.path.join(HERE, "hunyuan_model.npz"), x=x.numpy(), t=tstep.numpy(), text_states=ts_.numpy(), text_mask=tmask.numpy(),
                        text_states_2=ts2.numpy(), cos=cos2.numpy(), sin=sin2.numpy(), out=y.numpy(), txt_refined=txt_ref.numpy())
    print("hunyuan_model: out", tuple(y.shape), "refined text", tuple(txt_ref.shape))
    pos = importlib.import_module("videotuna.models.hunyuan.hyvideo_t2v.modules.posemb_layers")
    rc, rs = pos.get_nd_rotary_pos_embed([16, 56, 56], [3, 4, 6], theta=256, use_real=True, theta_rescale_factor=1)
    np.savez_compressed(os.path.join(HERE, "hunyuan_rope.npz"), cos=rc.numpy(), sin=rs.numpy())


if __name__ == "__main__":
    main()
