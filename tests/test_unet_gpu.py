"""VideoCrafter2 UNet on the device (vt355.unet through the C-ABI kernels) against the CPU oracle (oracle/unet_oracle.py, itself
pinned to the imported reference UNetModel by tests/golden/unet_*.npz) -- BASELINE configs[3], SURVEY 8(a) a11-a13, a15."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tiny(dev, seed=11):
    import unet_oracle as U
    from vt355.unet import UNetModel
    cfg = U.tiny_config()
    m = UNetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, model_channels=cfg.model_channels,
                  attention_resolutions=list(cfg.attention_resolutions), num_res_blocks=cfg.num_res_blocks,
                  channel_mult=list(cfg.channel_mult), num_head_channels=64, transformer_depth=1, context_dim=cfg.context_dim,
                  use_linear=True, use_checkpoint=True, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
                  use_relative_position=False, use_causal_attention=False, temporal_length=cfg.temporal_length,
                  addition_attention=True, fps_cond=True)
    P = U.init_params(cfg, seed=seed)
    assert list(P) == list(m.state_dict()), "parameter names / order differ from the reference's state_dict"
    m.load_state_dict(P)
    m.to(dev)
    Pr = {k: v.detach().float().cpu().double() for k, v in m.state_dict().items()}       # the bf16-rounded weights the device uses
    return U, cfg, m, Pr


def _relerr(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_tiny_unet_forward_matches_golden_and_oracle(dev):
    """forward on the golden inputs of the REFERENCE UNetModel run (tests/golden/unet_tiny.npz): vs the reference output itself
    (weights differ only by bf16 rounding) and vs the oracle on the rounded weights"""
    U, cfg, m, Pr = _tiny(dev)
    g = np.load(os.path.join(G, "unet_tiny.npz"))
    x = torch.from_numpy(g["x"]); ctx = torch.from_numpy(g["context"]); t = torch.from_numpy(g["t"]); fps = torch.from_numpy(g["fps"])
    with torch.no_grad():
        out = m(x.to(dev, BF), t.to(dev), context=ctx.to(dev, BF), fps=fps.to(dev))
    ref = U.unet_forward(Pr, cfg, x.to(BF).double(), t, ctx.to(BF).double(), fps=fps)
    e_or, e_gold = _relerr(out, ref), _relerr(out, torch.from_numpy(g["out"]))
    print(f"[unet tiny fwd] rel-L2 vs oracle {e_or:.3e}, vs reference golden (fp32 weights) {e_gold:.3e}")
    assert e_or < 3e-2 and e_gold < 5e-2


def test_tiny_unet_train_step_matches_oracle(dev):
    """eps-MSE loss, every parameter gradient and one AdamW step of the tiny UNet vs the fp64 oracle"""
    from vt355 import ops
    from vt355.optim import FusedAdamW
    U, cfg, m, Pr = _tiny(dev)
    ts = m.enable_training()
    g = np.load(os.path.join(G, "unet_tiny.npz"))
    x = torch.from_numpy(g["x"]); ctx = torch.from_numpy(g["context"]); t = torch.from_numpy(g["t"]); fps = torch.from_numpy(g["fps"])
    noise = torch.from_numpy(g["noise"])
    out = m(x.to(dev, BF), t.to(dev), context=ctx.to(dev, BF), fps=fps.to(dev))
    loss = torch.empty(1, device=dev); dp = torch.empty(out.shape, dtype=BF, device=dev)
    ops.mse_loss(out.detach().contiguous(), noise.to(dev), loss, dp)
    out.backward(dp)
    for v in Pr.values():
        v.requires_grad_(True)
    ref = U.unet_forward(Pr, cfg, x.to(BF).double(), t, ctx.to(BF).double(), fps=fps)
    lref = U.lvdm_loss(ref, noise.double())
    lref.backward()
    assert abs(loss.item() - lref.item()) < 2e-2 * lref.item(), (loss.item(), lref.item())
    worst, bad = 0.0, []
    tot_n = tot_d = 0.0
    for n in m.shapes:
        gd = m._view(ts.grad, n).detach().double().cpu()
        gr = Pr[n].grad
        e = (gd - gr).norm().item(); d = gr.norm().item()
        tot_n += e * e; tot_d += d * d
        rel = e / max(d, 1e-12)
        cos = torch.nn.functional.cosine_similarity(gd.flatten(), gr.flatten(), dim=0).item()
        if cos < 0.98 or rel > 0.2:
            bad.append((n, rel, cos))
        worst = max(worst, rel)
    overall = (tot_n / tot_d) ** 0.5
    print(f"[unet tiny train] loss dev {loss.item():.6f} oracle {lref.item():.6f}; grads: overall rel-L2 {overall:.3e}, worst per-parameter {worst:.3e}")
    assert not bad, bad[:10]
    assert overall < 5e-2
    opt = FusedAdamW(ts.params, lr=1e-3, fullft_state=ts)
    before = ts.flat.clone()
    opt.step()
    assert torch.isfinite(ts.flat).all() and (ts.flat - before).abs().max().item() > 0
    assert torch.equal(ts.flat_bf16.float(), ts.flat.to(BF).float())


# ------------------------------------------------------------------------------------------------ each block alone
def _cl(x5):
    """[B, C, T, H, W] -> channels-last rows [B*T*H*W, C]"""
    return x5.permute(0, 2, 3, 4, 1).reshape(-1, x5.shape[1])


def _block_case(dev, kind):
    """one layer of the tiny UNet through the engine (forward, input gradient, parameter gradients) vs the oracle's function for
    that block -- the oracle functions are pinned to the reference's own blocks by tests/test_oracle_golden.py
    (tests/golden/unet_blocks.npz: ResBlock + TemporalConvBlock, SpatialTransformer, TemporalTransformer, Downsample, Upsample)"""
    from vt355.unet import _Run, _Var
    U, cfg, m, Pr = _tiny(dev)
    ts = m.enable_training()
    st = m.structure
    g = torch.Generator().manual_seed(hash(kind) % 1000)
    B, T, H, W = 2, 4, 8, 8
    layer = {"res": st.input[3][0], "res_same": st.input[1][0], "st": st.input[1][1], "tt": st.input[1][2], "init_tt": st.init_attn,
             "down": st.input[2][0], "up": st.output[1][3]}[kind]
    if kind in ("res", "up"):
        H, W = 4, 4
    cin = getattr(layer, "cin", None) or layer.c
    x = (torch.randn(B, cin, T, H, W, generator=g)).to(BF).float()
    emb = torch.randn(B, 4 * cfg.model_channels, generator=g).to(BF).float()
    ctx = torch.randn(B, 77, cfg.context_dim, generator=g).to(BF).float()
    run = _Run(m, save=True)
    xv = _Var(_cl(x).to(dev, BF).contiguous())
    shape = [B, T, H, W]
    for v in Pr.values():
        v.requires_grad_(True)
    xr = x.double().requires_grad_(True)
    x4 = xr.permute(0, 2, 1, 3, 4).reshape(B * T, cin, H, W)
    if kind in ("res", "res_same"):
        se = torch.nn.functional.silu(emb)
        sev = _Var(se.to(dev, BF)); demb = torch.zeros(B, emb.shape[1], device=dev)
        yv = run.res_block(layer, xv, shape, sev, demb)
        # the oracle applies SiLU itself: feed it the pre-activation whose SiLU is `se` rounded to bf16 -> pass se through a patched call
        er = se.to(BF).double().repeat_interleave(T, dim=0).requires_grad_(True)
        import torch.nn.functional as F
        real_silu = F.silu
        try:
            F.silu = lambda t, *a, **k: t if t is er else real_silu(t, *a, **k)
            ref4 = U.res_block(x4, er, Pr, layer.pre, B, True)
        finally:
            F.silu = real_silu
    elif kind == "st":
        ctxv = _Var(ctx.to(dev, BF).view(B * 77, -1).contiguous()); ctxv.g = False
        yv = run.spatial_transformer(layer, xv, shape, ctxv, 77)
        ref4 = U.spatial_transformer(x4, ctx.double().repeat_interleave(T, dim=0), Pr, layer.pre, layer.heads)
    elif kind in ("tt", "init_tt"):
        yv = run.temporal_transformer(layer, xv, shape)
        ref4 = None
        ref5 = U.temporal_transformer(xr, Pr, layer.pre, layer.heads)
    else:
        pytest.skip("down / up run inside forward(): covered by the whole-network test")
    cout = yv.d.shape[1]
    ref5 = ref5 if kind in ("tt", "init_tt") else ref4.reshape(B, T, cout, ref4.shape[2], ref4.shape[3]).permute(0, 2, 1, 3, 4)
    gy = torch.randn(ref5.shape, generator=g).to(BF).float()
    (ref5 * gy.double()).sum().backward()
    e_out = _relerr(yv.d, _cl(ref5))
    yv.g = _cl(gy).to(dev, BF).contiguous()
    while run.tape:
        run.tape.pop()()
    e_dx = _relerr(xv.g, _cl(xr.grad))
    worst = 0.0
    names = [n for n in m.shapes if n.startswith(layer.pre + ".")]
    for n in names:
        gd = m._view(ts.grad, n).detach().double().cpu()
        rel = (gd - Pr[n].grad).norm().item() / max(Pr[n].grad.norm().item(), 1e-12)
        worst = max(worst, rel)
    print(f"[unet block {kind}] out rel-L2 {e_out:.3e}, dx {e_dx:.3e}, worst parameter gradient {worst:.3e} over {len(names)} tensors")
    assert e_out < 2e-2 and e_dx < 4e-2 and worst < 6e-2
    if kind in ("res", "res_same"):
        dse = er.grad.view(B, T, -1).sum(1)
        assert _relerr(demb, dse) < 4e-2


@pytest.mark.parametrize("kind", ["res", "res_same", "st", "tt", "init_tt"])
def test_unet_block_alone(dev, kind):
    _block_case(dev, kind)
