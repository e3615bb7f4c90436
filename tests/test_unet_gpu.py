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
