#!/usr/bin/env python3
"""Golden vectors for the VideoCrafter2 UNet path, produced by IMPORTING the reference's own modules (build container only):
  videotuna/models/lvdm/modules/networks/openaimodel3d.py  UNetModel (313-694), ResBlock (123-255), TemporalConvBlock (258-310),
                                                           Downsample / Upsample (56-120)
  videotuna/models/lvdm/modules/attention.py               CrossAttention (45-242, einsum path: xformers is absent),
                                                           SpatialTransformer (313-392), TemporalTransformer (395-519)
  videotuna/schedulers/ddpm.py                             LDDPM.q_sample + schedule (the eps-prediction loss inputs)
Non-arithmetic imports absent here are stubbed (colorama, omegaconf, loguru, cv2 -- SURVEY Appendix C).  Weights come from
oracle/unet_oracle.init_params (seeded; zero-initialised reference layers get random weights so gradients flow), so the fixture
holds only inputs, outputs and gradients.  Modules run in eval() mode: TemporalConvBlock's dropout(0.1) is then the identity.

    python tests/golden/make_golden_unet.py   -> tests/golden/unet_tiny.npz, unet_blocks.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import make_golden as MG  # noqa: E402
import unet_oracle as U   # noqa: E402


def grads_of(module, names):
    sd = dict(module.named_parameters())
    return {n: sd[n].grad.detach().numpy().copy() for n in names}


def main():
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    from videotuna.models.lvdm.modules.networks import openaimodel3d as om
    from videotuna.models.lvdm.modules import attention as at
    assert not at.XFORMERS_IS_AVAILBLE
    cfg = U.tiny_config()
    net = om.UNetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, model_channels=cfg.model_channels,
                       attention_resolutions=list(cfg.attention_resolutions), num_res_blocks=cfg.num_res_blocks,
                       channel_mult=list(cfg.channel_mult), num_head_channels=cfg.num_head_channels, transformer_depth=1,
                       context_dim=cfg.context_dim, use_linear=True, use_checkpoint=True, temporal_conv=True, temporal_attention=True,
                       temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False,
                       temporal_length=cfg.temporal_length, addition_attention=True, fps_cond=True).eval()
    P = U.init_params(cfg, seed=11)
    assert list(P) == [k for k, _ in net.named_parameters()], "oracle parameter list / order differs from the reference module"
    net.load_state_dict(P, strict=True)
    g = torch.Generator().manual_seed(5)
    B, T, H, W = 2, cfg.temporal_length, 8, 8
    x = torch.randn(B, cfg.in_channels, T, H, W, generator=g)
    ctx = torch.randn(B, 80, cfg.context_dim, generator=g)           # 80 > 77: the [:, :77] slice of CrossAttention is exercised
    t = torch.tensor([37, 912])
    noise = torch.randn(x.shape, generator=g)
    fps = torch.tensor([24, 8])
    out = net(x, t, context=ctx, fps=fps)
    loss = ((out - noise) ** 2).mean(dim=(1, 2, 3, 4)).mean()
    loss.backward()
    names = [n for n, _ in net.named_parameters()]
    rec = dict(x=x.numpy(), context=ctx.numpy(), t=t.numpy(), fps=fps.numpy(), noise=noise.numpy(), out=out.detach().numpy(),
               loss=np.float64(loss.item()))
    gsum = np.array([float(p.grad.double().sum()) for _, p in net.named_parameters()])
    gabs = np.array([float(p.grad.double().abs().sum()) for _, p in net.named_parameters()])
    rec["grad_sum"], rec["grad_abs_sum"] = gsum, gabs
    keep = ["input_blocks.0.0.weight", "time_embed.0.weight", "fps_embedding.2.bias", "init_attn.0.transformer_blocks.0.attn1.to_q.weight",
            "input_blocks.1.0.in_layers.2.weight", "input_blocks.1.0.temopral_conv.conv3.3.weight", "input_blocks.1.1.norm.weight",
            "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight", "input_blocks.1.2.transformer_blocks.0.ff.net.0.proj.weight",
            "input_blocks.2.0.op.weight", "middle_block.1.transformer_blocks.0.norm2.weight", "output_blocks.1.0.skip_connection.weight",
            "output_blocks.1.3.conv.weight", "out.2.weight", "out.0.bias"]
    for n in keep:
        assert n in names, n
        rec["grad." + n] = dict(net.named_parameters())[n].grad.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "unet_tiny.npz"), **rec)
    print("unet_tiny: out", tuple(out.shape), "loss", loss.item(), "params", len(names), sum(p.numel() for p in net.parameters()))

    # ---------------- each block alone: output, input gradient, parameter gradients ----------------
    blk = {}

    def run(tag, mod, args, wrt, fn=None):
        for p in mod.parameters():
            p.grad = None
        ins = [a.clone().requires_grad_(True) if (torch.is_tensor(a) and a.is_floating_point() and i in wrt) else a for i, a in enumerate(args)]
        y = (fn or mod)(*ins)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        blk[tag + ".y"] = y.detach().numpy(); blk[tag + ".gy"] = gy.numpy()
        for i, a in enumerate(args):
            if torch.is_tensor(a):
                blk[f"{tag}.in{i}"] = a.numpy()
                if i in wrt:
                    blk[f"{tag}.gin{i}"] = ins[i].grad.numpy()
        for n, p in mod.named_parameters():
            blk[f"{tag}.g.{n}"] = p.grad.detach().numpy()

    # ResBlock 64 -> 128 with temporal conv (1x1 skip), 2 samples x 4 frames
    rb = net.input_blocks[3][0]
    assert isinstance(rb, om.ResBlock) and rb.channels == 64 and rb.out_channels == 128
    run("res", rb, (torch.randn(8, 64, 4, 4, generator=g), torch.randn(8, 256, generator=g)), (0, 1), fn=lambda a, e: rb(a, e, batch_size=2))
    tcb = net.input_blocks[1][0].temopral_conv
    run("tconv", tcb, (torch.randn(2, 64, 4, 8, 8, generator=g),), (0,))
    st = net.input_blocks[1][1]
    assert isinstance(st, at.SpatialTransformer)
    run("st", st, (torch.randn(8, 64, 8, 8, generator=g), torch.randn(8, 80, 64, generator=g)), (0, 1))
    tt = net.input_blocks[1][2]
    assert isinstance(tt, at.TemporalTransformer)
    run("tt", tt, (torch.randn(2, 64, 4, 8, 8, generator=g),), (0,))
    ca = st.transformer_blocks[0].attn2
    run("xattn", ca, (torch.randn(3, 20, 64, generator=g), torch.randn(3, 80, 64, generator=g)), (0, 1))
    dn, up = net.input_blocks[2][0], net.output_blocks[1][3]
    assert isinstance(dn, om.Downsample) and isinstance(up, om.Upsample)
    run("down", dn, (torch.randn(4, 64, 8, 8, generator=g),), (0,))
    run("up", up, (torch.randn(4, 128, 4, 4, generator=g),), (0,))
    np.savez_compressed(os.path.join(HERE, "unet_blocks.npz"), **blk)
    print("unet_blocks:", len(blk), "arrays")

    # ---------------- LDDPM q_sample / schedule / scale_arr for the loss path ----------------
    from videotuna.schedulers.ddpm import LDDPM
    try:
        sch = LDDPM(timesteps=1000, linear_start=0.00085, linear_end=0.012)
    except TypeError:
        from videotuna.schedulers.ddpm import DDPM
        sch = DDPM(timesteps=1000, beta_schedule="linear", linear_start=0.00085, linear_end=0.012)
    xs = torch.randn(2, 4, 4, 8, 8, generator=g)
    nz = torch.randn(xs.shape, generator=g)
    tt_ = torch.tensor([5, 800])
    np.savez_compressed(os.path.join(HERE, "unet_loss.npz"), x0=xs.numpy(), noise=nz.numpy(), t=tt_.numpy(),
                        alphas_cumprod=sch.alphas_cumprod.double().numpy(), q_sample=sch.q_sample(xs, tt_, nz).numpy())
    print("unet_loss: ok")


if __name__ == "__main__":
    main()
