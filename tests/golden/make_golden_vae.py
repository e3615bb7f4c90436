#!/usr/bin/env python3
"""Generates tests/golden/vae_encoder_tiny.npz by running the reference's own ContextParallelEncoder3D
(videotuna/models/cogvideo_sat/vae_modules/cp_enc_dec.py) on seeded weights in the build container.  Non-arithmetic imports are
stubbed (beartype, sgm.util's context-parallel getters -> world 1 / rank 0, vae_modules.utils.SafeConv3d -> torch.nn.Conv3d, which
it subclasses without changing the arithmetic below 2 GB); a 1-rank gloo group serves torch.distributed.get_rank().
Run:  python tests/golden/make_golden_vae.py   (needs /root/reference; the .npz it writes is what travels)"""
import importlib.util
import os
import sys
import types
import typing

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import vae_oracle as V


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    bt = stub("beartype", beartype=lambda f: f)
    stub("beartype.typing", **{k: getattr(typing, k) for k in ("List", "Optional", "Tuple", "Union")})
    stub("sgm")
    stub("sgm.util", get_context_parallel_group=lambda: None, get_context_parallel_group_rank=lambda: 0,
         get_context_parallel_rank=lambda: 0, get_context_parallel_world_size=lambda: 1)
    stub("vae_modules")
    stub("vae_modules.utils", SafeConv3d=torch.nn.Conv3d)
    spec = importlib.util.spec_from_file_location("cp_enc_dec", os.path.join(REF, "videotuna/models/cogvideo_sat/vae_modules/cp_enc_dec.py"))
    mod = importlib.util.module_from_spec(spec); sys.modules["cp_enc_dec"] = mod
    spec.loader.exec_module(mod)
    cfg = V.tiny_config()
    enc = mod.ContextParallelEncoder3D(ch=cfg.ch, out_ch=3, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks, attn_resolutions=[],
                                       in_channels=cfg.in_channels, resolution=16, z_channels=cfg.z_channels, double_z=cfg.double_z,
                                       temporal_compress_times=cfg.temporal_compress_times).eval()
    P = V.init_params(cfg, seed=4)
    missing, unexpected = enc.load_state_dict(P, strict=True)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 3, 9, 16, 24, generator=g)          # 9 frames -> 5 -> 3 latent frames; 16x24 -> 4x6
    with torch.no_grad():
        out = enc(x)
    np.savez_compressed(os.path.join(HERE, "vae_encoder_tiny.npz"), x=x.numpy(), out=out.numpy(), **{"P." + k: v.numpy() for k, v in P.items()})
    print("wrote vae_encoder_tiny.npz", tuple(out.shape), float(out.abs().max()))
    # DownSample3D alone, both temporal branches (cp_enc_dec.py:640-676): rank-0 / fake_cp (first frame kept) and the plain
    # avg_pool1d branch (fake_cp=False), on an odd and an even frame count
    ds = mod.DownSample3D(32, with_conv=True, compress_time=True).eval()
    gd = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p_ in ds.parameters():
            p_.copy_(torch.randn(p_.shape, generator=gd) * 0.1)
    rec = {"ds.conv.weight": ds.conv.weight.detach().numpy(), "ds.conv.bias": ds.conv.bias.detach().numpy()}
    for T in (9, 8, 4):
        xd = torch.randn(1, 32, T, 6, 8, generator=gd)
        rec[f"x{T}"] = xd.numpy()
        with torch.no_grad():
            rec[f"keep_first{T}"] = ds(xd, fake_cp=True).numpy()
            rec[f"all_pairs{T}"] = ds(xd, fake_cp=False).numpy()
    np.savez_compressed(os.path.join(HERE, "vae_downsample3d.npz"), **rec)
    print("wrote vae_downsample3d.npz", {k: v.shape for k, v in rec.items()})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
