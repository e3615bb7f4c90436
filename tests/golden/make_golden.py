#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own Python modules.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
Only the *outputs* (small .npz data files) are committed; no reference source is copied.

Reference modules exercised (all under /root/reference):
  videotuna/utils/diffusion_utils.py ............ timestep_embedding (9-33), make_beta_schedule (36-45)
  videotuna/schedulers/ddpm.py .................. DDPM.q_sample (216-222), DDPM.get_v (224-228)
  videotuna/models/cogvideo_sat/dit_video_concat.py
        get_3d_sincos_pos_embed (59-105), modulate (430-431), unpatchify (434-454)
  videotuna/models/cogvideo_sat/sgm/modules/diffusionmodules/discretizer.py
        ZeroSNRDDPMDiscretization (80-140)  -- CogVideoX noise schedule (shift 3.0, zero-terminal SNR)
  videotuna/models/opensora/models/layers/blocks.py
        Attention (139-225, fp32-softmax path), t2i_modulate (75-76), approx_gelu MLP
  videotuna/models/lvdm/modules/attention.py
        CrossAttention einsum path (126-149)

Non-arithmetic third-party imports that are absent here (colorama, omegaconf, loguru, cv2, sat,
timm, xformers, mmengine ...) are replaced by inert stubs, exactly as SURVEY.md Appendix C describes.

usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]                       # behaves as a pass-through decorator
        return self

    def __getattr__(self, k):
        return _Any()


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def load_file(modname, path, package=None):
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    if package is not None:
        m.__package__ = package
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def install_stubs():
    stub("colorama", Fore=_Any(), Style=_Any())
    stub("omegaconf", DictConfig=dict, ListConfig=list, OmegaConf=_Any())
    stub("loguru", logger=_Any())
    stub("cv2")
    # sat (SwissArmyTransformer) -- only class/decorator names are needed at import time
    class _Base(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    stub("sat")
    stub("sat.model")
    stub("sat.model.base_model", BaseModel=_Base, non_conflict=lambda f: f)
    stub("sat.model.mixins", BaseMixin=_Base)
    stub("sat.mpu")
    stub("sat.mpu.layers", ColumnParallelLinear=_Base)
    stub("sat.ops")
    stub("sat.ops.layernorm", LayerNorm=torch.nn.LayerNorm, RMSNorm=_Base)
    stub("sat.transformer_defaults", HOOKS_DEFAULT={}, attention_fn_default=_Any())
    # sgm package skeleton; real util.py / discretizer.py loaded by path below
    sat_dir = os.path.join(REF, "videotuna/models/cogvideo_sat")
    stub("sgm")
    stub("sgm.util", instantiate_from_config=_Any(), append_zero=lambda x: torch.cat([x, x.new_zeros([1])]))
    stub("sgm.modules")
    stub("sgm.modules.diffusionmodules")
    load_file("sgm.modules.diffusionmodules.util",
              os.path.join(sat_dir, "sgm/modules/diffusionmodules/util.py"))
    stub("sgm.modules.diffusionmodules.openaimodel", Timestep=_Base)
    return sat_dir


def main():
    torch.manual_seed(20230211)
    sys.path.insert(0, REF)
    sat_dir = install_stubs()

    # ---------------- 1. timestep sinusoid ----------------
    from videotuna.utils.diffusion_utils import timestep_embedding
    t = torch.tensor([0, 1, 17, 500, 999], dtype=torch.int64)
    np.savez(os.path.join(OUT, "timestep_embedding.npz"), t=t.numpy(),
             emb1920=timestep_embedding(t, 1920).numpy(), emb64=timestep_embedding(t, 64).numpy())

    # ---------------- 2. 3D sincos positional table ----------------
    dvc = load_file("ref_dit_video_concat", os.path.join(sat_dir, "dit_video_concat.py"))
    small = dvc.get_3d_sincos_pos_embed(128, 6, 8, 3, height_interpolation=1.875, width_interpolation=1.875,
                                        time_interpolation=1.0)
    full = dvc.get_3d_sincos_pos_embed(1920, 30, 45, 13, height_interpolation=1.875, width_interpolation=1.875,
                                       time_interpolation=1.0)                # [13, 1350, 1920] float64
    rows = np.array([0, 1, 44, 45, 1349, 1350, 7000, 17549])
    flat = full.reshape(-1, 1920)
    np.savez(os.path.join(OUT, "pos_embed_3d.npz"), small=small.astype(np.float64),
             full_rows_idx=rows, full_rows=flat[rows].astype(np.float64),
             full_colsum=flat.sum(0), full_rowsum=flat.sum(1))

    # modulate + unpatchify (shape/order conventions)
    x = torch.randn(2, 12, 64)
    sh, sc = torch.randn(2, 64), torch.randn(2, 64)
    mod = dvc.modulate(x, sh, sc)
    tok = torch.randn(2, 3 * 2 * 4, 16 * 1 * 2 * 2)
    unp = dvc.unpatchify(tok, c=16, patch_size=(1, 2, 2), w=None, h=None, rope_T=3, rope_H=2, rope_W=4)
    np.savez(os.path.join(OUT, "modulate_unpatchify.npz"), x=x.numpy(), shift=sh.numpy(), scale=sc.numpy(),
             modulated=mod.numpy(), tokens=tok.numpy(), unpatchified=unp.numpy())

    # ---------------- 2b. 3D rotary tables + rotation (CogVideoX-5B; SAT Rotary3DPositionEmbeddingMixin) ----------------
    # The HF workflow (cogvideo_pl.py:442-473) takes its tables from diffusers' get_3d_rotary_pos_embed, which is not
    # in the reference tree; the SAT twin dit_video_concat.py:263-341 holds the same construction in-tree
    # (t:h:w = 16:24:24 of the 64-wide head, repeat-interleaved frequencies, rotate_half on (2i, 2i+1) pairs).
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # the mixin moves its tables to a GPU in __init__; keep them here
    try:
        rp_small = dvc.Rotary3DPositionEmbeddingMixin(4, 5, 3, 128, 64, text_length=6)
        rp_full = dvc.Rotary3DPositionEmbeddingMixin(30, 45, 13, 3072, 64, text_length=226)
    finally:
        torch.Tensor.cuda = _cuda
    tq = torch.randn(2, 2, 3 * 4 * 5, 64)
    rot = rp_small.rotary(tq, rope_T=3, rope_H=4, rope_W=5)
    fc, fs = rp_full.freqs_cos.reshape(-1, 64), rp_full.freqs_sin.reshape(-1, 64)
    np.savez(os.path.join(OUT, "rope_3d.npz"), small_cos=rp_small.freqs_cos.reshape(-1, 64).numpy(),
             small_sin=rp_small.freqs_sin.reshape(-1, 64).numpy(), x=tq.numpy(), rotated=rot.numpy(),
             full_rows_idx=rows, full_cos_rows=fc[rows].numpy(), full_sin_rows=fs[rows].numpy(),
             full_cos_colsum=fc.double().sum(0).numpy(), full_sin_colsum=fs.double().sum(0).numpy())
    cu = load_file("ref_common_utils", os.path.join(REF, "videotuna", "utils", "common_utils.py"))
    cases = [((30, 45), (30, 45)), ((60, 90), (30, 45)), ((32, 32), (30, 45)), ((48, 30), (30, 45)), ((17, 45), (30, 45))]
    np.savez(os.path.join(OUT, "crop_region.npz"), src=np.array([c[0] for c in cases]), tgt=np.array([c[1] for c in cases]),
             region=np.array([cu.get_resize_crop_region_for_grid(*c) for c in cases]))
    print("rope: ok")

    # ---------------- 3. CogVideoX noise schedule ----------------
    disc = load_file("sgm.modules.diffusionmodules.discretizer",
                     os.path.join(sat_dir, "sgm/modules/diffusionmodules/discretizer.py"),
                     package="sgm.modules.diffusionmodules")
    out = {}
    for s in (1.0, 3.0):
        d = disc.ZeroSNRDDPMDiscretization(linear_start=0.00085, linear_end=0.012, num_timesteps=1000,
                                           shift_scale=s)
        sq = d.get_sigmas(1000)                       # flipped sqrt(abar): index 0 <-> t=999
        out[f"sqrt_abar_shift{int(s)}"] = torch.flip(sq, (0,)).numpy()
        out[f"abar_preshift_rescale{int(s)}"] = np.asarray(d.alphas_cumprod, dtype=np.float64)
    np.savez(os.path.join(OUT, "schedule_cogvideox.npz"), **out)

    # ---------------- 4. q_sample / get_v ----------------
    from videotuna.schedulers.ddpm import DDPM
    sched = DDPM(timesteps=1000, beta_schedule="linear", linear_start=0.00085, linear_end=0.012)
    x0 = torch.randn(3, 4, 2, 5, 6)
    nz = torch.randn_like(x0)
    tt = torch.tensor([3, 400, 998])
    np.savez(os.path.join(OUT, "ddpm_qsample_getv.npz"), x0=x0.numpy(), noise=nz.numpy(), t=tt.numpy(),
             alphas_cumprod=sched.alphas_cumprod.double().numpy(),
             q_sample=sched.q_sample(x0, tt, nz).numpy(), get_v=sched.get_v(x0, nz, tt).numpy())

    # ---------------- 5a. lvdm CrossAttention (einsum path, self-attention, no bias on qkv) ------------
    try:
        from videotuna.models.lvdm.modules.attention import CrossAttention
        ca = CrossAttention(query_dim=128, heads=2, dim_head=64).eval()
        xc = torch.randn(2, 20, 128)
        with torch.no_grad():
            yc = ca(xc)
        np.savez(os.path.join(OUT, "lvdm_crossattention.npz"), x=xc.numpy(), y=yc.numpy(),
                 **{k.replace(".", "_"): v.numpy() for k, v in ca.state_dict().items()})
        print("lvdm CrossAttention: ok")
    except Exception as e:
        print("lvdm CrossAttention skipped:", type(e).__name__, e)

    # ---------------- 5b. attention (+qk LayerNorm) / modulate / MLP primitives (opensora blocks.py) ----------------
    class _Mlp(torch.nn.Module):          # timm.models.vision_transformer.Mlp restated
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=torch.nn.GELU,
                     drop=0.0, **kw):
            super().__init__()
            self.fc1 = torch.nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = torch.nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    stub("timm")
    stub("timm.models")
    stub("timm.models.layers", DropPath=torch.nn.Identity)
    stub("timm.models.vision_transformer", Mlp=_Mlp)
    stub("xformers")
    sys.modules["xformers"].ops = stub("xformers.ops", memory_efficient_attention=_Any(), fmha=_Any())
    stub("mmengine")
    stub("mmengine.registry", Registry=_Any)
    blocks_path = os.path.join(REF, "videotuna/models/opensora/models/layers/blocks.py")
    stub("videotuna.models.opensora.acceleration.communications", all_to_all=_Any(),
         split_forward_gather_backward=_Any(), gather_forward_split_backward=_Any())
    stub("videotuna.models.opensora.acceleration.parallel_states", get_sequence_parallel_group=_Any())
    try:
        blk = load_file("ref_opensora_blocks", blocks_path, package="videotuna.models.opensora.models.layers")
        att = blk.Attention(dim=128, num_heads=2, qkv_bias=True, qk_norm=True, norm_layer=torch.nn.LayerNorm,
                            enable_flash_attn=False).eval()
        with torch.no_grad():     # non-trivial affine so the per-head LayerNorm(64) is really exercised
            for n_ in (att.q_norm, att.k_norm):
                n_.weight.normal_(1.0, 0.2)
                n_.bias.normal_(0.0, 0.2)
        xa = torch.randn(2, 24, 128)
        with torch.no_grad():
            ya = att(xa)
        sd = {k: v.numpy() for k, v in att.state_dict().items()}
        xm = torch.randn(2, 24, 128)
        shm, scm = torch.randn(2, 1, 128), torch.randn(2, 1, 128)
        ln = torch.nn.LayerNorm(128, eps=1e-6, elementwise_affine=False)
        ym = blk.t2i_modulate(ln(xm), shm, scm)
        mlp = _Mlp(128, 512, act_layer=blk.approx_gelu)
        with torch.no_grad():
            yg = mlp(xm)
        np.savez(os.path.join(OUT, "opensora_primitives.npz"), attn_x=xa.numpy(), attn_y=ya.numpy(),
                 **{"attn_" + k: v for k, v in sd.items()},
                 mod_x=xm.numpy(), mod_shift=shm.numpy(), mod_scale=scm.numpy(), mod_y=ym.detach().numpy(),
                 mlp_fc1_w=mlp.fc1.weight.detach().numpy(), mlp_fc1_b=mlp.fc1.bias.detach().numpy(),
                 mlp_fc2_w=mlp.fc2.weight.detach().numpy(), mlp_fc2_b=mlp.fc2.bias.detach().numpy(),
                 mlp_y=yg.numpy())
        print("opensora primitives: ok")
    except Exception as e:           # ordinary import error -> documented, not fatal
        print("opensora primitives skipped:", type(e).__name__, e)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
