#!/usr/bin/env python3
"""Golden vectors for the OpenSora v1.0 path (BASELINE configs[0]), produced by RUNNING the reference's own code
in the build container:
  * STDiT (videotuna/models/opensora/models/stdit/stdit.py:34-416) and its layers (models/layers/blocks.py): imported, with the absent
    third-party modules stubbed as SURVEY Appendix C describes -- timm (Mlp / DropPath restated), mmengine Registry, colossalai-backed
    ckpt_utils, and xformers: `memory_efficient_attention` + `BlockDiagonalMask.from_seqlens` of the varlen text cross-attention
    (blocks.py:497-500) are RE-EXPRESSED with per-sample SDPA (xformers is a missing binary): that one op is pinned by restatement.
    A tiny model (depth 2, hidden 576 = 8 heads x 72, 4x8x8 latents, 12 text tokens of width 64, ragged mask, class_dropout_prob 0).
  * the training loss of LatentDiffusion.p_losses (models/iddpm3d.py:1332-1413) with the learned-variance VB term
    (_vb_terms_bpd :1543-1583) through OpenSoraScheduler.p_mean_variance (:444-519, whose mean-type branch is inverted for
    EPSILON models, :497-500) on the schedule of IDDPMScheduler.register_schedule (:188-290) / get_named_beta_schedule (:100-125).
    iddpm3d.py cannot be imported whole (its imports chain into peft, decord, torchvision, a broken `models.` package path ...), so the
    definitions this path runs are taken from the file's AST and executed against the importable base class
    videotuna.schedulers.diffusion_schedulers.DDPMScheduler -- reference code executed as is, nothing rewritten.

    python tests/golden/make_golden_stdit.py   -> tests/golden/stdit_tiny.npz, stdit_loss.npz
"""
import ast
import enum
import math
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import make_golden as MG  # noqa: E402


class _Mlp(nn.Module):          # timm.models.vision_transformer.Mlp (attribute names fc1 / fc2 matter: stdit.py:406-407)
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, **kw):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _BlockDiag:
    def __init__(self, q, kv):
        self.q, self.kv = list(q), list(kv)

    @classmethod
    def from_seqlens(cls, q_seqlen, kv_seqlen=None):
        return cls(q_seqlen, kv_seqlen if kv_seqlen is not None else q_seqlen)


def _mea(q, k, v, p=0.0, attn_bias=None, **kw):
    """xformers.ops.memory_efficient_attention re-expressed: q [1, sum N, H, D], k / v [1, sum L, H, D]; block-diagonal = per sample"""
    if attn_bias is None:
        return F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    outs, qo, ko = [], 0, 0
    for nq, nk in zip(attn_bias.q, attn_bias.kv):
        o = F.scaled_dot_product_attention(q[:, qo:qo + nq].transpose(1, 2), k[:, ko:ko + nk].transpose(1, 2), v[:, ko:ko + nk].transpose(1, 2))
        outs.append(o.transpose(1, 2)); qo += nq; ko += nk
    return torch.cat(outs, dim=1)


def import_stdit():
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    MG.stub("timm"); MG.stub("timm.models"); MG.stub("timm.models.layers", DropPath=nn.Identity)
    MG.stub("timm.models.vision_transformer", Mlp=_Mlp)
    MG.stub("xformers")
    fm = MG.stub("xformers.ops.fmha", BlockDiagonalMask=_BlockDiag)
    sys.modules["xformers"].ops = MG.stub("xformers.ops", memory_efficient_attention=_mea, fmha=fm)
    MG.stub("mmengine"); MG.stub("mmengine.registry", Registry=MG._Any)
    MG.stub("videotuna.models.opensora.acceleration.communications", all_to_all=MG._Any(), split_forward_gather_backward=MG._Any(),
            gather_forward_split_backward=MG._Any())
    MG.stub("videotuna.models.opensora.acceleration.parallel_states", get_sequence_parallel_group=MG._Any())
    MG.stub("videotuna.models.opensora.acceleration.checkpoint", auto_grad_checkpoint=lambda m, *a, **k: m(*a, **k))
    MG.stub("videotuna.models.opensora.registry", MODELS=MG._Any())
    MG.stub("videotuna.models.opensora.utils"); MG.stub("videotuna.models.opensora.utils.ckpt_utils", load_checkpoint=MG._Any())
    base = os.path.join(MG.REF, "videotuna/models/opensora/models")
    MG.load_file("videotuna.models.opensora.models.layers.blocks", os.path.join(base, "layers/blocks.py"),
                 package="videotuna.models.opensora.models.layers")
    return MG.load_file("ref_stdit", os.path.join(base, "stdit/stdit.py"), package="videotuna.models.opensora.models.stdit")


def scheduler_namespace():
    """the definitions of iddpm3d.py that the loss path runs, executed from the file's AST (see the module docstring)"""
    class _LM(nn.Module):
        @property
        def device(self):
            return torch.device("cpu")
    MG.stub("pytorch_lightning", LightningModule=_LM)
    import videotuna.schedulers.diffusion_schedulers as ds
    from videotuna.utils.diffusion_utils import discretized_gaussian_log_likelihood
    from videotuna.utils.distributions import normal_kl
    from videotuna.models.lvdm.modules.utils import default, exists, extract_into_tensor
    from einops import rearrange
    src = open(os.path.join(MG.REF, "videotuna/models/opensora/models/iddpm3d.py")).read()
    tree = ast.parse(src)
    want_top = {"mean_flat", "get_beta_schedule", "get_named_beta_schedule", "ModelMeanType", "ModelVarType", "LossType", "IDDPMScheduler",
                "OpenSoraScheduler"}
    body = [n for n in tree.body if getattr(n, "name", None) in want_top]
    lat = [n for n in tree.body if getattr(n, "name", None) == "LatentDiffusion"][0]
    methods = {n.name: n for n in lat.body if isinstance(n, ast.FunctionDef) and n.name in ("p_losses", "_vb_terms_bpd")}
    holder = ast.ClassDef(name="LossHolder", bases=[], keywords=[], body=[methods["p_losses"], methods["_vb_terms_bpd"]], decorator_list=[])
    mod = ast.Module(body=body + [holder], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(np=np, torch=torch, math=math, enum=enum, nn=nn, partial=partial, rearrange=rearrange, DDPMScheduler=ds.DDPMScheduler,
              ListConfig=list, default=default, exists=exists, extract_into_tensor=extract_into_tensor, normal_kl=normal_kl,
              discretized_gaussian_log_likelihood=discretized_gaussian_log_likelihood)
    exec(compile(mod, "iddpm3d_extract", "exec"), ns)
    return types.SimpleNamespace(**ns)


def main():
    torch.manual_seed(20230211)
    import stdit_oracle as SO
    st = import_stdit()
    cfg = SO.tiny_config()
    net = st.STDiT(input_size=cfg.input_size, in_channels=cfg.in_channels, patch_size=cfg.patch_size, hidden_size=cfg.hidden_size,
                   depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, class_dropout_prob=0.0, pred_sigma=True,
                   caption_channels=cfg.caption_channels, model_max_length=cfg.model_max_length, dtype=torch.float32,
                   space_scale=cfg.space_scale, time_scale=cfg.time_scale, enable_flashattn=False).eval()
    P = SO.init_params(cfg, seed=3)
    names = [k for k, _ in net.named_parameters()]
    assert list(P) == names, (set(P) ^ set(names))
    net.load_state_dict(P, strict=False)          # buffers (pos_embed, pos_embed_temporal, y_embedding) keep the reference's values
    g = torch.Generator().manual_seed(8)
    B = 2
    x = torch.randn(B, cfg.in_channels, *cfg.input_size, generator=g)
    y = torch.randn(B, 1, cfg.model_max_length, cfg.caption_channels, generator=g)
    mask = torch.zeros(B, cfg.model_max_length, dtype=torch.int64); mask[0, :5] = 1; mask[1, :12] = 1
    t = torch.tensor([3, 777])
    out = net(x, t, y, mask)
    gy = torch.randn(out.shape, generator=g)
    (out * gy).sum().backward()
    rec = dict(x=x.numpy(), y=y.numpy(), mask=mask.numpy(), t=t.numpy(), out=out.detach().numpy(), gy=gy.numpy(),
               pos_embed=net.pos_embed.numpy(), pos_embed_temporal=net.pos_embed_temporal.numpy())
    rec["grad_abs_sum"] = np.array([float(p.grad.double().abs().sum()) for _, p in net.named_parameters()])
    rec["grad_sum"] = np.array([float(p.grad.double().sum()) for _, p in net.named_parameters()])
    for n in ("x_embedder.proj.weight", "t_block.1.weight", "y_embedder.y_proj.fc1.weight", "blocks.0.scale_shift_table",
              "blocks.0.attn.qkv.weight", "blocks.0.attn_temp.proj.bias", "blocks.1.cross_attn.kv_linear.weight", "blocks.1.mlp.fc2.weight",
              "final_layer.scale_shift_table", "final_layer.linear.weight"):
        gr = dict(net.named_parameters())[n].grad.detach()
        gr = gr.reshape(gr.shape[0], -1)
        rec["grad." + n] = (gr if gr.numel() <= 20000 else gr[:8]).numpy()          # large matrices: the first 8 rows (+ the checksums above)
    # full-size positional tables of STDiT-XL/2 at 16x256x256 (space_scale 0.5, time_scale 1.0): checksums + a few rows
    big = st.STDiT(input_size=(16, 32, 32), depth=0, space_scale=0.5, time_scale=1.0, dtype=torch.float32)
    pe, pt = big.pos_embed[0].double().numpy(), big.pos_embed_temporal[0].double().numpy()
    rec.update(xl_pos_rows=pe[[0, 1, 17, 255]], xl_pos_colsum=pe.sum(0), xl_tpe_rows=pt[[0, 1, 15]], xl_tpe_colsum=pt.sum(0))
    np.savez_compressed(os.path.join(HERE, "stdit_tiny.npz"), **rec)
    print("stdit_tiny: out", tuple(out.shape), "params", len(names), sum(p.numel() for p in net.parameters()))

    # ---------------- loss ----------------
    S = scheduler_namespace()
    sch = S.OpenSoraScheduler(given_betas=S.get_named_beta_schedule("linear", 1000), timesteps=1000)
    sch.model_mean_type, sch.model_var_type = S.ModelMeanType.EPSILON, S.ModelVarType.LEARNED_RANGE
    holder = S.LossHolder()
    holder.diffusion_scheduler = sch
    holder.loss_type, holder.model_mean_type, holder.model_var_type = S.LossType.MSE, S.ModelMeanType.EPSILON, S.ModelVarType.LEARNED_RANGE
    holder.num_timesteps, holder.training = 1000, True
    x0 = torch.randn(3, 4, 4, 8, 8, generator=g)
    nz = torch.randn(x0.shape, generator=g)
    tt = torch.tensor([0, 40, 999])
    mo = torch.randn(3, 8, 4, 8, 8, generator=g).requires_grad_(True)
    holder.apply_model = lambda x_t, t_, cond, **kw: kw["model"](x_t, t_) if "model" in kw else mo
    sch.apply_model = lambda x_t, t_, cond, **kw: kw["model"](x_t, t_)
    loss, ld = holder.p_losses(x0, None, tt, noise=nz)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "stdit_loss.npz"), x0=x0.numpy(), noise=nz.numpy(), t=tt.numpy(), model_out=mo.detach().numpy(),
                        loss=np.float64(loss.item()), loss_mse=np.float64(ld["train/loss_mse"].item()), loss_vb=np.float64(ld["train/loss_vb"].item()),
                        dmodel_out=mo.grad.numpy(), betas=sch.betas.numpy(), alphas_cumprod=sch.alphas_cumprod.numpy(),
                        posterior_log_variance_clipped=sch.posterior_log_variance_clipped.numpy(),
                        x_t=sch.q_sample(x_start=x0, t=tt, noise=nz).numpy())
    print("stdit_loss:", loss.item(), {k: float(v) for k, v in ld.items()})


if __name__ == "__main__":
    main()
