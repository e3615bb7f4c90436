"""VideoCrafter2 training workflow on the vt355 UNet: the part of ``LVDMFlow`` / ``VideocrafterFlow`` that runs inside
``training_step`` (videotuna/models/lvdm/ddpm3d.py:732-741, 787-868; videotuna/flow/videocrafter.py:418-487) --
``t ~ U{0..T-1}`` -> dynamic rescaling ``x * scale_arr[t]`` (use_scale) -> ``q_sample`` -> UNet -> eps / v / x0 target ->
``mean_b mean_rest (pred - target)^2`` (logvar == 0, l_simple_weight 1, original_elbo_weight 0).

Same constructor keys as the reference classes for what this path needs (``unet_config`` / ``denoiser_config``,
``scheduler_config`` / ``diffusion_scheduler_config``, ``use_scale``, ``scale_a``, ``scale_b``, ``parameterization``, ...), so that
``configs/001_videocrafter2/vc2_t2v_320x512.yaml`` (flow style) and ``vc2_t2v_lora.yaml`` (model style) instantiate through
``vt355.config.instantiate_from_config``; everything else in those nodes (VAE, CLIP embedder, EMA, logging keys) is accepted and
ignored: the frozen first / cond stages are outside this path, batches carry pre-encoded ``{"latents" [B,4,T,H,W], "context"
[B,77,1024][, "fps"]}``.  Full fine-tuning only (the flow-style recipe); ``lora_args`` raises -- the UNet LoRA variant is not built.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import instantiate_from_config
from .optim import FusedAdamW


class LDDPM:
    """schedule tables of videotuna/schedulers/ddpm.py:64-153 (beta_schedule "linear": linspace(sqrt(s), sqrt(e))^2, float64) and
    q_sample / get_v (:216-228)"""

    def __init__(self, timesteps: int = 1000, beta_schedule: str = "linear", linear_start: float = 1e-4, linear_end: float = 2e-2,
                 parameterization: str = "eps", **unused):
        if beta_schedule != "linear":
            raise NotImplementedError("only the 'linear' beta schedule of the VideoCrafter2 recipes is built")
        self.num_timesteps = int(timesteps)
        self.parameterization = parameterization
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        self.alphas_cumprod = torch.from_numpy(np.cumprod(1.0 - betas, axis=0))
        self.logvar = torch.zeros(timesteps)
        self.learn_logvar = False
        self._dev = {}

    def coefficients(self, t: torch.Tensor):
        """(sqrt(abar_t), sqrt(1 - abar_t)) as fp32 device vectors [B]"""
        dev = t.device
        if dev not in self._dev:
            a = self.alphas_cumprod
            self._dev[dev] = torch.stack([a.sqrt(), (1 - a).sqrt()]).to(torch.float32).to(dev)
        sel = self._dev[dev][:, t]
        return sel[0].contiguous(), sel[1].contiguous()


LDMScheduler = LDDPM          # videotuna.schedulers.diffusion_schedulers.LDMScheduler: the same tables as a LightningModule


class _EpsLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dp = torch.empty_like(pred)
        ops.mse_loss(pred.contiguous(), target, loss, dp)
        ctx.save_for_backward(dp)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (dp,) = ctx.saved_tensors
        return dp * gout.to(dp.dtype), None


class LVDMFlow(nn.Module):
    def __init__(self, unet_config=None, denoiser_config=None, scheduler_config=None, diffusion_scheduler_config=None,
                 use_scale: bool = False, scale_a: float = 1.0, scale_b: float = 0.3, mid_step: int = 400, fix_scale_bug: bool = False,
                 parameterization: Optional[str] = None, base_learning_rate: float = 6e-6, lora_args=None, logdir=None,
                 l_simple_weight: float = 1.0, original_elbo_weight: float = 0.0, **ignored):
        super().__init__()
        if lora_args:
            raise NotImplementedError("lora_args: the LoRA variant of the VideoCrafter2 UNet is not built (full fine-tuning only)")
        if original_elbo_weight != 0.0 or l_simple_weight != 1.0:
            raise NotImplementedError("only l_simple_weight 1 / original_elbo_weight 0 (the shipped recipes) are built")
        node = denoiser_config if denoiser_config is not None else unet_config
        self.model = instantiate_from_config(node)
        self.model.bfloat16()
        sch = scheduler_config if scheduler_config is not None else diffusion_scheduler_config
        self.scheduler = instantiate_from_config(sch) if sch is not None else LDDPM(linear_start=0.00085, linear_end=0.012)
        self.diffusion_scheduler = self.scheduler
        self.parameterization = parameterization or getattr(self.scheduler, "parameterization", "eps")
        self.num_timesteps = self.scheduler.num_timesteps
        self.use_scale = use_scale
        if use_scale:      # ddpm3d.py:500-514 (the reference's default keeps the 1400-long table: fix_scale_bug False)
            step = self.num_timesteps - mid_step if fix_scale_bug else self.num_timesteps
            arr = np.concatenate((np.linspace(scale_a, scale_b, mid_step), np.full(step, scale_b)))
            self.register_buffer("scale_arr", torch.tensor(arr, dtype=torch.float32))
        self.learning_rate = base_learning_rate
        self.logdir = logdir
        self.lora_args = []
        self.global_step = 0

    @property
    def device(self):
        return self.model.device

    def configure_optimizers(self):
        ts = self.model.enable_training()
        return FusedAdamW(ts.params, lr=self.learning_rate, fullft_state=ts)

    def p_losses(self, x_start, context, t, noise, fps=16):
        """x_start fp32 [B,C,T,H,W] (already multiplied by scale_arr[t] when use_scale -- done in forward(), ddpm3d.py:740-741)"""
        sa, sb = self.scheduler.coefficients(t)
        x_noisy = torch.empty(x_start.shape, dtype=torch.bfloat16, device=x_start.device)
        ops.q_sample(x_start, noise, sa, sb, None, x_noisy)
        out = self.model(x_noisy, t, context=context, fps=fps)
        if self.parameterization == "eps":
            target = noise
        elif self.parameterization == "x0":
            target = x_start
        else:                       # "v": sqrt(abar) eps - sqrt(1-abar) x0   (schedulers/ddpm.py:224-228)
            target = sa.view(-1, 1, 1, 1, 1) * noise - sb.view(-1, 1, 1, 1, 1) * x_start
        return _EpsLoss.apply(out, target.contiguous())

    def loss_from(self, z, context, t, noise, fps=16):
        """deterministic core of training_step (fixed t / noise): what the parity tests call"""
        z = z.to(torch.float32)
        if self.use_scale:
            z = z * self.scale_arr[t].view(-1, 1, 1, 1, 1)
        return self.p_losses(z.contiguous(), context, t, noise.to(torch.float32).contiguous(), fps)

    def training_step(self, batch, batch_idx=0):
        if "latents" not in batch:
            raise RuntimeError("the VideoCrafter2 path takes pre-encoded batches {'latents','context'[,'fps']}: the frozen 2-D VAE and "
                               "the OpenCLIP embedder of the reference recipe are outside this engine's hot path (SURVEY 8(f))")
        z = batch["latents"]
        B = z.shape[0]
        t = torch.randint(0, self.num_timesteps, (B,), device=z.device).long()
        noise = torch.randn(z.shape, dtype=torch.float32, device=z.device)
        return self.loss_from(z, batch["context"], t, noise, batch.get("fps", 16))


VideocrafterFlow = LVDMFlow
