"""Noise schedule with the surface of ``diffusers.CogVideoXDPMScheduler`` that training touches
(cogvideo_pl.py:838-877): ``config.num_train_timesteps``, ``alphas_cumprod``, ``add_noise``, ``get_velocity``.

abar_t: scaled-linear betas (0.00085 -> 0.012, 1000 steps, fp64) -> cumprod -> SNR shift
abar/(s + (1-s) abar) with s = snr_shift_scale (3.0 for CogVideoX-2b) -> zero-terminal-SNR rescale.
In-tree twin: videotuna/models/cogvideo_sat/sgm/modules/diffusionmodules/discretizer.py:80-140 (pinned by
tests/golden/schedule_cogvideox.npz).  The values of the real checkpoint's scheduler_config.json are not available
offline, so they are explicit constructor arguments here.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from . import ops


class CogVideoXDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", snr_shift_scale: float = 3.0, rescale_betas_zero_snr: bool = True,
                 prediction_type: str = "v_prediction", **unused):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(beta_schedule)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, snr_shift_scale=snr_shift_scale,
                                      rescale_betas_zero_snr=rescale_betas_zero_snr, prediction_type=prediction_type)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1.0 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:
            s = ac.sqrt()
            s0, sT = s[0].clone(), s[-1].clone()
            s = (s - sT) * (s0 / (s0 - sT))
            ac = s ** 2
        self.alphas_cumprod = ac                      # fp64, CPU (like diffusers); device copies are cached
        self._dev = {}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kw):
        root = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        with open(os.path.join(root, "scheduler_config.json")) as f:
            return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})

    def coefficients(self, timesteps: torch.Tensor):
        """(sqrt(abar_t), sqrt(1-abar_t), 1/(1-abar_t)) as fp32 device vectors [B]."""
        dev = timesteps.device
        if dev not in self._dev:
            a = self.alphas_cumprod
            tab = torch.stack([a.sqrt(), (1 - a).sqrt(), 1 / (1 - a)]).to(torch.float32).to(dev)
            self._dev[dev] = tab
        tab = self._dev[dev]
        sel = tab[:, timesteps]                      # tiny gather on [3, B]
        return sel[0].contiguous(), sel[1].contiguous(), sel[2].contiguous()

    def add_noise(self, original_samples, noise, timesteps):
        """fp32 x0 / noise in, bf16 noisy sample out (HIP kernel)."""
        sa, sb, _ = self.coefficients(timesteps)
        out = torch.empty(original_samples.shape, dtype=torch.bfloat16, device=original_samples.device)
        ops.add_noise(original_samples.contiguous(), noise.contiguous(), sa, sb, out)
        return out
