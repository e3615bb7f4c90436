"""Full fine-tuning state of the DiT (BASELINE config 3: every weight trainable, as the reference's `-fullft` recipes do
through PL + torch.optim.AdamW, cogvideo_pl.py:774-779).

All parameters are re-homed into ONE flat bf16 buffer (the nn.Parameters become views, HF key names unchanged) with a
parallel flat fp32 master copy and a flat fp32 gradient buffer:
  * the optimizer is one fused AdamW launch over 1.69 G elements (fp32 master + moments, bf16 compute copy refreshed
    in the same pass),
  * data-parallel training all-reduces contiguous slices of one buffer, in reverse execution order, overlapped with
    the backward (vt355.ddp.BucketedReducer),
  * groups that the engine wants stacked -- [to_q|to_k|to_v] weights and biases, the four qk-LayerNorm vectors, all
    adaLN linears of the network -- are laid out contiguously so fused kernels write their gradients in place.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import ops

BF16 = torch.bfloat16


def _layer_order(i: int) -> List[str]:
    b = f"transformer_blocks.{i}."
    a = b + "attn1."
    return [a + "to_q.weight", a + "to_k.weight", a + "to_v.weight",
            a + "to_q.bias", a + "to_k.bias", a + "to_v.bias",
            a + "to_out.0.weight", a + "to_out.0.bias",
            b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight", b + "ff.net.2.bias",
            b + "norm1.norm.weight", b + "norm1.norm.bias", b + "norm2.norm.weight", b + "norm2.norm.bias",
            a + "norm_q.weight", a + "norm_q.bias", a + "norm_k.weight", a + "norm_k.bias"]


class FullFTState:
    def __init__(self, model):
        if model.lora is not None:
            raise NotImplementedError("LoRA adapters and full fine-tuning of the base weights are mutually exclusive here")
        L = model.config.num_layers
        named = dict(model.named_parameters())
        order: List[str] = []
        for suffix in ("weight", "bias"):                         # 1-2. every adaLN linear of the network, stacked
            for i in range(L):
                order += [f"transformer_blocks.{i}.norm1.linear.{suffix}", f"transformer_blocks.{i}.norm2.linear.{suffix}"]
            order.append(f"norm_out.linear.{suffix}")
        for i in range(L):                                        # 3. per block
            order += _layer_order(i)
        order += ["patch_embed.proj.weight", "patch_embed.proj.bias", "patch_embed.text_proj.weight", "patch_embed.text_proj.bias",
                  "time_embedding.linear_1.weight", "time_embedding.linear_1.bias", "time_embedding.linear_2.weight",
                  "time_embedding.linear_2.bias", "norm_final.weight", "norm_final.bias", "norm_out.norm.weight",
                  "norm_out.norm.bias", "proj_out.weight", "proj_out.bias"]
        assert set(order) == set(named), set(order) ^ set(named)
        self.names = order
        self.offsets: Dict[str, Tuple[int, torch.Size]] = {}
        off = 0
        for n in order:
            p = named[n]
            if p.dtype != BF16:
                raise TypeError(f"{n} is {p.dtype}: call .bfloat16() first")
            if p.numel() % 8:
                raise ValueError(f"{n}: size {p.numel()} is not a multiple of 8")
            self.offsets[n] = (off, p.shape)
            off += p.numel()
        self.numel = off
        dev = model.device
        self.flat_bf16 = torch.empty(off, dtype=BF16, device=dev)
        self.flat = torch.empty(off, dtype=torch.float32, device=dev)           # fp32 master
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n in order:
                o, shp = self.offsets[n]
                self.flat_bf16[o:o + named[n].numel()].copy_(named[n].detach().reshape(-1))
            self.flat.copy_(self.flat_bf16)                                       # (a dtype copy: plumbing, once)
            for n in order:
                o, shp = self.offsets[n]
                named[n].data = self.flat_bf16[o:o + named[n].numel()].view(shp)
                named[n].requires_grad_(True)
        self.params = [named[n] for n in order]
        self.version = 0
        self.on_grads_ready = None      # callable(lo, hi) fired by the backward when grad[lo:hi] is final (DDP overlap)
        self.model = model
        model.fullft = self
        model._packed = None

    # views ------------------------------------------------------------------------------------------
    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        o, shp = self.offsets[name]
        n = 1
        for s in shp:
            n *= s
        return buf[o:o + n].view(shp)

    def span(self, buf: torch.Tensor, first: str, last: str, shape) -> torch.Tensor:
        """contiguous view covering parameters first..last (adjacent in the flat layout)"""
        o0, _ = self.offsets[first]
        o1, s1 = self.offsets[last]
        n1 = 1
        for s in s1:
            n1 *= s
        return buf[o0:o1 + n1].view(shape)

    def g(self, name: str) -> torch.Tensor:
        return self.view(self.grad, name)

    def named_grads(self):
        return {n: self.g(n) for n in self.names}

    def mark_changed(self):
        self.version += 1


def enable_full_finetune(model) -> FullFTState:
    return FullFTState(model)
