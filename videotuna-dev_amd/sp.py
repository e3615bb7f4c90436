"""Ulysses sequence parallelism for the joint [image; text] attention of HunyuanVideo (BASELINE configs[4]).

The reference shards the image tokens over ``ulysses_degree`` ranks for sampling (hyvideo_t2v/inference.py:46-80 splits the latent and
the rotary tables along one spatial axis) and calls xfuser's ``xFuserLongContextAttention`` with the text tokens as the ``joint_tensor_*``
operands, ``joint_strategy="rear"`` (modules/attenion.py:157-215): an all-to-all turns "my sequence shard, all heads" into "the whole
sequence, my heads", the replicated text tokens contribute their slice of the heads at the rear of the sequence, attention runs locally,
and a second all-to-all brings the image rows back to sequence shards.  xfuser is a dependency that is not in this image; this module
states that exchange directly on ``torch.distributed`` (RCCL over xGMI on the device: ONE all_to_all_single per operand and direction --
each rank exchanges (P-1)/P of its shard over the P-1 point-to-point links at once, no ring) with its backward, so the same attention
also trains.

Gradient convention ("replicated-full"): the text rows and their gradients are replicated on every rank -- every rank holds the complete
gradient of a text tensor, never a partial sum.  Image rows are sharded.  Under it
  * heads slice of a replicated text tensor      -> backward all-gathers the head slices,
  * all-gather (over heads) of the text output   -> backward takes this rank's head slice,
  * the two image all-to-alls                    -> backward is the opposite all-to-all.
Parameters that only see text rows then have identical gradients on every rank; parameters that see image rows need a sum over the group
(the caller's job, with text-row contributions pre-scaled by 1/P where one matrix serves both).

``core(q, k, v, kv_len) -> out`` is the local attention over [B, S, h, d] operands (on the device: vt355.ops.attn128_fwd / _bwd wrapped
in an autograd.Function; in the CPU tests: a dense torch statement).  Nothing here touches ``oracle/``.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def _a2a(x: torch.Tensor, group) -> torch.Tensor:
    out = torch.empty_like(x)
    dist.all_to_all_single(out, x, group=group)
    return out


def _seq_to_head(x: torch.Tensor, group) -> torch.Tensor:
    """[B, S/P, H, d] (my rows, all heads) -> [B, S, H/P, d] (all rows, my heads); rows ordered by source rank"""
    P = dist.get_world_size(group)
    B, Sl, H, d = x.shape
    if H % P:
        raise ValueError(f"{H} heads do not split over {P} ranks")
    send = x.view(B, Sl, P, H // P, d).permute(2, 0, 1, 3, 4).contiguous()           # [dest rank = head chunk, B, Sl, H/P, d]
    recv = _a2a(send, group)                                                         # [source rank = row chunk, B, Sl, H/P, d]
    return recv.permute(1, 0, 2, 3, 4).reshape(B, P * Sl, H // P, d)


def _head_to_seq(x: torch.Tensor, group) -> torch.Tensor:
    """[B, S, H/P, d] -> [B, S/P, H, d]: the inverse exchange"""
    P = dist.get_world_size(group)
    B, S, h, d = x.shape
    if S % P:
        raise ValueError(f"{S} rows do not split over {P} ranks")
    send = x.view(B, P, S // P, h, d).permute(1, 0, 2, 3, 4).contiguous()            # [dest rank = row chunk, B, Sl, h, d]
    recv = _a2a(send, group)                                                         # [source rank = head chunk, B, Sl, h, d]
    return recv.permute(1, 2, 0, 3, 4).reshape(B, S // P, P * h, d)


class SeqToHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _seq_to_head(x, group)

    @staticmethod
    def backward(ctx, g):
        return _head_to_seq(g.contiguous(), ctx.group), None


class HeadToSeq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _head_to_seq(x, group)

    @staticmethod
    def backward(ctx, g):
        return _seq_to_head(g.contiguous(), ctx.group), None


def _gather_heads(x: torch.Tensor, group) -> torch.Tensor:
    P = dist.get_world_size(group)
    parts = [torch.empty_like(x) for _ in range(P)]
    dist.all_gather(parts, x.contiguous(), group=group)
    return torch.cat(parts, dim=2)


class SliceHeads(torch.autograd.Function):
    """this rank's heads of a replicated [B, L, H, d] tensor; backward: all-gather of the slices (replicated-full gradients)"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        P, r = dist.get_world_size(group), dist.get_rank(group)
        h = x.shape[2] // P
        return x[:, :, r * h:(r + 1) * h].contiguous()

    @staticmethod
    def backward(ctx, g):
        return _gather_heads(g, ctx.group), None


class GatherHeads(torch.autograd.Function):
    """[B, L, H/P, d] -> replicated [B, L, H, d]; backward: this rank's slice of the (replicated) gradient"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _gather_heads(x, group)

    @staticmethod
    def backward(ctx, g):
        P, r = dist.get_world_size(ctx.group), dist.get_rank(ctx.group)
        h = g.shape[2] // P
        return g[:, :, r * h:(r + 1) * h].contiguous(), None


def ulysses_joint_attention(core: Callable, q_img, k_img, v_img, q_txt, k_txt, v_txt, txt_valid: Optional[torch.Tensor] = None, group=None):
    """q/k/v_img [B, Li/P, H, d]: this rank's image rows (after q/k norm and rotary embedding with this rank's rows of the tables);
    q/k/v_txt [B, Lt, H, d]: replicated text rows; txt_valid int [B] valid text rows (keys beyond Li + txt_valid are masked by ``core``).
    -> (out_img [B, Li/P, H, d] sharded, out_txt [B, Lt, H, d] replicated)"""
    group = group if group is not None else dist.group.WORLD
    P = dist.get_world_size(group)
    Li = q_img.shape[1] * P
    if P == 1:
        q, k, v = (torch.cat([a, b], 1) for a, b in ((q_img, q_txt), (k_img, k_txt), (v_img, v_txt)))
        out = core(q, k, v, None if txt_valid is None else txt_valid + Li)
        return out[:, :Li], out[:, Li:]
    q, k, v = (torch.cat([SeqToHead.apply(a.contiguous(), group), SliceHeads.apply(b, group)], 1)
               for a, b in ((q_img, q_txt), (k_img, k_txt), (v_img, v_txt)))
    out = core(q, k, v, None if txt_valid is None else txt_valid + Li)              # [B, Li + Lt, H/P, d]
    return HeadToSeq.apply(out[:, :Li].contiguous(), group), GatherHeads.apply(out[:, Li:].contiguous(), group)


class _DeviceCore(torch.autograd.Function):
    """vt_attn128 (long-sequence head_dim 128) as the local attention: q, k, v [B, S, h, 128] bf16, kv_len int32 [B] | None"""

    @staticmethod
    def forward(ctx, q, k, v, kv_len):
        from . import ops
        B, S, h, d = q.shape
        if d != 128 or not q.is_cuda:
            raise RuntimeError("vt355 device attention core: head_dim 128 on an MI355X device only (no CPU fallback)")
        q3, k3, v3 = (t.contiguous().view(B, S, h * d) for t in (q, k, v))
        o = torch.empty_like(q3)
        lse = torch.empty(B, h, S, dtype=torch.float32, device=q.device)
        ops.attn128_fwd(q3, k3, v3, o, lse, h, d ** -0.5, kv_len=kv_len)
        ctx.save_for_backward(q3, k3, v3, o, lse)
        ctx.kv_len, ctx.h = kv_len, h
        return o.view(B, S, h, d)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        q3, k3, v3, o, lse = ctx.saved_tensors
        B, S, C = q3.shape
        dq = torch.empty_like(q3); dk = torch.empty_like(q3); dv = torch.empty_like(q3)
        ops.attn128_bwd(q3, k3, v3, o, g.contiguous().view(B, S, C), lse, dq, dk, dv, ctx.h, 128 ** -0.5, kv_len=ctx.kv_len)
        sh = (B, S, ctx.h, 128)
        return dq.view(sh), dk.view(sh), dv.view(sh), None


def device_core(q, k, v, kv_len):
    return _DeviceCore.apply(q, k, v, None if kv_len is None else kv_len.to(torch.int32).contiguous())
