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
    if x.is_cuda and dist.get_backend(group) == "gloo":          # rehearsal of several ranks on one card: gloo exchanges host memory
        out = torch.empty(x.shape, dtype=x.dtype)
        dist.all_to_all_single(out, x.cpu(), group=group)
        return out.to(x.device)
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


# ----------------------------------------------------------------------------------------------------------------------------------------
# The same exchange for the tape engine of hunyuan.py (no torch autograd): four explicit steps on the engine's joint buffers.
#
# Layout on a rank: joint rows of sample b are [its Li/P image rows ; the Lt text rows] (``Ljl = Li/P + Lt``), columns [q | k | v] of all
# H heads.  The attention runs on ``joint2`` [B, Li + Lt, 3 * (H/P) * d]: every image row (ordered by source rank = global order) and the
# text rows, this rank's H/P heads.
#
# Gradient convention here: PARTIAL SUMS.  The text stream is computed P times (once per rank, identical values); the exchange is
# differentiated as the graph it really is -- rank r's copy of the text rows feeds the attention only through head slice r, and rank r's
# attention output for the text rows is broadcast into every rank's copy.  So the backward of the all-gather is a sum over the ranks'
# incoming gradients (then this rank's head slice), and the backward of the head slice is a zero-padded placement.  Text-row gradients
# (and everything downstream of them, the modulation vector's gradient included) are then partial on each rank and their SUM over the group
# is the gradient -- exactly how a data-parallel reducer treats ranks: with each rank back-propagating the mean loss of ITS image rows, the
# group AVERAGE of every parameter gradient is the gradient of the mean loss over all rows.  No 1/P bookkeeping anywhere.
# ----------------------------------------------------------------------------------------------------------------------------------------
def _geom(group, H: int):
    group = group if group is not None else dist.group.WORLD
    P, r = dist.get_world_size(group), dist.get_rank(group)
    if H % P:
        raise ValueError(f"{H} heads do not split over {P} ranks")
    return group, P, r, H // P


def joint_to_heads(joint: torch.Tensor, B: int, Lil: int, Lt: int, H: int, d: int, group=None) -> torch.Tensor:
    """joint [B * (Lil + Lt), 3 H d] (local rows, all heads) -> joint2 [B, P Lil + Lt, 3 (H/P) d]: ONE all-to-all for q, k and v together"""
    group, P, r, h = _geom(group, H)
    j5 = joint.view(B, Lil + Lt, 3, P, h * d)
    send = j5[:, :Lil].permute(3, 0, 1, 2, 4).contiguous()                           # [dest rank = head chunk, B, Lil, 3, h d]
    recv = _a2a(send, group)                                                         # [source rank = row chunk, B, Lil, 3, h d]
    joint2 = torch.empty(B, P * Lil + Lt, 3 * h * d, dtype=joint.dtype, device=joint.device)
    joint2[:, :P * Lil].view(B, P, Lil, 3, h * d).copy_(recv.permute(1, 0, 2, 3, 4))
    joint2[:, P * Lil:].view(B, Lt, 3, h * d).copy_(j5[:, Lil:, :, r])
    return joint2


def heads_to_rows(o2: torch.Tensor, o3: torch.Tensor, B: int, Lil: int, Lt: int, H: int, d: int, group=None) -> None:
    """o2 [B, P Lil + Lt, (H/P) d] (all rows, my heads) -> o3 [B, Lil + Lt, H d] view of the local output (rows may be strided)"""
    group, P, r, h = _geom(group, H)
    Li = P * Lil
    send = o2[:, :Li].view(B, P, Lil, h * d).permute(1, 0, 2, 3).contiguous()         # [dest rank = row chunk, B, Lil, h d]
    recv = _a2a(send, group)                                                         # [source rank = head chunk, B, Lil, h d]
    o3[:, :Lil].copy_(recv.permute(1, 2, 0, 3).reshape(B, Lil, H * d))
    txt = o2[:, Li:].contiguous()                                                    # [B, Lt, h d]: every rank's copy of the text rows needs all heads
    parts = [torch.empty_like(txt) for _ in range(P)]
    dist.all_gather(parts, txt, group=group)
    o3[:, Lil:].copy_(torch.stack(parts, 2).reshape(B, Lt, H * d))


def rows_grad_to_heads(g3: torch.Tensor, B: int, Lil: int, Lt: int, H: int, d: int, group=None) -> torch.Tensor:
    """backward of heads_to_rows: g3 [B, Lil + Lt, H d] view (local rows; text rows PARTIAL) -> go2 [B, P Lil + Lt, (H/P) d]"""
    group, P, r, h = _geom(group, H)
    send = g3[:, :Lil].reshape(B, Lil, P, h * d).permute(2, 0, 1, 3).contiguous()     # [dest rank = head chunk, B, Lil, h d]
    recv = _a2a(send, group)                                                         # [source rank = row chunk, B, Lil, h d]
    go2 = torch.empty(B, P * Lil + Lt, h * d, dtype=g3.dtype, device=g3.device)
    go2[:, :P * Lil].view(B, P, Lil, h * d).copy_(recv.permute(1, 0, 2, 3))
    acc = torch.float32 if g3.dtype in (torch.bfloat16, torch.float16) else g3.dtype
    gt = g3[:, Lil:].to(acc).contiguous()                                            # sum of the ranks' partial text gradients, in fp32
    dist.all_reduce(gt, group=group)
    go2[:, P * Lil:].copy_(gt.view(B, Lt, P, h * d)[:, :, r])
    return go2


def heads_grad_to_joint(dj2: torch.Tensor, B: int, Lil: int, Lt: int, H: int, d: int, group=None) -> torch.Tensor:
    """backward of joint_to_heads: dj2 [B, P Lil + Lt, 3 (H/P) d] -> dj [B * (Lil + Lt), 3 H d] (text rows: this rank's heads, zeros elsewhere)"""
    group, P, r, h = _geom(group, H)
    send = dj2[:, :P * Lil].view(B, P, Lil, 3, h * d).permute(1, 0, 2, 3, 4).contiguous()   # [dest rank = row chunk, B, Lil, 3, h d]
    recv = _a2a(send, group)                                                         # [source rank = head chunk, B, Lil, 3, h d]
    dj = torch.empty(B * (Lil + Lt), 3 * H * d, dtype=dj2.dtype, device=dj2.device)
    d5 = dj.view(B, Lil + Lt, 3, P, h * d)
    d5[:, :Lil].copy_(recv.permute(1, 2, 3, 0, 4))
    d5[:, Lil:].zero_()
    d5[:, Lil:, :, r].copy_(dj2[:, P * Lil:].view(B, Lt, 3, h * d))
    return dj
