"""HunyuanVideo trunk under Ulysses sequence parallelism (vt355.hunyuan.HunyuanBlocks.set_sequence_parallel + vt355.sp): two ranks that share
the one card of the test box (gloo moves the exchanged rows through host memory here; on a node the same calls are RCCL all-to-alls over
xGMI) against the same blocks run unsharded in the same process.  What must hold under sp.py's partial-sum convention: the local image rows
and the (replicated) text rows of the output equal the unsharded output's; image-row input gradients equal their shard; text-row, modulation
-vector and PARAMETER gradients summed over the ranks equal the unsharded ones.  The reference trains HunyuanVideo without sequence
parallelism; the unsharded path is the one pinned to it (tests/test_hunyuan_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _worker(rank, world, port, lora, res):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vt355.hunyuan import HunyuanBlocks
        dev = torch.device("cuda:0")
        D, H, B, Li, Lt = 256, 2, 2, 96, 12
        m = HunyuanBlocks(hidden_size=D, heads_num=H, mm_double_blocks_depth=2, mm_single_blocks_depth=2, lora_rank=4 if lora else 0, lora_alpha=2.0)
        m.init_weights(3)
        if lora:
            m.lora.init_weights(5, zero_b=False)
        m.to(dev)
        ts = m.enable_lora_training() if lora else m.enable_training()
        owner = m.lora if lora else m
        g = torch.Generator().manual_seed(17)
        img, txt, vec = (torch.randn(B, Li, D, generator=g) * 0.5).to(BF), (torch.randn(B, Lt, D, generator=g) * 0.5).to(BF), torch.randn(B, D, generator=g).to(BF)
        ang = torch.rand(Li, 64, generator=g) * 6.28
        cos, sin = torch.repeat_interleave(ang.cos(), 2, dim=1).contiguous(), torch.repeat_interleave(ang.sin(), 2, dim=1).contiguous()
        tv = torch.tensor([12, 7])
        gout = (torch.randn(B, Li + Lt, D, generator=g) * 0.1).to(BF)
        gout[1, Li + 7:] = 0                                                        # padding text rows carry no gradient

        def run(sl, group):
            m.set_sequence_parallel(group)
            ts.grad.zero_()
            xi, xt, xv = [t.to(dev).requires_grad_(True) for t in (img[:, sl], txt, vec)]
            out = m(xi, xt, xv, tv.to(dev), (cos[sl].to(dev), sin[sl].to(dev)))
            go = torch.cat([gout[:, sl], gout[:, Li:]], 1) if group is not None else gout
            if group is not None:                                                   # the text rows' incoming gradient arrives ONCE in total: rank 0 carries it
                go = go.clone()
                if rank != 0:
                    go[:, sl.stop - sl.start:] = 0
            out.backward(go.to(dev))
            torch.cuda.synchronize()
            return out.detach().float().cpu(), xi.grad.float().cpu(), xt.grad.float().cpu(), xv.grad.float().cpu(), ts.grad.detach().clone()

        full = run(slice(0, Li), None)
        n = Li // world
        sl = slice(rank * n, (rank + 1) * n)
        part = run(sl, dist.group.WORLD)
        m.set_sequence_parallel(None)
        valid = torch.ones(B, Lt, 1); valid[1, 7:] = 0
        errs = {"out_img": _rel(part[0][:, :n], full[0][:, sl]), "out_txt": _rel(part[0][:, n:] * valid, full[0][:, Li:] * valid),
                "dimg": _rel(part[1], full[1][:, sl])}
        for name, idx in (("dtxt", 2), ("dvec", 3)):
            tot = part[idx].to(dev); dist.all_reduce(tot)
            errs[name] = _rel(tot.cpu() * (valid if idx == 2 else 1), full[idx] * (valid if idx == 2 else 1))
        gsum = part[4].clone(); dist.all_reduce(gsum)
        errs["params"] = _rel(gsum, full[4])
        worst = 0.0
        for name in owner.shapes:
            a, b = owner._view(gsum, name), owner._view(full[4], name)
            if b.norm().item() > 0:
                worst = max(worst, _rel(a, b))
        errs["worst_param"] = worst
        res[rank] = errs
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lora", [False, True])
def test_blocks_under_ulysses_world2_match_unsharded(dev, lora):
    mgr = mp.Manager(); res = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), lora, res), nprocs=2, join=True)
    for r in (0, 1):
        print(f"[hunyuan sp-2 {'lora' if lora else 'fullft'}] rank {r}: " + ", ".join(f"{k} {v:.2e}" for k, v in res[r].items()))
        for k, v in res[r].items():
            assert v < (5e-2 if k == "worst_param" else 1e-2), (r, k, v)


def _worker_model(rank, world, port, res):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vt355.hunyuan import HYVideoDiffusionTransformer, HunyuanVideoFlow
        dev = torch.device("cuda:0")
        m = HYVideoDiffusionTransformer(in_channels=4, hidden_size=256, heads_num=2, mm_double_blocks_depth=2, mm_single_blocks_depth=2,
                                        text_states_dim=64, text_states_dim_2=32, lora_rank=4, lora_alpha=2.0)
        m.init_weights(3); m.lora.init_weights(5, zero_b=False)
        m.to(dev)
        flow = HunyuanVideoFlow(model=m, learning_rate=1e-3).to(dev)
        flow.configure_optimizers()
        ts = m.lora.train_state
        g = torch.Generator().manual_seed(23)
        B, L = 2, 10
        x0 = torch.randn(B, 4, 3, 8, 12, generator=g).to(dev)
        noise = torch.randn(B, 4, 3, 8, 12, generator=g).to(dev)
        text = torch.randn(B, L, 64, generator=g).to(dev); pooled = torch.randn(B, 32, generator=g).to(dev)
        mask = torch.ones(B, L, dtype=torch.long); mask[1, 6:] = 0
        sigma = torch.tensor([0.3, 0.8], device=dev)

        def run(group):
            m.set_sequence_parallel(group)
            ts.grad.zero_()
            loss = flow.loss_from(x0, text, mask.to(dev), pooled, sigma, noise)
            loss.backward()
            torch.cuda.synchronize()
            return loss.detach().float(), ts.grad.detach().clone()

        l_full, g_full = run(None)
        l_part, g_part = run(dist.group.WORLD)
        dist.all_reduce(l_part); dist.all_reduce(g_part)                             # what the data-parallel reducer does: the average over ranks
        l_part /= world; g_part /= world
        errs = {"loss": abs(l_part.item() - l_full.item()) / l_full.item(), "adapter_grads": _rel(g_part, g_full)}
        # training_step: the ranks of a group draw ONE sigma and ONE noise
        torch.manual_seed(100 + rank)
        batch = {"latents": x0, "prompt_embeds": text, "prompt_attention_mask": mask.to(dev), "pooled_prompt_embeds": pooled}
        l_step = flow.training_step(batch)
        l_step.backward()
        assert torch.isfinite(l_step)
        m.set_sequence_parallel(None)
        res[rank] = errs
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_whole_transformer_lora_step_under_ulysses_world2(dev):
    """HunyuanVideoFlow.loss_from with the image tokens split over two ranks: mean of the ranks' losses = the unsharded loss, mean of the
    ranks' adapter gradients = the unsharded gradient (the all-reduce a data-parallel reducer performs anyway)"""
    mgr = mp.Manager(); res = mgr.dict()
    mp.spawn(_worker_model, args=(2, _free_port(), res), nprocs=2, join=True)
    for r in (0, 1):
        print(f"[hunyuan sp-2 model] rank {r}: " + ", ".join(f"{k} {v:.2e}" for k, v in res[r].items()))
        assert res[r]["loss"] < 2e-3 and res[r]["adapter_grads"] < 2e-2, res[r]
