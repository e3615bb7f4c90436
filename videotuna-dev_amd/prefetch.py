"""Overlap of the frozen encoders with the DiT step (BASELINE north_star: "next micro-batch's VAE encode overlapped with
the gradient all-reduce").

The reference encodes every micro-batch inside ``training_step`` (``get_batch_input`` -> per-sample VAE encode,
cogvideo_pl.py:797-813, then T5, 822-831) on the same stream as the DiT, so the denoiser waits for two frozen,
gradient-free networks.  Here the raw batch i+1 is pushed through the user-supplied encoders on a SIDE HIP stream as soon
as batch i is handed to the trainer: its kernels fill the gaps of the main stream (the RCCL all-reduce of step i, the
optimizer) and the DiT of step i+1 finds ``{"latents", "prompt_embeds"}`` ready.  The encoders themselves (3D causal VAE,
T5-XXL) are outside this engine (8(f)); anything callable works.

    for batch in EncoderPrefetcher(loader, encode=workflow.encode_raw_batch, device=dev):
        loss = workflow.training_step(batch)      # batch is already {"latents", "prompt_embeds"[, "image_latents"]}
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterable, Iterator, Optional

import torch


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class EncoderPrefetcher:
    def __init__(self, batches: Iterable[Any], encode: Callable[[Any], Dict[str, Any]], device=None, depth: int = 1):
        """batches: iterable of raw batches (reference schema {"video", "caption"[, "image"]});  encode: raw batch ->
        encoded batch, run under torch.no_grad() on the side stream;  depth: how many batches are encoded ahead."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.batches, self.encode, self.depth = batches, encode, depth
        self.device = torch.device(device) if device is not None else None
        self.on_gpu = self.device is not None and self.device.type == "cuda"
        self.side: Optional[torch.cuda.Stream] = torch.cuda.Stream(device=self.device) if self.on_gpu else None

    def _launch(self, raw):
        if not self.on_gpu:                        # host-only use (tests of the sequencing): same order, no overlap
            with torch.no_grad():
                return self.encode(raw), None
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)                # inputs produced on the main stream (e.g. H2D copies) are visible
        for t in _tensors(raw):
            if t.is_cuda:
                t.record_stream(self.side)         # ... and must not be recycled by the allocator before the side stream read them
        with torch.cuda.stream(self.side), torch.no_grad():
            out = self.encode(raw)
            ev = torch.cuda.Event()
            ev.record(self.side)
        return out, ev

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.batches)
        queue = []
        if self.on_gpu:                 # encoder kernels will share the CUs with the DiT's backward from here on: the
            from . import ops           # attention backward must not rely on all its persistent workgroups being resident
            ops.declare_side_stream(True)
        try:
            for raw in it:
                queue.append(self._launch(raw))
                if len(queue) > self.depth:
                    yield self._hand_over(*queue.pop(0))
            while queue:
                yield self._hand_over(*queue.pop(0))
        finally:
            if self.on_gpu:
                ops.declare_side_stream(False)

    def _hand_over(self, out, ev):
        if ev is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)                    # device-side dependency only: the host never blocks
            for t in _tensors(out):
                if t.is_cuda:
                    t.record_stream(main)          # the caching allocator must not recycle them under the side stream
        return out


class EncodingCache:
    """Per-sample cache of what the frozen encoders produce (the reference re-runs the VAE and T5-XXL on every
    step, cogvideo_pl.py:797-813).  Two stores, both on the device (a 49x480x720 clip is 2.2 MB of latent moments x 2 and 1.9 MB of
    prompt embedding; 288 GB hold tens of thousands):
      * prompt embeddings keyed by the caption string -- exact: the text encoder is frozen and deterministic;
      * latent MOMENTS (mean, std of the VAE posterior) keyed by the batch's optional ``"index"`` entry -- the sample is re-drawn on
        every hit, so the step sees the same distribution as the reference's ``latent_dist.sample()``; only valid when the loader
        applies no random crop / frame jitter to that index, hence opt-in by providing the key.
    ``max_bytes`` bounds the store (oldest entries go first)."""

    def __init__(self, max_bytes: int = 64 << 30):
        self.max_bytes, self.bytes = max_bytes, 0
        self.text: Dict[str, torch.Tensor] = {}
        self.moments: Dict[Any, tuple] = {}
        self.hits = {"text": 0, "latent": 0}
        self.misses = {"text": 0, "latent": 0}

    def _room(self, n: int):
        self.bytes += n
        for store in (self.moments, self.text):
            while self.bytes > self.max_bytes and store:
                v = store.pop(next(iter(store)))
                self.bytes -= sum(t.numel() * t.element_size() for t in (v if isinstance(v, tuple) else (v,)))

    def prompt_embeds(self, captions, encode: Callable[[list], torch.Tensor]) -> torch.Tensor:
        miss = [c for c in dict.fromkeys(captions) if c not in self.text]
        self.misses["text"] += len(miss); self.hits["text"] += len(captions) - len(miss)
        now = {c: self.text[c] for c in captions if c in self.text}
        if miss:
            emb = encode(miss)
            for c, e in zip(miss, emb):
                e = e.detach().clone()
                now[c] = e
                self._room(e.numel() * e.element_size())
                self.text[c] = e
        return torch.stack([now[c] for c in captions], 0)

    def latents(self, indices, videos, posterior: Callable[[torch.Tensor], Any], scaling_factor: float, generator=None) -> torch.Tensor:
        """posterior(video [1,3,T,H,W]) -> object with .mean / .std (vt355.vae.DiagonalGaussianDistribution)"""
        out = []
        for n, i in enumerate(indices):                                              # videos[n] is only touched on a miss (it may be lazy)
            key = int(i)
            if key not in self.moments:
                self.misses["latent"] += 1
                v = videos[n]
                d = posterior(v.unsqueeze(0) if v.dim() == 4 else v)
                m, sdev = d.mean.detach().clone(), d.std.detach().clone()
                self._room((m.numel() + sdev.numel()) * m.element_size())
                self.moments[key] = (m, sdev)
            else:
                self.hits["latent"] += 1
                m, sdev = self.moments[key]
            eps = torch.randn(m.shape, generator=generator, device=m.device, dtype=m.dtype)
            out.append((m + sdev * eps) * scaling_factor)
        return torch.cat(out, 0)
