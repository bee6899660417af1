"""Host -> device input staging for the training loop: the reference moves every batch with blocking `.cuda()` calls inside
`Batch.to_tuple()` (dataset/batching.py:67-87).  With the GPU step at ~140 ms for 8 x 16 x 256 x 256 frames (100 MB per batch) a blocking copy
would cost ~4 ms per step; this iterator keeps ONE batch in flight: pinned host buffers, an asynchronous copy on its own HIP stream, and an event
the consumer's stream waits on -- the copy of batch i+1 overlaps the training step of batch i.

    for batch_tuple in DevicePrefetcher(dataloader, device):          # yields (observations, actions, rewards, dones) on `device`
        trainer.compute_losses(model, batch_tuple, ...)
"""
from typing import Iterable, Iterator, Optional, Tuple

import torch


def _as_tuple(batch) -> Tuple:
    """`Batch` objects of the reference expose to_tuple(cuda=...) (dataset/batching.py:67); tuples / lists pass through."""
    if hasattr(batch, "to_tuple"):
        try:
            return tuple(batch.to_tuple(cuda=False))
        except TypeError:
            return tuple(batch.to_tuple())
    return tuple(batch)


class DevicePrefetcher:
    def __init__(self, loader: Iterable, device, pin: bool = True):
        self.loader = loader
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.pin = pin and self.on_gpu
        self.stream: Optional[torch.cuda.Stream] = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._pinned = {}          # (slot, index) -> pinned staging tensor, reused across batches of the same shape
        self._slot_event = [None, None]   # last copy issued from each slot's pinned buffers (host waits on it before overwriting them)

    def _stage(self, batch, slot):
        items = _as_tuple(batch)
        out = []
        for i, t in enumerate(items):
            if not torch.is_tensor(t):
                out.append(t)
                continue
            if not self.on_gpu:
                out.append(t.to(self.device))
                continue
            src = t
            if self.pin and not t.is_pinned() and t.device.type == "cpu":
                key = (slot, i)
                buf = self._pinned.get(key)
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    self._pinned[key] = buf
                buf.copy_(t)
                src = buf
            out.append(src.to(self.device, non_blocking=True))
        return tuple(out)

    def __iter__(self) -> Iterator[Tuple]:
        it = iter(self.loader)
        slot = 0

        def fetch():
            nonlocal slot
            try:
                b = next(it)
            except StopIteration:
                return None, None
            if not self.on_gpu:
                return self._stage(b, 0), None
            if self._slot_event[slot] is not None:
                self._slot_event[slot].synchronize()       # the DMA that read this slot's pinned buffers two batches ago must be done
            with torch.cuda.stream(self.stream):
                staged = self._stage(b, slot)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self._slot_event[slot] = ev
            slot ^= 1
            return staged, ev

        nxt, ev = fetch()
        while nxt is not None:
            cur, cur_ev = nxt, ev
            if cur_ev is not None:
                torch.cuda.current_stream(self.device).wait_event(cur_ev)
                for t in cur:
                    if torch.is_tensor(t):
                        t.record_stream(torch.cuda.current_stream(self.device))
            nxt, ev = fetch()          # batch i+1 starts copying before batch i is consumed
            yield cur
