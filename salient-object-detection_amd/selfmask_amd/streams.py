"""Batches in flight on several HIP streams.

One batch's decoder and metric kernels are short, dependent launches that leave most of the 256 CUs idle; another
batch's encoder GEMMs fill them.  ``StreamRing`` deals consecutive batches round-robin onto ``n`` streams (each stream
gets its own forward workspace, see MaskFormer._get_workspace; per-call tensors come from torch's stream-aware caching
allocator) and joins them back into the caller's stream.  Results do not depend on ``n``: every batch runs the same
kernels in the same order on its own stream.  Measured on MI355X (B=64, ViT-S/16 224^2): 1 / 2 / 3 / 4 streams =
12.5k / 14.9k / 15.7k / 14.8k images/s."""
from contextlib import contextmanager

import torch

DEFAULT_STREAMS = 3


class StreamRing:
    def __init__(self, device: torch.device, n: int = DEFAULT_STREAMS):
        if n < 1:
            raise ValueError("StreamRing needs at least one stream")
        self.device = device
        self.home = torch.cuda.current_stream(device)
        # experiment hook (round 3, measured and left off: see DESIGN.md): SM_STREAM_PRIORITIES="-1,0,0" gives the ring's streams
        # HIP priorities (lower = more urgent)
        import os
        prio = [int(v) for v in os.environ.get("SM_STREAM_PRIORITIES", "").split(",") if v.strip()]
        self.streams = ([torch.cuda.Stream(device=device, priority=prio[i % len(prio)]) if prio else torch.cuda.Stream(device=device)
                         for i in range(n)] if n > 1 else [self.home])
        self._k = 0
        self.fork()

    def fork(self):
        """Work queued on the caller's stream so far (inputs, weights) is visible to every ring stream."""
        for s in self.streams:
            if s is not self.home:
                s.wait_stream(self.home)

    @contextmanager
    def next(self):
        s = self.streams[self._k % len(self.streams)]
        self._k += 1
        with torch.cuda.stream(s):
            yield s

    def join(self):
        """The caller's stream waits for everything queued on the ring."""
        for s in self.streams:
            if s is not self.home:
                self.home.wait_stream(s)
