"""hipGraph replay of MaskFormer.forward.

One forward is ~170 kernel launches issued by sm_maskformer_forward; the decoder's ~80 of them run 5-12 us each, so on
a busy or slow host the launch path (not the GPU) can set the pace.  The forward allocates nothing and never
synchronises, so it captures as is: ``GraphedForward`` records it once per (input shape, stream) into a HIP graph
(torch.cuda.CUDAGraph, i.e. hipStreamBeginCapture / hipGraphLaunch underneath) and afterwards replays it - one launch
per batch.

Admission / eviction (``GraphCachePolicy``, pure Python, unit-tested on CPU): a key is captured only at its
``admit_after + 1``-th sighting (default: third), so one-off shapes - a ragged last batch, native-resolution images -
stay eager and never pay a capture (device-wide synchronise + allocator work); at most ``max_graphs`` graphs live,
evicted least-recently-used, and an evicted key starts counting from zero again.

The graphs bake in raw pointers to the model's packed (split) weights: ``model.weights_generation`` changes whenever
those are rebuilt (``load_state_dict``, ``.to()``, ``refresh_packed()``), and every graph captured under an older
generation is destroyed before the next call.

The outputs of a replay live in the graph's memory pool and are overwritten by the next replay on the same key;
consume them on the same stream before calling again (the Evaluator and bench.py do)."""
from collections import OrderedDict
from typing import Callable, Dict, Hashable, Optional

import torch


class GraphCachePolicy:
    """Which calls run eagerly, which capture, which replay.  ``decide(key)`` returns "eager", "capture" or "replay";
    after a successful capture the owner calls ``admit(key, entry)`` (which may evict the least recently used entry
    through ``on_evict``); ``entry(key)`` returns the stored object."""

    def __init__(self, max_entries: int = 8, admit_after: int = 2, max_seen: int = 4096,
                 on_evict: Optional[Callable[[Hashable, object], None]] = None):
        assert max_entries >= 1 and admit_after >= 0
        self.max_entries, self.admit_after, self.max_seen = max_entries, admit_after, max_seen
        self.on_evict = on_evict
        self._seen: Dict[Hashable, int] = {}
        self._entries: "OrderedDict[Hashable, object]" = OrderedDict()
        self.evictions = 0

    def __len__(self):
        return len(self._entries)

    def decide(self, key: Hashable) -> str:
        if key in self._entries:
            self._entries.move_to_end(key)  # LRU: a hit makes the entry the youngest
            return "replay"
        n = self._seen.get(key, 0)
        if len(self._seen) >= self.max_seen and key not in self._seen:
            self._seen.clear()  # a stream of one-off shapes must not grow this table without bound
        self._seen[key] = n + 1
        return "capture" if n >= self.admit_after else "eager"

    def entry(self, key: Hashable):
        return self._entries[key]

    def admit(self, key: Hashable, entry: object) -> None:
        while len(self._entries) >= self.max_entries:
            old_key, old = self._entries.popitem(last=False)
            self._seen.pop(old_key, None)  # an evicted key earns its next capture from scratch
            self.evictions += 1
            if self.on_evict is not None:
                self.on_evict(old_key, old)
        self._entries[key] = entry

    def clear(self) -> None:
        while self._entries:
            old_key, old = self._entries.popitem(last=False)
            if self.on_evict is not None:
                self.on_evict(old_key, old)
        self._seen.clear()


class GraphedForward:
    def __init__(self, model, enabled: bool = True, max_graphs: int = 8, admit_after: int = 2):
        self.model = model
        self.enabled = enabled
        self._pending_sync = False
        self.policy = GraphCachePolicy(max_graphs, admit_after, on_evict=self._on_evict)
        self._generation = getattr(model, "weights_generation", 0)
        self.captures = 0
        self.replays = 0
        self.failed = None  # first capture error, kept for the caller to report; eager launches take over

    def _on_evict(self, key, entry):
        # the graph's replays may still be in flight on its stream: drain the device before its pool is released
        torch.cuda.synchronize(key[0])

    @property
    def max_graphs(self):
        return self.policy.max_entries

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, **kw):
        if not self.enabled or kw or self.failed is not None or not x.is_cuda:
            return self.model(x, **kw)
        gen = getattr(self.model, "weights_generation", 0)
        if gen != self._generation:  # packed weights were rebuilt: every captured pointer is stale
            self.policy.clear()
            self._generation = gen
        key = (x.device, tuple(x.shape), x.dtype, torch.cuda.current_stream(x.device).cuda_stream,
               getattr(self.model, "attention_path", "auto"))  # a graph bakes in the kernels of one attention path
        what = self.policy.decide(key)
        if what == "eager":
            return self.model(x)
        if what == "capture":
            ent = self._capture(x, key)
            if ent is None:
                return self.model(x)
        else:
            ent = self.policy.entry(key)
        graph, static_x, out = ent[:3]
        static_x.copy_(x, non_blocking=True)
        graph.replay()
        self.replays += 1
        return out

    def _capture(self, x, key):
        try:
            static_x = x.contiguous().float().clone()
            # torch captures every graph on one shared side stream, so the model's per-stream workspace cache would hand
            # all graphs the same scratch: each graph owns its workspace instead
            ws = self.model.new_workspace(static_x)
            graph = torch.cuda.CUDAGraph()
            # thread_local: the capture only polices this thread - an RCCL watchdog thread polling its events while we
            # capture (torch.distributed is initialised in multi-GPU runs) must not invalidate it
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.model(static_x, workspace=ws)
        except Exception as e:  # capture unsupported here: keep the eager HIP path, say why once
            self.failed = f"{type(e).__name__}: {e}"
            return None
        ent = (graph, static_x, out, ws)
        self.policy.admit(key, ent)
        self.captures += 1
        return ent
