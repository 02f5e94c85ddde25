"""A pool of decode worker PROCESSES writing straight into shared memory (why processes: sm_decode_worker.py).

``DecodePool(n)`` starts ``n`` children (``python sm_decode_worker.py``: numpy + Pillow only, no torch, no GPU - started with
``subprocess``, never forked from this process, which may already hold a HIP context).  ``BatchSlots`` is a ring of shared
segments (files under /dev/shm), one per batch in flight; sample i of a batch owns a fixed window of its slot, so a worker needs
no coordination to write its pixels.  ``decode_batch(slot, paths)`` returns zero-copy numpy views of the decoded images and
ground truths inside the slot - exactly what ``pipeline.pack_images`` / ``pack_gts`` take - valid until the slot is reused.

A sample larger than its window (``max_side``) is decoded in-process instead (same function, same bytes - but on the GIL-holding
consumer thread and after the worker has decoded it once for nothing: a warning says so once; size ``max_side`` for the dataset).
A request travels as a JSON line whenever a path holds a tab, a newline or a backslash (tab-separated otherwise: 0.5 us against 2.9).  When a ring of slots is closed its files are unlinked
and the workers are told to unmap them (``DecodePool.drop``): a worker keeps no mapping of a finished loader.
"""
import importlib.util
import json
import os
import queue
import subprocess
import sys
import threading
import uuid
from concurrent.futures import Future
from typing import List, Optional, Sequence, Tuple

import numpy as np

_WORKER = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sm_decode_worker.py"))


def _worker_module():
    """sm_decode_worker.py loaded from ITS path (not through sys.path: the file sits beside the package under a generic name)."""
    mod = sys.modules.get("sm_decode_worker")
    if mod is None or os.path.normpath(getattr(mod, "__file__", "")) != _WORKER:
        spec = importlib.util.spec_from_file_location("sm_decode_worker", _WORKER)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        sys.modules["sm_decode_worker"] = mod
    return mod


def decode_item(p_img, p_gt=None):
    return _worker_module().decode_item(p_img, p_gt)


_WARNED_BIG = False


def _PLAIN(path: str) -> bool:
    """may travel tab-separated: no tab, newline or backslash, and not something the worker would take for a JSON line"""
    return not ("\t" in path or "\n" in path or "\\" in path or "\r" in path) and not path.startswith("J[")


def _cgroup_cpus() -> Optional[float]:
    """CPU quota of this container (cgroup v2 cpu.max / v1 cfs_quota): a box may show 64 cores in its affinity mask and still
    be allowed 16 cores' worth of time - more busy workers than that only add context switches."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        if q > 0:
            return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    except (OSError, ValueError):
        pass
    return None


def default_workers() -> int:
    """Cores this rank may use for decoding, one left for the main thread, at most 32: its share of the CPU affinity (already
    narrowed to the rank by distributed.pin_rank_cores, or divided by LOCAL_WORLD_SIZE here) and of the container's CPU
    quota (always divided: the quota is the node's).  SM_DECODE_WORKERS overrides."""
    if os.environ.get("SM_DECODE_WORKERS"):
        return max(1, int(os.environ["SM_DECODE_WORKERS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 4
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    if os.environ.get("SM_RANK_CORES_PINNED") != "1":
        n = max(1, n // local_world)
    q = _cgroup_cpus()
    if q is not None:
        n = max(1, min(n, int(q) // local_world))
    return max(1, min(32, n - (1 if n > 2 else 0)))


class BatchSlots:
    """``n_slots`` shared segments of ``batch`` windows each: [rgb: max_side^2 * 3 bytes | gt: max_side^2 bytes] per sample."""

    def __init__(self, n_slots: int, batch: int, max_side: int = 640):
        self.batch, self.rgb_cap, self.gt_cap = batch, max_side * max_side * 3, max_side * max_side
        self.stride = self.rgb_cap + self.gt_cap
        self.files, self.maps = [], []
        tag = uuid.uuid4().hex[:12]
        import mmap
        for k in range(n_slots):
            path = f"/dev/shm/sm_decode_{os.getpid()}_{tag}_{k}"
            with open(path, "w+b") as f:
                f.truncate(self.stride * batch)  # sparse: pages materialise as the workers write them
                self.maps.append(mmap.mmap(f.fileno(), 0))
            self.files.append(path)

    def close(self, pool: Optional["DecodePool"] = None) -> None:
        """unlink the segments (the parent's mappings of views still alive stay valid; the memory goes when they do) and, given the
        pool that wrote into them, have every worker drop its mapping too"""
        files, self.files = getattr(self, "files", []), []
        for p in files:
            try:
                os.unlink(p)
            except OSError:
                pass
        if pool is not None and files:
            pool.drop(files)

    def __del__(self):
        self.close()


class DecodePool:
    def __init__(self, workers: Optional[int] = None):
        self.n = workers or default_workers()
        self._q: "queue.Queue" = queue.Queue()
        self._procs, self._threads, self._locks = [], [], []
        self._closed = False
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = "1"
        for _ in range(self.n):
            p = subprocess.Popen([sys.executable, "-u", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                 bufsize=1, env=env, close_fds=True)
            lock = threading.Lock()  # one conversation at a time on a worker's pipes (its feeder thread, or drop())
            t = threading.Thread(target=self._feed, args=(p, lock), daemon=True)
            t.start()
            self._procs.append(p)
            self._threads.append(t)
            self._locks.append(lock)

    def _feed(self, p, lock) -> None:
        """One thread per worker: blocked on the queue or on the worker's pipe, i.e. outside the GIL almost always."""
        while True:
            item = self._q.get()
            if item is None:
                return
            lines, fut = item  # a chunk of samples: one write, then one reply line per sample
            try:
                with lock:
                    p.stdin.write("".join(lines))
                    p.stdin.flush()
                    replies = []
                    for _ in lines:
                        reply = p.stdout.readline()
                        if not reply:
                            raise RuntimeError("decode worker exited")
                        replies.append(reply.rstrip("\n").split("\t"))
                fut.set_result(replies)
            except Exception as e:  # noqa: BLE001
                fut.set_exception(e)

    def drop(self, files: Sequence[str]) -> int:
        """Tell every worker that these shared segments are gone; returns the number of mappings released.  Best effort: a worker
        that has died is skipped (shared_pool replaces the pool on its next use)."""
        released = 0
        if self._closed:
            return 0
        msg = "J" + json.dumps(["drop"] + list(files)) + "\n"
        for p, lock in zip(self._procs, self._locks):
            try:
                with lock:
                    p.stdin.write(msg)
                    p.stdin.flush()
                    reply = p.stdout.readline()
                if reply:
                    released += int(reply.split("\t")[1])
            except (OSError, ValueError, IndexError):
                pass
        return released

    def decode_batch(self, slots: BatchSlots, slot: int, paths: Sequence[Tuple[str, Optional[str]]]):
        """Decode ``paths`` = [(image path, GT path or None)] into windows 0.. of ``slots`` segment ``slot``.  Returns a Future-like
        callable: call it to wait and get (list of rgb views (H, W, 3), list of GT views (H, W) or None)."""
        assert len(paths) <= slots.batch and not self._closed
        # chunks of consecutive samples, about two per worker and batch: the per-message cost (queue, pipe, thread wake-up) is
        # paid per chunk, and the chunks still balance the workers
        chunk = max(1, -(-len(paths) // (2 * self.n)))
        futs: List[Future] = []
        for c0 in range(0, len(paths), chunk):
            lines = []
            for i in range(c0, min(c0 + chunk, len(paths))):
                pi, pg = paths[i]
                ro = i * slots.stride
                if _PLAIN(pi) and (pg is None or (_PLAIN(pg) and pg != "-")):
                    lines.append(f"{pi}\t{pg or '-'}\t{slots.files[slot]}\t{ro}\t{slots.rgb_cap}\t{ro + slots.rgb_cap}\t{slots.gt_cap}\n")
                else:
                    lines.append("J" + json.dumps([pi, pg, slots.files[slot], ro, slots.rgb_cap, ro + slots.rgb_cap, slots.gt_cap]) + "\n")
            f: Future = Future()
            self._q.put((lines, f))
            futs.append(f)
        mm = slots.maps[slot]

        def result():
            rgbs, gts = [], []
            replies = [r for f in futs for r in f.result()]
            for i, r in enumerate(replies):
                if r[0] == "err":
                    raise RuntimeError(f"decode worker: {r[1]} ({paths[i][0]})")
                h, w, gh, gw = (int(v) for v in r[1:5])
                if r[0] == "big":  # beyond the window: decode here (same function)
                    global _WARNED_BIG
                    if not _WARNED_BIG:
                        import warnings
                        _WARNED_BIG = True
                        warnings.warn(f"decode_pool: {paths[i][0]} ({h}x{w}) does not fit its {slots.rgb_cap // 3}-pixel shared-memory window; such "
                                      f"samples are decoded a second time on the consumer thread (BatchSlots(max_side=...) sizes the window)")
                    rgb, m = decode_item(*paths[i])
                else:
                    ro = i * slots.stride
                    rgb = np.frombuffer(mm, np.uint8, h * w * 3, ro).reshape(h, w, 3)
                    m = np.frombuffer(mm, np.uint8, gh * gw, ro + slots.rgb_cap).reshape(gh, gw) if paths[i][1] else None
                rgbs.append(rgb)
                gts.append(m)
            return rgbs, gts
        return result

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        for _ in self._threads:
            self._q.put(None)
        for p in self._procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()

    def __del__(self):
        self.close()


_SHARED = {}
_SHARED_LOCK = threading.Lock()


def shared_pool(workers: Optional[int] = None) -> DecodePool:
    """A process-wide pool per worker count, started on first use and closed at interpreter exit: an evaluation of a few hundred
    images must not pay the start of 15 Python processes (~0.15 s) every call.  ``decode_batch`` is thread-safe."""
    import atexit
    n = workers or default_workers()
    with _SHARED_LOCK:
        pool = _SHARED.get(n)
        if pool is None or pool._closed or any(p.poll() is not None for p in pool._procs):
            if pool is not None:
                pool.close()
            pool = _SHARED[n] = DecodePool(n)
            atexit.register(pool.close)
        return pool
