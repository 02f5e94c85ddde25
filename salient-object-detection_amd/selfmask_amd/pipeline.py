"""Input pipeline of the evaluator / serving path: JPEG decode on a pool of host threads, everything
after it - the reference's ``T.Resize`` (PIL bilinear), ``ToTensor`` and ``Normalize`` - on the MI355X
(``csrc/preprocess.hip``), with pinned staging buffers so decode, H2D copy and the model overlap.

Reference semantics kept (datasets/base_dataset.py:228-256, duts.py:108-147, custom_dataset.py:26-32, app.py:198-205):
RGB convert; fixed S x S resize = Pillow's BILINEAR resample (anti-aliased triangle filter, 22-bit fixed-point taps, uint8
after each pass); ``/255``; mean (0.485, 0.456, 0.406) / std (0.229, 0.224, 0.225); GT in mode "L", ``m > 0`` when its max
exceeds 1.  The host computes only what depends on sizes (tap tables) or on nothing (the 768-entry normalisation table);
per-pixel arithmetic is the device's.
"""
import math
import os
from concurrent.futures import ThreadPoolExecutor
from functools import lru_cache
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

from . import _native as N
from .datasets import MEAN, STD

PRECISION_BITS = 32 - 8 - 2  # Pillow, libImaging/Resample.c


def normalize_lut() -> np.ndarray:
    """lut[c*256 + v] = (v / 255 - mean_c) / std_c in fp32: ToTensor (``.div(255)``) then Normalize (sub, div), the
    reference's own expressions evaluated for every possible uint8 value."""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return np.concatenate([((v - np.float32(MEAN[c])) / np.float32(STD[c])).astype(np.float32) for c in range(3)])


@lru_cache(maxsize=4096)
def pil_resize_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0), box = the whole axis.
    Returns (bounds int32 (out, 2) = first input index / tap count, taps int32 (out, ks), ks).  Plain Python floats are
    IEEE doubles evaluated in the C code's order, so the fixed-point taps are Pillow's bit for bit."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ks = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    taps = np.zeros((out_size, ks), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            wv = 1.0 - t if t < 1.0 else 0.0
            w.append(wv)
            ww += wv
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            taps[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, taps, ks


def resize_reference_numpy(img: np.ndarray, S: int) -> np.ndarray:
    """The same integer arithmetic on the host (numpy): used by the CPU tests to pin ``pil_resize_coeffs`` against
    ``PIL.Image.resize`` without a GPU.  img (H, W, 3) uint8 -> (S, S, 3) uint8."""
    H, W, _ = img.shape

    def one_pass(a, n_in, axis_len):  # a: (rows, n_in, 3) -> (rows, S, 3)
        bounds, taps, ks = pil_resize_coeffs(n_in, S)
        out = np.empty((a.shape[0], S, 3), np.uint8)
        for xx in range(S):
            x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
            acc = (a[:, x0:x0 + n, :].astype(np.int64) * taps[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        return out

    tmp = one_pass(img, W, S)                                       # horizontal first (Resample.c: ImagingResampleInner)
    return one_pass(tmp.transpose(1, 0, 2), H, S).transpose(1, 0, 2)  # then vertical on the uint8 intermediate


class PinnedPool:
    """Reusable page-locked staging buffers.  ``pin_memory()`` per batch costs milliseconds (and now and then tens of
    milliseconds) of hipHostMalloc; the pool hands out a buffer again once the event recorded behind its last H2D copy has
    completed, and grows only when every buffer of a size class is still in flight."""

    _BUSY = object()  # handed out, release_after not called yet

    def __init__(self):
        import threading
        self._free = {}  # (dtype, rounded size) -> list of [tensor, None (free) | _BUSY | event behind the last copy]
        self._by_ptr = {}  # data_ptr of a buffer -> its entry (release_after is called twice per image at batch 1)
        self._lock = threading.Lock()

    @staticmethod
    def _round(n: int) -> int:
        r = 4096
        while r < n:
            r *= 2
        return r

    def get(self, n: int, dtype) -> torch.Tensor:
        """Thread-safe: the loader's packing thread fills buffers for batch k + 1 while the consumer still owns batch k's."""
        key = (dtype, self._round(max(n, 1)))
        with self._lock:
            for ent in self._free.setdefault(key, []):
                if ent[1] is None or (ent[1] is not self._BUSY and ent[1].query()):
                    ent[1] = self._BUSY
                    return ent[0][:n]
            ent = [torch.empty(key[1], dtype=dtype).pin_memory(), self._BUSY]
            self._free[key].append(ent)
            self._by_ptr[ent[0].data_ptr()] = ent
            return ent[0][:n]

    def release_after(self, tensors, stream) -> None:
        """The buffers behind ``tensors`` may be reused once the work queued on ``stream`` so far has finished."""
        ev = torch.cuda.Event()
        ev.record(stream)
        with self._lock:
            for t in tensors:
                ent = self._by_ptr.get(t.data_ptr())  # a handed-out view starts at its buffer's first byte
                if ent is not None:
                    ent[1] = ev


_POOL = PinnedPool()


def pack_images(images: Sequence[np.ndarray], S: Optional[int], pinned: bool = False):
    """images: list of (H, W, 3) uint8 arrays -> (pixels u8 tensor, coef int32 tensor, descr u8 tensor, max_h, max_pixels,
    out_elems).  S = output side for the resize path, None for native resolution.  ``pinned``: buffers come from the
    page-locked pool (the caller must call ``_POOL.release_after`` once the H2D copies are queued)."""
    import ctypes
    B = len(images)
    descr = (N.PreImage * B)()
    offs, off = [], 0
    for im in images:
        offs.append(off)
        off += (im.shape[0] * im.shape[1] * 3 + 15) & ~15
    mk = (lambda n, dt: _POOL.get(n, dt)) if pinned else (lambda n, dt: torch.empty(n, dtype=dt))
    pixels = mk(max(off, 16), torch.uint8)
    pv = pixels.numpy()
    coef_parts, coef_index, ci = [], {}, 0
    out_off = 0
    for b, im in enumerate(images):
        assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, "decoded images must be (H, W, 3) uint8"
        h, w = im.shape[:2]
        pv[offs[b]:offs[b] + h * w * 3] = im.reshape(-1)
        d = descr[b]
        d.off, d.H, d.W, d.out_off = offs[b], h, w, out_off
        out_off += 3 * h * w
        if S is not None:
            for n_in, key in ((w, "x"), (h, "y")):
                if n_in not in coef_index:
                    bounds, taps, ks = pil_resize_coeffs(n_in, S)
                    coef_index[n_in] = (ci, ks)
                    coef_parts += [bounds.reshape(-1), taps.reshape(-1)]
                    ci += bounds.size + taps.size
                o, ks = coef_index[n_in]
                if key == "x":
                    d.coef_x, d.ksx = o, ks
                else:
                    d.coef_y, d.ksy = o, ks
    coef = mk(max(ci, 1), torch.int32)
    if ci:
        coef.numpy()[:ci] = np.concatenate(coef_parts)
    dt = mk(ctypes.sizeof(descr), torch.uint8)
    dt.numpy()[:] = np.frombuffer(bytes(descr), np.uint8)
    return pixels, coef, dt, max(im.shape[0] for im in images), max(im.shape[0] * im.shape[1] for im in images), out_off


_LUT = {}


def _lut(device):
    t = _LUT.get(device)
    if t is None:
        t = _LUT[device] = torch.from_numpy(normalize_lut()).to(device)
    return t


def pack_gts(gts: Sequence[np.ndarray]):
    """Ground-truth masks of a batch -> (flat uint8 pinned tensor, descriptor bytes pinned tensor, shapes): the host half of
    ops.GtBatch, done on the loader's packing thread."""
    import ctypes
    B = len(gts)
    descr = (N.EvalImage * B)()
    off = 0
    for b, g in enumerate(gts):
        assert g.dtype == np.uint8 and g.ndim == 2, "ground-truth masks must be 2-D uint8 arrays"
        descr[b].gt_off, descr[b].H, descr[b].W = off, g.shape[0], g.shape[1]
        off += g.size
    flat = _POOL.get(max(off, 1), torch.uint8)
    fv = flat.numpy()
    o = 0
    for g in gts:
        fv[o:o + g.size] = g.reshape(-1)
        o += g.size
    dt = _POOL.get(ctypes.sizeof(descr), torch.uint8)
    dt.numpy()[:] = np.frombuffer(bytes(descr), np.uint8)
    return flat, dt, [(int(g.shape[0]), int(g.shape[1])) for g in gts]


def native_buckets(sizes: Sequence[Tuple[int, int]], patch: int, max_batch: int) -> List[List[int]]:
    """Native-resolution evaluation in token-grid buckets: positions (into ``sizes``, a list of (H, W)) grouped by the patch
    grid (ceil(H / P), ceil(W / P)) their image pads to, cut into batches of at most ``max_batch``, in order of first
    appearance; inside a batch the dataset order is kept.  Every position appears exactly once."""
    groups = {}
    for i, (h, w) in enumerate(sizes):
        groups.setdefault((-(-h // patch), -(-w // patch)), []).append(i)
    out = []
    for idx in groups.values():
        out += [idx[s:s + max_batch] for s in range(0, len(idx), max_batch)]
    return out


def preprocess_on_device(images, S: Optional[int], device, pinned: bool = False, return_u8: bool = False, packed=None,
                         pad_to: Optional[Tuple[int, int]] = None):
    """Decoded uint8 images -> normalised fp32 model input on ``device``.  S given: (B, 3, S, S) after the PIL-exact
    bilinear resize (``return_u8``: also the resized uint8 images (B, S, S, 3)); S None: a list of (1, 3, H, W) tensors at
    native resolution (views of one buffer), or - ``pad_to=(Hp, Wp)`` - ONE (B, 3, Hp, Wp) batch with every image in the
    top-left corner and zeros elsewhere (what make_input_divisible builds per image, vision_transformer.py:260-267)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("the input pipeline's resize / normalise kernels run on a HIP device (no CPU fallback)")
    # ``packed``: the result of pack_images(images, S, pinned=True) prepared ahead of time (PrefetchingLoader(pack_size=...));
    # ``images`` then only needs the (H, W) of every image (arrays or shape tuples)
    if packed is not None:
        pinned = True
    pixels, coef, descr, max_h, max_px, out_elems = packed if packed is not None else pack_images(images, S, pinned)
    st = torch.cuda.current_stream(device).cuda_stream
    pd, cd, dd = (t.to(device, non_blocking=True) for t in (pixels, coef, descr))
    if pinned:
        _POOL.release_after((pixels, coef, descr), torch.cuda.current_stream(device))
    lib = N.load()
    B = len(images)
    if S is not None:
        tmp = torch.empty((B, max_h * S * 3), dtype=torch.uint8, device=device)
        out = torch.empty((B, 3, S, S), dtype=torch.float32, device=device)
        u8 = torch.empty((B, S, S, 3), dtype=torch.uint8, device=device) if return_u8 else None
        N.check(lib.sm_preprocess_resize_u8(pd.data_ptr(), dd.data_ptr(), cd.data_ptr(), _lut(device).data_ptr(), tmp.data_ptr(),
                                            tmp.stride(0), out.data_ptr(), u8.data_ptr() if return_u8 else None, B, S, max_h, st),
                "sm_preprocess_resize_u8")
        return (out, u8) if return_u8 else out
    if pad_to is not None:
        Hp, Wp = pad_to
        for im in images:
            h, w = (im.shape[:2] if hasattr(im, "shape") else im[:2])
            assert h <= Hp and w <= Wp, "pad_to must cover every image of the batch"
        out = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=device)
        N.check(lib.sm_preprocess_normalize_pad_u8(pd.data_ptr(), dd.data_ptr(), _lut(device).data_ptr(), out.data_ptr(), B, Hp, Wp, st),
                "sm_preprocess_normalize_pad_u8")
        return out
    out = torch.empty(out_elems, dtype=torch.float32, device=device)
    N.check(lib.sm_preprocess_normalize_u8(pd.data_ptr(), dd.data_ptr(), _lut(device).data_ptr(), out.data_ptr(), B, max_px, st),
            "sm_preprocess_normalize_u8")
    views, o = [], 0
    for im in images:
        h, w = (im.shape[:2] if hasattr(im, "shape") else im[:2])
        views.append(out[o:o + 3 * h * w].view(1, 3, h, w))
        o += 3 * h * w
    return views


from .decode_pool import decode_item  # noqa: E402  (sm_decode_worker.decode_item, loaded from the file's own path)


class PrefetchingLoader:
    """Batches of a SaliencyTestDataset decoded by ``workers`` host processes (or threads), ``depth`` batches ahead of the consumer.
    Iterating yields (list of rgb uint8 arrays, list of GT uint8 arrays, list of dataset indices).  Lifetime: without ``pack`` the
    arrays are the caller's own (copied out of the shared-memory slots: ``list(loader)`` is safe); with ``pack=True`` the yielded
    page-locked staging buffers belong to the pipeline and are recycled once the copy they feed has run - consume a batch before
    asking for the one ``depth`` further on.  If /dev/shm cannot hold the slots the loader falls back to the thread decoder."""

    def __init__(self, dataset, indices: Sequence[int], batch_size: int, workers: Optional[int] = None, depth: int = 3,
                 pack: bool = False, pack_size: Optional[int] = None, batches: Optional[Sequence[Sequence[int]]] = None,
                 decode: Optional[str] = None):
        """``pack``: one more host thread assembles every decoded batch into the page-locked staging buffers of the device
        pipeline (pack_images(..., pack_size, pinned=True) + pack_gts) while the consumer is still busy with the previous
        batch; iterating then yields ((packed images, image shapes), packed GTs, indices) for
        ``preprocess_on_device(shapes, S, device, packed=...)`` / ``ops.GtBatch.from_packed``."""
        self.ds, self.idx, self.bs, self.depth = dataset, list(indices), batch_size, max(1, depth)
        self.pack, self.pack_size = pack, pack_size
        # ``batches``: explicit lists of dataset indices (native-resolution buckets) instead of consecutive slices of ``indices``
        self.batches = [list(b) for b in batches] if batches is not None else None
        # ``decode``: "process" (default) = worker processes writing into shared memory (decode_pool.py): Pillow holds the GIL for
        # about half of a sample's host time, so threads stop scaling at 1 / that (595 images/s on one thread, 480 on eight);
        # "thread" = the round-2 thread pool (SM_DECODE=thread selects it globally).  ``workers`` defaults to this rank's share
        # of the node's cores (CPU affinity / LOCAL_WORLD_SIZE).
        self.decode = decode or os.environ.get("SM_DECODE", "process")
        assert self.decode in ("process", "thread"), self.decode
        if workers is None:
            from .decode_pool import default_workers
            workers = default_workers()
        self.workers = workers

    def __len__(self):
        return len(self.batches) if self.batches is not None else -(-len(self.idx) // self.bs)

    def __iter__(self):
        batches = self.batches if self.batches is not None else [self.idx[s:s + self.bs] for s in range(0, len(self.idx), self.bs)]
        if not batches:
            return
        slots = None
        if self.decode == "process":
            from .decode_pool import BatchSlots, shared_pool
            try:
                dpool, slots = shared_pool(self.workers), BatchSlots(self.depth + 1, max(len(b) for b in batches))
            except OSError as e:  # no /dev/shm, or not enough of it: the thread decoder needs none
                import warnings
                warnings.warn(f"PrefetchingLoader: shared-memory slots unavailable ({e}); decoding on threads instead")
        if slots is not None:
            try:
                def submit(k):  # -> callable returning (rgb views, GT views) inside shared slot k % (depth + 1)
                    return dpool.decode_batch(slots, k % (self.depth + 1), [(self.ds.p_imgs[i], self.ds.p_gts[i]) for i in batches[k]])
                yield from self._run(batches, submit)
            finally:
                slots.close(dpool)  # files unlinked, the workers unmap them (the processes stay for the next loader)
            return
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            def submit(k):
                futs = [pool.submit(decode_item, self.ds.p_imgs[i], self.ds.p_gts[i]) for i in batches[k]]

                def result():
                    items = [f.result() for f in futs]
                    return [it[0] for it in items], [it[1] for it in items]
                return result
            yield from self._run(batches, submit)

    def _run(self, batches, submit):
        """``submit(k)`` starts the decode of batch k and returns a callable that waits for it: (rgb arrays, GT arrays).  The
        arrays of batch k may be overwritten once batch k + depth + 1 is submitted (shared-memory slots)."""
        if not self.pack:
            inflight = [submit(k) for k in range(min(self.depth, len(batches)))]
            for k in range(len(batches)):
                wait = inflight.pop(0)
                if k + self.depth < len(batches):
                    inflight.append(submit(k + self.depth))
                rgbs, gts = wait()
                # the caller's own arrays: the views of a shared-memory slot are overwritten depth + 1 batches later
                yield [np.array(r) for r in rgbs], [None if g is None else np.array(g) for g in gts], batches[k]
            return

        def assemble(wait):  # runs on the packing thread: waits for the batch's decodes, copies into pinned staging
            rgbs, gts = wait()
            shapes = [(int(r.shape[0]), int(r.shape[1])) for r in rgbs]
            return (pack_images(rgbs, self.pack_size, pinned=True), shapes), (pack_gts(gts) if gts[0] is not None else None)

        # ONE packing thread: assembling a batch is ~30 MB of numpy copies (GIL released), 16k images/s on one thread - what 15
        # decode processes deliver.  Measured over 12 288 files, two alternations (profiles/r03_pack_threads.log): 1 thread
        # 14.5 / 14.0 k images/s end to end, 2: 14.0 / 12.8 k, 3: 12.2 / 14.2 k, 4: 11.8 / 11.2 k - every extra thread competes with
        # the decode processes for the rank's cores.  (The first, 0.3-s version of the leg had favoured two.)
        with ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("SM_PACK_THREADS", "1")))) as packer:
            def submit_packed(k):
                return packer.submit(assemble, submit(k))
            inflight = [submit_packed(k) for k in range(min(self.depth, len(batches)))]
            for k in range(len(batches)):
                fut = inflight.pop(0)
                if k + self.depth < len(batches):
                    inflight.append(submit_packed(k + self.depth))
                imgs, gts = fut.result()
                yield imgs, gts, batches[k]
