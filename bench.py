#!/usr/bin/env python3
"""Headline benchmark: images/sec of the SelfMask inference hot path (ViT-S/16, 224^2, nq=20, batch 64 per GPU).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one evaluator iteration - MaskFormer.forward + mask post-processing + the 14 metrics - over one batch of
synthetic images already resident in HBM (weights = synthetic checkpoint seed 0; there is no dataset or checkpoint
offline).  Consecutive steps are dealt onto --streams HIP streams (3 batches in flight, selfmask_amd/streams.py), as
the shipped Evaluator does.  Images shard across ranks with no data-path collective (weak scaling); the only exchange
is the evaluator's end-of-run all-gather of per-image result rows (RCCL), issued once after the K steps inside the
timed region.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      - the kernel holding the largest share of the forward, timed in situ: HIP events around each of its
                  launches on the launch stream inside real forwards (sm_forward_timing); `traffic` = HBM bytes per launch
                  from the committed PMC passes, quoted only if they were taken on exactly these kernel sources;
  sustained     - the same step loop run for --sustained-steps more steps (the K-step figure is a short burst);
  cpu_baseline  - the CPU oracle (torch-CPU restatement of the reference) on this box's host cores, rank 0, N=1 only:
                  the same work as a GPU step (forward + post-processing + 14 metrics) on all cores (`value`), the
                  forward alone on all cores and on one thread, and the reference's own operating point (batch 1);
  end_to_end    - the real Evaluator over a generated DUTS-layout JPEG tree (decode pool -> device resize / normalise ->
                  forward -> metrics): what a user sees, bounded by JPEG decoding on the host cores;
  serving       - SelfMaskInference.predict_tensors at batch 1 (app.py's call pattern): p50 / p99 latency.
--quick skips everything but the timed steps and the roofline (profiling runs).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "salient-object-detection_amd"), REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD, 256 CUs @ 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16); the split GEMM issues 3 MFMA flops per
                               # algorithmic flop, so its fp32-equivalent roof is 833 TFLOP/s
HBM_PEAK_GBS = 8000.0


def forward_flops_per_image(P, S, L=6, nq=20):
    """SURVEY.md 8d formulae (algorithmic FLOPs of one MaskFormer.forward)."""
    g = S // P
    n, N = g * g, g * g + 1
    enc = 2 * n * 3 * P * P * 384 + 12 * (2 * N * 384 * 1152 + 4 * N * N * 384 + 2 * N * 384 * 384 + 4 * N * 384 * 1536)
    dec = L * (12 * nq * 384 * 384 + 4 * nq * nq * 384 + 4 * n * 384 * 384 + 4 * nq * n * 384 + 4 * nq * 384 * 1536)
    head = L * 2 * nq * 384 * 4 * n + L * nq * (4 * 384 * 384 + 2 * 384)
    return enc + dec + head


def time_forward_kernels(model, x, forwards=3):
    """Per-kernel timing of the real forward: the library brackets each GEMM / attention / LayerNorm launch with two
    HIP events on the launch stream (sm_forward_timing, include/selfmask_hip.h) while `forwards` forwards of the bench
    batch run on one stream right after the timed steps.  Returns {kernel name: dict}, largest time first."""
    from selfmask_amd import _native as Nn
    lib = Nn.load()
    torch.cuda.synchronize()
    Nn.check(lib.sm_forward_timing(1), "sm_forward_timing")
    for _ in range(forwards):
        model(x)
    torch.cuda.synchronize()
    buf = (Nn.KernelTime * 32)()
    n = lib.sm_forward_timing_read(buf, 32)
    lib.sm_forward_timing(0)
    if n < 0:
        raise RuntimeError(lib.sm_last_error().decode())
    out = {}
    for e in buf[:n]:
        t = e.total_us * 1e-6
        out[e.name.decode()] = {
            "launches_per_forward": e.launches / forwards, "avg_launch_us": e.total_us / e.launches,
            "total_ms_per_forward": e.total_us / forwards * 1e-3, "event_overhead_us": e.overhead_us,
            "flops_per_launch": e.flops / e.launches, "bytes_per_launch": e.bytes / e.launches,
            "achieved_tflops": e.flops / t / 1e12, "achieved_gbs": e.bytes / t / 1e9}
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms_per_forward"]))


def time_taps(fn, reps=3):
    """The same event taps around the launch sequences of the clusterer / the bilateral solver (per phase; one-launch phases carry
    their kernel's name): {name: {"ms_per_call", "launch_pairs_per_call", "algorithmic_gbs" (when the phase states its bytes)}}."""
    from selfmask_amd import _native as Nn
    lib = Nn.load()
    fn()
    torch.cuda.synchronize()
    Nn.check(lib.sm_forward_timing(1), "sm_forward_timing")
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    buf = (Nn.KernelTime * 32)()
    n = lib.sm_forward_timing_read(buf, 32)
    lib.sm_forward_timing(0)
    if n < 0:
        raise RuntimeError(lib.sm_last_error().decode())
    out = {}
    for e in buf[:n]:
        d = {"ms_per_call": round(e.total_us / reps * 1e-3, 4), "event_pairs_per_call": e.launches / reps}
        if e.bytes > 0 and e.total_us > 0:
            d["algorithmic_bytes_per_call"] = round(e.bytes / reps)
            d["algorithmic_gbs"] = round(e.bytes / (e.total_us * 1e-6) / 1e9, 2)
        if e.flops > 0 and e.total_us > 0:
            d["achieved_tflops"] = round(e.flops / (e.total_us * 1e-6) / 1e12, 2)
        out[e.name.decode()] = d
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_call"]))


def source_hash() -> str:
    """Hash of the kernel sources: PMC measurements are only quoted for exactly the code they were taken on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(REPO, "salient-object-detection_amd", "csrc", "*"))) + \
        [os.path.join(REPO, "include", "selfmask_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(cores, int(os.environ.get("SM_CPU_BASELINE_THREADS", "16")))  # a 1-GPU box's CPU share is 16 cores


def cpu_baseline(P, S, budget_s=7.0, legs=("step", "forward", "one_thread", "batch1"), Bc=16):
    """The CPU oracle (torch-CPU restatement, fp32) on this host: bounded samples of the same workload."""
    import numpy as np
    from oracle import selfmask_oracle as O  # measured as the BASELINE only, never on the product path
    from oracle import evaluator_oracle as E
    from selfmask_amd import synthetic_state_dict, synthetic_images
    cores = _cores()
    sd = synthetic_state_dict(0, "soft", patch_size=P)
    x = torch.from_numpy(synthetic_images(1234, (Bc, 3, S, S)))
    rng = np.random.Generator(np.random.PCG64(99))
    gts = []
    for _ in range(Bc):
        h, w = (int(v) for v in rng.integers(300, 401, size=2))
        yy, xx = np.mgrid[:h, :w]
        gts.append(torch.from_numpy((((yy - h * .5) / (h * .2)) ** 2 + ((xx - w * .5) / (w * .25)) ** 2 <= 1).astype(np.int64)))

    def timed(fn, per_call, threads):
        torch.set_num_threads(threads)
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            fn()
            n += per_call
        return n, time.perf_counter() - t0

    def step_all():  # what one GPU step does: forward, post-processing to each GT's size, both selections, 14 metrics
        out = O.forward(x, sd, P)
        for b in range(Bc):
            pm, q, ub, _ = E.postprocess(out["mask_pred"][b, -1], out["objectness"][b, -1, :, 0], gts[b], scale_factor=None)
            E.all_metrics(pm[q], gts[b]); E.all_metrics(pm[ub], gts[b])

    n_all, t_all = timed(step_all, Bc, cores)
    res = {"value": round(n_all / t_all, 2), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": f"{n_all} images (batches of {Bc}, ViT-S/{P} {S}x{S}: fp32 torch-CPU oracle forward + post-processing + "
                     f"14 metrics per image) in {t_all:.1f}s"}
    if "forward" in legs:
        n_fwd, t_fwd = timed(lambda: O.forward(x, sd, P), Bc, cores)
        res["forward_only_all_cores"] = {"value": round(n_fwd / t_fwd, 2), "cores": cores, "sample": f"{n_fwd} images in {t_fwd:.1f}s"}
    if "one_thread" in legs:
        n_one, t_one = timed(lambda: O.forward(x[:4], sd, P), 4, 1)
        res["forward_only_one_thread"] = {"value": round(n_one / t_one, 2), "cores": 1, "sample": f"{n_one} images (batches of 4) in {t_one:.1f}s"}
    if "batch1" in legs:
        n_b1, t_b1 = timed(lambda: O.forward(x[:1], sd, P), 1, cores)
        res["forward_only_batch1_all_cores"] = {"value": round(n_b1 / t_b1, 2), "cores": cores,
                                                "sample": f"{n_b1} images at batch 1 (the reference evaluator's operating point) in {t_b1:.1f}s"}
    torch.set_num_threads(cores)
    return res


def end_to_end(model, dev, P, S, B, streams, n_images=768, world=1, rank=0):
    """The real Evaluator over a generated DUTS-layout tree of JPEG / PNG files (300-400 px): decode on the
    host worker processes, resize + normalise + forward + metrics on the device.  With several ranks the tree holds
    n_images x world files, the Evaluator shards them (one RCCL all-gather of the result rows) and every rank decodes its
    share on its own block of host cores."""
    import shutil
    import tempfile
    from selfmask_amd import datasets as DS
    from selfmask_amd.decode_pool import default_workers
    from selfmask_amd.evaluator import Evaluator
    # `distinct` generated files, listed `repeat` times through symbolic links (writing JPEGs costs 2.5 ms each).  16 x 768 = 12 288
    # images: a second of work at 12 k images/s - the 3 072 of the first version lasted 0.3 s, of which the fill and the drain
    # of the decode pipeline were a tenth, and the figure moved by 30 % between runs.  The native-resolution legs (1.1 k images/s
    # at batch 1) keep 3 072 images in a second tree.
    distinct, repeat, repeat_native = n_images, 16, 4
    n_native = n_images * repeat_native
    n_images *= world * repeat
    box = [tempfile.mkdtemp(prefix="sm_bench_ds_") if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    root = box[0]
    root_native = os.path.join(root, "native_tree")
    try:
        if rank == 0:
            DS.write_synthetic_dataset(root, "duts", distinct * world, seed=7)
            sub, di, _, dg, _ = DS.LAYOUTS["duts"]
            for i in range(distinct * world, n_images):
                for d_, ext in ((di, "jpg"), (dg, "png")):
                    os.symlink(os.path.join(root, sub, d_, f"{i % (distinct * world):05d}.{ext}"), os.path.join(root, sub, d_, f"{i:05d}.{ext}"))
            if world == 1:
                for d_ in (di, dg):
                    os.makedirs(os.path.join(root_native, sub, d_))
                for i in range(n_native):
                    for d_, ext in ((di, "jpg"), (dg, "png")):
                        os.symlink(os.path.join(root, sub, d_, f"{i % distinct:05d}.{ext}"), os.path.join(root_native, sub, d_, f"{i:05d}.{ext}"))
        if world > 1:
            dist.barrier()
        ev = Evaluator(network=model, dir_dataset=root)
        ev.device = dev
        ev("duts", dir_ckpt=os.path.join(root, "ckpt"), img_size=S, batch_size=B, device=dev, streams=streams)  # warm: page cache, graphs
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        res = ev("duts", dir_ckpt=os.path.join(root, "ckpt"), img_size=S, batch_size=B, device=dev, streams=streams)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the host part alone: decode of this rank's files on the same pool, nothing else
        from selfmask_amd.distributed import shard_indices
        from selfmask_amd.pipeline import PrefetchingLoader
        ds = DS.get_dataset(root, "duts", eval_img_size=S)
        mine = shard_indices(len(ds), rank, world)
        t1 = time.perf_counter()
        for _ in PrefetchingLoader(ds, mine, B, depth=streams + 1):
            pass
        dt_dec = time.perf_counter() - t1
        t1 = time.perf_counter()
        for _ in PrefetchingLoader(ds, mine[:256], B, depth=streams + 1, decode="thread"):
            pass
        dt_thr = time.perf_counter() - t1
        if world > 1:
            dt, dt_dec = max_over_ranks([dt, dt_dec], dev)
        out = {"end_to_end_images_per_sec": round(n_images / dt, 1), "images": n_images, "distinct_files": distinct * world, "ranks": world,
               "decode_workers_per_rank": default_workers(), "decode": "worker processes + shared memory (decode_pool.py)",
               "host_decode_only_images_per_sec": round(n_images / dt_dec, 1),
               "host_decode_only_thread_pool_images_per_sec_per_rank": round(len(mine[:256]) / dt_thr, 1), "iou": res["iou"],
               "what": "Evaluator('duts', img_size=%d, batch_size=%d): JPEG/PNG decode on host worker processes, Pillow-exact resize + "
                       "normalise + forward + metrics on the device" % (S, B)}
        if world > 1:
            return out
        # The reference's OWN operating point (evaluator.pyc@L373): native resolution.  Batch 1 as the reference runs it, and
        # the same rows from token-grid buckets (images that pad to the same patch grid share a batch; bit-identical rows)
        import numpy as np

        ev_n = Evaluator(network=model, dir_dataset=root_native)
        ev_n.device = dev

        def native(bs):
            ev_n("duts", dir_ckpt=os.path.join(root, "ckpt_n"), batch_size=bs, device=dev, streams=streams)  # warm: graphs, page cache
            torch.cuda.synchronize()
            t = time.perf_counter()
            ev_n("duts", dir_ckpt=os.path.join(root, "ckpt_n"), batch_size=bs, device=dev, streams=streams)
            torch.cuda.synchronize()
            return time.perf_counter() - t, ev_n.last_rows.copy(), dict(ev_n.graph_stats)

        t1, rows1, _ = native(1)
        t16, rows16, g16 = native(16)
        out["native_resolution"] = {
            "images": n_native, "sizes": "300-400 px per side, uniformly random (49 token grids at patch %d)" % P,
            "batch1_images_per_sec": round(n_native / t1, 1), "bucketed_batch16_images_per_sec": round(n_native / t16, 1),
            "rows_bit_identical": bool(np.array_equal(rows1, rows16)), "hip_graph_bucketed": g16,
            "what": "Evaluator('duts', img_size=None): batch_size=1 (the reference's mode: eager launches, one image per forward) vs "
                    "batch_size=16 (token-grid buckets, zero-padded like make_input_divisible, graph replay per bucket shape); both "
                    "include the host JPEG/PNG decode; the CPU oracle at batch 1 is cpu_baseline.forward_only_batch1_all_cores"}
        return out
    finally:
        if world > 1:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)


def serving_latency(model, dev, n=200):
    """SelfMaskInference.predict_tensors (app.py:241-284) at batch 1 on a 300x400 image: decoded array in -> (index, scores,
    mask) out, per-request wall time including the final device->host copies."""
    import io
    import numpy as np
    from argparse import Namespace
    from PIL import Image
    from selfmask_amd.inference import SelfMaskInference
    inf = SelfMaskInference(None, Namespace(), device=dev, model=model)
    rng = np.random.Generator(np.random.PCG64(5))
    rgb = rng.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", quality=92)
    jpeg = buf.getvalue()
    for _ in range(10):
        inf.predict_tensors(rgb)
    lat, lat_jpeg = [], []
    for _ in range(n):
        t0 = time.perf_counter(); inf.predict_tensors(rgb); lat.append(time.perf_counter() - t0)
    for _ in range(n // 4):
        t0 = time.perf_counter(); inf.predict_tensors(io.BytesIO(jpeg)); lat_jpeg.append(time.perf_counter() - t0)
    q = lambda v, p: round(float(np.percentile(np.array(v) * 1e3, p)), 3)
    g = inf.base_structure._graphed
    return {"batch": 1, "requests": n, "p50_ms": q(lat, 50), "p99_ms": q(lat, 99), "with_jpeg_decode_p50_ms": q(lat_jpeg, 50),
            "with_jpeg_decode_p99_ms": q(lat_jpeg, 99), "hip_graph_replays": g.replays, "hip_graph_failed": g.failed,
            "what": "SelfMaskInference.predict_tensors: 300x400 RGB -> resize 224 + normalise (HIP) -> graph-replayed forward -> "
                    "arg-max objectness + mask (HIP) -> D2H"}


class Workload:
    """One configuration of the step loop: model + a resident batch + synthetic ground truth + graphs + stream ring."""

    def __init__(self, dev, P, S, B, rank=0, gemm_mode=None, streams=3, graph=True, zero_data=False, host_input=False,
                 forward_only=False):
        import numpy as np
        from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images, ops
        from selfmask_amd.graphs import GraphedForward
        from selfmask_amd.streams import StreamRing
        self.dev, self.P, self.S, self.B, self.ops, self.forward_only = dev, P, S, B, ops, forward_only
        model = MaskFormer(n_queries=20, patch_size=P, n_decoder_layers=6, return_intermediate=True,
                           use_binary_classifier=True, gemm_mode=gemm_mode)
        model.load_state_dict(synthetic_state_dict(0, "soft", patch_size=P), strict=True)
        self.model = model.to(dev)
        # every rank owns a different shard of the (synthetic) image list: rank-strided seeds
        self.x = torch.from_numpy(synthetic_images(1234 + rank, (B, 3, S, S))).to(dev)
        if zero_data:
            with torch.no_grad():
                for p_ in self.model.parameters():
                    p_.zero_()
            self.model.refresh_packed()
            self.x.zero_()
        # synthetic ground truth at DUTS-like native sizes (300-400 px ellipses), packed once and resident in HBM
        rng = np.random.Generator(np.random.PCG64(99 + rank))
        gts = []
        for _ in range(B):
            h, w = (int(v) for v in rng.integers(300, 401, size=2))
            yy, xx = np.mgrid[:h, :w]
            gts.append(torch.from_numpy(((((yy - h * rng.uniform(.3, .7)) / (h * rng.uniform(.1, .3))) ** 2 +
                                          ((xx - w * rng.uniform(.3, .7)) / (w * rng.uniform(.1, .3))) ** 2) <= 1)
                                        .astype(np.uint8)))
        self.gt_batch = ops.GtBatch(gts, dev)
        self.x_host = self.x.cpu().pin_memory() if host_input else None
        self.fwd = GraphedForward(self.model, enabled=graph)
        self.ring = StreamRing(dev, max(1, streams))

    def step(self):
        # one evaluator iteration over a batch (evaluator.pyc@L193-228, batched mode): forward, last decoder
        # layer, up-sample to each GT's size, upper-bound + arg-max-objectness query, 7 metrics x 2 -> 16 floats/image
        out = self.fwd(self.x_host.to(self.dev, non_blocking=True) if self.x_host is not None else self.x)
        if self.forward_only:
            return out["objectness"][:, -1, :16, 0]
        return self.ops.evaluate_masks(out["mask_pred"][:, -1], out["objectness"][:, -1, :, 0], self.gt_batch, scale=0.0)

    def run_steps(self, n, dst=None):
        self.ring.fork()
        for k in range(n):
            with self.ring.next():
                r = self.step()
                if dst is not None:
                    dst[k * self.B:(k + 1) * self.B] = r
        self.ring.join()

    def prime(self, warmup):
        # untimed, not part of W: every stream sees the batch shape often enough to be admitted to the graph cache, so its
        # hipGraph is captured here and never inside the timed region, whatever --warmup is
        self.run_steps((self.fwd.policy.admit_after + 1) * len(self.ring.streams))
        self.run_steps(warmup)

    def timed(self, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run_steps(steps)
        torch.cuda.synchronize()
        return time.perf_counter() - t0


def roofline_of(kern, gemm_mode, name=None):
    """roofline dict of one tapped kernel (default: the one holding the largest share of the forward)."""
    peak = F32_MFMA_PEAK_TFLOPS if gemm_mode == "fp32" else F16_MFMA_PEAK_TFLOPS
    issue = 3.0 if gemm_mode in ("w16", "f16x2") else 1.0  # MFMA FLOPs issued per algorithmic FLOP (hi*hi, hi*lo, lo*hi)
    if name is None:
        name = next(iter(kern))
    v = kern[name]
    # An event pair around a launch also measures the command processor's event handling (event_overhead_us, what an EMPTY
    # pair reads); it is subtracted, and a launch shorter than three times that overhead is below what the taps resolve:
    # no rate is quoted for it (rocprofv3's kernel trace under profiles/ is the source for such kernels)
    resolved = v["avg_launch_us"] >= 3.0 * v["event_overhead_us"]
    if "layernorm" in name:
        r = {"bound": "hbm", "achieved": round(v["achieved_gbs"], 1) if resolved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(v["achieved_gbs"] / HBM_PEAK_GBS, 4) if resolved else None}
    else:
        r = {"bound": "mfma", "achieved": round(v["achieved_tflops"], 2) if resolved else None, "peak": peak, "unit": "TFLOP/s",
             "frac": round(v["achieved_tflops"] / peak, 4) if resolved else None,
             "mfma_issued_tflops": round(v["achieved_tflops"] * issue, 2) if resolved else None}
        if "attention" in name and resolved:  # 49 FLOP/B at N=197, d=64: below the machine balance (312), so also quote bytes
            r.update(hbm_gbs=round(v["achieved_gbs"], 1), hbm_frac=round(v["achieved_gbs"] / HBM_PEAK_GBS, 4))
    if not resolved:
        r["note"] = "launch shorter than 3x the event-pair overhead: below the taps' resolution, no rate quoted"
    r.update(kernel=name, avg_launch_us=round(v["avg_launch_us"], 2), launches_per_forward=v["launches_per_forward"],
             flops_per_launch=v["flops_per_launch"], event_overhead_us=round(v["event_overhead_us"], 2),
             share_of_forward=round(v["total_ms_per_forward"] / sum(u["total_ms_per_forward"] for u in kern.values()), 3))
    return r


def shape_leg(dev, P, S, B, streams, steps=20, warmup=5, cpu=True):
    """The same step loop at another shape of BASELINE.json / the shipped YAML (ViT-S/8 224^2: the shipped checkpoint's
    patch size, N = 785; 384^2: configs[2], N = 577): images/s, the dominant kernel's roofline, the CPU oracle beside it."""
    w = Workload(dev, P, S, B, streams=streams)
    w.prime(warmup)
    dt = w.timed(steps)
    kern = time_forward_kernels(w.model, w.x)
    value = steps * B / dt
    flops_img = forward_flops_per_image(P, S)
    res = {"workload": f"ViT-S/{P} {S}x{S}, nq=20, batch={B}, forward + evaluator post-processing and metrics",
           "tokens": (S // P) ** 2 + 1, "value": round(value, 1), "unit": "images/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
           "model_tflops": round(value * flops_img / 1e12, 2), "roofline": roofline_of(kern, w.model.gemm_mode),
           "roofline_other_kernels": {k: roofline_of(kern, w.model.gemm_mode, k) for k in list(kern)[1:]}}
    if cpu:
        res["cpu_baseline"] = cpu_baseline(P, S, budget_s=4.0, legs=("step",), Bc=8)
    del w
    torch.cuda.empty_cache()
    return res


def _event_ms(fn, reps, dev):
    """average device time of fn() over `reps` calls on the current stream (HIP events on that stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def refine_leg(dev, streams, P=16, S=384, B=32, steps=12, warmup=3, cpu=True):
    """BASELINE.json configs[2]: forward at 384^2 -> picked mask -> bilateral-solver refinement (bilateral_solver.py:152-193) ->
    metrics, as Evaluator(img_size=384, refine="bilateral") runs it per batch; the same loop without the refinement beside it; the
    batched solver alone with its algorithmic bytes / time; the CPU oracle solver on one image."""
    import numpy as np
    from selfmask_amd import ops
    from selfmask_amd.bilateral_solver import bilateral_solver_batch_device
    from selfmask_amd.datasets import MEAN, STD, synthetic_scene
    w = Workload(dev, P, S, B, streams=streams)
    rng = np.random.Generator(np.random.PCG64(77))
    scenes = [synthetic_scene(rng, S, S) for _ in range(B)]
    u8 = torch.from_numpy(np.stack([im for im, _ in scenes])).to(dev)  # (B, S, S, 3): what the solver refines against
    mean, std = torch.tensor(MEAN, device=dev), torch.tensor(STD, device=dev)
    w.x = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
    w.gt_batch = ops.GtBatch([torch.from_numpy(g.astype(np.uint8)) for _, g in scenes], dev)
    state = {}

    def step_refined():
        out = w.fwd(w.x)
        mp, ob = out["mask_pred"][:, -1], out["objectness"][:, -1, :, 0]
        rows = ops.evaluate_masks(mp, ob, w.gt_batch, scale=0.0)
        target = ops.upsample_selected(mp, rows, (S, S), "pick")
        _, binary, info = bilateral_solver_batch_device(u8, target, return_info=True)
        refined = ops.mask_u8_to_f32(binary).unsqueeze(1)
        state.update(target=target, info=info)
        return ops.evaluate_masks(refined, rows[:, 0:1], w.gt_batch, scale=0.0)

    plain = w.step
    w.prime(warmup)
    dt_plain = w.timed(steps)
    w.step = step_refined
    w.run_steps(2 * len(w.ring.streams))
    dt_ref = w.timed(steps)
    w.step = plain
    target, info = state["target"], state["info"].cpu().numpy()
    ms_solver = _event_ms(lambda: bilateral_solver_batch_device(u8, target), 5, dev)
    phases = time_taps(lambda: bilateral_solver_batch_device(u8, target))
    V, iters = info[:, 0].astype(np.float64), info[:, 1].astype(np.float64)
    npx = float(S * S)
    nnz = 6.0 * V  # nnz ~ V + sum of the blur matrices' entries ~ 6 V
    alg = npx * (3 + 5 * 4 + 8 + 8) + 4 * npx * 16 + 4 * npx * (8 + 4) + 11 * (nnz * 12 + 3 * V * 8) + iters * (nnz * 12 + 10 * V * 8)
    res = {"workload": f"ViT-S/{P} {S}x{S}, nq=20, batch={B}: forward + metrics + bilateral-solver refinement of the picked mask + metrics of "
                       f"the refined mask (Evaluator(img_size={S}, refine='bilateral') per batch; synthetic ellipse scenes)",
           "value": round(steps * B / dt_ref, 1), "unit": "images/sec", "steps": steps, "ms_per_step": round(dt_ref / steps * 1e3, 3),
           "without_refinement_images_per_sec": round(steps * B / dt_plain, 1),
           "refined_over_plain": round(dt_plain / dt_ref, 3),
           "solver_alone": {"batch": B, "ms_per_batch": round(ms_solver, 3), "ms_per_image": round(ms_solver / B, 4),
                            "images_per_sec": round(B / ms_solver * 1e3, 1), "vertices_mean": round(float(V.mean()), 1),
                            "pcg_iterations_mean": round(float(iters.mean()), 2),
                            "algorithmic_bytes_per_image": round(float(alg.mean())),
                            "phases_ms_per_batch": phases,
                            "roofline": {"bound": "hbm", "achieved": round(float(alg.sum()) / (ms_solver * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(float(alg.sum()) / (ms_solver * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                         "note": "latency / irregularity-bound by design (SURVEY.md 8d): a few thousand lattice vertices per "
                                                 "image, 11 blurs + <= 25 PCG iterations of dependent sparse mat-vecs"}}}
    if cpu:
        from oracle import bilateral_oracle as BO  # the BASELINE beside it, never on the product path
        img0, t0 = u8[0].cpu().numpy(), target[0].cpu().numpy()
        BO.bilateral_solver_output(img0, t0)
        n, ts = 0, time.perf_counter()
        while time.perf_counter() - ts < 4.0:
            BO.bilateral_solver_output(img0, t0)
            n += 1
        dtc = time.perf_counter() - ts
        res["cpu_baseline"] = {"value": round(n / dtc, 2), "unit": "images/sec", "cores": 1, "kind": "port",
                               "sample": f"{n} solves of one {S}x{S} image by oracle/bilateral_oracle.py (numpy / scipy restatement of "
                                         f"bilateral_solver_output, single thread as the reference) in {dtc:.1f}s"}
    del w
    torch.cuda.empty_cache()
    return res


def pseudo_masks_leg(dev, streams, P=16, S=224, B=128, steps=12, warmup=3, cpu=True):
    """BASELINE.json configs[4], DINO branch: encoder -> bilinear x2 -> spectral clustering for k = 2, 3, 4 (ONE eigen-solve per
    image) -> 9 candidate masks -> vote (mask_generator.pyc@L136-230), as MaskGenerator.__call__ runs it: `streams` batches in flight,
    each batch's votes read back (one copy) `streams` batches later.  images/s of the whole chain, the same on one stream, the
    clusterer alone, and the CPU restatement (oracle/cluster_oracle.py: dense eigh) beside it.  The clusterer is parity UNPINNED
    (absent from the reference)."""
    import numpy as np
    from collections import deque
    from selfmask_amd import voting as VT
    w = Workload(dev, P, S, B, streams=streams, forward_only=True, graph=False)
    model, x = w.model, w.x

    def run(n_steps, ring, xb):
        pending = deque()
        ring.fork()
        for _ in range(n_steps):
            with ring.next():
                pending.append(VT.vote_mask_batch_async(VT.extract_candidate_masks(model, xb), winners=True))  # (B, 9, H, W) -> votes
            if len(pending) >= len(ring.streams):
                pending.popleft().winners_host()
        while pending:
            pending.popleft().winners_host()
        ring.join()

    def timed(ring, xb=None, n_steps=steps):
        xb = x if xb is None else xb
        run(warmup + len(ring.streams), ring, xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n_steps, ring, xb)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    from selfmask_amd.streams import StreamRing
    dt1 = timed(StreamRing(dev, 1))
    dt = timed(w.ring) if len(w.ring.streams) > 1 else dt1
    tok = model(x, encoder_only=True)["patch_tokens"]
    gh, gw = tok.shape[1:3]
    feats = VT.upsample_tokens_aligned(tok.reshape(B, gh * gw, 384), gh, gw, 2).reshape(B, 4 * gh * gw, 384)
    ms_enc = _event_ms(lambda: model(x, encoder_only=True), 10, dev)
    ms_spec = _event_ms(lambda: VT.spectral_cluster(feats, (2, 3, 4)), 10, dev)
    ms_km = _event_ms(lambda: [VT.kmeans(feats, k) for k in (2, 3, 4)], 5, dev)
    _, det = VT.spectral_cluster(feats, (2, 3, 4), return_details=True)
    info = det["info"].cpu().numpy()
    n = 4 * gh * gw
    phases = time_taps(lambda: VT.spectral_cluster(feats, (2, 3, 4)))
    dom_name = next(iter(phases))
    dom = dict(phases[dom_name])
    if dom_name.startswith("spectral_embed_kernel"):
        # one launch = B workgroups (one per image, one per CU), each running its image's whole eigen-solve out of its CU's LDS.  LDS bytes
        # of ONE block mat-vec (8 columns): every list entry (2 m per row: m out- + m in-neighbours, a mutual pair twice) gathers a row's
        # 8 doubles and reads its 2-byte offset; the row's own X and Y are read and X written
        mv, m = float(info[:, 1].sum()), 9
        per_mv = n * (2.0 * m * 64 + 2.0 * m * 2 + 3 * 64)
        lds_peak = min(B, 256) * 128 * 2.4  # GB/s: 128 B / clk per CU at 2.4 GHz, on the CUs the launch occupies
        gbs = mv * per_mv / (dom["ms_per_call"] * 1e-3) / 1e9
        dom.update({"kernel": dom_name, "bound": "lds: one workgroup (8 waves) per image on its own CU, dependent sparse mat-vecs gathered from the LDS "
                                                 "(fp64, 16-B random reads) - not an HBM or MFMA roofline kernel",
                    "block_matvecs_per_launch": mv, "lds_bytes_per_block_matvec": per_mv, "achieved_lds_gbs": round(gbs, 1),
                    "lds_peak_gbs_on_the_cus_used": round(lds_peak, 1), "lds_frac": round(gbs / lds_peak, 3),
                    "us_per_block_matvec_per_image": round(dom["ms_per_call"] * 1e3 / (mv / B), 3)})
    res = {"workload": f"ViT-S/{P} {S}x{S}, batch={B}: encoder -> bilinear x2 ({n} points x 384) -> spectral clustering k=2,3,4 (10-NN graph, "
                       f"normalised Laplacian, 4 eigenvectors, k-means) -> 9 candidates -> vote",
           "value": round(steps * B / dt, 1), "unit": "images/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
           "batches_in_flight": len(w.ring.streams), "one_stream_images_per_sec": round(steps * B / dt1, 1),
           "encoder_ms_per_batch": round(ms_enc, 3), "spectral_cluster_ms_per_batch": round(ms_spec, 3),
           "spectral_cluster_images_per_sec": round(B / ms_spec * 1e3, 1),
           "kmeans_option_ms_per_batch": round(ms_km, 3),
           "eigensolver": {"outer_iterations_mean": round(float(info[:, 0].mean()), 2), "block_matvecs_mean": round(float(info[:, 1].mean()), 1),
                           "converged": int(info[:, 2].sum()), "of": B, "max_residual": float(det["residuals"].max())},
           "clusterer_phases_ms_per_batch": phases, "dominant_kernel": dom,
           "parity": "UNPINNED: the reference's `clusterings` module is absent in every form; scikit-learn is the witness (tests)"}
    # the generator's own operating point: images at their NATIVE size (mask_generator.pyc@L136-200) - DUTS-TR is mostly 400 x 300:
    # 19 x 25 patches -> 38 x 50 = 1900 points per image (the eigen-solver keeps 2 columns of its blocks in the LDS there)
    Hn, Wn, Bn = 300, 400, 128
    from selfmask_amd import synthetic_images
    xn = torch.from_numpy(synthetic_images(4321, (Bn, 3, Hn, Wn))).to(dev)
    sn = max(4, steps // 3)
    dtn = timed(w.ring, xn, sn)
    tokn = model(xn, encoder_only=True)["patch_tokens"]
    ghn, gwn = tokn.shape[1:3]
    featn = VT.upsample_tokens_aligned(tokn.reshape(Bn, ghn * gwn, 384), ghn, gwn, 2).reshape(Bn, 4 * ghn * gwn, 384)
    _, detn = VT.spectral_cluster(featn, (2, 3, 4), return_details=True)
    infon = detn["info"].cpu().numpy()
    res["native_300x400"] = {"workload": f"the same chain on {Hn}x{Wn} images ({4 * ghn * gwn} points each), batch {Bn}, {len(w.ring.streams)} in flight",
                             "value": round(sn * Bn / dtn, 1), "unit": "images/sec", "steps": sn, "ms_per_step": round(dtn / sn * 1e3, 3),
                             "encoder_ms_per_batch": round(_event_ms(lambda: model(xn, encoder_only=True), 5, dev), 3),
                             "spectral_cluster_ms_per_batch": round(_event_ms(lambda: VT.spectral_cluster(featn, (2, 3, 4)), 5, dev), 3),
                             "clusterer_phases_ms_per_batch": time_taps(lambda: VT.spectral_cluster(featn, (2, 3, 4))),
                             "eigensolver": {"block_matvecs_mean": round(float(infon[:, 1].mean()), 1), "converged": int(infon[:, 2].sum()), "of": Bn}}
    del xn, tokn, featn, detn
    # and from FILES, as MaskGenerator(...)(p_images) is called: JPEGs of 300-400 px in both directions (every size different: 49 patch
    # grids share the batches), decoded by the worker processes, run-length codes out
    import shutil
    import tempfile
    from selfmask_amd import datasets as DS
    from selfmask_amd.decode_pool import default_workers
    from selfmask_amd.mask_generator import MaskGenerator
    root = tempfile.mkdtemp(prefix="sm_bench_pm_")
    try:
        distinct, repeat = 256, 24
        DS.write_synthetic_dataset(root, "duts", distinct, seed=11)
        sub, di = DS.LAYOUTS["duts"][:2]
        for i in range(distinct, distinct * repeat):
            os.symlink(os.path.join(root, sub, di, f"{i % distinct:05d}.jpg"), os.path.join(root, sub, di, f"{i:05d}.jpg"))
        files = [os.path.join(root, sub, di, f"{i:05d}.jpg") for i in range(distinct * repeat)]
        gen = MaskGenerator(network=model, device=dev, streams=len(w.ring.streams))
        gen(files[:distinct])  # warm: page cache, workspaces
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        codes = gen(files)
        dtf = time.perf_counter() - t0
        res["from_files"] = {"workload": f"MaskGenerator(...)(p_images): {len(files)} JPEG files of 300-400 px ({distinct} distinct, every size "
                                         f"different) -> decode (worker processes) -> normalise + pad to the patch grid (device) -> chain -> "
                                         f"run-length codes (device run boundaries); batches of <= {gen.batch_size} per patch grid, "
                                         f"{gen.streams} in flight",
                             "value": round(len(files) / dtf, 1), "unit": "images/sec", "files": len(codes), "decode_workers": default_workers()}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    if cpu:
        from oracle import cluster_oracle as CO
        f0 = feats[0].cpu().numpy()
        CO.spectral_cluster(f0, (2, 3, 4), 10)
        c, ts = 0, time.perf_counter()
        while time.perf_counter() - ts < 4.0:
            CO.spectral_cluster(f0, (2, 3, 4), 10)
            c += 1
        dtc = time.perf_counter() - ts
        res["cpu_baseline"] = {"value": round(c / dtc, 2), "unit": "images/sec", "cores": _cores(), "kind": "port",
                               "sample": f"{c} runs of oracle/cluster_oracle.spectral_cluster on one image's {n} x 384 features (numpy k-NN + "
                                         f"scipy dense eigh + numpy k-means; clustering only, no encoder) in {dtc:.1f}s"}
    del w
    torch.cuda.empty_cache()
    return res


def throughput_mode_leg(dev, ref, steps=30, warmup=6):
    """SURVEY.md 7.2 (b) diagnostic - NOT the metric: the same pipeline with ONE f16 MFMA per product in the weight GEMMs and the
    encoder attention (gemm_mode "f16": plain f16 operands, fp32 accumulate, fp32 LayerNorm / softmax statistics, fp32-grade
    mask einsum).  Reported: images/s, the dominant kernel's fraction of the f16 roof (what the kernel STRUCTURE reaches with
    the x3 removed), and how far the results move from the fp32-grade path on the bench batch."""
    w = Workload(dev, ref.P, ref.S, ref.B, gemm_mode="f16", streams=len(ref.ring.streams))
    w.prime(warmup)
    dt = w.timed(steps)
    kern = time_forward_kernels(w.model, w.x)
    a = ref.model(ref.x, return_logits=True)
    b = w.model(w.x, return_logits=True)
    la, lb = a["mask_logits"][:, -1], b["mask_logits"][:, -1]
    rows_a = ref.ops.evaluate_masks(a["mask_pred"][:, -1], a["objectness"][:, -1, :, 0], ref.gt_batch, scale=0.0)
    rows_b = w.ops.evaluate_masks(b["mask_pred"][:, -1], b["objectness"][:, -1, :, 0], w.gt_batch, scale=0.0)
    torch.cuda.synchronize()
    iou_a, iou_b = rows_a[:, 0].double().mean().item(), rows_b[:, 0].double().mean().item()  # arg-max-objectness query's IoU
    res = {"what": "diagnostic, not the metric: gemm_mode='f16' (one MFMA per product); headline stays fp32-grade (w16)",
           "value": round(steps * w.B / dt, 1), "unit": "images/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
           "roofline": roofline_of(kern, "f16"),
           "roofline_other_kernels": {k: roofline_of(kern, "f16", k) for k in list(kern)[1:4]},
           "vs_fp32_grade_on_bench_batch": {
               "max_abs_logit_diff": float((la - lb).abs().max()), "logit_absmax": float(la.abs().max()),
               "pixel_flip_rate": float(((la >= 0) != (lb >= 0)).float().mean()),
               "argmax_objectness_changed": int((rows_a[:, 14] != rows_b[:, 14]).sum()),
               "mean_iou_fp32_grade": round(iou_a, 3), "mean_iou_f16": round(iou_b, 3),
               "iou_identical_to_3dp": round(iou_a, 3) == round(iou_b, 3)}}
    del w
    torch.cuda.empty_cache()
    return res


def _gloo() -> bool:
    return dist.is_initialized() and dist.get_backend() == "gloo"


def all_gather_rows(rows: torch.Tensor, world: int) -> torch.Tensor:
    """the path's one exchange: all-gather of the per-image result rows - RCCL on device tensors; on host copies
    under the gloo rehearsal (SM_BENCH_REHEARSAL=1)"""
    src = rows.cpu() if _gloo() else rows
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    return out


def max_over_ranks(values, dev) -> list:
    t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if _gloo() else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()


def spawn_ranks(n: int, argv) -> int:
    """Start `n` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and
    wait for them.  The parent never touches the GPU: ranks are fresh processes, nothing is exec'ed after HIP init."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(a, world: int, rank: int) -> None:
    """The launch / timing / gather skeleton of main() on CPU with gloo: K steps of a trivial per-rank "evaluation"
    (rank-strided rows), barrier on both sides, one all-gather inside the timed region, MAX over ranks."""
    import numpy as np
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus
    B = a.batch
    rows = torch.zeros((a.steps * B, 16))

    def sync():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        torch.ones(8).sum()
    sync()
    t0 = time.perf_counter()
    for k in range(a.steps):  # row i of rank r stands for global image (k*B + i) * world + r
        rows[k * B:(k + 1) * B, 0] = torch.arange(k * B, (k + 1) * B, dtype=torch.float32) * world + rank
    gathered = rows
    if world > 1:
        gathered = torch.empty((world * rows.shape[0], 16))
        dist.all_gather_into_tensor(gathered, rows)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ids = np.sort(gathered[:, 0].numpy())
    ok = bool((ids == np.arange(world * a.steps * B)).all())  # every global image exactly once after the gather
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest (gloo, CPU)", "value": world * a.steps * B / dt, "unit": "rows/sec",
                          "n_gpus": world, "rccl_ranks": 0, "gloo_ranks": world, "steps": a.steps, "warmup": a.warmup,
                          "gather_complete": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--patch", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight per GPU: step k runs on HIP stream k %% streams (own workspace), so one "
                         "batch's latency-bound decoder / metrics kernels fill CUs beside another's encoder GEMMs")
    ap.add_argument("--no-graph", action="store_true", help="launch the forward's kernels eagerly instead of replaying "
                                                            "one captured hipGraph per stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained-steps", type=int, default=500, help="extra untimed-by-the-contract leg: the same loop for this many steps")
    ap.add_argument("--quick", action="store_true", help="timed steps + roofline only (no sustained / CPU / end-to-end / serving legs)")
    ap.add_argument("--host-input", action="store_true",
                    help="diagnostic: every step copies its batch from pinned host memory first (PCIe-inclusive rate; "
                         "the metric keeps inputs resident in HBM)")
    ap.add_argument("--forward-only", action="store_true", help="diagnostic: skip the evaluator kernels (not the metric)")
    ap.add_argument("--zero-data", action="store_true",
                    help="diagnostic: all-zero weights and images - the same instruction stream with (almost) no switching "
                         "activity in the matrix cores; how far the result rises above the metric is how power-limited it is")
    ap.add_argument("--gemm-mode", default=None, choices=["w16", "f16x2", "fp32", "f16"],
                    help="GEMM back end (default w16; f16 = the one-MFMA-per-product diagnostic, not the metric)")
    ap.add_argument("--e2e-ranks", action="store_true", help="with --gpus N > 1: also run the end-to-end leg (files -> metrics) on all N ranks")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the ViT-S/8 224^2 and ViT-S/16 384^2 legs")
    ap.add_argument("--only-leg", default=None, choices=["refine_384", "pseudo_masks"],
                    help="run ONE of the extra legs and print its JSON (profiling runs); not the driver's contract")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU rehearsal of the N-rank launch path (gloo, no GPU): same spawn, barrier, gather and "
                         "max-over-ranks timing around a trivial step; used by tests/test_bench_launcher_cpu.py")
    a = ap.parse_args()

    # ---- N ranks: one process per GPU ------------------------------------------------------------------------------
    # `python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself, as FRESH child
    # processes (torch.distributed.run), before this process has made any GPU call; it relays rank 0's JSON line and
    # exits with the children's status.  Launched by torch.distributed.run directly (the driver's form) it is a rank.
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report n_gpus != requested")
    if a.launcher_selftest:
        return launcher_selftest(a, world, rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from selfmask_amd.distributed import pin_rank_cores
        pin_rank_cores()  # this rank's block of the host's cores: its decode workers inherit the affinity
        if os.environ.get("SM_BENCH_REHEARSAL") == "1":
            # logic rehearsal where only ONE GPU exists: every rank on cuda:0, gloo instead of RCCL (collectives on host copies)
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    if a.only_leg:
        leg = {"refine_384": refine_leg, "pseudo_masks": pseudo_masks_leg}[a.only_leg]
        print(json.dumps({a.only_leg: leg(dev, a.streams, cpu=not a.no_cpu_baseline)}), flush=True)
        return
    P, S, B = a.patch, a.size, a.batch
    wl = Workload(dev, P, S, B, rank=rank, gemm_mode=a.gemm_mode, streams=a.streams, graph=not a.no_graph, zero_data=a.zero_data,
                  host_input=a.host_input, forward_only=a.forward_only)
    model, x, fwd, ring, run_steps = wl.model, wl.x, wl.fwd, wl.ring, wl.run_steps

    # priming (untimed, not part of W): every stream sees the batch shape often enough to be admitted to the graph
    # cache, so its hipGraph is captured here and never inside the timed region, whatever --warmup is
    run_steps((fwd.policy.admit_after + 1) * len(ring.streams))
    run_steps(a.warmup)
    rows = torch.zeros((a.steps * B, 16), device=dev)  # per-image result rows (14 metrics + the two query ids)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    run_steps(a.steps, rows)
    t_enqueued = time.perf_counter() - t0  # host time to issue the K steps (launch-bound if close to dt)
    if world > 1:  # the path's one exchange: all-gather of the per-image rows
        gathered = all_gather_rows(rows, world)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks([dt], dev)[0]

    # per-kernel taps right behind the timed steps (same clocks / cache state as the K steps; agreement with rocprofv3's
    # kernel-trace averages of the same command is checked in profiles/)
    kern = time_forward_kernels(model, x) if rank == 0 else None

    # sustained leg (outside the contract's timed region): the K-step figure above is a ~0.2 s burst
    sustained = None
    if a.sustained_steps > 0 and not a.quick:
        sync()
        t1 = time.perf_counter()
        run_steps(a.sustained_steps)
        sync()
        ds_ = time.perf_counter() - t1
        if world > 1:
            ds_ = max_over_ranks([ds_], dev)[0]
        sustained = {"steps": a.sustained_steps, "value": round(world * a.sustained_steps * B / ds_, 1), "unit": "images/sec",
                     "ms_per_step": round(ds_ / a.sustained_steps * 1e3, 3)}

    # end to end (files -> metrics).  With several ranks every rank takes part (the Evaluator shards the files and gathers the rows
    # over RCCL) - opt-in there (--e2e-ranks): the driver's scaling runs measure the contract's K steps, and a leg that walks a
    # file tree on N ranks has no place inside them
    e2e = None
    if not a.quick and (world == 1 or a.e2e_ranks):
        e2e = end_to_end(model, dev, P, S, B, len(ring.streams), world=world, rank=rank)

    if rank == 0:
        value = world * a.steps * B / dt
        flops_img = forward_flops_per_image(P, S)
        dom_name, dom = next(iter(kern.items()))
        # HBM bytes per launch from the PMC passes of this same command (scripts/pmc_traffic.sh writes
        # profiles/r04_pmc_traffic.json with the hash of the kernel sources it ran on).  A profiler cannot run inside the
        # timed process, so the committed measurement is quoted - only if it was taken on exactly these sources, else null.
        traffic, traffic_note = None, "no PMC file for these kernel sources"
        for rnd in ("r04", "r03"):  # the newest PMC passes whose kernel-source hash matches
            pf = f"profiles/{rnd}_pmc_traffic.json"
            try:
                with open(os.path.join(REPO, pf)) as f:
                    pmc = json.load(f)
            except (OSError, ValueError):
                continue
            if pmc.get("source_hash") == source_hash():
                traffic = pmc.get("kernels", {}).get(dom_name, {}).get("hbm_bytes_per_launch")
                traffic_note = f"{pf} (separate FETCH_SIZE / WRITE_SIZE passes, FETCH x2 on gfx950)"
                break
            traffic_note = f"{pf} was taken on other kernel sources (hash mismatch): not quoted"
        peak = F32_MFMA_PEAK_TFLOPS if model.gemm_mode == "fp32" else F16_MFMA_PEAK_TFLOPS
        issue = 3.0 if model.gemm_mode in ("w16", "f16x2") else 1.0  # MFMA FLOPs issued per algorithmic FLOP (hi*hi, hi*lo, lo*hi)

        res = {
            "metric": "images/sec (224^2, nq=20)", "value": round(value, 1), "unit": "images/sec", "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(t_enqueued / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if model.gemm_mode == "fp32" else "f16 (diagnostic throughput mode: one MFMA per product)" if model.gemm_mode == "f16"
                     else "f32 (GEMM operands split into two f16 halves, 3 f16 MFMAs per product, f32 accumulate)",
            "data": "synthetic" if not a.zero_data else "all zeros (diagnostic, not the metric)",
            "config": {"workload": f"DUTS-TE-shaped synthetic images, ViT-S/{P} {S}x{S}, nq=20, batch={B}/GPU, "
                                   f"MaskFormer.forward + evaluator post-processing and metrics (BASELINE.json configs[1])",
                       "patch": P, "image_size": S, "batch_per_gpu": B, "n_queries": 20, "gemm_mode": model.gemm_mode,
                       "streams": len(ring.streams), "host_input": bool(a.host_input),
                       "hip_graph": {"captures": fwd.captures, "replays": fwd.replays, "failed": fwd.failed},
                       "parallelism": f"images sharded x{world}, one all-gather of result rows"},
            "model_tflops": round(value * flops_img / 1e12, 2),
            "roofline": dict(roofline_of(kern, model.gemm_mode, dom_name), traffic=traffic, traffic_source=traffic_note,
                             how="HIP events around every launch of this kernel inside 3 real forwards on ONE stream "
                                 "(sm_forward_timing); an empty event pair's time (event_overhead_us) is subtracted per launch",
                             whole_step={"achieved": round(value / world * flops_img / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                         "frac": round(value / world * flops_img / 1e12 / peak, 4),
                                         "mfma_issued_tflops": round(value / world * flops_img / 1e12 * issue, 2),
                                         "what": "algorithmic FLOPs of the forward x images/s of the timed region, per GPU: the rate the "
                                                 "three-batches-in-flight pipeline sustains (the tile shapes are chosen for THIS figure; "
                                                 "a lone launch of the dominant kernel is slower than with smaller tiles)"}),
            "roofline_other_kernels": {k: roofline_of(kern, model.gemm_mode, k) for k in list(kern)[1:]},
        }
        res["sustained"] = sustained
        res["end_to_end"] = e2e
        if world == 1 and not a.quick:
            res["serving"] = serving_latency(model, dev)
            if model.gemm_mode == "w16" and not a.zero_data:
                res["throughput_mode"] = throughput_mode_leg(dev, wl)
            if (P, S) == (16, 224) and not a.no_other_shapes:
                # the shapes the shipped checkpoint (ViT-S/8, configs/duts-...yaml:39) and configs[2] (384^2) run at
                # ViT-S/8 at batch 32 since round 4: its attention launch is 7 groups x 6 heads x B workgroups on 512 slots - 1.31 rounds of
                # work in 2 at batch 16 (4.89 k images/s; 5.07 k at 24, 5.10 k at 32: profiles/r04_vit_s8_by_batch.log)
                res["other_shapes"] = {"vit_s8_224": shape_leg(dev, 8, 224, 32, len(ring.streams), cpu=not a.no_cpu_baseline),
                                       "vit_s16_384": shape_leg(dev, 16, 384, 32, len(ring.streams), cpu=not a.no_cpu_baseline)}
                res["refine_384"] = refine_leg(dev, len(ring.streams), cpu=not a.no_cpu_baseline)       # configs[2]
                res["pseudo_masks"] = pseudo_masks_leg(dev, len(ring.streams), cpu=not a.no_cpu_baseline)  # configs[4]
        if world == 1 and not a.no_cpu_baseline and not a.quick:
            res["cpu_baseline"] = cpu_baseline(P, S)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
