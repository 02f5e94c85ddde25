#!/usr/bin/env python3
"""Headline benchmark: images/sec of the SelfMask inference hot path (ViT-S/16, 224^2, nq=20, batch 64 per GPU).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" is one MaskFormer.forward over one batch of synthetic images already resident in HBM (weights = synthetic
checkpoint seed 0; there is no dataset or checkpoint offline).  Images shard across ranks with no data-path
collective (weak scaling); the only exchange is the evaluator's end-of-run all-gather of per-image result rows
(RCCL), issued once after the K steps inside the timed region.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      - the dominant kernel (fp32-MFMA GEMM family) timed live with HIP events on the launch stream;
  cpu_baseline  - the CPU oracle (torch-CPU restatement of the reference) on this box's host cores, rank 0, N=1 only.
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


def forward_gemm_launches(B, P, S, L=6, nq=20):
    """Every GEMM launch of one MaskFormer.forward:
    (name, M, N, K, epilogue, batch, launches per forward, split_k, F16X2 output in split mode)."""
    from selfmask_amd import _native as Nn
    g = S // P
    n, N = g * g, g * g + 1
    M, Mp, Md, Mo = B * N, B * n, B * nq, B * nq * L
    return [
        ("patch_embed", Mp, 384, 3 * P * P, Nn.EPI_BIAS, 1, 1, 1, False),
        ("enc.qkv", M, 1152, 384, Nn.EPI_BIAS, 1, 12, 1, True), ("enc.proj", M, 384, 384, Nn.EPI_RESIDUAL, 1, 12, 1, False),
        ("enc.fc1", M, 1536, 384, Nn.EPI_GELU, 1, 12, 1, True), ("enc.fc2", M, 384, 1536, Nn.EPI_RESIDUAL, 1, 12, 1, False),
        ("dec.ca_kv_all_layers", Mp, L * 768, 384, Nn.EPI_BIAS, 1, 1, 1, True),
        ("dec.sa_qkv", Md, 1152, 384, Nn.EPI_BIAS, 1, L, 1, True), ("dec.sa_out", Md, 384, 384, Nn.EPI_RESIDUAL, 1, L, 1, False),
        ("dec.ca_q", Md, 384, 384, Nn.EPI_BIAS, 1, L, 1, True), ("dec.ca_out", Md, 384, 384, Nn.EPI_RESIDUAL, 1, L, 1, False),
        ("dec.lin1", Md, 1536, 384, Nn.EPI_RELU, 1, L, 1, True), ("dec.lin2_splitk4", Md, 384, 1536, Nn.EPI_BIAS, 1, L, 4, False),
        ("mask_einsum", L * nq, 4 * n, 384, Nn.EPI_BIAS, B, 1, 1, False),
        ("obj.ffn0", Mo, 384, 384, Nn.EPI_RELU, 1, 1, 1, True), ("obj.ffn1", Mo, 384, 384, Nn.EPI_RELU, 1, 1, 1, True),
    ]


def time_gemm_kernels(B, P, S, mode, iters=3):
    """Live per-kernel timing of the GEMM instantiations (the kernels that hold most of the forward): for each
    kernel name (= back end + workgroup tile) replay exactly the launch mix one forward issues, bracketed by HIP events
    on the stream the library launches on (torch's current stream).  Returns {kernel: dict} sorted by time."""
    import ctypes
    from selfmask_amd import ops, _native as Nn
    lib = Nn.load()
    dev = "cuda"
    split_mode = mode == "f16x2"
    groups = {}
    for name, M, N, K, epi, batch, cnt, split, osplit in forward_gemm_launches(B, P, S):
        ga = Nn.GemmArgs()
        ga.M, ga.N, ga.K, ga.batch, ga.split_k = M, N, K, batch, split
        bm, bn, nst = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if split_mode:
            Nn.check(lib.sm_gemm_f16x2_pick_tile(ga, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nst)))
            kname = f"gemm_f16x2_kernel<{bm.value}, {bn.value}, {nst.value}, 2, 2, 3>"  # 2x2 waves, 3 workgroups/CU
        else:
            Nn.check(lib.sm_gemm_f32_pick_tile(ga, ctypes.byref(bm), ctypes.byref(bn)))
            nst_f32 = {(128, 128): 2, (128, 64): 3, (64, 64): 4}[(bm.value, bn.value)]
            kname = f"gemm_f32_kernel<{bm.value}, {bn.value}, {nst_f32}>"
        a = torch.randn(batch, M, K, device=dev)
        w = torch.randn(batch, N, K, device=dev) * 0.03
        if split_mode:  # operands in the F16X2 split format, as the forward's producers write them
            a, w = ops.split_f16x2(a), ops.split_f16x2(w)
        bias = torch.zeros(N, device=dev) if split == 1 else None
        c = torch.empty(max(batch, split), M, N, device=dev)
        r = torch.randn(batch, M, N, device=dev) if epi == Nn.EPI_RESIDUAL else None
        groups.setdefault((kname, bm.value, bn.value), []).append((name, a, w, bias, c, r, epi, cnt, 2.0 * M * N * K * batch, split, osplit))
    out = {}
    for (kname, bm, bn), items in groups.items():
        def run_mix():
            for name, a, w, bias, c, r, epi, cnt, fl, split, osplit in items:
                for _ in range(cnt):
                    if split_mode:
                        batched = a.shape[0] > 1
                        ops.gemm_f16x2(a if batched else a[0], w if batched else w[0], bias, epilogue=epi,
                                       residual=None if r is None else (r if batched else r[0]), tile=(bm, bn),
                                       out=c if (split > 1 or batched) else c[0], split_k=split, out_f16x2=osplit)
                    else:
                        ops.gemm(a, w, bias, epilogue=epi, residual=r, out=c, tile=(bm, bn), split_k=split)
        run_mix()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run_mix()
        e1.record()
        torch.cuda.synchronize()
        launches = sum(i[7] for i in items)
        flops = sum(i[7] * i[8] for i in items)
        total_s = e0.elapsed_time(e1) * 1e-3 / iters
        out[kname] = {
            "launches_per_forward": launches, "avg_launch_us": total_s / launches * 1e6,
            "flops_per_launch": flops / launches, "total_ms_per_forward": total_s * 1e3,
            "achieved_tflops": flops / total_s / 1e12, "mix": [i[0] for i in items]}
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms_per_forward"]))


def cpu_baseline(P, S, budget_s=15.0):
    """The CPU oracle (torch-CPU restatement, fp32) on this host: bounded sample of the same workload."""
    from oracle import selfmask_oracle as O  # measured as the BASELINE only, never on the product path
    from selfmask_amd import synthetic_state_dict, synthetic_images
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("SM_CPU_BASELINE_THREADS", "16")))  # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(0, "soft", patch_size=P)
    Bc = 16
    x = torch.from_numpy(synthetic_images(1234, (Bc, 3, S, S)))
    O.forward(x, sd, P)  # warm-up
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        O.forward(x, sd, P)
        n += Bc
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} images (batches of {Bc}, ViT-S/{P} {S}x{S}, fp32 torch-CPU oracle) in {dt:.1f}s"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true", help="diagnostic: skip the evaluator kernels (not the metric)")
    ap.add_argument("--gemm-mode", default=None, choices=["f16x2", "fp32"], help="GEMM back end (default f16x2)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images
    P, S, B = a.patch, a.size, a.batch
    model = MaskFormer(n_queries=20, patch_size=P, n_decoder_layers=6, return_intermediate=True,
                       use_binary_classifier=True, gemm_mode=a.gemm_mode)
    model.load_state_dict(synthetic_state_dict(0, "soft", patch_size=P), strict=True)
    model = model.to(dev)
    # every rank owns a different shard of the (synthetic) image list: rank-strided seeds
    x = torch.from_numpy(synthetic_images(1234 + rank, (B, 3, S, S))).to(dev)
    # synthetic ground truth at DUTS-like native sizes (300-400 px ellipses), packed once and resident in HBM
    import numpy as np
    from selfmask_amd import ops
    rng = np.random.Generator(np.random.PCG64(99 + rank))
    gts = []
    for _ in range(B):
        h, w = (int(v) for v in rng.integers(300, 401, size=2))
        yy, xx = np.mgrid[:h, :w]
        gts.append(torch.from_numpy(((((yy - h * rng.uniform(.3, .7)) / (h * rng.uniform(.1, .3))) ** 2 +
                                      ((xx - w * rng.uniform(.3, .7)) / (w * rng.uniform(.1, .3))) ** 2) <= 1)
                                    .astype(np.uint8)))
    gt_batch = ops.GtBatch(gts, dev)

    def step():
        # one evaluator iteration over a batch (evaluator.pyc@L193-228, batched mode): forward, last decoder
        # layer, up-sample to each GT's size, upper-bound + arg-max-objectness query, 7 metrics x 2 -> 16 floats/image
        out = model(x)
        if a.forward_only:
            return out["objectness"][:, -1, :16, 0]
        return ops.evaluate_masks(out["mask_pred"][:, -1], out["objectness"][:, -1, :, 0], gt_batch, scale=0.0)

    from selfmask_amd.streams import StreamRing
    ring = StreamRing(dev, max(1, a.streams))

    def run_steps(n, dst=None):
        ring.fork()
        for k in range(n):
            with ring.next():
                r = step()
                if dst is not None:
                    dst[k * B:(k + 1) * B] = r
        ring.join()

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
    if world > 1:  # the path's one exchange: all-gather of the per-image rows
        gathered = torch.empty((world * rows.shape[0], 16), device=dev)
        dist.all_gather_into_tensor(gathered, rows)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        value = world * a.steps * B / dt
        flops_img = forward_flops_per_image(P, S)
        kern = time_gemm_kernels(B, P, S, model.gemm_mode)
        dom_name, dom = next(iter(kern.items()))
        ach = dom["achieved_tflops"]
        peak = F32_MFMA_PEAK_TFLOPS if model.gemm_mode == "fp32" else F16_MFMA_PEAK_TFLOPS
        res = {
            "metric": "images/sec (224^2, nq=20)", "value": round(value, 1), "unit": "images/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if model.gemm_mode == "fp32" else "f32 (GEMM operands split into two f16 halves, f16 MFMA, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"DUTS-TE-shaped synthetic images, ViT-S/{P} {S}x{S}, nq=20, batch={B}/GPU, "
                                   f"MaskFormer.forward + evaluator post-processing and metrics (BASELINE.json configs[1])",
                       "patch": P, "image_size": S, "batch_per_gpu": B, "n_queries": 20, "gemm_mode": model.gemm_mode,
                       "streams": len(ring.streams),
                       "parallelism": f"images sharded x{world}, one all-gather of result rows"},
            "model_tflops": round(value * flops_img / 1e12, 2),
            "roofline": {"bound": "mfma", "kernel": dom_name, "launch_mix": dom["mix"],
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": None,
                         "mfma_issued_tflops": round(ach * (3.0 if model.gemm_mode == "f16x2" else 1.0), 2),
                         "avg_launch_us": round(dom["avg_launch_us"], 2),
                         "launches_per_forward": dom["launches_per_forward"],
                         "flops_per_launch": dom["flops_per_launch"]},
            "roofline_other_kernels": {k: {"achieved": round(v["achieved_tflops"], 2),
                                           "frac": round(v["achieved_tflops"] / peak, 4),
                                           "avg_launch_us": round(v["avg_launch_us"], 2),
                                           "launches_per_forward": v["launches_per_forward"]}
                                       for k, v in list(kern.items())[1:]},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(P, S)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
