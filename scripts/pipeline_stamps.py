#!/usr/bin/env python3
"""The same in-kernel stamps as gemm_stamps.py / qkv_stamps.py, but read after the bench's three-batches-in-flight pipeline
(tuning build): how long a tile's prologue / K loop / epilogue take when other streams' kernels share the chip.
usage: pipeline_stamps.py N K [bench args]   (N, K of the GEMM launches to stamp, e.g. 1536 384 = fc1)"""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
Nf, Kf = int(sys.argv[1]), int(sys.argv[2])
from selfmask_amd import _native as N
lib = N.load()
assert lib.sm_gemm_stamp_filter(Nf, Kf, 64 * 197) == 0
import bench, torch
WAVES, WG = 7, 384
buf = (ctypes.c_ulonglong * (2048 * 4))()
buf2 = (ctypes.c_ulonglong * (WG * WAVES * 8))()
orig = bench.time_forward_kernels


def snapshot_then(*args, **kw):  # the single-stream kernel taps run after the timed steps: read the stamps before them
    torch.cuda.synchronize()
    assert lib.sm_gemm_stamps(buf, 2048 * 4) == 0 and lib.sm_qkv_stamps(buf2, WG * WAVES * 8) == 0
    assert lib.sm_gemm_stamp_filter(-1, 0, 0) == 0
    return orig(*args, **kw)


bench.time_forward_kernels = snapshot_then
sys.argv = ["bench.py", "--quick", "--steps", "60"] + sys.argv[3:]
bench.main()
ntiles = int(os.environ.get("TILES", "150"))
t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4).astype(np.int64)[:ntiles]
t = t[(t[:, 0] > 0) & (t[:, 3] > t[:, 0])]
d = np.diff(t, axis=1)
print(f"GEMM N={Nf} K={Kf} in the pipeline, {len(t)} tiles of the last launch(es): prologue | K loop | epilogue | life (median ticks)")
print("   " + " | ".join(f"{np.median(d[:, i]):7.0f}" for i in range(3)) + f" | {np.median(t[:, 3] - t[:, 0]):7.0f}"
      + f"    p90 life {np.percentile(t[:, 3] - t[:, 0], 90):.0f}")
q = np.frombuffer(buf2, dtype=np.uint64).reshape(WG, WAVES, 8).astype(np.int64)
dq = np.diff(q[:, :, :7], axis=2)
print("fused QKV+attention in the pipeline: prologue | projection | drain | convert | attention | store | workgroup life")
print("   " + " | ".join(f"{np.median(dq[:, :, i]):7.0f}" for i in range(6)) + f" | {np.median(q[:, :, 6].max(1) - q[:, :, 0].min(1)):7.0f}")
