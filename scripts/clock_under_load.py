#!/usr/bin/env python3
"""What clock does the chip hold under the encoder GEMMs?  A one-wave probe (tuning build: sm_clock_probe) samples s_memtime
against the 100 MHz s_memrealtime counter on a side stream while the main stream runs (a) nothing, (b) the W16 qkv GEMM back to
back on random operands, (c) the same on all-zero operands, (d) the fused QKV+attention kernel."""
import os, sys, ctypes
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))
import torch
from selfmask_amd import ops, _native as N

lib = N.load()
lib.sm_clock_probe.restype = ctypes.c_int
lib.sm_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
side = torch.cuda.Stream()
M = 12608
g = torch.Generator().manual_seed(1)


def operands(zero):
    a = ops.split_f16x2((torch.randn(M, 384, generator=g) * (0.0 if zero else 1.0)).to(dev))
    w = (torch.randn(1152, 384, generator=g) * 0.03).to(dev)
    w16, ws = ops.split_w16(w * 0.0 + 1e-30 if zero else w)
    return a, w16, ws, torch.randn(1152, generator=g).to(dev), torch.empty(1, M, 1152, device=dev)


def probe(load, label, ms=400):
    out = torch.zeros(8 * 4, dtype=torch.int64, device=dev)
    for _ in range(50):
        load()                      # warm: clocks settle under the load before the probe starts
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        lib.sm_clock_probe(out.data_ptr(), 20000, 8, side.cuda_stream)  # 8 probes in sequence, ~20000 x 2048 clocks each
    n = 0
    ev = torch.cuda.Event(); 
    with torch.cuda.stream(side):
        ev.record()
    while not ev.query():
        load(); n += 1
    torch.cuda.synchronize()
    o = out.cpu().view(8, 4)
    mhz = [float((r[2] - r[0]) / max(1, (r[3] - r[1])) * 100.0) for r in o]
    print(f"{label:44s} shader clock {min(mhz):7.0f} .. {max(mhz):7.0f} MHz over 8 probes ({n} launches of the load meanwhile)")


a, w16, ws, b, c = operands(False)
az, w16z, wsz, bz, cz = operands(True)
xn = torch.randn(64 * 197, 384, device=dev)
wq = torch.randn(1152, 384, device=dev) * 0.05
bq = torch.zeros(1152, device=dev)
probe(lambda: None if torch.cuda._sleep(100000) else None, "idle (a sleep kernel on the main stream)")
probe(lambda: ops.gemm_w16(a, w16, ws, b, variant=2, out=c, out_f16x2=True), "W16 qkv GEMM 128x128, random operands")
probe(lambda: ops.gemm_w16(az, w16z, wsz, bz, variant=2, out=cz, out_f16x2=True), "W16 qkv GEMM 128x128, zero operands")
probe(lambda: ops.gemm_w16(a, w16, ws, b, variant=31, out=c, out_f16x2=True), "W16 qkv GEMM 256x128 deep ring, random")
probe(lambda: ops.qkv_attention(xn, wq, bq, 64), "fused QKV+attention (incl. its split kernels)")
