import os, sys
sys.path[:0] = ["/root/repo/salient-object-detection_amd", "/root/repo"]
import torch
from selfmask_amd import ops, _native as N
torch.manual_seed(0)
m, n, k = 64 * 197, 1152, 384
a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
a_s, w_s = ops.split_f16x2(a), ops.split_f16x2(w)
c = torch.empty(m, n, device="cuda")
tile = tuple(int(v) for v in os.environ.get("TILE", "128,64").split(","))
for _ in range(5):
    ops.gemm_f16x2(a_s, w_s, b, tile=tile, out=c)
torch.cuda.synchronize()
