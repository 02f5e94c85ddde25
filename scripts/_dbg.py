import os, sys
sys.path[:0] = ["/root/repo/salient-object-detection_amd", "/root/repo"]
import torch
from selfmask_amd import ops, _native as N
torch.manual_seed(0)
M=300
a = torch.randn(M, 384, device="cuda"); w1 = torch.randn(1536, 384, device="cuda") * 0.05
epi=N.EPI_GELU; tile=(64,64)
c32 = ops.gemm_f16x2(ops.split_f16x2(a), ops.split_f16x2(w1), None, epilogue=epi, tile=tile)
cs = ops.gemm_f16x2(ops.split_f16x2(a), ops.split_f16x2(w1), None, epilogue=epi, out_f16x2=True, tile=tile)
ref = ops.split_f16x2(c32)
A = cs.view(torch.float16).view(M, -1, 2, 8).float(); R = ref.view(torch.float16).view(M, -1, 2, 8).float()
rec = lambda T: (T[:,:,0] + T[:,:,1]/2048).reshape(M,-1)
print("reconstructed max diff epilogue-split vs fp32:", (rec(A)-c32).abs().max().item(), " kernel-split vs fp32:", (rec(R)-c32).abs().max().item())
d=(rec(A)-c32).abs(); i=d.argmax(); m,n=divmod(i.item(), c32.shape[1]); print(m,n,c32[m,n].item(), rec(A)[m,n].item(), A[m,n//8,:,n%8], R[m,n//8,:,n%8])
print("hi mismatch frac", (A[:,:,0]!=R[:,:,0]).float().mean().item(), "lo mismatch frac", (A[:,:,1]!=R[:,:,1]).float().mean().item())
