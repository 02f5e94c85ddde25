#!/usr/bin/env python3
"""BASELINE.json configs[2] on one MI355X: 384x384 inputs (ViT-S/16 -> 577 tokens), nq = 20, forward + the evaluator
kernels, then the bilateral-solver refinement of the picked mask per image (device-resident; the CPU oracle's solver
beside it on a few images).  Prints images/s for the forward+metrics part and ms/image for the solver."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import numpy as np
import torch
from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images, GraphedForward, StreamRing, ops
from selfmask_amd.bilateral_solver import bilateral_solver_output_device, bilateral_solver_batch_device

dev = torch.device("cuda:0")
B, S, P = 32, 384, 16
m = MaskFormer(n_queries=20, patch_size=P, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
m.load_state_dict(synthetic_state_dict(0, "soft", patch_size=P)); m = m.to(dev)
x = torch.from_numpy(synthetic_images(5, (B, 3, S, S))).to(dev)
rng = np.random.Generator(np.random.PCG64(3))
gts = []
for _ in range(B):
    yy, xx = np.mgrid[:S, :S]
    gts.append(torch.from_numpy((((yy - S * rng.uniform(.3, .7)) / (S * .2)) ** 2 + ((xx - S * rng.uniform(.3, .7)) / (S * .25)) ** 2 <= 1).astype(np.uint8)))
gb = ops.GtBatch(gts, dev)
fwd, ring = GraphedForward(m), StreamRing(dev, 3)

def step():
    out = fwd(x)
    return ops.evaluate_masks(out["mask_pred"][:, -1], out["objectness"][:, -1, :, 0], gb, scale=0.0), out

def run(n):
    ring.fork()
    for _ in range(n):
        with ring.next():
            step()
    ring.join()

run(9); torch.cuda.synchronize()
t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"forward + metrics @384^2, B={B}, 3 streams: {30 * B / dt:.1f} images/s ({dt / 30 * 1e3:.2f} ms per step)")

rows, out = step()
masks = (out["mask_pred"][:, -1] >= 0.5)
img_u8 = ((x.permute(0, 2, 3, 1) * 0.22 + 0.45).clamp(0, 1) * 255).to(torch.uint8).contiguous()
up = torch.nn.functional.interpolate(out["mask_pred"][:, -1], size=(S, S), mode="bilinear", align_corners=False)
q = rows[:, 14].long()
tgt = torch.stack([(up[i, q[i]] >= 0.5).double() for i in range(B)])
for _ in range(2):
    for i in range(B): bilateral_solver_output_device(img_u8[i], tgt[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(B): bilateral_solver_output_device(img_u8[i], tgt[i])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"bilateral solver @384^2 on the device: {dt / B * 1e3:.2f} ms per image ({B / dt:.1f} images/s, one stream)")
for nb in (32, 128, 256):
    rep = nb // B
    big_i, big_t = img_u8.repeat(rep, 1, 1, 1), tgt.repeat(rep, 1, 1)
    bilateral_solver_batch_device(big_i, big_t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    soft_b, bin_b = bilateral_solver_batch_device(big_i, big_t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"bilateral solver @384^2, batch of {nb}: {dt * 1e3:.1f} ms = {dt / nb * 1e3:.3f} ms per image ({nb / dt:.0f} images/s)")
ref_s, ref_b = bilateral_solver_output_device(img_u8[5], tgt[5])
print("batched == single:", bool(torch.equal(soft_b[5], ref_s) and torch.equal(bin_b[5], ref_b)))
try:
    from oracle import bilateral_oracle as BO
    t0 = time.perf_counter()
    for i in range(3): BO.bilateral_solver_output(img_u8[i].cpu().numpy(), tgt[i].cpu().numpy())
    print(f"CPU oracle solver: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per image (1 core)")
except Exception as e:
    print("oracle solver not timed:", e)
