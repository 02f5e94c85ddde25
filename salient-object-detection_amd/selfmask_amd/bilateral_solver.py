"""Mirror of the reference's ``bilateral_solver.bilateral_solver_output`` (bilateral_solver.py:152-193) on the MI355X.

Same signature and return value: ``(output_solver: float64 (H,W), binary_solver: bool (H,W))``.  ``img`` may be a PIL
image (as in the reference) or an (H,W,3) uint8 array; ``target`` an (H,W) array or tensor.  The whole pipeline (grid
construction, bistochastisation, PCG, slicing, hole filling, component selection) runs in libselfmask_hip.so; only
the two result arrays come back to the host.
"""
from typing import Tuple

import numpy as np
import torch

from . import _native as N


def bilateral_solver_output_device(img_u8: torch.Tensor, target: torch.Tensor, sigma_spatial=16, sigma_luma=16,
                                   sigma_chroma=8, return_info: bool = False):
    """Device-resident form: img_u8 (H,W,3) uint8, target (H,W) float64, both on the HIP device.  Returns device
    tensors (soft float64 (H,W), binary uint8 (H,W)[, info int32 (4)])."""
    if not (img_u8.is_cuda and target.is_cuda):
        raise RuntimeError("bilateral solver (MI355X) needs device tensors; there is no CPU fallback")
    H, W = target.shape
    assert img_u8.shape == (H, W, 3) and img_u8.dtype == torch.uint8
    img_u8, target = img_u8.contiguous(), target.contiguous().to(torch.float64)
    lib = N.load()
    nbytes = lib.sm_bilateral_workspace_bytes(H, W, float(sigma_spatial), float(sigma_luma), float(sigma_chroma))
    if nbytes == 0:
        raise ValueError("unsupported size / sigmas for the bilateral lattice")
    dev = target.device
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    soft = torch.empty((H, W), dtype=torch.float64, device=dev)
    binary = torch.empty((H, W), dtype=torch.uint8, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    a = N.BilateralArgs()
    a.img, a.target, a.soft, a.binary, a.info = img_u8.data_ptr(), target.data_ptr(), soft.data_ptr(), binary.data_ptr(), info.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    a.sigma_spatial, a.sigma_luma, a.sigma_chroma = float(sigma_spatial), float(sigma_luma), float(sigma_chroma)
    a.lam, a.a_diag_min, a.cg_tol, a.confidence, a.cg_maxiter = 256.0, 1e-5, 1e-5, 0.999, 25  # bs_params (:170-175)
    a.H, a.W = H, W
    N.check(lib.sm_bilateral_solver_f64(a, torch.cuda.current_stream().cuda_stream), "sm_bilateral_solver_f64")
    return (soft, binary, info) if return_info else (soft, binary)


def bilateral_solver_batch_device(imgs_u8: torch.Tensor, targets: torch.Tensor, sigma_spatial=16, sigma_luma=16,
                                  sigma_chroma=8, return_info: bool = False):
    """Many images of one size in one launch sequence (sm_bilateral_solver_batch_f64): imgs_u8 (B,H,W,3) uint8,
    targets (B,H,W) float64 on the device -> (soft (B,H,W) float64, binary (B,H,W) uint8[, info (B,4) int32]).
    The solver's two long kernels (bistochastize + PCG; hole filling + component labelling) run in one workgroup per
    image - latency-bound by design, the lattice has a few thousand vertices - so a lone solve occupies one of the 256
    CUs and a batch fills them.  Results are bit-identical to per-image calls."""
    if not (imgs_u8.is_cuda and targets.is_cuda):
        raise RuntimeError("bilateral solver (MI355X) needs device tensors; there is no CPU fallback")
    B, H, W = targets.shape
    assert imgs_u8.shape == (B, H, W, 3) and imgs_u8.dtype == torch.uint8
    imgs_u8, targets = imgs_u8.contiguous(), targets.contiguous().to(torch.float64)
    lib = N.load()
    nbytes = lib.sm_bilateral_workspace_bytes(H, W, float(sigma_spatial), float(sigma_luma), float(sigma_chroma))
    if nbytes == 0:
        raise ValueError("unsupported size / sigmas for the bilateral lattice")
    dev = targets.device
    ws = torch.empty(nbytes * B, dtype=torch.uint8, device=dev)
    soft = torch.empty((B, H, W), dtype=torch.float64, device=dev)
    binary = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    a = N.BilateralArgs()
    a.img, a.target, a.soft, a.binary, a.info = imgs_u8.data_ptr(), targets.data_ptr(), soft.data_ptr(), binary.data_ptr(), info.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes * B
    a.sigma_spatial, a.sigma_luma, a.sigma_chroma = float(sigma_spatial), float(sigma_luma), float(sigma_chroma)
    a.lam, a.a_diag_min, a.cg_tol, a.confidence, a.cg_maxiter = 256.0, 1e-5, 1e-5, 0.999, 25  # bs_params (:170-175)
    a.H, a.W = H, W
    N.check(lib.sm_bilateral_solver_batch_f64(a, B, torch.cuda.current_stream().cuda_stream), "sm_bilateral_solver_batch_f64")
    return (soft, binary, info) if return_info else (soft, binary)


def bilateral_solver_output(img, target, sigma_spatial=16, sigma_luma=16, sigma_chroma=8,
                            device="cuda:0") -> Tuple[np.ndarray, np.ndarray]:
    reference = np.array(img)  # PIL image or array (:159)
    t = target.detach().cpu().numpy() if torch.is_tensor(target) else np.asarray(target)
    soft, binary = bilateral_solver_output_device(torch.from_numpy(np.ascontiguousarray(reference)).to(device),
                                                  torch.from_numpy(t.astype(np.float64)).to(device),
                                                  sigma_spatial, sigma_luma, sigma_chroma)
    return soft.cpu().numpy(), binary.cpu().numpy().astype(bool)
