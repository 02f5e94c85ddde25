"""The split-operand (F16X2) GEMM through the C ABI against fp64: every workgroup tile, epilogue and output format,
ragged M / N, batched launches, split-K, and the fused residual + LayerNorm epilogue of the 64 x 384 full-row tile."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from selfmask_amd import ops, _native as N  # noqa: E402

DEV = "cuda:0"
TILES = [(256, 128), (128, 128), (128, 64), (64, 64), (64, 384)]


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _ref(a, w, b, epi, r):
    y = a.double() @ w.double().T + (b.double() if b is not None else 0.0)
    if epi == N.EPI_GELU:
        y = F.gelu(y)
    elif epi == N.EPI_RELU:
        y = F.relu(y)
    elif epi == N.EPI_RESIDUAL:
        y = y + r.double()
    return y


def test_split_round_trip():
    x = _rand(37, 384, seed=1, scale=30.0).to(DEV)
    back = ops.unsplit_f16x2(ops.split_f16x2(x))
    assert ((back - x).abs() <= x.abs() * 2.0 ** -21 + 1e-7).all()  # 22 significant bits (f16 range permitting)


def test_split_bits_match_the_definition():
    """hi = f16(x) (round to nearest even), lo = f16((x - hi) * 2^11): the packed-convert / fma-mix form of the kernels
    (common.h split2) must give exactly these bits - every producer (LayerNorm, GEMM / attention epilogues) relies on it."""
    import numpy as np
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(64, 384, generator=g) * s for s in (1e-6, 1e-3, 1.0, 30.0, 3000.0)])
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0 / 2, 2.0 ** -14, 2.0 ** -24, 1.0 + 2.0 ** -11])
    got = ops.split_f16x2(x.to(DEV)).cpu().contiguous().view(torch.float16).reshape(x.shape[0], -1, 2, 8).numpy()
    xn = x.numpy()
    hi = xn.astype(np.float16)
    lo = ((xn - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    want = np.stack([hi.reshape(x.shape[0], -1, 8), lo.reshape(x.shape[0], -1, 8)], axis=2)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("tile", TILES, ids=[f"{a}x{b}" for a, b in TILES])
@pytest.mark.parametrize("M,Nn,K,epi,osplit", [
    (197 * 3, 384, 384, N.EPI_RESIDUAL, False),   # proj-like, ragged M
    (197 * 3, 1536, 384, N.EPI_GELU, True),       # fc1: GELU + F16X2 output
    (300, 384, 1536, N.EPI_RESIDUAL, False),      # fc2: long K
    (130, 1152, 384, N.EPI_BIAS, True),           # qkv, F16X2 output
    (61, 200, 64, N.EPI_RELU, False),             # N not a multiple of any tile, short K
    (20, 384, 384, N.EPI_BIAS, False),            # one partial tile
])
def test_tiles_epilogues_formats(tile, M, Nn, K, epi, osplit):
    a, w, b = _rand(M, K, seed=2), _rand(Nn, K, seed=3, scale=0.05), _rand(Nn, seed=4)
    r = _rand(M, Nn, seed=5) if epi == N.EPI_RESIDUAL else None
    c = ops.gemm_f16x2(ops.split_f16x2(a.to(DEV)), ops.split_f16x2(w.to(DEV)), b.to(DEV), epilogue=epi,
                       residual=None if r is None else r.to(DEV), tile=tile, out_f16x2=osplit)
    got = ops.unsplit_f16x2(c) if osplit else c
    ref = _ref(a, w, b, epi, r)
    ref32 = _ref(a.float(), w.float(), b.float(), epi, r).float() if False else None
    err = (got.double().cpu() - ref).abs().max().item()
    assert err <= 4e-6 * max(1.0, ref.abs().max().item()), err  # fp32-grade (a torch fp32 GEMM is at 2-7e-6 here)


def test_batched_and_split_k_and_second_a_operand():
    a, w = _rand(4, 120, 384, seed=6), _rand(4, 196, 384, seed=7, scale=0.05)
    c = ops.gemm_f16x2(ops.split_f16x2(a.to(DEV)), ops.split_f16x2(w.to(DEV)), None, tile=(64, 64))
    ref = torch.einsum("bmk,bnk->bmn", a.double(), w.double())
    assert (c.double().cpu() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    a2, w2 = _rand(1280, 1536, seed=8), _rand(384, 1536, seed=9, scale=0.03)
    parts = ops.gemm_f16x2(ops.split_f16x2(a2.to(DEV)), ops.split_f16x2(w2.to(DEV)), None, tile=(64, 64), split_k=4)
    ref2 = a2.double() @ w2.double().T
    assert (parts.sum(0).double().cpu() - ref2).abs().max().item() <= 4e-6 * ref2.abs().max().item()


@pytest.mark.parametrize("M,K", [(197 * 2, 384), (64, 1536), (1000, 1536), (7, 384)])
def test_fused_residual_layernorm_epilogue(M, K):
    """SM_EPI_RESIDUAL_LN on the 64 x 384 full-row tile: C equals the plain residual epilogue bit for bit, C2 equals
    sm_layernorm_rows_f32 applied to that C (same lanes, same reduction tree) and LayerNorm in fp64 to 2e-6."""
    a, w, b = _rand(M, K, seed=10), _rand(384, K, seed=11, scale=0.05), _rand(384, seed=12)
    r, gam, bet = _rand(M, 384, seed=13, scale=3.0), _rand(384, seed=14) * 0.2 + 1.0, _rand(384, seed=15) * 0.1
    a_s, w_s = ops.split_f16x2(a.to(DEV)), ops.split_f16x2(w.to(DEV))
    c, xn = ops.gemm_f16x2(a_s, w_s, b.to(DEV), residual=r.to(DEV), tile=(64, 384), ln=(gam.to(DEV), bet.to(DEV), 1e-6))
    plain = ops.gemm_f16x2(a_s, w_s, b.to(DEV), epilogue=N.EPI_RESIDUAL, residual=r.to(DEV), tile=(64, 384))
    assert torch.equal(c, plain)
    assert torch.equal(c, ops.gemm_f16x2(a_s, w_s, b.to(DEV), epilogue=N.EPI_RESIDUAL, residual=r.to(DEV), tile=(128, 128)))
    ref = F.layer_norm(c.double().cpu(), (384,), gam.double(), bet.double(), 1e-6)
    got = ops.unsplit_f16x2(xn).double().cpu()
    assert (got - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    ln_kernel = ops.layernorm(c, gam.to(DEV), bet.to(DEV), 1e-6)  # fp32 output of the stand-alone kernel
    assert (ops.unsplit_f16x2(xn) - ln_kernel).abs().max().item() <= 2.0 ** -20 * max(1.0, ln_kernel.abs().max().item())
