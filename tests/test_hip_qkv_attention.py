"""The fused QKV-projection + attention kernel (qkv_attention.hip, the north-star kernel) through the C ABI against an
fp64 evaluation of Attention.forward up to the output projection (vision_transformer.py:113-131), and against the
unfused pair of launches it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from selfmask_amd import ops, _native as N  # noqa: E402

DEV = "cuda:0"


def _ref(xn, w, b, B, scale):
    M = xn.shape[0]
    n = M // B
    qkv = (xn.double() @ w.double().T + b.double()).reshape(B, n, 3, 6, 64).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = torch.softmax((q @ k.transpose(-2, -1)) * scale, dim=-1)
    return (attn @ v).transpose(1, 2).reshape(M, 384)


@pytest.mark.parametrize("B,n,wstd,osplit", [
    (3, 197, 0.02, False),    # the headline shape: ViT-S/16 224^2
    (2, 197, 0.08, True),     # peaky softmax (far from uniform), F16X2 output
    (8, 197, 0.05, False),    # (B*6) % 8 == 0: the XCD-aware (image, head) order
    (1, 208, 0.05, False),    # the largest grid the kernel takes
    (2, 193, 0.05, True),     # one token in the last query block
    (2, 65, 0.05, False),     # small grids: the upper waves hold no queries
    (5, 1, 0.05, False),      # a single token
])
def test_fused_matches_fp64_and_unfused(B, n, wstd, osplit):
    g = torch.Generator().manual_seed(100 + n)
    xn = torch.randn(B * n, 384, generator=g)
    w = torch.randn(1152, 384, generator=g) * wstd
    b = torch.randn(1152, generator=g) * 0.1
    ref = _ref(xn, w, b, B, 0.125)
    got = ops.qkv_attention(xn.to(DEV), w.to(DEV), b.to(DEV), B, 0.125, out_f16x2=osplit)
    got = (ops.unsplit_f16x2(got) if osplit else got).double().cpu()
    err = (got - ref).abs().max().item()
    # the unfused product path: W16 GEMM (F16X2 output) + F16X2 attention
    w16, ws = ops.split_w16(w.to(DEV))
    qkv = ops.gemm_w16(ops.split_f16x2(xn.to(DEV)), w16, ws, b.to(DEV), out_f16x2=True)
    qkv5 = ops.unsplit_f16x2(qkv).reshape(B, n, 3, 6, 64)
    unf = ops.attention(qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2], 0.125, split=True).reshape(B * n, 384).double().cpu()
    err_unf = (unf - ref).abs().max().item()
    print(f"\nB={B} N={n} wstd={wstd}: fused-fp64 {err:.2e}  unfused-fp64 {err_unf:.2e}  max|ref| {ref.abs().max():.2f}")
    assert err <= 3e-6 * max(1.0, ref.abs().max().item())
    assert err <= 3.0 * err_unf + 1e-6


def test_rescale_branch_and_masked_tail():
    """One key whose score towers over the rest late in the sequence forces the lazy-maximum rescale branch; the tail
    keys 197..207 (stored duplicates of the last token) and the LDS rows past them must never leak into the result."""
    B, n = 2, 197
    g = torch.Generator().manual_seed(7)
    xn = torch.randn(B * n, 384, generator=g)
    w = torch.randn(1152, 384, generator=g) * 0.05
    b = torch.zeros(1152)
    xn[150] *= 6.0      # a huge key (and query) in the third 64-key chunk of image 0
    xn[n + 196] *= 8.0  # and the very last token of image 1
    ref = _ref(xn, w, b, B, 0.125)
    got = ops.qkv_attention(xn.to(DEV), w.to(DEV), b.to(DEV), B, 0.125).double().cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item())


def test_token_limit_is_reported():
    lib = N.load()
    assert lib.sm_qkv_attention_max_tokens() == 208
    with pytest.raises(RuntimeError, match="N=209"):
        ops.qkv_attention(torch.zeros(209, 384, device=DEV), torch.zeros(1152, 384, device=DEV) + 0.01,
                          torch.zeros(1152, device=DEV), 1)


@pytest.mark.parametrize("B,n", [(3, 197), (16, 197), (2, 65)])
def test_fused_kernel_with_folded_layernorm(B, n):
    """norm1 folded into the projection (sm_qkv_attn_args.ln_stats): raw residual stream in, the norm's gain in the weights,
    (mu, r) per token from the producer's partial statistics - against LayerNorm + Attention.forward in fp64."""
    g = torch.Generator().manual_seed(300 + n)
    x = torch.randn(B * n, 384, generator=g) * 1.5 + torch.randn(B * n, 1, generator=g) * 0.5
    gamma, beta = torch.rand(384, generator=g) + 0.5, torch.randn(384, generator=g) * 0.2
    w = torch.randn(1152, 384, generator=g) * 0.05
    b = torch.randn(1152, generator=g) * 0.1
    xd = x.double()
    xn = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-6) * gamma.double() + beta.double()
    ref = _ref(xn, w, b, B, 0.125)
    seg = x.reshape(B * n, 12, 32)
    mean = seg.mean(2)
    stats = torch.stack([mean, ((seg - mean[..., None]) ** 2).sum(2)], dim=2).contiguous()   # what the residual GEMM emits
    got = ops.qkv_attention(x.to(DEV), w.to(DEV), b.to(DEV), B, 0.125, ln=(gamma.to(DEV), beta.to(DEV), 1e-6, stats.to(DEV)))
    err = (got.double().cpu() - ref).abs().max().item()
    plain = ops.qkv_attention(xn.float().to(DEV), w.to(DEV), b.to(DEV), B, 0.125).double().cpu()
    err_plain = (plain - ref).abs().max().item()
    print(f"\nfolded LN, B={B} N={n}: folded-fp64 {err:.2e}  LayerNorm-first-fp64 {err_plain:.2e}  max|ref| {ref.abs().max():.2f}")
    assert err <= 4e-6 * max(1.0, ref.abs().max().item())
    assert err <= 3.0 * err_plain + 1e-6
