"""Per-operator Python entry points over the C ABI (torch tensors in, torch tensors out, current HIP stream).

PyTorch is plumbing here: device memory + streams.  Every function launches hand-written gfx950 kernels from
libselfmask_hip.so and raises if handed a CPU tensor (there is no fallback path).
"""
from typing import Optional, Tuple

import torch

from . import _native as N


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise RuntimeError("selfmask_amd ops need float32 tensors on a HIP device (no CPU fallback)")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = N.EPI_BIAS,
         residual: Optional[torch.Tensor] = None,
         tile: Optional[Tuple[int, int]] = None, out: Optional[torch.Tensor] = None,
         out2: Optional[torch.Tensor] = None, a_alt: Optional[torch.Tensor] = None, alt_from_n: int = 0,
         split_k: int = 1) -> torch.Tensor:
    """C = epilogue(A W^T + bias).  a: (M,K) or (batch,M,K); w: (N,K) or (batch,N,K) (torch Linear layout)."""
    _dev(a, w, bias, residual)
    lib = N.load()
    a3 = a if a.dim() == 3 else a.unsqueeze(0)
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    assert a3.stride(-1) == 1 and w3.stride(-1) == 1
    batch = max(a3.shape[0], w3.shape[0])
    M, K = a3.shape[1], a3.shape[2]
    Nn = w3.shape[1]
    c = out if out is not None else torch.empty((max(batch, split_k), M, Nn), device=a.device, dtype=torch.float32)
    c3 = c if c.dim() == 3 else c.unsqueeze(0)
    g = N.GemmArgs()
    g.A, g.W, g.bias, g.C = a3.data_ptr(), w3.data_ptr(), _ptr(bias), c3.data_ptr()
    g.strideA = a3.stride(0) if a3.shape[0] > 1 else 0
    g.strideW = w3.stride(0) if w3.shape[0] > 1 else 0
    g.strideC = c3.stride(0)
    g.M, g.N, g.K = M, Nn, K
    g.lda, g.ldw, g.ldc = a3.stride(1), w3.stride(1), c3.stride(1)
    g.batch, g.epilogue = batch, epilogue
    if split_k > 1:  # slice s of the K range lands in c[s] (raw partial products; the caller sums the slices)
        assert batch == 1 and c3.shape[0] >= split_k
        g.split_k = split_k
    if a_alt is not None:
        g.A_alt, g.alt_from_n = a_alt.data_ptr(), alt_from_n
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        g.R, g.ldr = r3.data_ptr(), r3.stride(1)
        g.strideR = r3.stride(0) if r3.shape[0] > 1 else 0
    if epilogue == N.EPI_SIGMOID2:
        c2 = out2 if out2 is not None else torch.empty_like(c)
        g.C2 = c2.data_ptr()
    if tile is None:
        N.check(lib.sm_gemm_f32(g, _stream()), "sm_gemm_f32")
    else:
        N.check(lib.sm_gemm_f32_tile(g, tile[0], tile[1], _stream()), "sm_gemm_f32_tile")
    res = c if a.dim() == 3 or w.dim() == 3 or out is not None or split_k > 1 else c[0]
    if epilogue == N.EPI_SIGMOID2:
        return res, (c2 if c2.dim() == res.dim() else c2[0])
    return res


def split_f16x2(x: torch.Tensor) -> torch.Tensor:
    """fp32 (..., K) -> F16X2 (same shape / dtype container: 4 B per element, hi/lo f16 halves per group of 8)."""
    _dev(x)
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.stride(1) == 1
    out = torch.empty((x2.shape[0], x2.shape[1]), device=x.device, dtype=torch.float32)
    N.check(N.load().sm_split_f16x2(x2.data_ptr(), x2.stride(0), out.data_ptr(), out.stride(0), x2.shape[0], x2.shape[1],
                                    _stream()), "sm_split_f16x2")
    return out.view(x.shape)


def unsplit_f16x2(t: torch.Tensor) -> torch.Tensor:
    """F16X2 (..., K) -> the fp32 values it stands for (hi + lo / 2048); inverse of split_f16x2 up to the format's 22 bits."""
    k = t.shape[-1]
    h = t.contiguous().view(torch.float16).reshape(*t.shape[:-1], k // 8, 2, 8).float()
    return (h[..., 0, :] + h[..., 1, :] / 2048.0).reshape(t.shape)


def gemm_f16x2(a_split: torch.Tensor, w_split: torch.Tensor, bias=None, epilogue: int = N.EPI_BIAS, residual=None,
               tile=(128, 128), out=None, out_f16x2: bool = False, split_k: int = 1, ln=None):
    """C = epilogue(A W^T + bias) with A, W in F16X2 format (see split_f16x2).  2-D operands, or 3-D (batch, rows, K)
    for a batched launch (no split_k then).  Test / tuning entry: the forward drives the kernel from C."""
    _dev(a_split, w_split, bias, residual)
    batched = a_split.dim() == 3
    assert not (batched and split_k > 1)
    a3 = a_split if batched else a_split.unsqueeze(0)
    w3 = w_split if batched else w_split.unsqueeze(0)
    nb, M, K = a3.shape
    Nn = w3.shape[1]
    c = out if out is not None else torch.empty((max(nb, split_k), M, Nn), device=a_split.device, dtype=torch.float32)
    c3 = c if c.dim() == 3 else c.unsqueeze(0)
    g = N.GemmArgs()
    g.A, g.W, g.bias, g.C = a3.data_ptr(), w3.data_ptr(), _ptr(bias), c3.data_ptr()
    g.strideA, g.strideW, g.strideC = a3.stride(0), w3.stride(0), c3.stride(0)
    g.M, g.N, g.K = M, Nn, K
    g.lda, g.ldw, g.ldc = a3.stride(1), w3.stride(1), c3.stride(1)
    g.batch, g.epilogue, g.split_k = nb, epilogue, split_k if split_k > 1 else 0
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        g.R, g.ldr, g.strideR = r3.data_ptr(), r3.stride(1), r3.stride(0)
    xn = None
    if ln is not None:  # (gamma, beta, eps): C = R + A W^T + bias and C2 = LayerNorm(C) in F16X2 (64x384 tile, N = 384)
        gamma, beta, eps = ln
        xn = torch.empty_like(c3)
        g.epilogue, g.C2, g.ln_gamma, g.ln_beta, g.ln_eps = N.EPI_RESIDUAL_LN, xn.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps
    N.check(N.load().sm_gemm_f16x2_tile(g, 1 if out_f16x2 else 0, tile[0], tile[1], _stream()), "sm_gemm_f16x2_tile")
    res = c if (out is not None or split_k > 1 or batched) else c[0]
    return (res, xn if batched else xn[0]) if ln is not None else res


def w16_scale_exponent(w: torch.Tensor) -> int:
    """s such that max|w| * 2^s lies in [2^13, 2^14): the per-tensor scaling of the W16 weight format."""
    import math
    m = float(w.detach().abs().max())
    if not math.isfinite(m):
        raise ValueError("weight tensor holds non-finite values")
    if m == 0.0:
        return 0
    return 13 - math.floor(math.log2(m))


def split_w16(w: torch.Tensor):
    """fp32 weight (rows, K) -> (W16 tensor, w_scale = 2^-s): scaled hi / UNSCALED lo halves (gemm_w16.hip)."""
    _dev(w)
    w2 = w.reshape(w.shape[0], -1).contiguous()
    s = w16_scale_exponent(w2)
    out = torch.empty_like(w2)
    N.check(N.load().sm_split_w16(w2.data_ptr(), w2.stride(0), out.data_ptr(), out.stride(0), w2.shape[0], w2.shape[1],
                                  float(2.0 ** s), _stream()), "sm_split_w16")
    return out, float(2.0 ** -s)


def gemm_w16(a_split: torch.Tensor, w16: torch.Tensor, w_scale: float, bias=None, epilogue: int = N.EPI_BIAS, residual=None,
             variant: Optional[int] = None, out=None, out_f16x2: bool = False, split_k: int = 1, a_alt=None, alt_from_n: int = 0,
             patch_n: int = 0, xs_out=None, stats_out=None, ln_stats=None, ln_c=None, ln_eps: float = 0.0):
    """C = epilogue(A W^T + bias), A in F16X2, W in W16 (split_w16); single-accumulator kernel.  variant None = the
    library's pick for the shape.  Test / tuning entry: the forward drives the kernel from C."""
    _dev(a_split, w16, bias, residual, a_alt)
    M, K = a_split.shape
    Nn = w16.shape[0]
    c = out if out is not None else torch.empty((max(1, split_k), M, Nn), device=a_split.device, dtype=torch.float32)
    c3 = c if c.dim() == 3 else c.unsqueeze(0)
    g = N.GemmArgs()
    g.A, g.W, g.bias, g.C = a_split.data_ptr(), w16.data_ptr(), _ptr(bias), c3.data_ptr()
    g.strideC = c3.stride(0)
    g.M, g.N, g.K = M, Nn, K
    g.lda, g.ldw, g.ldc = a_split.stride(0), w16.stride(0), c3.stride(1)
    g.batch, g.epilogue, g.split_k, g.w_scale = 1, epilogue, split_k if split_k > 1 else 0, w_scale
    if a_alt is not None:
        g.A_alt, g.alt_from_n = a_alt.data_ptr(), alt_from_n
    if residual is not None:
        g.R, g.ldr = residual.data_ptr(), residual.stride(0)
    if epilogue == N.EPI_PATCH:
        g.patch_n = patch_n
    # LayerNorm folded into the GEMMs around it: producer outputs (F16X2 copy + row statistics) / consumer inputs (see the header)
    g.C2, g.ln_stats_out, g.ln_stats, g.ln_c, g.ln_eps = _ptr(xs_out), _ptr(stats_out), _ptr(ln_stats), _ptr(ln_c), ln_eps
    lib = N.load()
    if variant is None:
        N.check(lib.sm_gemm_w16(g, 1 if out_f16x2 else 0, _stream()), "sm_gemm_w16")
    else:
        N.check(lib.sm_gemm_w16_tile(g, 1 if out_f16x2 else 0, variant, _stream()), "sm_gemm_w16_tile")
    return c if (out is not None or split_k > 1) else c[0]


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
              add: Optional[torch.Tensor] = None, in_map=(0, 0, 0), out_map=(0, 0, 0), rows: Optional[int] = None,
              out_rows: Optional[int] = None):
    """y = LayerNorm(x) (+ optional y2 = y + add[r % len(add)]); optional grouped row remaps (see the header)."""
    _dev(x, gamma, beta, add)
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.stride(1) == 1 and x2.shape[1] == N.EMBED
    rows = x2.shape[0] if rows is None else rows
    y = torch.empty((rows if out_rows is None else out_rows, N.EMBED), device=x.device, dtype=torch.float32)
    a = N.LnArgs()
    a.x, a.ldx, a.gamma, a.beta, a.y, a.ldy = x2.data_ptr(), x2.stride(0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), N.EMBED
    a.in_map, a.out_map = N.RowMap(*in_map), N.RowMap(*out_map)
    a.rows, a.eps = rows, eps
    y2 = None
    if add is not None:
        add = add.contiguous()
        y2 = torch.empty((rows, N.EMBED), device=x.device, dtype=torch.float32)
        a.y2, a.ldy2, a.add, a.add_rows = y2.data_ptr(), N.EMBED, add.data_ptr(), add.shape[0]
    N.check(N.load().sm_layernorm_rows_f32(a, _stream()), "sm_layernorm_rows_f32")
    if in_map == (0, 0, 0) and out_map == (0, 0, 0):
        y = y.view(x.shape)
    return (y, y2) if add is not None else y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float = 0.125, split: bool = False) -> torch.Tensor:
    """q (B,Nq,H,64), k/v (B,Nk,H,64) views (last two dims contiguous) -> (B,Nq,H*64).
    split=True: the operands are first converted to F16X2 rows and sm_attention_f16x2 (f16 matrix cores) runs."""
    _dev(q, k, v)
    B, nq, H, dh = q.shape
    nk = k.shape[1]
    if split:  # F16X2 images of the (B, N, H*64) rows; the (B,N,H,64) view keeps float-unit strides
        q, k, v = (split_f16x2(t.reshape(t.shape[0], t.shape[1], H * dh).contiguous()).view(t.shape[0], t.shape[1], H, dh)
                   for t in (q, k, v))
    assert dh == 64 and q.stride(3) == 1 and q.stride(2) == 64 and k.stride(2) == 64 and v.stride(2) == 64
    o = torch.empty((B, nq, H * dh), device=q.device, dtype=torch.float32)
    a = N.AttnArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    a.sQb, a.sQr, a.sKb, a.sKr, a.sVb, a.sVr = q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1)
    a.sOb, a.sOr = o.stride(0), o.stride(1)
    a.batch, a.heads, a.n_q, a.n_k, a.scale = B, H, nq, nk, scale
    if split:
        N.check(N.load().sm_attention_f16x2(a, _stream()), "sm_attention_f16x2")
    else:
        N.check(N.load().sm_attention_f32(a, _stream()), "sm_attention_f32")
    return o


def fold_layernorm(weight: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm(gamma, beta) folded into the Linear(weight, bias) it feeds -> (W16 tensor of weight * gamma, its 2^-s, folded
    bias b + W beta, row sums c of the gain-scaled weight as rounded to W16): LN(x) W^T + b = r (x W'^T - mu c) + b'."""
    w16, ws = split_w16((weight * gamma[None, :]).contiguous())
    rows, K = weight.shape
    h = w16.view(torch.float16).reshape(rows, K // 8, 2, 8).double()
    c = ((h[:, :, 0, :] + h[:, :, 1, :]).reshape(rows, K).sum(1) * ws).float().contiguous()
    # (an elementwise product + row sum, not `@`: weight packing stays off the vendor BLAS - one-time per checkpoint, not timed)
    b2 = (bias.double() + (weight.double() * beta.double()[None, :]).sum(1)).float().contiguous()
    return w16, ws, b2, c


def qkv_attention(xn: torch.Tensor, w_qkv: torch.Tensor, b_qkv: torch.Tensor, B: int, scale: float = 0.125,
                  out_f16x2: bool = False, ln=None) -> torch.Tensor:
    """Fused qkv Linear + softmax attention of an encoder block (sm_qkv_attention_w16): xn (B*N, 384) fp32 LayerNorm output
    (converted to F16X2 here), w_qkv (1152, 384) / b_qkv (1152) fp32 -> (B*N, 384) merged-head attention output."""
    _dev(xn, w_qkv, b_qkv)
    M = xn.shape[0]
    assert M % B == 0 and xn.shape[1] == N.EMBED
    xs = split_f16x2(xn.contiguous())
    o = torch.empty((M, N.EMBED), device=xn.device, dtype=torch.float32)
    a = N.QkvAttnArgs()
    if ln is not None:  # (gamma, beta, eps, stats): xn is then the RAW stream whose LayerNorm is folded into the projection
        gamma, beta, eps, stats = ln
        w16, ws, b2, cvec = fold_layernorm(w_qkv, b_qkv, gamma, beta)
        a.ln_stats, a.ln_c, a.ln_eps = stats.data_ptr(), cvec.data_ptr(), eps
        b_qkv = b2
    else:
        w16, ws = split_w16(w_qkv)
    a.Xn, a.Wqkv, a.bias, a.O = xs.data_ptr(), w16.data_ptr(), b_qkv.data_ptr(), o.data_ptr()
    a.ldx, a.ldo, a.B, a.N, a.w_scale, a.scale, a.out_f16x2 = N.EMBED, N.EMBED, B, M // B, ws, scale, 1 if out_f16x2 else 0
    N.check(N.load().sm_qkv_attention_w16(a, _stream()), "sm_qkv_attention_w16")
    return o


def im2col_patches(img: torch.Tensor, patch: int) -> torch.Tensor:
    _dev(img)
    img = img.contiguous()
    B, _, H, W = img.shape
    gh, gw = -(-H // patch), -(-W // patch)
    cols = torch.empty((B * gh * gw, 3 * patch * patch), device=img.device, dtype=torch.float32)
    N.check(N.load().sm_im2col_patches_f32(img.data_ptr(), cols.data_ptr(), B, H, W, patch, _stream()), "sm_im2col")
    return cols


def pos_embed_bicubic(pos: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    _dev(pos)
    pos = pos.reshape(-1, N.EMBED).contiguous()
    g0 = int(round((pos.shape[0] - 1) ** 0.5))
    out = torch.empty((1 + gh * gw, N.EMBED), device=pos.device, dtype=torch.float32)
    N.check(N.load().sm_pos_embed_bicubic_f32(pos.data_ptr(), g0, out.data_ptr(), gh, gw, _stream()), "sm_pos_bicubic")
    return out


def upsample2x_tokens(tok: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """tok (B, gh*gw, 384) -> (B, 4*gh*gw, 384), channels-last bilinear x2."""
    _dev(tok)
    assert tok.stride(2) == 1 and tok.stride(1) == N.EMBED
    B = tok.shape[0]
    up = torch.empty((B, 4 * gh * gw, N.EMBED), device=tok.device, dtype=torch.float32)
    N.check(N.load().sm_upsample2x_tokens_f32(tok.data_ptr(), tok.stride(0), up.data_ptr(), B, gh, gw, _stream()),
            "sm_upsample2x")
    return up


def rowdot_sigmoid(h: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _dev(h, w, b)
    h2 = h.reshape(-1, N.EMBED).contiguous()
    out = torch.empty((h2.shape[0],), device=h.device, dtype=torch.float32)
    N.check(N.load().sm_rowdot_sigmoid_f32(h2.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), h2.shape[0],
                                           _stream()), "sm_rowdot_sigmoid")
    return out


_THRESHOLDS = {}


def f_max_thresholds(device) -> torch.Tensor:
    """The 255 strict thresholds of metrics/f_measure.py:65 as the float32 values torch.arange produces (constant)."""
    t = _THRESHOLDS.get(device)
    if t is None:
        t = torch.arange(0, 1, 1 / 255).to(device)
        assert t.numel() == 255 and t.dtype == torch.float32
        _THRESHOLDS[device] = t
    return t


class GtBatch:
    """Ground-truth masks of one batch packed for sm_evaluate_masks_f32: one uint8 buffer + a device descriptor
    array.  Build it once per batch (the evaluator's data loader side), reuse it across calls."""

    def __init__(self, gts, device):
        B = len(gts)
        descr = (N.EvalImage * B)()
        off, flat = 0, []
        for b, g in enumerate(gts):
            if g.dtype != torch.uint8 or g.dim() != 2:
                raise RuntimeError("ground-truth masks must be 2-D uint8 tensors")
            descr[b].gt_off, descr[b].H, descr[b].W = off, g.shape[0], g.shape[1]
            off += g.numel()
            flat.append(g.reshape(-1))
        self.B = B
        self.shapes = [(int(g.shape[0]), int(g.shape[1])) for g in gts]
        self.gt_all = (flat[0] if B == 1 else torch.cat(flat)).to(device)
        self.images = torch.frombuffer(bytearray(bytes(descr)), dtype=torch.uint8).to(device)


def _gtbatch_from_packed(packed, device) -> "GtBatch":
    """GtBatch from pipeline.pack_gts' page-locked buffers: two asynchronous H2D copies on the current stream."""
    from .pipeline import _POOL
    flat, descr, shapes = packed
    gb = GtBatch.__new__(GtBatch)
    gb.B, gb.shapes = len(shapes), shapes
    gb.gt_all = flat.to(device, non_blocking=True)
    gb.images = descr.to(device, non_blocking=True)
    _POOL.release_after((flat, descr), torch.cuda.current_stream(device))
    return gb


GtBatch.from_packed = staticmethod(_gtbatch_from_packed)


def evaluate_masks(mask_pred_last: torch.Tensor, objectness_last: torch.Tensor, gts, scale: float = 0.0,
                   return_ious: bool = False):
    """Evaluator post-processing + 14 metrics per image on the device (sm_evaluate_masks_f32).

    mask_pred_last (B, nq, mh, mw) probabilities (any batch stride, e.g. ``out["mask_pred"][:, -1]``),
    objectness_last (B, nq), gts: list of B uint8 {0,1} tensors (H_b, W_b) or a prepacked GtBatch.
    scale > 0: reference mode (F.interpolate(scale_factor=scale)[..., :H, :W]); 0: resize to each GT's size.
    Returns rows (B, 16) float32 [7 metrics of the picked mask, 7 of the upper bound, q*, ub] (+ ious (B, nq))."""
    _dev(mask_pred_last, objectness_last)
    B, nq, mh, mw = mask_pred_last.shape
    assert mask_pred_last.stride(3) == 1 and mask_pred_last.stride(2) == mw and mask_pred_last.stride(1) == mh * mw
    assert objectness_last.stride(1) == 1
    dev = mask_pred_last.device
    gb = gts if isinstance(gts, GtBatch) else GtBatch(gts, dev)
    assert gb.B == B
    if scale > 0:
        for (h, w) in gb.shapes:
            assert h <= int(mh * scale) and w <= int(mw * scale), "GT larger than the up-sampled mask"
    rows = torch.empty((B, 16), device=dev, dtype=torch.float32)
    ious = torch.empty((B, nq), device=dev, dtype=torch.float32) if return_ious else None
    lib = N.load()
    max_pixels = max(h * w for (h, w) in gb.shapes)
    wsb = lib.sm_evaluate_workspace_bytes(B, nq, mh, mw, max_pixels)
    if wsb == 0:
        raise ValueError("unsupported evaluate_masks shape")
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    a = N.EvalArgs()
    a.mask_pred, a.mask_stride_b = mask_pred_last.data_ptr(), mask_pred_last.stride(0)
    a.objectness, a.obj_stride_b = objectness_last.data_ptr(), objectness_last.stride(0)
    a.gt, a.images, a.thresholds = gb.gt_all.data_ptr(), gb.images.data_ptr(), f_max_thresholds(dev).data_ptr()
    a.rows, a.ious, a.workspace, a.workspace_bytes = rows.data_ptr(), _ptr(ious), ws.data_ptr(), wsb
    a.B, a.nq, a.mh, a.mw, a.scale = B, nq, mh, mw, float(scale)
    a.max_pixels = max_pixels
    N.check(lib.sm_evaluate_masks_f32(a, _stream()), "sm_evaluate_masks_f32")
    return (rows, ious) if return_ious else rows


def upsample_selected(mask_pred_last: torch.Tensor, rows: torch.Tensor, size, which: str = "pick") -> torch.Tensor:
    """(B, nq, mh, mw) probabilities + the (B, 16) rows of evaluate_masks -> (B, OH, OW) float64: the picked ("pick") or
    upper-bound ("ub") query's mask up-sampled bilinearly to ``size`` - the bilateral solver's target."""
    _dev(mask_pred_last, rows)
    B, nq, mh, mw = mask_pred_last.shape
    assert mask_pred_last.stride(3) == 1 and mask_pred_last.stride(2) == mw and mask_pred_last.stride(1) == mh * mw
    assert rows.shape == (B, 16) and rows.is_contiguous()
    out = torch.empty((B, size[0], size[1]), dtype=torch.float64, device=mask_pred_last.device)
    N.check(N.load().sm_upsample_selected_f64(mask_pred_last.data_ptr(), mask_pred_last.stride(0), rows.data_ptr(),
                                              14 if which == "pick" else 15, out.data_ptr(), B, mh, mw, size[0], size[1],
                                              _stream()), "sm_upsample_selected_f64")
    return out


def mask_u8_to_f32(m: torch.Tensor) -> torch.Tensor:
    if not m.is_cuda or m.dtype != torch.uint8:
        raise RuntimeError("mask_u8_to_f32 takes a uint8 tensor on a HIP device")
    m = m.contiguous()
    out = torch.empty(m.shape, dtype=torch.float32, device=m.device)
    N.check(N.load().sm_mask_u8_to_f32(m.data_ptr(), out.data_ptr(), m.numel(), _stream()), "sm_mask_u8_to_f32")
    return out
