// HBM-bound helper kernels of the SelfMask path: im2col for the patch embedding, cls rows, bicubic position-grid
// resize, bilinear x2 of the token grid, objectness row-dot + sigmoid, query mean, thread-local error text.
#include "common.h"
#include <math.h>
#include <string.h>

namespace sm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- im2col: cols[(b,gy,gx)][(c,i,j)] = img[b][c][gy*P+i][gx*P+j], zero beyond H/W ------------------------------
// one thread per 4 consecutive j (16 B of one image row); consecutive threads walk k fastest so the cols rows are
// written as full contiguous lines; image reads are P*4-byte runs.
// (the patch size is a template parameter and the indices are 32-bit: the first version's four 64-bit divisions per float4 cost
// more than the copy)
template <bool SPLIT, int P>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, float* __restrict__ cols, int B,
                                                     int H, int W, int gh, int gw, unsigned total4) {
    constexpr int K = 3 * P * P, K4 = K / 4;
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total4; t += gridDim.x * 256u) {
        const unsigned m = t / K4;
        const int k4 = (int)(t - m * K4);
        const unsigned row = m / (unsigned)gw;  // (b, gy)
        const int gx = (int)(m - row * gw), b = (int)(row / (unsigned)gh), gy = (int)(row - (unsigned)b * gh);
        const int k = k4 * 4, c = k / (P * P), i = (k / P) % P, j = k % P;
        const int y = gy * P + i, x = gx * P + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < H) {
            const float* src = img + (((int64_t)b * 3 + c) * H + y) * W + x;
            if (x + 3 < W && (((uintptr_t)src) & 15) == 0) {
                v = *reinterpret_cast<const float4*>(src);
            } else {
                if (x < W) v.x = src[0];
                if (x + 1 < W) v.y = src[1];
                if (x + 2 < W) v.z = src[2];
                if (x + 3 < W) v.w = src[3];
            }
        }
        if constexpr (SPLIT) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            store_f16x2_4(cols + (int64_t)m * K, k, vv);
        } else {
            *reinterpret_cast<float4*>(cols + (int64_t)m * K + k) = v;
        }
    }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* tokens, int B,
                                int N) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < SM_EMBED; c += blockDim.x) tokens[(int64_t)b * N * SM_EMBED + c] = cls[c] + pos[c];
}

// ---- bicubic (Keys, A = -0.75, align_corners = False), separable: along x for the 4 tap rows, then along y -------
__device__ __forceinline__ float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ __launch_bounds__(128) void pos_bicubic_kernel(const float* __restrict__ pin, int g0, float* __restrict__ pout,
                                                          int gh, int gw) {
    const int p = blockIdx.x;  // output token (0 = cls)
    if (p == 0) {
        for (int c = threadIdx.x; c < SM_EMBED; c += blockDim.x) pout[c] = pin[c];
        return;
    }
    const int oy = (p - 1) / gw, ox = (p - 1) % gw;
    const float sy = (float)g0 / (float)gh, sx = (float)g0 / (float)gw;
    const float ry = sy * (oy + 0.5f) - 0.5f, rx = sx * (ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    const float ty = ry - fy, tx = rx - fx;
    const float wy[4] = {cubic2(ty + 1.f), cubic1(ty), cubic1(1.f - ty), cubic2(2.f - ty)};
    const float wx[4] = {cubic2(tx + 1.f), cubic1(tx), cubic1(1.f - tx), cubic2(2.f - tx)};
    const float* grid = pin + SM_EMBED;
    for (int c = threadIdx.x; c < SM_EMBED; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), g0 - 1);
            float rowv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), g0 - 1);
                rowv += wx[j] * grid[((int64_t)yy * g0 + xx) * SM_EMBED + c];
            }
            acc += wy[i] * rowv;
        }
        pout[(int64_t)p * SM_EMBED + c] = acc;
    }
}

// ---- bilinear x sf (the model's scale_factor: 2 as shipped), align_corners=False, channels-last: up[b][(oy,ox)][c] -------
// source index as F.interpolate(scale_factor=sf) computes it (ATen area_pixel_compute_source_index with the scale 1 / sf
// rounded to fp32): max(inv * (o + 0.5) - 0.5, 0); sf = 2 gives the constants of round 1 bit for bit, sf = 1 the identity
template <bool SPLIT>
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ tok, int64_t strideb,
                                                         float* __restrict__ up, int gh, int gw, int64_t total4, int sf, float inv) {
    const int oh = sf * gh, ow = sf * gw, C4 = SM_EMBED / 4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % C4) * 4;
        const int64_t px = t / C4;
        const int ox = (int)(px % ow), oy = (int)((px / ow) % oh), b = (int)(px / ((int64_t)ow * oh));
        float syf = inv * (oy + 0.5f) - 0.5f, sxf = inv * (ox + 0.5f) - 0.5f;
        syf = syf < 0.f ? 0.f : syf;
        sxf = sxf < 0.f ? 0.f : sxf;
        const int y0 = (int)syf, x0 = (int)sxf;
        const int y1 = y0 + (y0 < gh - 1 ? 1 : 0), x1 = x0 + (x0 < gw - 1 ? 1 : 0);
        const float ly1 = syf - y0, ly0 = 1.f - ly1, lx1 = sxf - x0, lx0 = 1.f - lx1;
        const float* base = tok + (int64_t)b * strideb + c;
        const float4 p00 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * gw + x0) * SM_EMBED);
        const float4 p01 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * gw + x1) * SM_EMBED);
        const float4 p10 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * gw + x0) * SM_EMBED);
        const float4 p11 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * gw + x1) * SM_EMBED);
        float4 o;
        o.x = ly0 * (lx0 * p00.x + lx1 * p01.x) + ly1 * (lx0 * p10.x + lx1 * p11.x);
        o.y = ly0 * (lx0 * p00.y + lx1 * p01.y) + ly1 * (lx0 * p10.y + lx1 * p11.y);
        o.z = ly0 * (lx0 * p00.z + lx1 * p01.z) + ly1 * (lx0 * p10.z + lx1 * p11.z);
        o.w = ly0 * (lx0 * p00.w + lx1 * p01.w) + ly1 * (lx0 * p10.w + lx1 * p11.w);
        if constexpr (SPLIT) {
            const float vv[4] = {o.x, o.y, o.z, o.w};
            store_f16x2_4(up + px * SM_EMBED, c, vv);
        } else {
            *reinterpret_cast<float4*>(up + px * SM_EMBED + c) = o;
        }
    }
}

// ---- bilinear x2 (align_corners=False) of low-resolution mask logits + sigmoid ---------------------------------------
// The mask einsum is linear in the up-sampled features and the up-sampling is linear in the tokens, so
// einsum(Q, up(tok)) = up(einsum(Q, tok)): the GEMM runs on the gh x gw grid (4x fewer FLOPs, no 4n x 384 feature
// map in HBM) and this kernel up-samples its (B, R, gh, gw) output with the same taps and weights as upsample2x_kernel.
// One thread per output pixel of a GROUP of planes (blockIdx.y): the taps and weights of the pixel are computed once and used
// for every plane of the group - the index arithmetic (the first version paid three 64-bit divisions per output: 28.6 us for
// the 24 MB of a batch of 64) is off the per-element path, stores stay coalesced along x.
constexpr int UPL_PLANES = 16;
__global__ __launch_bounds__(256) void upsample2x_logits_kernel(const float* __restrict__ low, float* __restrict__ logits,
                                                                float* __restrict__ prob, int gh, int gw, int64_t planes, int sf, float inv) {
    const int oh = sf * gh, ow = sf * gw, opix = oh * ow, lpix = gh * gw;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= opix) return;
    const int oy = i / ow, ox = i - oy * ow;
    float syf = inv * (oy + 0.5f) - 0.5f, sxf = inv * (ox + 0.5f) - 0.5f;
    syf = syf < 0.f ? 0.f : syf;
    sxf = sxf < 0.f ? 0.f : sxf;
    const int y0 = (int)syf, x0 = (int)sxf;
    const int y1 = y0 + (y0 < gh - 1 ? 1 : 0), x1 = x0 + (x0 < gw - 1 ? 1 : 0);
    const float ly1 = syf - y0, ly0 = 1.f - ly1, lx1 = sxf - x0, lx0 = 1.f - lx1;
    const int o00 = y0 * gw + x0, o01 = y0 * gw + x1, o10 = y1 * gw + x0, o11 = y1 * gw + x1;
    const int64_t p0 = (int64_t)blockIdx.y * UPL_PLANES;
    const int np = planes - p0 < UPL_PLANES ? (int)(planes - p0) : UPL_PLANES;
    const float* base = low + p0 * lpix;
    float* lo = logits ? logits + p0 * opix + i : nullptr;
    float* pr = prob + p0 * opix + i;
#pragma unroll 4
    for (int k = 0; k < np; ++k, base += lpix, pr += opix) {
        const float p00 = base[o00], p01 = base[o01], p10 = base[o10], p11 = base[o11];
        const float o = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
        if (lo) { *lo = o; lo += opix; }
        *pr = 1.0f / (1.0f + expf(-o));
    }
}

// ---- out[row] = sigmoid(h[row] . w + b): one wave per row ---------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_sigmoid_kernel(const float* __restrict__ hbuf, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* hr = hbuf + (int64_t)row * SM_EMBED;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float2 a = *reinterpret_cast<const float2*>(hr + i * 128 + lane * 2);
        const float2 ww = *reinterpret_cast<const float2*>(w + i * 128 + lane * 2);
        s += a.x * ww.x + a.y * ww.y;
    }
    s = wave_sum(s) + bias[0];
    if (lane == 0) out[row] = 1.0f / (1.0f + expf(-s));
}

__global__ void query_mean_kernel(const float* __restrict__ q, float* __restrict__ f, int L, int nq) {
    // one thread per (image, channel); the loads of a group of 8 queries are independent of the running sum, so they
    // are all in flight before the (in-order) adds
    const int b = blockIdx.x / (SM_EMBED / 128), c = (blockIdx.x % (SM_EMBED / 128)) * 128 + threadIdx.x;
    const float* src = q + (((int64_t)b * L + (L - 1)) * nq) * SM_EMBED + c;
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= nq; i += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(i + j) * SM_EMBED];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; i < nq; ++i) s += src[(int64_t)i * SM_EMBED];
    f[(int64_t)b * SM_EMBED + c] = s / (float)nq;
}

static inline int grid_for(int64_t work, int per_block = 256) {
    int64_t g = (work + per_block - 1) / per_block;
    return (int)(g > 256 * 8 ? 256 * 8 : (g < 1 ? 1 : g));  // cap at ~8 blocks per CU, grid-stride the rest
}

}  // namespace sm

extern "C" int sm_version(void) { return 100; }
extern "C" const char* sm_last_error(void) { return sm::g_err; }

static int im2col_impl(const float* img, float* cols, int32_t B, int32_t H, int32_t W, int32_t P, void* stream, bool split) {
    SM_REQUIRE(img && cols, "sm_im2col_patches_f32: null pointer");
    SM_REQUIRE(B > 0 && H > 0 && W > 0 && (P == 8 || P == 16), "sm_im2col_patches_f32: bad shape B=%d H=%d W=%d P=%d", B,
               H, W, P);
    const int gh = (H + P - 1) / P, gw = (W + P - 1) / P;
    const int64_t total4 = (int64_t)B * gh * gw * (3 * P * P / 4);
    SM_REQUIRE(total4 < ((int64_t)1 << 31), "sm_im2col_patches_f32: B=%d H=%d W=%d is beyond the kernel's 32-bit element index", B, H, W);
    const dim3 grid(sm::grid_for(total4));
    const hipStream_t st = (hipStream_t)stream;
    const unsigned n = (unsigned)total4;
    if (split && P == 16) hipLaunchKernelGGL((sm::im2col_kernel<true, 16>), grid, dim3(256), 0, st, img, cols, B, H, W, gh, gw, n);
    else if (split) hipLaunchKernelGGL((sm::im2col_kernel<true, 8>), grid, dim3(256), 0, st, img, cols, B, H, W, gh, gw, n);
    else if (P == 16) hipLaunchKernelGGL((sm::im2col_kernel<false, 16>), grid, dim3(256), 0, st, img, cols, B, H, W, gh, gw, n);
    else hipLaunchKernelGGL((sm::im2col_kernel<false, 8>), grid, dim3(256), 0, st, img, cols, B, H, W, gh, gw, n);
    return sm::check_launch("sm_im2col_patches_f32");
}
extern "C" int sm_im2col_patches_f32(const float* img, float* cols, int32_t B, int32_t H, int32_t W, int32_t P, void* stream) {
    return im2col_impl(img, cols, B, H, W, P, stream, false);
}
extern "C" int sm_im2col_patches_f16x2(const float* img, float* cols, int32_t B, int32_t H, int32_t W, int32_t P, void* stream) {
    return im2col_impl(img, cols, B, H, W, P, stream, true);
}

extern "C" int sm_cls_rows_f32(const float* cls, const float* pos, float* tokens, int32_t B, int32_t N, void* stream) {
    SM_REQUIRE(cls && pos && tokens && B > 0 && N > 0, "sm_cls_rows_f32: bad arguments");
    hipLaunchKernelGGL(sm::cls_rows_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, cls, pos, tokens, B, N);
    return sm::check_launch("sm_cls_rows_f32");
}

extern "C" int sm_pos_embed_bicubic_f32(const float* pos_in, int32_t g0, float* pos_out, int32_t gh, int32_t gw,
                                        void* stream) {
    SM_REQUIRE(pos_in && pos_out && g0 > 0 && gh > 0 && gw > 0, "sm_pos_embed_bicubic_f32: bad arguments");
    hipLaunchKernelGGL(sm::pos_bicubic_kernel, dim3(1 + gh * gw), dim3(128), 0, (hipStream_t)stream, pos_in, g0, pos_out,
                       gh, gw);
    return sm::check_launch("sm_pos_embed_bicubic_f32");
}

static int upsample_impl(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw, void* stream,
                         bool split, int sf = 2) {
    SM_REQUIRE(tok && up && B > 0 && gh > 0 && gw > 0 && strideb % 4 == 0 && sf >= 1 && sf <= 16, "sm_upsample_tokens: bad arguments");
    const int64_t total4 = (int64_t)B * sf * sf * gh * gw * (SM_EMBED / 4);
    const float inv = (float)(1.0 / sf);
    if (split)
        hipLaunchKernelGGL(sm::upsample2x_kernel<true>, dim3(sm::grid_for(total4)), dim3(256), 0, (hipStream_t)stream, tok,
                           strideb, up, gh, gw, total4, sf, inv);
    else
        hipLaunchKernelGGL(sm::upsample2x_kernel<false>, dim3(sm::grid_for(total4)), dim3(256), 0, (hipStream_t)stream, tok,
                           strideb, up, gh, gw, total4, sf, inv);
    return sm::check_launch("sm_upsample_tokens");
}
extern "C" int sm_upsample_tokens_f32(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw, int32_t scale,
                                      void* stream) {
    return upsample_impl(tok, strideb, up, B, gh, gw, stream, false, scale);
}
extern "C" int sm_upsample_tokens_f16x2(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw, int32_t scale,
                                        void* stream) {
    return upsample_impl(tok, strideb, up, B, gh, gw, stream, true, scale);
}
extern "C" int sm_upsample_logits_sigmoid_f32(const float* low, float* logits, float* prob, int64_t planes, int32_t gh, int32_t gw,
                                              int32_t scale, void* stream) {
    SM_REQUIRE(low && prob && planes > 0 && gh > 0 && gw > 0 && scale >= 1 && scale <= 16, "sm_upsample_logits_sigmoid_f32: bad arguments");
    const int64_t opix = (int64_t)scale * scale * gh * gw, groups = (planes + sm::UPL_PLANES - 1) / sm::UPL_PLANES;
    SM_REQUIRE(opix < (1 << 30) && groups < 65536, "sm_upsample_logits_sigmoid_f32: %lld planes of %lld pixels are beyond the launch grid",
               (long long)planes, (long long)opix);
    hipLaunchKernelGGL(sm::upsample2x_logits_kernel, dim3((unsigned)((opix + 255) / 256), (unsigned)groups), dim3(256), 0, (hipStream_t)stream,
                       low, logits, prob, gh, gw, planes, scale, (float)(1.0 / scale));
    return sm::check_launch("sm_upsample_logits_sigmoid_f32");
}
extern "C" int sm_upsample2x_tokens_f32(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw,
                                        void* stream) {
    return upsample_impl(tok, strideb, up, B, gh, gw, stream, false);
}
extern "C" int sm_upsample2x_tokens_f16x2(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw,
                                          void* stream) {
    return upsample_impl(tok, strideb, up, B, gh, gw, stream, true);
}

extern "C" int sm_upsample2x_logits_sigmoid_f32(const float* low, float* logits, float* prob, int64_t planes, int32_t gh,
                                                int32_t gw, void* stream) {
    return sm_upsample_logits_sigmoid_f32(low, logits, prob, planes, gh, gw, 2, stream);
}

extern "C" int sm_rowdot_sigmoid_f32(const float* h, const float* w, const float* b, float* out, int32_t rows,
                                     void* stream) {
    SM_REQUIRE(h && w && b && out && rows > 0, "sm_rowdot_sigmoid_f32: bad arguments");
    hipLaunchKernelGGL(sm::rowdot_sigmoid_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, w, b, out,
                       rows);
    return sm::check_launch("sm_rowdot_sigmoid_f32");
}

namespace sm {
// serving path (app.py:266-284): best = argmax(objectness) (first maximum, as torch.argmax), out = clip(mask[best], 0, 1)
__global__ __launch_bounds__(256) void pick_mask_kernel(const float* __restrict__ masks, int64_t mask_stride_b,
                                                       const float* __restrict__ obj, int64_t obj_stride_b, float* __restrict__ out,
                                                       int* __restrict__ best_out, int nq, int hw) {
    const int b = blockIdx.y;
    const float* o = obj + (int64_t)b * obj_stride_b;
    int best = 0;
    for (int q = 1; q < nq; ++q)
        if (o[q] > o[best]) best = q;  // every thread repeats the nq-long scan (nq = 20): no shared memory, no divergence
    if (blockIdx.x == 0 && threadIdx.x == 0) best_out[b] = best;
    const float* m = masks + (int64_t)b * mask_stride_b + (int64_t)best * hw;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) out[(int64_t)b * hw + p] = fminf(fmaxf(m[p], 0.f), 1.f);
}
}  // namespace sm

extern "C" int sm_pick_mask_f32(const float* masks, int64_t mask_stride_b, const float* objectness, int64_t obj_stride_b,
                                float* out, int32_t* best, int32_t B, int32_t nq, int32_t hw, void* stream) {
    SM_REQUIRE(masks && objectness && out && best && B > 0 && nq > 0 && hw > 0, "sm_pick_mask_f32: bad arguments");
    hipLaunchKernelGGL(sm::pick_mask_kernel, dim3((hw + 255) / 256 < 64 ? (hw + 255) / 256 : 64, B), dim3(256), 0,
                       (hipStream_t)stream, masks, mask_stride_b, objectness, obj_stride_b, out, best, nq, hw);
    return sm::check_launch("sm_pick_mask_f32");
}

extern "C" int sm_query_mean_f32(const float* queries, float* features, int32_t B, int32_t L, int32_t nq, void* stream) {
    SM_REQUIRE(queries && features && B > 0 && L > 0 && nq > 0, "sm_query_mean_f32: bad arguments");
    hipLaunchKernelGGL(sm::query_mean_kernel, dim3(B * (SM_EMBED / 128)), dim3(128), 0, (hipStream_t)stream, queries, features, L, nq);
    return sm::check_launch("sm_query_mean_f32");
}

#ifdef SM_TUNING
// Diagnostic (tuning build only): one wave samples the shader clock counter (s_memtime) against the constant 100 MHz
// reference counter (s_memrealtime) for `spins` iterations; out[0..3] = first / last pair.  Run beside a kernel under test on
// another stream: (d memtime / d memrealtime) x 100 MHz is the clock the chip holds under that load.
namespace sm {
__global__ void clock_probe_kernel(unsigned long long* out, int spins) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = t0, r1 = r0;
    for (int i = 0; i < spins; ++i) {
        __builtin_amdgcn_s_sleep(32);
        t1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) { out[0] = t0; out[1] = r0; out[2] = t1; out[3] = r1; }
}
}  // namespace sm
extern "C" int sm_clock_probe(unsigned long long* out, int spins, int blocks, void* stream) {
    for (int b = 0; b < blocks; ++b)
        hipLaunchKernelGGL(sm::clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out + 4 * b, spins);
    return sm::check_launch("sm_clock_probe");
}
#endif
