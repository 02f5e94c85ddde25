// LayerNorm over rows of 384 floats: four rows per 64-lane wave (16 lanes x 24 register-resident elements per row),
// mean and (two-pass, biased) variance by shuffle butterflies.  HBM-bound: reads x once, writes y once.
// Reference: nn.LayerNorm at vision_transformer.py:165,169,299 (eps 1e-6), transformer_decoder.py:280,290,295,
// 139 (eps 1e-5).
#include "common.h"

namespace sm {

__device__ __forceinline__ int64_t map_row(int r, sm_row_map m) {
    return m.group > 0 ? (int64_t)(r / m.group) * m.stride + m.offset + r % m.group : r;
}

// 16 lanes per row (four rows per wave), each lane owning the three 8-element groups g = l16 + 16 i: every access is a
// 16-B (fp32: 2 x 16 B) piece and an F16X2 group leaves as one contiguous 32-B store; statistics are 4-step butterflies
// inside the 16-lane group.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}

__global__ __launch_bounds__(256) void layernorm384_kernel(sm_ln_args a) {
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < a.rows;
    const int rrow = live ? row : a.rows - 1;  // dead groups shadow the last row (they join the shuffles, never store)
    const float* xr = a.x + map_row(rrow, a.in_map) * a.ldx;
    float v[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 p0 = *reinterpret_cast<const float4*>(xr + (l16 + 16 * i) * 8);
        const float4 p1 = *reinterpret_cast<const float4*>(xr + (l16 + 16 * i) * 8 + 4);
        v[i][0] = p0.x; v[i][1] = p0.y; v[i][2] = p0.z; v[i][3] = p0.w;
        v[i][4] = p1.x; v[i][5] = p1.y; v[i][6] = p1.z; v[i][7] = p1.w;
    }
    if (a.n_partials > 0) {  // fused split-K reduction: slices in order, then bias, then the residual
        for (int sidx = 1; sidx < a.n_partials; ++sidx) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float* t = xr + sidx * a.partial_stride + (l16 + 16 * i) * 8;
                const float4 p0 = *reinterpret_cast<const float4*>(t), p1 = *reinterpret_cast<const float4*>(t + 4);
                v[i][0] += p0.x; v[i][1] += p0.y; v[i][2] += p0.z; v[i][3] += p0.w;
                v[i][4] += p1.x; v[i][5] += p1.y; v[i][6] += p1.z; v[i][7] += p1.w;
            }
        }
        const float* rr = a.residual + map_row(rrow, a.in_map) * a.ldx;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = (l16 + 16 * i) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 bb = *reinterpret_cast<const float4*>(a.pre_bias + k + 4 * h);
                const float4 t = *reinterpret_cast<const float4*>(rr + k + 4 * h);
                v[i][4 * h + 0] = t.x + (v[i][4 * h + 0] + bb.x); v[i][4 * h + 1] = t.y + (v[i][4 * h + 1] + bb.y);
                v[i][4 * h + 2] = t.z + (v[i][4 * h + 2] + bb.z); v[i][4 * h + 3] = t.w + (v[i][4 * h + 3] + bb.w);
            }
        }
    }
    if (a.raw && live) {  // the value itself (a pre-norm block's residual stream), before it is centred in place below
        float* rw = a.raw + map_row(row, a.in_map) * a.ldx;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = (l16 + 16 * i) * 8;
            *reinterpret_cast<float4*>(rw + k) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
            *reinterpret_cast<float4*>(rw + k + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    const float mean = group16_sum(s) * (1.0f / 384.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] -= mean;
            q += v[i][e] * v[i][e];
        }
    const float rstd = 1.0f / sqrtf(group16_sum(q) * (1.0f / 384.0f) + a.eps);
    if (!live) return;
    const int64_t orow = map_row(row, a.out_map);
    float* yr = a.y ? a.y + orow * a.ldy : nullptr;
    float* ysr = a.ys ? a.ys + orow * a.ldy : nullptr;
    float* y2r = a.y2 ? a.y2 + (int64_t)row * a.ldy2 : nullptr;
    const float* ar = a.y2 ? a.add + (int64_t)(row % a.add_rows) * SM_EMBED : nullptr;
    const bool chain = a.chain_gamma != nullptr;  // (kernel-uniform)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = (l16 + 16 * i) * 8;
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 gm = *reinterpret_cast<const float4*>(a.gamma + k + 4 * h);
            const float4 bt = *reinterpret_cast<const float4*>(a.beta + k + 4 * h);
            o[4 * h + 0] = v[i][4 * h + 0] * rstd * gm.x + bt.x; o[4 * h + 1] = v[i][4 * h + 1] * rstd * gm.y + bt.y;
            o[4 * h + 2] = v[i][4 * h + 2] * rstd * gm.z + bt.z; o[4 * h + 3] = v[i][4 * h + 3] * rstd * gm.w + bt.w;
        }
        if (chain) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = o[e];  // the chained norm's input: y as stored
        }
        const float (&o0)[4] = *reinterpret_cast<const float (*)[4]>(&o[0]);
        const float (&o1)[4] = *reinterpret_cast<const float (*)[4]>(&o[4]);
        if (yr) {
            *reinterpret_cast<float4*>(yr + k) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(yr + k + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (ysr) store_f16x2_8(ysr, k, o0, o1);
        if (y2r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 ad = *reinterpret_cast<const float4*>(ar + k + 4 * h);
                o[4 * h + 0] += ad.x; o[4 * h + 1] += ad.y; o[4 * h + 2] += ad.z; o[4 * h + 3] += ad.w;
            }
            if (a.y2_f16x2) store_f16x2_8(y2r, k, o0, o1);
            else {
                *reinterpret_cast<float4*>(y2r + k) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(y2r + k + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    }
    if (!chain) return;
    // chained norm: the same arithmetic, in the same order, as a launch of its own reading y (every lane of a live row's group is live)
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) s2 += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
    const float mean2 = group16_sum(s2) * (1.0f / 384.0f);
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[i][e] -= mean2;
            q2 += v[i][e] * v[i][e];
        }
    const float rstd2 = 1.0f / sqrtf(group16_sum(q2) * (1.0f / 384.0f) + a.chain_eps);
    const int64_t crow = map_row(row, a.chain_map);
    float* cyr = a.chain_y ? a.chain_y + crow * a.chain_ldy : nullptr;
    float* cysr = a.chain_ys ? a.chain_ys + crow * a.chain_ldy : nullptr;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = (l16 + 16 * i) * 8;
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 gm = *reinterpret_cast<const float4*>(a.chain_gamma + k + 4 * h);
            const float4 bt = *reinterpret_cast<const float4*>(a.chain_beta + k + 4 * h);
            o[4 * h + 0] = v[i][4 * h + 0] * rstd2 * gm.x + bt.x; o[4 * h + 1] = v[i][4 * h + 1] * rstd2 * gm.y + bt.y;
            o[4 * h + 2] = v[i][4 * h + 2] * rstd2 * gm.z + bt.z; o[4 * h + 3] = v[i][4 * h + 3] * rstd2 * gm.w + bt.w;
        }
        const float (&o0)[4] = *reinterpret_cast<const float (*)[4]>(&o[0]);
        const float (&o1)[4] = *reinterpret_cast<const float (*)[4]>(&o[4]);
        if (cyr) {
            *reinterpret_cast<float4*>(cyr + k) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(cyr + k + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (cysr) store_f16x2_8(cysr, k, o0, o1);
    }
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int rows_per, int64_t total4) {
    const int64_t per4 = (int64_t)rows_per * (SM_EMBED / 4);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256)
        reinterpret_cast<float4*>(dst)[t] = reinterpret_cast<const float4*>(src)[t % per4];
}

}  // namespace sm

extern "C" int sm_layernorm_rows_f32(const sm_ln_args* a, void* stream) {
    SM_REQUIRE(a && a->x && a->gamma && a->beta && (a->y || a->ys), "sm_layernorm_f32: null pointer");
    if (a->ys || a->y2_f16x2) SM_REQUIRE(a->ldy % 8 == 0 && a->ldy2 % 8 == 0, "sm_layernorm_f32: F16X2 outputs need ld %% 8 == 0");
    SM_REQUIRE(a->rows >= 0 && a->ldx >= SM_EMBED && a->ldy >= SM_EMBED && a->ldx % 4 == 0 && a->ldy % 4 == 0,
               "sm_layernorm_f32: bad rows/strides (row strides must be multiples of 4 floats)");
    SM_REQUIRE(((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->ys | (uintptr_t)a->y2 | (uintptr_t)a->gamma | (uintptr_t)a->beta |
                (uintptr_t)a->add | (uintptr_t)a->pre_bias | (uintptr_t)a->residual) % 16 == 0,
               "sm_layernorm_f32: pointers must be 16-B aligned");
    SM_REQUIRE(a->in_map.group >= 0 && a->out_map.group >= 0, "sm_layernorm_f32: bad row map");
    if (a->n_partials > 0)
        SM_REQUIRE(a->pre_bias && a->residual && a->partial_stride > 0 && a->partial_stride % 4 == 0, "sm_layernorm_f32: bad partials");
    if (a->y2) SM_REQUIRE(a->add && a->add_rows > 0 && a->ldy2 >= SM_EMBED && a->ldy2 % 4 == 0, "sm_layernorm_f32: bad y2/add");
    if (a->chain_gamma)
        SM_REQUIRE(a->chain_beta && (a->chain_y || a->chain_ys) && a->chain_ldy >= SM_EMBED && a->chain_ldy % 8 == 0 && a->chain_map.group >= 0 &&
                       ((uintptr_t)a->chain_gamma | (uintptr_t)a->chain_beta | (uintptr_t)a->chain_y | (uintptr_t)a->chain_ys) % 16 == 0,
                   "sm_layernorm_f32: bad chained norm (gamma, beta, an output, ld %% 8 == 0, 16-B aligned)");
    if (a->rows == 0) return SM_OK;
    hipLaunchKernelGGL(sm::layernorm384_kernel, dim3((a->rows + 15) / 16), dim3(256), 0, (hipStream_t)stream, *a);
    return sm::check_launch("sm_layernorm_f32");
}

extern "C" int sm_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                                int64_t ldy, int32_t rows, int32_t cols, float eps, void* stream) {
    SM_REQUIRE(cols == SM_EMBED, "sm_layernorm_f32: cols=%d, only 384 is supported", cols);
    sm_ln_args a = {};
    a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.y = y; a.ldy = ldy; a.rows = rows; a.eps = eps;
    return sm_layernorm_rows_f32(&a, stream);
}

extern "C" int sm_broadcast_rows_f32(const float* src, float* dst, int32_t rows_per, int32_t B, void* stream) {
    SM_REQUIRE(src && dst && rows_per > 0 && B > 0, "sm_broadcast_rows_f32: bad arguments");
    const int64_t total4 = (int64_t)B * rows_per * (SM_EMBED / 4);
    int64_t grid = (total4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sm::broadcast_rows_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, dst, rows_per,
                       total4);
    return sm::check_launch("sm_broadcast_rows_f32");
}
