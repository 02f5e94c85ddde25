// LayerNorm over rows of 384 floats: one 64-lane wave per row, 6 elements per lane held in registers,
// mean and (two-pass, biased) variance by wavefront shuffle butterflies.  HBM-bound: reads x once, writes y once.
// Reference: nn.LayerNorm at vision_transformer.py:165,169,299 (eps 1e-6), transformer_decoder.py:280,290,295,
// 139 (eps 1e-5).
#include "common.h"

namespace sm {

__device__ __forceinline__ int64_t map_row(int r, sm_row_map m) {
    return m.group > 0 ? (int64_t)(r / m.group) * m.stride + m.offset + r % m.group : r;
}

__global__ __launch_bounds__(256) void layernorm384_kernel(sm_ln_args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;  // whole wave exits together
    const float* xr = a.x + map_row(row, a.in_map) * a.ldx;
    float2 v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const float2*>(xr + i * 128 + lane * 2);
    if (a.n_partials > 0) {  // fused split-K reduction: slices in order, then bias, then the residual
        for (int sidx = 1; sidx < a.n_partials; ++sidx) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float2 t = *reinterpret_cast<const float2*>(xr + sidx * a.partial_stride + i * 128 + lane * 2);
                v[i].x += t.x; v[i].y += t.y;
            }
        }
        const float* rr = a.residual + map_row(row, a.in_map) * a.ldx;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float2 bb = *reinterpret_cast<const float2*>(a.pre_bias + i * 128 + lane * 2);
            const float2 t = *reinterpret_cast<const float2*>(rr + i * 128 + lane * 2);
            v[i].x = t.x + (v[i].x + bb.x); v[i].y = t.y + (v[i].y + bb.y);
        }
    }
    float s = (v[0].x + v[0].y) + (v[1].x + v[1].y) + (v[2].x + v[2].y);
    const float mean = wave_sum(s) * (1.0f / 384.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i].x -= mean;
        v[i].y -= mean;
        q += v[i].x * v[i].x + v[i].y * v[i].y;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 384.0f) + a.eps);
    const int64_t orow = map_row(row, a.out_map);
    float* yr = a.y ? a.y + orow * a.ldy : nullptr;
    float* ysr = a.ys ? a.ys + orow * a.ldy : nullptr;
    float* y2r = a.y2 ? a.y2 + (int64_t)row * a.ldy2 : nullptr;
    const float* ar = a.y2 ? a.add + (int64_t)(row % a.add_rows) * SM_EMBED : nullptr;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float2 gm = *reinterpret_cast<const float2*>(a.gamma + i * 128 + lane * 2);
        const float2 bt = *reinterpret_cast<const float2*>(a.beta + i * 128 + lane * 2);
        float2 o;
        o.x = v[i].x * rstd * gm.x + bt.x;
        o.y = v[i].y * rstd * gm.y + bt.y;
        if (yr) *reinterpret_cast<float2*>(yr + i * 128 + lane * 2) = o;
        if (ysr) store_f16x2_2(ysr, i * 128 + lane * 2, o.x, o.y);
        if (y2r) {
            const float2 ad = *reinterpret_cast<const float2*>(ar + i * 128 + lane * 2);
            o.x += ad.x;
            o.y += ad.y;
            if (a.y2_f16x2) store_f16x2_2(y2r, i * 128 + lane * 2, o.x, o.y);
            else *reinterpret_cast<float2*>(y2r + i * 128 + lane * 2) = o;
        }
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
    SM_REQUIRE(a->rows >= 0 && a->ldx >= SM_EMBED && a->ldy >= SM_EMBED && a->ldx % 2 == 0 && a->ldy % 2 == 0,
               "sm_layernorm_f32: bad rows/strides");
    SM_REQUIRE(a->in_map.group >= 0 && a->out_map.group >= 0, "sm_layernorm_f32: bad row map");
    if (a->n_partials > 0) SM_REQUIRE(a->pre_bias && a->residual && a->partial_stride > 0, "sm_layernorm_f32: bad partials");
    if (a->y2) SM_REQUIRE(a->add && a->add_rows > 0 && a->ldy2 >= SM_EMBED && a->ldy2 % 2 == 0, "sm_layernorm_f32: bad y2/add");
    if (a->rows == 0) return SM_OK;
    hipLaunchKernelGGL(sm::layernorm384_kernel, dim3((a->rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a);
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
