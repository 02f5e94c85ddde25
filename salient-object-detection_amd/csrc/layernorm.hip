// LayerNorm over rows of 384 floats: one 64-lane wave per row, 6 elements per lane held in registers,
// mean and (two-pass, biased) variance by wavefront shuffle butterflies.  HBM-bound: reads x once, writes y once.
// Reference: nn.LayerNorm at vision_transformer.py:165,169,299 (eps 1e-6), transformer_decoder.py:280,290,295,
// 139 (eps 1e-5).
#include "common.h"

namespace sm {

__device__ __forceinline__ int64_t map_row(int r, sm_row_map m) {
    return m.group > 0 ? (int64_t)(r / m.group) * m.stride + m.offset + r % m.group : r;
}

__global__ __launch_bounds__(256) void layernorm384_kernel(const float* __restrict__ x, int64_t ldx, sm_row_map in_map,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* y, int64_t ldy,
                                                           sm_row_map out_map, int rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // whole wave exits together
    const float* xr = x + map_row(row, in_map) * ldx;
    float2 v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const float2*>(xr + i * 128 + lane * 2);
    float s = (v[0].x + v[0].y) + (v[1].x + v[1].y) + (v[2].x + v[2].y);
    const float mean = wave_sum(s) * (1.0f / 384.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i].x -= mean;
        v[i].y -= mean;
        q += v[i].x * v[i].x + v[i].y * v[i].y;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 384.0f) + eps);
    float* yr = y + map_row(row, out_map) * ldy;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float2 gm = *reinterpret_cast<const float2*>(gamma + i * 128 + lane * 2);
        const float2 bt = *reinterpret_cast<const float2*>(beta + i * 128 + lane * 2);
        float2 o;
        o.x = v[i].x * rstd * gm.x + bt.x;
        o.y = v[i].y * rstd * gm.y + bt.y;
        *reinterpret_cast<float2*>(yr + i * 128 + lane * 2) = o;
    }
}

}  // namespace sm

extern "C" int sm_layernorm_rows_f32(const float* x, int64_t ldx, sm_row_map in_map, const float* gamma,
                                     const float* beta, float* y, int64_t ldy, sm_row_map out_map, int32_t rows,
                                     float eps, void* stream) {
    SM_REQUIRE(x && gamma && beta && y, "sm_layernorm_f32: null pointer");
    SM_REQUIRE(rows >= 0 && ldx >= SM_EMBED && ldy >= SM_EMBED && ldx % 2 == 0 && ldy % 2 == 0,
               "sm_layernorm_f32: bad rows/strides");
    SM_REQUIRE(in_map.group >= 0 && out_map.group >= 0, "sm_layernorm_f32: bad row map");
    if (rows == 0) return SM_OK;
    hipLaunchKernelGGL(sm::layernorm384_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, in_map,
                       gamma, beta, y, ldy, out_map, rows, eps);
    return sm::check_launch("sm_layernorm_f32");
}

extern "C" int sm_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                                int64_t ldy, int32_t rows, int32_t cols, float eps, void* stream) {
    SM_REQUIRE(cols == SM_EMBED, "sm_layernorm_f32: cols=%d, only 384 is supported", cols);
    const sm_row_map id = {0, 0, 0};
    return sm_layernorm_rows_f32(x, ldx, id, gamma, beta, y, ldy, id, rows, eps, stream);
}
