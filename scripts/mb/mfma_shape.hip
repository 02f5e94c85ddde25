// Micro-benchmark: v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 at EQUAL FLOPs per wave, random operands, with the
// W16 GEMM's LDS read ratio (one ds_read_b128 per MFMA-equivalent of work), 2 and 4 waves per SIMD, ~30 ms per run so the
// chip settles at its power-limited clock.  Prints TFLOP/s: under a power limit the shape that needs less energy per FLOP wins
// (MI355X_MICROARCH.md, DVFS give-back item 7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LDS>  // SHAPE 32: four 32x32 accumulators; SHAPE 16: sixteen 16x16 accumulators (same 64 registers)
__global__ __launch_bounds__(256) void loop(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[16384];
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)sm)[i] = in[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    const char* p = sm + (threadIdx.x & 63) * 16;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)in[threadIdx.x + e]; y[e] = (_Float16)in[threadIdx.x + 8 + e]; }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (LDS) x = *reinterpret_cast<const f16x8*>(p + ((it + a) & 7) * 1024);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int v = 0; v < 16; ++v) s += acc[a][v];
    } else {
        f32x4 acc[16];
        for (int a = 0; a < 16; ++a) for (int v = 0; v < 4; ++v) acc[a][v] = 0.f;
        for (int it = 0; it < iters / 2; ++it) {  // 16 x (16x16x32) = 8 x (32x32x16) FLOPs per iteration, half the iterations
#pragma unroll
            for (int a = 0; a < 16; ++a) {  // one read per two MFMAs = the same LDS bytes per FLOP as the 32x32 loop
                if (LDS && (a & 1) == 0) x = *reinterpret_cast<const f16x8*>(p + ((it + a) & 7) * 1024);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[a], 0, 0, 0);
            }
        }
        for (int a = 0; a < 16; ++a) for (int v = 0; v < 4; ++v) s += acc[a][v];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, bool LDS>
double run(int wg_per_cu, const float* in, float* out) {
    const int iters = 40000, grid = 256 * wg_per_cu;
    loop<SHAPE, LDS><<<grid, 256>>>(in, out, 100);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<SHAPE, LDS><<<grid, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    printf("shape %2d lds=%d waves/SIMD=%d: %8.3f ms  %7.1f TFLOP/s\n", SHAPE, (int)LDS, wg_per_cu, ms, fl / ms / 1e9);
    return fl / ms / 1e9;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22);
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.0f;
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 2; w <= 4; w += 2) {
            run<32, false>(w, in, out); run<16, false>(w, in, out);
            run<32, true>(w, in, out); run<16, true>(w, in, out);
        }
    return 0;
}
