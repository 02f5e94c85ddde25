// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate (operands in registers, random data), 1-3 waves per SIMD,
// with and without ds_read_b128 traffic at the split GEMM's ratio (4 reads per 3 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void loop(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[16384];
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)in[threadIdx.x + e]; y[e] = (_Float16)in[threadIdx.x + 8 + e]; }
    for (int i = threadIdx.x; i < 4096; i += 256) ((float*)sm)[i] = in[i & 1023];
    __syncthreads();
    const char* p = sm + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (LDS) {
                // one read per MFMA (+1 per 3): the fragment feeds the NEXT MFMAs like in a software pipeline
                f16x8 r = *reinterpret_cast<const f16x8*>(p + ((it + a) & 7) * 1024);
                x = r;
            }
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int v = 0; v < 16; ++v) s += acc[a][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(int wg_per_cu, const float* in, float* out) {
    const int iters = 2000, grid = 256 * wg_per_cu;
    loop<NACC, LDS><<<grid, 256>>>(in, out, 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<NACC, LDS><<<grid, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("NACC=%d lds=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", NACC, (int)LDS, wg_per_cu, ms, fl / ms / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22);
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int w = 1; w <= 3; ++w) { run<4, false>(w, in, out); run<8, false>(w, in, out); run<4, true>(w, in, out); run<8, true>(w, in, out); }
    return 0;
}
