// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on this device (operands in registers, random data).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    float x = in[threadIdx.x], y = in[threadIdx.x + 256];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int v = 0; v < 16; ++v) s += acc[a][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu) {
    int grid = 256 * wg_per_cu, iters = 2000;
    float *in, *out;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, grid * 256 * 4);
    std::vector<float> h(512); for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<grid, 256>>>(in, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<grid, 256>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d wg/cu=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, ms, flops / ms / 1e9);
}
int main() {
    run<1>(1); run<4>(1); run<4>(2); run<2>(2); run<1>(2);
    return 0;
}
