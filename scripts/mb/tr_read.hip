// What does ds_read_b64_tr_b16 deliver?  LDS holds M[row][col] = row*100 + col as f16 (32 rows x 64 cols, row stride 128 B).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int row_stride_bytes) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32 * 64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) lds[i] = (_Float16)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x;
    // candidate addressing: within each 16-lane group, lane 4q+p -> row q, cols 4p..4p+3 of a 4x16 block
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int row0 = 0, col0 = 16 * grp;   // each group reads its own 16 columns, rows 0..3
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (row0 + q) * row_stride_bytes + (col0 + 4 * p) * 2;
    h4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    k<<<1, 64>>>(d, 128);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
