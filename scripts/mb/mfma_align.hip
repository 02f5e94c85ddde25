// Does the alignment (mod 4) of the VGPR tuples holding the A / B operands of v_mfma_f32_16x16x32_f16 matter?
// Pure-asm loops, one wave per SIMD x 2, 16 independent accumulators, operands never rewritten (values: whatever the
// registers hold after a few v_mov of lane-dependent data).  hipcc --offload-arch=gfx950 -O3 scripts/mb/mfma_align.hip -o mfma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

#define MF(acc, a, b) "v_mfma_f32_16x16x32_f16 v[" #acc ":" #acc "+3], v[" a "], v[" b "], v[" #acc ":" #acc "+3]\n"
// A operand tuple, then eight B tuples
#define BODY(A, B0, B1, B2, B3, B4, B5, B6, B7)                                                                     \
    MF(0, A, B0) MF(4, A, B1) MF(8, A, B2) MF(12, A, B3) MF(16, A, B4) MF(20, A, B5) MF(24, A, B6) MF(28, A, B7)     \
    MF(32, A, B0) MF(36, A, B1) MF(40, A, B2) MF(44, A, B3) MF(48, A, B4) MF(52, A, B5) MF(56, A, B6) MF(60, A, B7)

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    // fill v64..v127 with lane-dependent junk so that the operands are not all zero (power / clock realism)
    asm volatile(
        "v_cvt_f32_u32 v64, v0\n v_mul_f32 v64, 0x3a83126f, v64\n v_cvt_pkrtz_f16_f32 v64, v64, v64\n"
        "v_mov_b32 v65, v64\n v_mov_b32 v66, v64\n v_mov_b32 v67, v64\n v_mov_b32 v68, v64\n v_mov_b32 v69, v64\n"
        "v_mov_b32 v70, v64\n v_mov_b32 v71, v64\n"
        ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
    for (int r = 72; r < 128; ++r) {}
    asm volatile(
        "v_mov_b32 v80, v64\n v_mov_b32 v81, v64\n v_mov_b32 v82, v64\n v_mov_b32 v83, v64\n v_mov_b32 v84, v64\n v_mov_b32 v85, v64\n"
        "v_mov_b32 v86, v64\n v_mov_b32 v87, v64\n v_mov_b32 v88, v64\n v_mov_b32 v89, v64\n v_mov_b32 v90, v64\n v_mov_b32 v91, v64\n"
        "v_mov_b32 v92, v64\n v_mov_b32 v93, v64\n v_mov_b32 v94, v64\n v_mov_b32 v95, v64\n v_mov_b32 v96, v64\n v_mov_b32 v97, v64\n"
        "v_mov_b32 v98, v64\n v_mov_b32 v99, v64\n v_mov_b32 v100, v64\n v_mov_b32 v101, v64\n v_mov_b32 v102, v64\n v_mov_b32 v103, v64\n"
        "v_mov_b32 v104, v64\n v_mov_b32 v105, v64\n v_mov_b32 v106, v64\n v_mov_b32 v107, v64\n v_mov_b32 v108, v64\n v_mov_b32 v109, v64\n"
        "v_mov_b32 v110, v64\n v_mov_b32 v111, v64\n v_mov_b32 v112, v64\n v_mov_b32 v113, v64\n v_mov_b32 v114, v64\n v_mov_b32 v115, v64\n"
        ::: "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",
            "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)       // everything 4-aligned (what the fast build of the GEMM got from the register allocator)
            asm volatile(BODY("64:67", "80:83", "84:87", "88:91", "92:95", "96:99", "100:103", "104:107", "108:111")
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
        else if (MODE == 1)  // four of the eight B tuples start at a register = 2 mod 4 (the slow build)
            asm volatile(BODY("64:67", "80:83", "84:87", "88:91", "92:95", "98:101", "102:105", "106:109", "110:113")
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
        else                 // A tuple = 2 mod 4 as well
            asm volatile(BODY("66:69", "80:83", "84:87", "88:91", "92:95", "98:101", "102:105", "106:109", "110:113")
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v32" : "=v"(r));
    if (r == 123456.0f) out[threadIdx.x] = r;
}

template <int MODE>
static int run(const char* what) {
    float* out;
    CHECK(hipMalloc(&out, 4096));
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * 8 * iters * 16 * 16384.0;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", what, ms, flops / ms / 1e9);
    return 0;
}
int main() {
    for (int r = 0; r < 2; ++r) {
        if (run<0>("A, B tuples 4-aligned")) return 1;
        if (run<1>("four of eight B tuples at 2 mod 4")) return 1;
        if (run<2>("A and four of eight B tuples at 2 mod 4")) return 1;
    }
    return 0;
}
