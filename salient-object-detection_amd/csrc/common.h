// Shared helpers for the gfx950 kernels of libselfmask_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/selfmask_hip.h"

namespace sm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SM_ELAUNCH;
    }
    return SM_OK;
}

#define SM_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            sm::set_error(__VA_ARGS__); \
            return SM_EINVAL;          \
        }                              \
    } while (0)

// 64-lane butterfly reductions (wavefront shuffles; every lane ends with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// row of a 32x32 MFMA accumulator held in register v by lane-half h  (cdna_hip_programming.md section 3)
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

}  // namespace sm
