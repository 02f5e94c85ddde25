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

// One wave-wide LDS-DMA: every lane fetches 16 B from its own global address, the 64 pieces land contiguously at LDS
// byte offset `lds_off` (wave-uniform, goes through M0) + 16 * lane.  Issued from inline asm (M0 saved / restored) so
// that hipcc cannot put a vmcnt(0) in front of later LDS reads; completion is tracked by the caller's s_waitcnt vmcnt.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_off)
        : "memory");
}

// The same with a wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit byte offset: half the address registers of the
// pointer form, and a K loop advances the scalar base instead of 64 per-lane pointers
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_off)
        : "memory");
}

// Reductions across the 16-lane rows / the two halves of a wave by register swaps (gfx950 v_permlane16_swap / v_permlane32_swap)
// instead of ds_bpermute round trips through the LDS: swapping a value with itself leaves rows {0,0,2,2} in one result and
// {1,1,3,3} in the other (halves: {lo,lo} and {hi,hi}), so one max / add per step gives every lane the same reduced value,
// summed in the same order on every lane.
__device__ __forceinline__ float halves_max(float x) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
__device__ __forceinline__ float halves_sum(float x) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ float rows4_max(float x) {  // over lanes l, l ^ 16, l ^ 32, l ^ 48
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return halves_max(fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1])));
}
__device__ __forceinline__ float rows4_sum(float x) {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return halves_sum(__uint_as_float(s[0]) + __uint_as_float(s[1]));
}

// ---- F16X2 split format (see gemm_f16x2.hip): hi = f16(x), lo = f16((x - hi) * 2^11) ---------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The hi conversion goes through inline asm so that the value stored and the value subtracted are ONE conversion
// result: left to itself hipcc emitted a packed round-toward-zero convert for a stored vector and a round-to-nearest
// one for the subtraction (lo then had the wrong sign whenever the two roundings differed).
__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    float hf;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(hi) : "v"(x));
    asm("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(hi));
    lo = (_Float16)((x - hf) * 2048.0f);
}
// Two elements in 5 VALU instructions instead of 12, same bits as split1: one packed round-to-nearest convert (gfx950
// v_cvt_pk_f16_f32) for the hi pair, then lo = f16(fma(f32(hi), -2048, 2048 x)) by v_fma_mixlo/mixhi_f16, which read the
// f16 halves of the hi register directly and write the halves of the lo register (2048 (x - hi) is exact in f32, so the
// single fma rounds like the subtract-multiply-convert chain).
__device__ __forceinline__ void split2(float x0, float x1, f16x2& hi, f16x2& lo) {
    unsigned h, l;
    const float m2048 = -2048.0f;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    const float y0 = x0 * 2048.0f, y1 = x1 * 2048.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "s"(m2048), "v"(y0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "s"(m2048), "v"(y1));
    hi = __builtin_bit_cast(f16x2, h);
    lo = __builtin_bit_cast(f16x2, l);
}
__device__ __forceinline__ void split4(const float (&x)[4], f16x4& hi, f16x4& lo) {
    f16x2 h0, l0, h1, l1;
    split2(x[0], x[1], h0, l0);
    split2(x[2], x[3], h1, l1);
    hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
    lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
}
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
    f16x4 h0, l0, h1, l1;
    const float a[4] = {x[0], x[1], x[2], x[3]}, b[4] = {x[4], x[5], x[6], x[7]};
    split4(a, h0, l0);
    split4(b, h1, l1);
    hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// The same split from instructions the compiler models (v_cvt_pk_f16_f32 from the vector conversion, the lo half as
// f16(fma(f32(hi), -2048, 2048 x)) - exact in f32, one rounding: the same bits as split2).  For results that feed an MFMA
// straight from registers: hipcc neither schedules around nor pads the hazards of an `asm` statement's instructions (VALU
// write -> MFMA operand read needs wait states, cdna_hip_programming.md 5.7 item 2), and with the attention step laid out
// as one basic block the scheduler does place the P V MFMAs right behind the split (measured: 2^-11-sized errors).
__device__ __forceinline__ void split2c(float x0, float x1, f16x2& hi, f16x2& lo) {
    const f32x2 v = {x0, x1};
    hi = __builtin_convertvector(v, f16x2);
    f16x2 l;
    l[0] = (_Float16)__builtin_fmaf((float)hi[0], -2048.0f, x0 * 2048.0f);
    l[1] = (_Float16)__builtin_fmaf((float)hi[1], -2048.0f, x1 * 2048.0f);
    lo = l;
}
__device__ __forceinline__ void split8c(const float (&x)[8], f16x8& hi, f16x8& lo) {
    f16x2 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2c(x[2 * i], x[2 * i + 1], h[i], l[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { hi[2 * i] = h[i][0]; hi[2 * i + 1] = h[i][1]; lo[2 * i] = l[i][0]; lo[2 * i + 1] = l[i][1]; }
}
// byte offset of element k of an F16X2 row: hi half; the lo half lives 16 bytes further
__device__ __forceinline__ int64_t f16x2_off(int64_t k) { return (k & ~(int64_t)7) * 4 + (k & 7) * 2; }
// store 4 consecutive elements k..k+3 (k % 4 == 0) of an F16X2 row
__device__ __forceinline__ void store_f16x2_4(void* row, int64_t k, const float (&x)[4]) {
    f16x4 hi, lo;
    split4(x, hi, lo);
    char* p = reinterpret_cast<char*>(row) + f16x2_off(k);
    *reinterpret_cast<f16x4*>(p) = hi;
    *reinterpret_cast<f16x4*>(p + 16) = lo;
}
// store 2 consecutive elements k, k+1 (k even)
__device__ __forceinline__ void store_f16x2_2(void* row, int64_t k, float x0, float x1) {
    f16x2 hi, lo;
    split2(x0, x1, hi, lo);
    char* p = reinterpret_cast<char*>(row) + f16x2_off(k);
    *reinterpret_cast<f16x2*>(p) = hi;
    *reinterpret_cast<f16x2*>(p + 16) = lo;
}

// Lanes l and l^32 each hold elements 4h..4h+3 (h = l >> 5) of two consecutive 8-element groups: x of group G, y of
// group G+1 (the 32x32 accumulator layout).  One v_permlane32_swap per register later lane h owns ALL of group G+h
// (x = elements 0..3, y = elements 4..7), so an F16X2 group leaves as one contiguous 32-B store instead of four 8-B ones.
__device__ __forceinline__ void pair_groups(float (&x)[4], float (&y)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[e]), __float_as_uint(y[e]), false, false);
        x[e] = __uint_as_float(sw[0]);
        y[e] = __uint_as_float(sw[1]);
    }
}
// store the 8 elements k..k+7 (k % 8 == 0) of an F16X2 row: x = elements 0..3, y = elements 4..7
__device__ __forceinline__ void store_f16x2_8(void* row, int64_t k, const float (&x)[4], const float (&y)[4]) {
    f16x4 h0, l0, h1, l1;
    split4(x, h0, l0);
    split4(y, h1, l1);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = h0[e]; hi[4 + e] = h1[e]; lo[e] = l0[e]; lo[4 + e] = l1[e]; }
    char* p = reinterpret_cast<char*>(row) + k * 4;
    *reinterpret_cast<f16x8*>(p) = hi;
    *reinterpret_cast<f16x8*>(p + 16) = lo;
}

// Branch-free erf for the GELU epilogue (the libm erff costs ~60 instructions per element with both range branches
// taken in a wave; this one ~22).  |x| <= 1: x * q(x^2);  1 < |x|: sign(x) * (1 - 2^(-p(min(|x|, 4)))).  Coefficients
// and their float32 check against scipy (max abs error 1.4e-7, ~2 ulp of 1) : scripts/fit_erf.py.
__device__ __forceinline__ float fast_erff(float x) {
    const float t = fabsf(x), s = x * x;
    float q = 7.882497448e-05f;
    q = fmaf(q, s, -8.018855006e-04f);
    q = fmaf(q, s, 5.189312156e-03f);
    q = fmaf(q, s, -2.685432881e-02f);
    q = fmaf(q, s, 1.128359735e-01f);
    q = fmaf(q, s, -3.761262596e-01f);
    q = fmaf(q, s, 1.128379107e+00f);
    const float tc = fminf(t, 4.0f);
    float p = -2.328598612e-06f;
    p = fmaf(p, tc, 6.577336899e-05f);
    p = fmaf(p, tc, -8.551856736e-04f);
    p = fmaf(p, tc, 6.838695146e-03f);
    p = fmaf(p, tc, -3.803624585e-02f);
    p = fmaf(p, tc, 1.586650759e-01f);
    p = fmaf(p, tc, 9.116925001e-01f);
    p = fmaf(p, tc, 1.630485892e+00f);
    p = fmaf(p, tc, -4.374439013e-04f);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(-p), x);
    return t < 1.0f ? x * q : big;
}

// The same erf on two values per instruction (v_pk_fma_f32 / v_pk_mul_f32: the fp32 vector rate doubles on packed
// operands), same operations in the same order per element - bitwise the scalar function's results.
__device__ __forceinline__ f32x2 fast_erff2(f32x2 x) {
    const f32x2 t = __builtin_elementwise_abs(x), s = x * x;
    f32x2 q = 7.882497448e-05f;
    q = __builtin_elementwise_fma(q, s, (f32x2)(-8.018855006e-04f));
    q = __builtin_elementwise_fma(q, s, (f32x2)(5.189312156e-03f));
    q = __builtin_elementwise_fma(q, s, (f32x2)(-2.685432881e-02f));
    q = __builtin_elementwise_fma(q, s, (f32x2)(1.128359735e-01f));
    q = __builtin_elementwise_fma(q, s, (f32x2)(-3.761262596e-01f));
    q = __builtin_elementwise_fma(q, s, (f32x2)(1.128379107e+00f));
    const f32x2 tc = __builtin_elementwise_min(t, (f32x2)(4.0f));
    f32x2 p = -2.328598612e-06f;
    p = __builtin_elementwise_fma(p, tc, (f32x2)(6.577336899e-05f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(-8.551856736e-04f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(6.838695146e-03f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(-3.803624585e-02f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(1.586650759e-01f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(9.116925001e-01f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(1.630485892e+00f));
    p = __builtin_elementwise_fma(p, tc, (f32x2)(-4.374439013e-04f));
    const f32x2 xq = x * q;
    f32x2 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(-p[i]), x[i]);
        r[i] = t[i] < 1.0f ? xq[i] : big;
    }
    return r;
}
// exact-erf GELU (nn.GELU default) of four values, in place
__device__ __forceinline__ void gelu4(float (&t)[4]) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const f32x2 x = {t[e], t[e + 1]};
        const f32x2 y = (0.5f * x) * (1.0f + fast_erff2(x * 0.70710678118654752440f));
        t[e] = y[0];
        t[e + 1] = y[1];
    }
}

// LDS image of a 128-B stage row (four 8-k groups x [hi | lo]) for the fragment reads of v_mfma_f32_16x16x32_f16 (lane =
// (row & 15, k-group)): chunk c = 2 kg + x (x = 0 hi / 1 lo) of a row lives in the 16-B slot c ^ swz(row), swz(row) = (row & 7) ^
// ((row >> 3) & 1).  Found by exhaustive search over the XOR swizzles that are linear in the row bits (round 4,
// profiles/r04_lds_slot_search.txt), for BOTH users of the image:
//   * the ds_read_b128 fragment reads - the hardware serves them in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...;
//     MI355X_MICROARCH.md, LDS), and the 16 lanes of every group must hit 16 different slots of the 256-B bank row: conflict-free
//     for rows of 128 B (GEMM stages), of 256 B with the half select by row parity (K of the fused kernel) and of 896 B (V^T);
//   * the ds_write_b128 stores of the fused kernel's K / V^T conversion - eight CONTIGUOUS lanes = eight consecutive rows of one
//     (kg, x), bank row 128 B for stores: the eight slots must differ.  The round-2/3 image (a rotation by row bit 3, an XOR by row
//     bit 1) served the reads but put those eight lanes on TWO slots - a 4-way conflict on every K / V^T store, the
//     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.17 of profiles/r03_pmc_sq_counters.txt.
#ifdef SM_M16_SLOT_R3  // the previous image, for A/B builds only (build.py --variant=slot_r3 -DSM_M16_SLOT_R3)
__device__ __forceinline__ int m16_slot(int row, int kg, int x) { return 2 * ((kg + 2 * ((row >> 3) & 1)) & 3) + (x ^ ((row >> 1) & 1)); }
__device__ __forceinline__ int m16_chunk_of_slot(int row, int p) { return 2 * (((p >> 1) + 2 * ((row >> 3) & 1)) & 3) + ((p & 1) ^ ((row >> 1) & 1)); }
#else
__device__ __forceinline__ int m16_swz(int row) { return (row & 7) ^ ((row >> 3) & 1); }
__device__ __forceinline__ int m16_slot(int row, int kg, int x) { return (2 * kg + x) ^ m16_swz(row); }
// inverse, for the LDS-DMA source address: the chunk (2 kg + x) that lives in slot p of `row` (an XOR is its own inverse)
__device__ __forceinline__ int m16_chunk_of_slot(int row, int p) { return p ^ m16_swz(row); }
#endif

// row of a 32x32 MFMA accumulator held in register v by lane-half h  (cdna_hip_programming.md section 3)
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// timing taps of sm_forward_timing (forward.hip) for the library's other launch sequences: an event pair on `stream` around what runs
// between tap_begin and tap_end while timing is on (handle -1 otherwise); read back, summed by name, with sm_forward_timing_read
int tap_begin(void* stream, const char* name, double flops, double bytes);
void tap_end(int handle);
struct TapGuard {
    int h;
    TapGuard(void* stream, const char* name, double flops = 0.0, double bytes = 0.0) : h(tap_begin(stream, name, flops, bytes)) {}
    ~TapGuard() { tap_end(h); }
};

}  // namespace sm

const char* sm_qkv_attention_kernel_name(int mfma_terms);  // qkv_attention.hip: rocprofv3 name of the fused kernel that is launched
