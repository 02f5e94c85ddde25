// fp32-grade GEMM on the f16 matrix cores: C = epilogue(A W^T + bias) with BOTH operands in the "F16X2" split format.
//
// Why: the 1e-4 logit gate rules out bf16/f16 operands (7e-2 off), and the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
// is 1/16 of the f16 rate.  Splitting every fp32 value x into two halves, hi = f16(x) and lo = f16((x - hi) * 2^11),
// keeps 22 significant bits; x*y ~= hi_x*hi_y + 2^-11 (hi_x*lo_y + lo_x*hi_y) needs THREE f16 MFMAs with two fp32
// accumulators (main, cross) and drops only the 2^-22 lo*lo term.  Simulated through the whole network this is as
// close to the fp64 truth as the fp32 reference itself (calib: 1.26e-5 vs 1.57e-5; soft: 6.0e-5 vs 7.9e-5), at
// 3 x 32 cycles per 32x32x16 block instead of 8 x 64: 5.3x the fp32-MFMA rate (833 TFLOP/s equivalent at spec).
//
// F16X2 layout of a row of K floats (K % 8 == 0): for every group of 8 consecutive k, 16 bytes of hi halves followed
// by 16 bytes of lo halves.  A row therefore occupies exactly the bytes of the fp32 row (4 B / element), a 32-k tile
// of a row is one 128-B line made of eight 16-B chunks, and the LDS-DMA ring / XOR swizzle of gemm.hip carry over
// unchanged.  Producers write this format directly (LayerNorm, attention, the GELU/ReLU epilogues, im2col, ...).
//
// MFMA orientation: the WEIGHT fragment is the A operand and the ACTIVATION fragment the B operand, so a lane ends
// up with ONE output row m and four consecutive output columns n per register quad (whole 16-B pieces of a row, and
// after one lane-pair exchange whole 32-B F16X2 groups); the epilogue then turns each wave's block through LDS so the
// global accesses are contiguous row segments.
//
// Measured context (MI355X): a bare loop of this MFMA sustains 1.6 PFLOP/s (0.95-1.35 with one
// ds_read_b128 per MFMA), and this kernel's LDS-read + MFMA loop alone runs at that rate; what the kernel adds on top
// - LDS-DMA issue, prologue latency, the epilogue's HBM writes - overlaps only across workgroups, so occupancy
// (__launch_bounds__) and the tile's DMA bytes per FLOP decide the default shapes.
#include "common.h"
#include <type_traits>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace sm {

constexpr int HBK = 32;  // k per pipeline stage (two 16-deep MFMA steps)

__device__ __forceinline__ void dma16h(const void* gsrc, unsigned lds_off) { lds_dma16(gsrc, lds_off); }  // common.h
template <int N>
__device__ __forceinline__ void wait_vmcnt_h() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NWM x NWN waves per workgroup, each owning a (BM/NWM) x (BN/NWN) block of the tile as TM x TN 32x32 accumulators.
// The kernel is fed from L2 at a roughly fixed rate per CU (~50 GB/s measured), so what a tile shape buys is MACs per
// byte staged: BM*BN/(BM+BN) - 43 for 128x64, 85 for 256x128.
// Rings: W tiles NST deep, A tiles NST + AX deep (AX = 0 or 1).  Weights are re-read by every workgroup and come from
// L2; activation panels come from HBM / the Infinity Cache with several times the latency, so the LDS that two
// workgroups per CU leave free (2 x 80 of 160 KiB with the 128x128 tile) can buy the A stream one more K-tile of
// run-ahead.  Measured with three batches in flight: 1 % SLOWER than AX = 0 (a CU whose LDS is full takes no workgroup
// of another stream's kernel), so the shipped shapes use AX = 0; SM_F16X2_VARIANT=a3 selects the AX = 1 variant.
template <int BM, int BN, int NST, int NWM, int NWN, int MINB, int AX>
__global__ __launch_bounds__(NWM * NWN * 64, MINB) void gemm_f16x2_kernel(sm_gemm_args g) {
    constexpr int NW = NWM * NWN, WTM = BM / NWM, WTN = BN / NWN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_INST = BM / 8 / NW, W_INST = BN / 8 / NW;  // 1-KiB LDS-DMA pieces (8 rows x 128 B) per wave and stage
    static_assert(A_INST * 8 * NW == BM && W_INST * 8 * NW == BN && TM * 32 * NWM == BM && TN * 32 * NWN == BN, "tile split");
    constexpr int NI = A_INST + W_INST;
    constexpr int NSTA = NST + AX;
    constexpr int A_STAGE = BM * 128, W_STAGE = BN * 128;  // bytes per ring slot (128 B per row per 32-k tile)
    constexpr int W_RING = NSTA * A_STAGE;                 // the W ring follows the A ring
    constexpr int RING_BYTES = NSTA * A_STAGE + NST * W_STAGE;
    static_assert(AX == 0 || AX == 1, "A runs at most one tile further ahead than W");
    extern __shared__ __attribute__((aligned(16))) char smemh[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order (1-D grid): workgroups are dealt round-robin over the 8 XCDs (id % 8 labels the XCD, each
    // with a private 4 MiB L2).  Give every XCD a CONTIGUOUS range of logical tiles (m-tile major, n-tile minor) so all
    // n-tiles of an A row-panel run on one XCD and re-read it from that L2 instead of the Infinity Cache.
    int tile_id = blockIdx.x;
    const int ntn = (g.N + BN - 1) / BN;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = tile_id & 7, slot = tile_id >> 3;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int n0 = (tile_id % ntn) * BN, m0 = (tile_id / ntn) * BM;
    const int M = g.M, N = g.N;
    const int split = g.split_k > 1 ? g.split_k : 1;
    const int64_t bz = split > 1 ? 0 : (int64_t)blockIdx.z;
    const int nk = g.K / HBK / split;
    const int k_begin = split > 1 ? (int)blockIdx.z * nk * HBK : 0;
    // operands are F16X2: same element count / strides as the fp32 tensors they mirror (4 B per element)
    const char* A = reinterpret_cast<const char*>(((g.alt_from_n > 0 && n0 >= g.alt_from_n) ? g.A_alt : g.A) + bz * g.strideA + k_begin);
    const char* W = reinterpret_cast<const char*>(g.W + bz * g.strideW + k_begin);

    const char* a_src[A_INST];
    const char* w_src[W_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = (wave * A_INST + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = A + ((int64_t)gm * g.lda) * 4 + c * 16;
    }
#pragma unroll
    for (int i = 0; i < W_INST; ++i) {
        const int row = (wave * W_INST + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;
        w_src[i] = W + ((int64_t)gn * g.ldw) * 4 + c * 16;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smemh;
    auto issue_a = [&](int kt, int slot) {
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds_base + slot * A_STAGE + wave * A_INST * 1024);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) dma16h(a_src[i] + kt * 128, sa + i * 1024);
    };
    auto issue_w = [&](int kt, int slot) {
        const unsigned sw = __builtin_amdgcn_readfirstlane(lds_base + W_RING + slot * W_STAGE + wave * W_INST * 1024);
#pragma unroll
        for (int i = 0; i < W_INST; ++i) dma16h(w_src[i] + kt * 128, sw + i * 1024);
    };
    // one pipeline step: the W tile NST-1 ahead, then the A tile NSTA-1 ahead (this order is what the counted wait below
    // relies on); tiles past the end re-fetch the last one so that every step issues the same number of pieces
    auto issue_step = [&](int kt) {
        const int tw = kt + NST - 1, ta = kt + NSTA - 1;
        if (tw >= 0) issue_w(tw < nk ? tw : nk - 1, tw % NST);
        if (ta >= 0) issue_a(ta < nk ? ta : nk - 1, ta % NSTA);
    };

    // fragment byte offsets inside a 128-B row: k16-step s, lane half h -> k-group 2s+h -> chunks 2(2s+h) [hi], +1 [lo]
    const int swz = (r >> 1) & 7;
    int off_hi[2], off_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        off_hi[s] = ((2 * (2 * s + h)) ^ swz) * 16;
        off_lo[s] = ((2 * (2 * s + h) + 1) ^ swz) * 16;
    }
    const int a_row = (wm * WTM + r) * 128;
    const int w_row = (wn * WTN + r) * 128;

    f32x16 acc[TM][TN], crs[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[i][j][v] = 0.f; crs[i][j][v] = 0.f; }

#pragma unroll
    for (int v = -(NSTA - 1); v < 0; ++v) issue_step(v);  // prologue = the virtual steps before kt = 0

#ifdef SM_TUNING  // timing-only ablations exist only in the tuning build (build.py --tuning), never in the product library
    const int ablate = g.patch_n <= -2 ? -g.patch_n : 0;  // 2 = no MFMA, 3 = no DMA in the loop, 4 = no epilogue, 5 = 3 + 4, 6 = 5 without the barrier
#else
    constexpr int ablate = 0;
#endif
    for (int kt = 0; kt < nk; ++kt) {
        if (ablate != 6) {
            // tiles kt of both rings have landed; what may still fly was issued after the later of the two: AX = 0 the
            // NST-2 steps since; AX = 1 the A tile issued right behind W(kt) plus those steps
            wait_vmcnt_h<(NST - 2) * NI + AX * A_INST>();
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ablate != 3 && ablate < 5) issue_step(kt);
        if (ablate == 2) continue;
        const char* sta = smemh + (kt % NSTA) * A_STAGE;
        const char* stw = smemh + W_RING + (kt % NST) * W_STAGE;
        // all fragment reads of the stage are issued up front: the second 16-k step's reads land under the first
        // step's MFMAs (LDS returns in order, so the compiler waits with a counted lgkmcnt for the first half only)
        f16x8 ah[2][TM], al[2][TM], wh[2][TN], wl[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[s][i] = *reinterpret_cast<const f16x8*>(sta + a_row + i * 32 * 128 + off_hi[s]);
                al[s][i] = *reinterpret_cast<const f16x8*>(sta + a_row + i * 32 * 128 + off_lo[s]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wh[s][j] = *reinterpret_cast<const f16x8*>(stw + w_row + j * 32 * 128 + off_hi[s]);
                wl[s][j] = *reinterpret_cast<const f16x8*>(stw + w_row + j * 32 * 128 + off_lo[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // D[n][m]: weights are the MFMA A operand (rows = n), activations the B operand (cols = m)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][j], ah[s][i], acc[i][j], 0, 0, 0);
                    crs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][j], al[s][i], crs[i][j], 0, 0, 0);
                    crs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s][j], ah[s][i], crs[i][j], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    wait_vmcnt_h<0>();
    if (ablate >= 4) return;  // timing-only: no epilogue

    float* C = g.C + (split > 1 ? (int64_t)blockIdx.z : bz) * g.strideC;
    const bool out_split = g.patch_n < 0;  // out-format flag travels in the sign of patch_n for non-PATCH epilogues

    // ---- epilogue ----------------------------------------------------------------------------------------------------
    // In the accumulator layout a lane owns one output ROW: stored from there, a wave-wide store touches 32 rows with
    // 16-B pieces (32 partial cache lines per instruction), and with three batches in flight those partial writes cost
    // as much as the whole K loop (removing the epilogue: +40 % images/s).  So each wave turns its 32 x WTN blocks
    // through a private piece of the (now idle) LDS ring: values - bias, GELU / ReLU applied, already in their final
    // fp32 or F16X2 byte layout - are written row-major, read back with consecutive lanes on consecutive 16-B pieces,
    // and leave as contiguous row segments (WTN * 4 B = 256 B for the 128x128 tile); the residual / position-embedding
    // rows are read the same way.  Same arithmetic, same order: bit-identical results.
    constexpr int EPLD = WTN * 4 + 16;            // bytes per staged row (+16: conflict-free b128 column writes)
    constexpr int PIECES = WTN / 4;               // 16-B pieces per row
    static_assert(NW * 32 * EPLD <= RING_BYTES, "epilogue staging must fit in the ring");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading the ring
    char* ep = smemh + wave * (32 * EPLD);

    if constexpr (BN == SM_EMBED) {
        // ---- residual + LayerNorm epilogue (SM_EPI_RESIDUAL_LN): the tile spans whole rows (N = BN = 384), so the
        // pre-norm of the NEXT block runs here instead of in its own launch.  All waves stage their fp32 blocks
        // (A W^T + bias) row-major in the idle ring, then every 16 lanes take one row exactly as layernorm384_kernel
        // does (three 8-element groups per lane, two-pass variance, 4-step butterflies): x = R + value goes to C
        // (fp32 residual stream), LayerNorm(x) to C2 in F16X2 - bit-identical to the unfused pair of launches.
        if (g.epilogue == SM_EPI_RESIDUAL_LN) {
            constexpr int LNLD = BN * 4 + 16;
            static_assert(BM * LNLD <= RING_BYTES, "row staging must fit in the ring");
            static_assert(TM == 1, "one 32-row block per wave");
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn * WTN + j * 32 + 8 * q + 4 * h;
                    float4 val;
                    float* vp = &val.x;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        vp[e] = (acc[0][j][4 * q + e] + crs[0][j][4 * q + e] * (1.0f / 2048.0f)) + (g.bias ? g.bias[nl + e] : 0.f);
                    *reinterpret_cast<float4*>(smemh + (wm * WTM + r) * LNLD + nl * 4) = val;
                }
            __syncthreads();
            const int l16 = lane & 15;
#pragma unroll 1
            for (int it = 0; it < BM / (NW * 4); ++it) {
                const int row = (it * NW + wave) * 4 + (lane >> 4);
                const int m = m0 + row;
                const bool live = m < M;
                const int mm = live ? m : M - 1;  // dead groups shadow the last row (they join the shuffles, never store)
                const char* sr = smemh + row * LNLD;
                const float* rr = (g.R + bz * g.strideR) + (int64_t)mm * g.ldr;
                float v[3][8];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int k = (l16 + 16 * i) * 8;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float4 p = *reinterpret_cast<const float4*>(sr + (k + 4 * hh) * 4);
                        const float4 t = *reinterpret_cast<const float4*>(rr + k + 4 * hh);
                        v[i][4 * hh + 0] = t.x + p.x; v[i][4 * hh + 1] = t.y + p.y;
                        v[i][4 * hh + 2] = t.z + p.z; v[i][4 * hh + 3] = t.w + p.w;
                    }
                }
                float* cr = C + (int64_t)mm * g.ldc;
                if (live) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int k = (l16 + 16 * i) * 8;
                        *reinterpret_cast<float4*>(cr + k) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
                        *reinterpret_cast<float4*>(cr + k + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
                    }
                }
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) sum += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
                const float mean = sum * (1.0f / 384.0f);
                float qv = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[i][e] -= mean;
                        qv += v[i][e] * v[i][e];
                    }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) qv += __shfl_xor(qv, o, 16);
                const float rstd = 1.0f / sqrtf(qv * (1.0f / 384.0f) + g.ln_eps);
                if (live) {
                    float* yr = (g.C2 + bz * g.strideC) + (int64_t)m * g.ldc;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int k = (l16 + 16 * i) * 8;
                        float o[8];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const float4 gm = *reinterpret_cast<const float4*>(g.ln_gamma + k + 4 * hh);
                            const float4 bt = *reinterpret_cast<const float4*>(g.ln_beta + k + 4 * hh);
                            o[4 * hh + 0] = v[i][4 * hh + 0] * rstd * gm.x + bt.x; o[4 * hh + 1] = v[i][4 * hh + 1] * rstd * gm.y + bt.y;
                            o[4 * hh + 2] = v[i][4 * hh + 2] * rstd * gm.z + bt.z; o[4 * hh + 3] = v[i][4 * hh + 3] * rstd * gm.w + bt.w;
                        }
                        const float (&o0)[4] = *reinterpret_cast<const float (*)[4]>(&o[0]);
                        const float (&o1)[4] = *reinterpret_cast<const float (*)[4]>(&o[4]);
                        store_f16x2_8(yr, k, o0, o1);
                    }
                }
            }
            return;
        }
    }

    auto run = [&](auto epi_tag, auto fmt_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr bool F = decltype(fmt_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // phase A: accumulator layout -> staged rows
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (F) {
                    // F16X2 output: lanes l / l^32 trade halves so each owns one whole 32-B group (N % 8 == 0)
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        float x[4], y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int nx = n0 + wn * WTN + j * 32 + 8 * q + 4 * h + e, ny = nx + 8;
                            x[e] = (acc[i][j][4 * q + e] + crs[i][j][4 * q + e] * (1.0f / 2048.0f)) + ((g.bias && nx < N) ? g.bias[nx] : 0.f);
                            y[e] = (acc[i][j][4 * q + 4 + e] + crs[i][j][4 * q + 4 + e] * (1.0f / 2048.0f)) + ((g.bias && ny < N) ? g.bias[ny] : 0.f);
                            if constexpr (EPI == SM_EPI_RELU) {
                                x[e] = fmaxf(x[e], 0.f);
                                y[e] = fmaxf(y[e], 0.f);
                            }
                        }
                        if constexpr (EPI == SM_EPI_GELU) { gelu4(x); gelu4(y); }
                        pair_groups(x, y);
                        store_f16x2_8(ep + r * EPLD, j * 32 + 8 * (q + h), x, y);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = j * 32 + 8 * q + 4 * h, n = n0 + wn * WTN + nl;
                        float4 val;
                        float* vp = &val.x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float b = (g.bias && n < N) ? g.bias[n + e] : 0.f;
                            float t = (acc[i][j][4 * q + e] + crs[i][j][4 * q + e] * (1.0f / 2048.0f)) + b;
                            if constexpr (EPI == SM_EPI_GELU) t = 0.5f * t * (1.0f + fast_erff(t * 0.70710678118654752440f));
                            else if constexpr (EPI == SM_EPI_RELU) t = fmaxf(t, 0.f);
                            vp[e] = t;
                        }
                        *reinterpret_cast<float4*>(ep + r * EPLD + nl * 4) = val;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes have landed (LDS is in-order)
            // phase B: staged rows -> global, lane = (row, 16-B piece)
#pragma unroll
            for (int it = 0; it < 32 * PIECES / 64; ++it) {
                const int idx = it * 64 + lane, row = idx / PIECES, pc = idx % PIECES;
                int m = m0 + wm * WTM + i * 32 + row;
                const int n = n0 + wn * WTN + pc * 4;
                if (m < M && n < N) {
                    float4 v = *reinterpret_cast<const float4*>(ep + row * EPLD + pc * 16);
                    if constexpr (EPI == SM_EPI_RESIDUAL) {
                        const float4 rr = *reinterpret_cast<const float4*>((g.R + bz * g.strideR) + (int64_t)m * g.ldr + n);
                        v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w;
                    } else if constexpr (EPI == SM_EPI_SIGMOID2) {
                        float4 sg;
                        sg.x = 1.0f / (1.0f + expf(-v.x)); sg.y = 1.0f / (1.0f + expf(-v.y));
                        sg.z = 1.0f / (1.0f + expf(-v.z)); sg.w = 1.0f / (1.0f + expf(-v.w));
                        *reinterpret_cast<float4*>((g.C2 + bz * g.strideC) + (int64_t)m * g.ldc + n) = sg;
                    } else if constexpr (EPI == SM_EPI_PATCH) {
                        const int img = m / g.patch_n, p = m - img * g.patch_n;
                        const float4 rr = *reinterpret_cast<const float4*>(g.R + (int64_t)(1 + p) * g.ldr + n);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                        m = img * (g.patch_n + 1) + 1 + p;
                    }
                    *reinterpret_cast<float4*>(C + (int64_t)m * g.ldc + n) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next block overwrites the rows
        }
    };
    using T = std::true_type;
    using Fa = std::false_type;
    switch (g.epilogue) {
        case SM_EPI_GELU: out_split ? run(std::integral_constant<int, SM_EPI_GELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_GELU>{}, Fa{}); break;
        case SM_EPI_RELU: out_split ? run(std::integral_constant<int, SM_EPI_RELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_RELU>{}, Fa{}); break;
        case SM_EPI_RESIDUAL: run(std::integral_constant<int, SM_EPI_RESIDUAL>{}, Fa{}); break;
        case SM_EPI_SIGMOID2: run(std::integral_constant<int, SM_EPI_SIGMOID2>{}, Fa{}); break;
        case SM_EPI_PATCH: run(std::integral_constant<int, SM_EPI_PATCH>{}, Fa{}); break;
        default: out_split ? run(std::integral_constant<int, SM_EPI_BIAS>{}, T{}) : run(std::integral_constant<int, SM_EPI_BIAS>{}, Fa{}); break;
    }
}

// fp32 (rows, K) -> F16X2, one thread per group of 8
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                          int64_t ldd, int K, int64_t total_groups) {
    const int gpr = K / 8;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total_groups; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / gpr;
        const int gidx = (int)(t - row * gpr);
        const float4 a = *reinterpret_cast<const float4*>(src + row * lds_ + gidx * 8);
        const float4 b = *reinterpret_cast<const float4*>(src + row * lds_ + gidx * 8 + 4);
        const float x0[4] = {a.x, a.y, a.z, a.w}, x1[4] = {b.x, b.y, b.z, b.w};
        store_f16x2_4(dst + row * ldd, gidx * 8, x0);
        store_f16x2_4(dst + row * ldd, gidx * 8 + 4, x1);
    }
}

template <int BM, int BN, int NST, int NWM = 2, int NWN = 2, int MINB = 1, int AX = 0>
static int launch_gemm_h(const sm_gemm_args& g, hipStream_t st) {
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.split_k > 1 ? g.split_k : g.batch);
    constexpr size_t lds = (size_t)((NST + AX) * BM + NST * BN) * 128;
    if (lds > 64 * 1024) {
        static std::once_flag attr_once;  // per instantiation; safe from several host threads
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_kernel<BM, BN, NST, NWM, NWN, MINB, AX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipGetLastError();
        });
    }
    hipLaunchKernelGGL((gemm_f16x2_kernel<BM, BN, NST, NWM, NWN, MINB, AX>), grid, dim3(NWM * NWN * 64), lds, st, g);
    return check_launch("sm_gemm_f16x2");
}

}  // namespace sm

extern "C" int sm_split_f16x2(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int32_t K,
                              void* stream) {
    SM_REQUIRE(src && dst && rows > 0 && K > 0 && K % 8 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= K &&
                   ld_dst >= K,
               "sm_split_f16x2: bad arguments (K %% 8 == 0, strides %% 4 == 0)");
    const int64_t groups = rows * (K / 8);
    int64_t grid = (groups + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sm::split_f16x2_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst,
                       K, groups);
    return sm::check_launch("sm_split_f16x2");
}

extern "C" int sm_gemm_f16x2_tile(const sm_gemm_args* g, int out_f16x2, int bm, int bn, void* stream) {
    SM_REQUIRE(g && g->A && g->W && g->C, "sm_gemm_f16x2: null pointer");
    SM_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0 && g->batch > 0 && g->K % sm::HBK == 0, "sm_gemm_f16x2: bad shape");
    SM_REQUIRE(g->N % 4 == 0 && g->ldc % 4 == 0 && ((uintptr_t)g->C % 16 == 0), "sm_gemm_f16x2: N, ldc must be multiples of 4");
    SM_REQUIRE(g->lda % 8 == 0 && g->ldw % 8 == 0 && ((uintptr_t)g->A % 16 == 0) && ((uintptr_t)g->W % 16 == 0),
               "sm_gemm_f16x2: lda/ldw must be multiples of 8, pointers 16-B aligned");
    SM_REQUIRE(g->epilogue >= 0 && g->epilogue <= SM_EPI_RESIDUAL_LN, "sm_gemm_f16x2: bad epilogue");
    if (g->epilogue == SM_EPI_RESIDUAL_LN)
        SM_REQUIRE(bm == 64 && bn == 384 && g->N == SM_EMBED && g->R && g->C2 && g->ln_gamma && g->ln_beta && !out_f16x2 &&
                       g->ldr % 4 == 0 && g->ldc % 8 == 0 && !(g->split_k > 1),
                   "sm_gemm_f16x2: RESIDUAL_LN needs the 64x384 tile, N == 384, R, C2, ln_gamma, ln_beta");
    if (out_f16x2)
        SM_REQUIRE(g->N % 8 == 0 && g->ldc % 8 == 0 && (g->epilogue == SM_EPI_BIAS || g->epilogue == SM_EPI_GELU ||
                                                      g->epilogue == SM_EPI_RELU) && !(g->split_k > 1),
                   "sm_gemm_f16x2: F16X2 output needs N %% 8 == 0 and a BIAS/GELU/RELU epilogue");
    if (g->epilogue == SM_EPI_RESIDUAL) SM_REQUIRE(g->R && g->ldr % 4 == 0, "sm_gemm_f16x2: residual needs R, ldr %% 4 == 0");
    if (g->epilogue == SM_EPI_SIGMOID2) SM_REQUIRE(g->C2, "sm_gemm_f16x2: SIGMOID2 needs C2");
    if (g->epilogue == SM_EPI_PATCH) SM_REQUIRE(g->R && g->patch_n > 0 && g->batch == 1, "sm_gemm_f16x2: PATCH needs R/patch_n");
    if (g->alt_from_n > 0) SM_REQUIRE(g->A_alt && g->alt_from_n % 128 == 0, "sm_gemm_f16x2: bad A_alt");
    if (g->split_k > 1)
        SM_REQUIRE(g->batch == 1 && g->epilogue == SM_EPI_BIAS && !g->bias && (g->K / sm::HBK) % g->split_k == 0,
                   "sm_gemm_f16x2: bad split_k");
    sm_gemm_args a = *g;
    if (out_f16x2) a.patch_n = -1;
#ifdef SM_TUNING
    if (const char* ab = getenv("SM_F16X2_ABLATE")) a.patch_n = -atoi(ab);  // timing-only (wrong results): tuning build only
#endif
    hipStream_t st = (hipStream_t)stream;
    // Occupancy is what this kernel lives on (scripts/gemm_f16x2_ablate.py): the DMA stream, the MFMA stream and the
    // epilogue of one workgroup only overlap with those of OTHER workgroups on the CU, so the default shapes are the
    // ones that fit three workgroups per CU - 48 KiB of LDS and, via __launch_bounds__(256, 3), <= 168 registers
    // (left alone hipcc spends 109 VGPR + 64 AGPR on the 128x64 tile = two per CU, and the kernel is 25 % slower).
    // tuning knobs (tuning build only): SM_F16X2_NST = ring depth (2..5 where instantiated), SM_F16X2_VARIANT = "a3" (128x128, four waves,
    // the A ring one tile deeper), "w4" (128x128 as four waves of 64x64), "w16" (256x128 as sixteen waves of 64x32)
#ifdef SM_TUNING
    const char* env = getenv("SM_F16X2_NST");
    const int nst = env ? atoi(env) : 0;
    const char* var = getenv("SM_F16X2_VARIANT");
#else
    constexpr int nst = 0;
    const char* var = nullptr;
#endif
    const bool var_a3 = var && !strcmp(var, "a3"), var_w4 = var && !strcmp(var, "w4");
    if (bm == 64 && bn == 384) return sm::launch_gemm_h<64, 384, 2, 2, 4, 1, 0>(a, st);  // full-row tile (LayerNorm epilogue)
    if (bm == 256 && bn == 128 && var && !strcmp(var, "w16")) return sm::launch_gemm_h<256, 128, 2, 4, 4, 1, 0>(a, st);  // 16 waves of 64x32
    if (bm == 256 && bn == 128) return nst == 2 ? sm::launch_gemm_h<256, 128, 2, 4, 2>(a, st) : sm::launch_gemm_h<256, 128, 3, 4, 2>(a, st);
    if (bm == 256 && bn == 64) return nst == 3 ? sm::launch_gemm_h<256, 64, 3, 4, 1>(a, st) : sm::launch_gemm_h<256, 64, 2, 4, 1>(a, st);
    if (bm == 128 && bn == 128 && var_a3) return sm::launch_gemm_h<128, 128, 2, 2, 2, 2, 1>(a, st);  // 80 KiB: -1 % end to end
    if (bm == 128 && bn == 128 && var_w4) return sm::launch_gemm_h<128, 128, 2, 2, 2, 2, 0>(a, st);  // -2.7 % end to end
    if (bm == 128 && bn == 128) return nst == 3 ? sm::launch_gemm_h<128, 128, 3>(a, st) : nst == 4 ? sm::launch_gemm_h<128, 128, 4>(a, st) : sm::launch_gemm_h<128, 128, 2, 2, 4, 2, 0>(a, st);
    if (bm == 128 && bn == 64) return nst == 3 ? sm::launch_gemm_h<128, 64, 3>(a, st) : nst == 4 ? sm::launch_gemm_h<128, 64, 4>(a, st) : sm::launch_gemm_h<128, 64, 2, 2, 2, 3>(a, st);
    if (bm == 64 && bn == 64) return nst == 4 ? sm::launch_gemm_h<64, 64, 4>(a, st) : nst == 5 ? sm::launch_gemm_h<64, 64, 5>(a, st) : nst == 2 ? sm::launch_gemm_h<64, 64, 2, 2, 2, 5>(a, st) : sm::launch_gemm_h<64, 64, 3, 2, 2, 3>(a, st);
    sm::set_error("sm_gemm_f16x2_tile: unsupported tile %dx%d", bm, bn);
    return SM_EINVAL;
}

extern "C" int sm_gemm_f16x2_pick_tile(const sm_gemm_args* g, int* bm, int* bn, int* nst) {
    SM_REQUIRE(g && bm && bn && nst, "sm_gemm_f16x2_pick_tile: null pointer");
    // Measured on MI355X (B=64 ViT-S/16 shapes).  A GEMM alone on the GPU is fastest as 128x64 tiles (three workgroups
    // per CU hide each other's prologue / epilogue), but the evaluator keeps three batches in flight on three streams
    // (streams.py), other kernels fill those gaps, and what counts is DMA instructions per FLOP: 128x128 tiles (two
    // workgroups per CU, 64 KiB of LDS each) gave +5 % end to end over 128x64, 256x128 +3 %, 64x64 -8 %.  The 128x128
    // tile runs as EIGHT waves of 64x32 (101 registers: both workgroups of a CU = 16 waves, four per SIMD): +2.7 % over
    // four waves of 64x64 - more waves to cover the DMA issue and the epilogue, although each re-reads more LDS per MFMA.
    const long nb = g->split_k > 1 ? g->split_k : g->batch;
    const long wg128 = (long)((g->M + 127) / 128) * ((g->N + 63) / 64) * nb;
    const long wg128x128 = (long)((g->M + 127) / 128) * ((g->N + 127) / 128) * nb;
    if (wg128x128 >= 256) { *bm = 128; *bn = 128; *nst = 2; }
    else if (wg128 >= 512) { *bm = 128; *bn = 64; *nst = 2; }
    else { *bm = 64; *bn = 64; *nst = 3; }
    // tuning knobs: "BMxBN" for the GEMMs with >= 512 workgroups, by output width (N >= 768 / narrower)
#ifdef SM_TUNING
    static const char* force_w = getenv("SM_F16X2_TILE_WIDE");
    static const char* force_n = getenv("SM_F16X2_TILE_NARROW");
#else
    const char* force_w = nullptr;
    const char* force_n = nullptr;
#endif
    const char* force = g->N >= 768 ? force_w : force_n;
    if (force && wg128 >= 512) {
        int fbm = 0, fbn = 0;
        if (sscanf(force, "%dx%d", &fbm, &fbn) == 2) { *bm = fbm; *bn = fbn; *nst = (fbm == 64) ? 3 : 2; }
    }
    return SM_OK;
}

extern "C" int sm_gemm_f16x2(const sm_gemm_args* g, int out_f16x2, void* stream) {
    int bm, bn, nst;
    int rc = sm_gemm_f16x2_pick_tile(g, &bm, &bn, &nst);
    if (rc) return rc;
    return sm_gemm_f16x2_tile(g, out_f16x2, bm, bn, stream);
}
