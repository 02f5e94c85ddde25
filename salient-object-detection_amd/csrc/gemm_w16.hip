// fp32-grade GEMM against a WEIGHT matrix on the f16 matrix cores with ONE fp32 accumulator: C = epilogue(A W^T + bias),
// A in the F16X2 split format (hi = f16(a), lo' = f16((a - hi) * 2^11), as every producer writes it), W in the
// "W16" format prepared once per checkpoint.
//
// Why a second weight format.  gemm_f16x2.hip evaluates a*w as hi*hi + 2^-11 (hi*lo' + lo'*hi) and therefore needs TWO
// accumulators (main, cross): 32 accumulator registers per 32x32 block.  That caps the tile a workgroup can own, and
// the kernel is fed through the texture path (LDS-DMA: ~16 cycles of the CU's single address unit per 1-KiB piece):
// at 128 x 128 the staging of both operands takes two thirds of the MFMA time and any stall shows.
// Weights are static, so they can be pre-scaled: W' = W * 2^s with s chosen per tensor so that max|W'| lies in
// [2^13, 2^14).  Then wl = f16(W' - f16(W')) is a NORMAL f16 number without its own 2^11 factor (or, for tiny weights,
// a subnormal whose absolute error is 2^-39 of the tensor's maximum), and
//     a * w' = ah*wh + ah*wl + al'*(wh * 2^-11)            (dropping al*wl, 2^-22 relative, as before)
// all three products in the SAME scale: one accumulator, three MFMAs.  whs = wh * 2^-11 is exact (wh >= 2^-3 after the
// scaling, far above the f16 subnormal range) and costs four v_pk_mul_f16 per weight fragment, hidden under the MFMAs.
// The epilogue multiplies the accumulator by 2^-s (exact) before the bias.  Half the accumulator registers buy a
// 256 x 128 tile per workgroup: 3/4 of the staged bytes per MFMA of the 128 x 128 tile at the same occupancy.
//
// Everything else follows gemm_f16x2.hip: LDS-DMA ring with counted vmcnt + one raw barrier per K-tile, XOR swizzle on
// the DMA source address and the fragment reads, weights as the MFMA A operand (a lane owns one output row), epilogue
// turned through the idle ring so global accesses are contiguous row segments, XCD-aware tile order.
#include "common.h"
#include <type_traits>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace sm {

template <int N>
__device__ __forceinline__ void wait_vmcnt_w() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KT = k per ring stage (16 or 32): a stage row is KT*4 bytes = CH 16-B chunks (per 8-k group: hi chunk, lo chunk).
// A 1-KiB LDS-DMA piece covers 1024 / (KT*4) rows.  Chunk c of row r sits at slot c ^ swz(r): with 128-B rows two rows
// share a 256-B bank row (swz = (r >> 1) & 7), with 64-B rows four do (swz = (r >> 2) & 3) - either way the 16 lanes of
// a ds_read_b128 group (distinct rows, one logical chunk) land on 16 different 16-B slots.
template <int KT>
__device__ __forceinline__ int stage_swz(int row) {
    if constexpr (KT == 32) return (row >> 1) & 7;
    else return (row >> 2) & 3;
}

#ifdef SM_TUNING  // the v_mfma_f32_32x32x16_f16 family of round 2 (software-pipelined, deep-ring forms): measured and rejected
// (the 16x16x32 kernels are 10-15 % faster alone, +4.4 % in the pipeline); kept in the tuning build as the comparison
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// WPS = waves per SIMD the register allocation must leave room for (second __launch_bounds__ argument).
// PIPE = 1: software-pipelined K loop - the fragments of stage kt+1 are read into a second register set and the next
// LDS-DMA pieces are issued BETWEEN the MFMA blocks of stage kt, so a wave's LDS latency and DMA issue time run under its
// own MFMAs instead of in front of them (needs NST >= 3: the slot refilled in iteration kt was consumed in kt-1).
template <int BM, int BN, int KT, int NST, int NWM, int NWN, int WPS, int PIPE = 0>
__global__ __launch_bounds__(NWM * NWN * 64, WPS) void gemm_w16_kernel(sm_gemm_args g) {
    constexpr int NW = NWM * NWN, WTM = BM / NWM, WTN = BN / NWN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWB = KT * 4, CH = KT / 4, RPP = 1024 / ROWB;       // bytes / chunks per stage row, rows per DMA piece
    constexpr int KS = KT / 16;                                          // 16-deep MFMA steps per stage
    constexpr int A_INST = BM / RPP / NW, W_INST = BN / RPP / NW;        // 1-KiB pieces per wave and stage
    static_assert(KT == 16 || KT == 32, "stage depth");
    static_assert(A_INST * RPP * NW == BM && W_INST * RPP * NW == BN && TM * 32 * NWM == BM && TN * 32 * NWN == BN, "tile split");
    static_assert(A_INST >= 1 && W_INST >= 1, "every wave issues at least one piece per operand");
    constexpr int NI = A_INST + W_INST;
    constexpr int A_STAGE = BM * ROWB, W_STAGE = BN * ROWB;
    constexpr int W_RING = NST * A_STAGE;
    constexpr int RING_BYTES = NST * (A_STAGE + W_STAGE);
    extern __shared__ __attribute__((aligned(16))) char smemw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware 1-D tile order: every XCD (id % 8) gets a contiguous range of logical tiles, m-tile major
    int tile_id = blockIdx.x;
    const int ntn = (g.N + BN - 1) / BN;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = tile_id & 7, slot = tile_id >> 3;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int n0 = (tile_id % ntn) * BN, m0 = (tile_id / ntn) * BM;
    const int M = g.M, N = g.N;
    const int split = g.split_k > 1 ? g.split_k : 1;
    const int nk = g.K / KT / split;
    const int k_begin = split > 1 ? (int)blockIdx.z * nk * KT : 0;
    const char* A = reinterpret_cast<const char*>(((g.alt_from_n > 0 && n0 >= g.alt_from_n) ? g.A_alt : g.A) + k_begin);
    const char* W = reinterpret_cast<const char*>(g.W + k_begin);

    const char* a_src[A_INST];
    const char* w_src[W_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = (wave * A_INST + i) * RPP + lane / CH;
        const int c = (lane % CH) ^ stage_swz<KT>(row);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = A + ((int64_t)gm * g.lda) * 4 + c * 16;
    }
#pragma unroll
    for (int i = 0; i < W_INST; ++i) {
        const int row = (wave * W_INST + i) * RPP + lane / CH;
        const int c = (lane % CH) ^ stage_swz<KT>(row);
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;
        w_src[i] = W + ((int64_t)gn * g.ldw) * 4 + c * 16;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smemw;
    auto issue_step = [&](int kt) {  // the tile NST-1 ahead of kt: W pieces, then A pieces (tiles past the end re-fetch the last)
        const int t = kt + NST - 1;
        if (t < 0) return;
        const int tt = t < nk ? t : nk - 1, slot = t % NST;
        const unsigned sw = __builtin_amdgcn_readfirstlane(lds_base + W_RING + slot * W_STAGE + wave * W_INST * 1024);
#pragma unroll
        for (int i = 0; i < W_INST; ++i) lds_dma16(w_src[i] + tt * ROWB, sw + i * 1024);
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds_base + slot * A_STAGE + wave * A_INST * 1024);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) lds_dma16(a_src[i] + tt * ROWB, sa + i * 1024);
    };

    // fragment byte offsets inside a stage row: 16-deep step s, lane half h -> k-group KS==2 ? 2s+h : h
    const int swz = stage_swz<KT>(r);
    int off_hi[KS], off_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        off_hi[s] = ((2 * (2 * s + h)) ^ swz) * 16;
        off_lo[s] = ((2 * (2 * s + h) + 1) ^ swz) * 16;
    }
    const int a_row = (wm * WTM + r) * ROWB;
    const int w_row = (wn * WTN + r) * ROWB;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

#pragma unroll
    for (int v = -(NST - 1); v < 0; ++v) issue_step(v);

    const f16x8 down = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                        (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};  // 2^-11
    if constexpr (PIPE) {
        static_assert(NST >= 3, "the pipelined loop refills the slot consumed one iteration earlier");
        struct Frag { f16x8 ah[KS][TM], al[KS][TM], wh[KS][TN], wl[KS][TN]; };
        constexpr int NB = KS * TN * TM;           // MFMA blocks (3 MFMAs each) per stage
        constexpr int NRD = KS * (TN + TM);        // hi/lo fragment pairs to read per stage
        constexpr int NITEM = NRD + NI;            // side work per stage: fragment pairs of the next stage, then DMA pieces
        constexpr int IPB = (NITEM + NB - 1) / NB; // side items placed behind each MFMA block
        auto read_pair = [&](Frag& f, int kt, auto item_tag) {
            constexpr int item = decltype(item_tag)::value;
            constexpr int s = item / (TN + TM), q = item % (TN + TM);
            if constexpr (q < TN) {
                const char* stw = smemw + W_RING + (kt % NST) * W_STAGE + w_row + q * 32 * ROWB;
                f.wh[s][q] = *reinterpret_cast<const f16x8*>(stw + off_hi[s]);
                f.wl[s][q] = *reinterpret_cast<const f16x8*>(stw + off_lo[s]);
            } else {
                const char* sta = smemw + (kt % NST) * A_STAGE + a_row + (q - TN) * 32 * ROWB;
                f.ah[s][q - TN] = *reinterpret_cast<const f16x8*>(sta + off_hi[s]);
                f.al[s][q - TN] = *reinterpret_cast<const f16x8*>(sta + off_lo[s]);
            }
        };
        auto issue_piece = [&](int kt, auto piece_tag) {  // piece p of the tile NST-1 ahead of kt: W pieces first
            constexpr int p = decltype(piece_tag)::value;
            const int t = kt + NST - 1, tt = t < nk ? t : nk - 1, slot = t % NST;
            if constexpr (p < W_INST)
                lds_dma16(w_src[p] + tt * ROWB, __builtin_amdgcn_readfirstlane(lds_base + W_RING + slot * W_STAGE + (wave * W_INST + p) * 1024));
            else
                lds_dma16(a_src[p - W_INST] + tt * ROWB, __builtin_amdgcn_readfirstlane(lds_base + slot * A_STAGE + (wave * A_INST + p - W_INST) * 1024));
        };
        auto body = [&](int kt, Frag& cur, Frag& nxt) {
            wait_vmcnt_w<(NST - 3) * NI>();  // tile kt+1 has landed (this wave's pieces)
            __builtin_amdgcn_s_barrier();    // ... for every wave; every wave has consumed tile kt-1 (its slot is refilled below)
            __builtin_amdgcn_sched_barrier(0);
            static_for<NB>([&](auto blk) {
                constexpr int b = decltype(blk)::value;
                constexpr int s = b / (TN * TM), j = (b / TM) % TN, i = b % TM;
                const f16x8 whs = cur.wh[s][j] * down;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh[s][j], cur.ah[s][i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl[s][j], cur.ah[s][i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whs, cur.al[s][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                static_for<IPB>([&](auto sub) {
                    constexpr int item = b * IPB + decltype(sub)::value;
                    if constexpr (item < NRD) read_pair(nxt, kt + 1, std::integral_constant<int, item>{});
                    else if constexpr (item < NITEM) issue_piece(kt, std::integral_constant<int, item - NRD>{});
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        Frag fa, fb;
        wait_vmcnt_w<(NST - 2) * NI>();  // tile 0 has landed
        __builtin_amdgcn_s_barrier();
        static_for<NRD>([&](auto it) { read_pair(fa, 0, it); });
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            body(kt, fa, fb);
            body(kt + 1, fb, fa);
        }
        if (kt < nk) body(kt, fa, fb);
    } else
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt_w<(NST - 2) * NI>();   // tile kt has landed (this wave's pieces); younger tiles may still fly
        __builtin_amdgcn_s_barrier();     // ... for every wave, and every wave is done reading the slot refilled next
        __builtin_amdgcn_sched_barrier(0);
        issue_step(kt);
        const char* sta = smemw + (kt % NST) * A_STAGE;
        const char* stw = smemw + W_RING + (kt % NST) * W_STAGE;
        f16x8 ah[KS][TM], al[KS][TM], wh[KS][TN], wl[KS][TN];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wh[s][j] = *reinterpret_cast<const f16x8*>(stw + w_row + j * 32 * ROWB + off_hi[s]);
                wl[s][j] = *reinterpret_cast<const f16x8*>(stw + w_row + j * 32 * ROWB + off_lo[s]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[s][i] = *reinterpret_cast<const f16x8*>(sta + a_row + i * 32 * ROWB + off_hi[s]);
                al[s][i] = *reinterpret_cast<const f16x8*>(sta + a_row + i * 32 * ROWB + off_lo[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f16x8 whs = wh[s][j] * down;  // exact: |wh| >= 2^-3 after the per-tensor scaling (v_pk_mul_f16 x 4)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    // D[n][m]: weights are the MFMA A operand (rows = n), activations the B operand (cols = m)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s][j], ah[s][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s][j], ah[s][i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whs, al[s][i], acc[i][j], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    wait_vmcnt_w<0>();

    float* C = g.C + (split > 1 ? (int64_t)blockIdx.z : 0) * g.strideC;
    const bool out_split = g.patch_n < 0;  // out-format flag travels in the sign of patch_n for non-PATCH epilogues
    const float ws = g.w_scale;            // 2^-s of the weight tensor (exact power of two)

    // ---- epilogue: each wave turns its 32 x WTN blocks through a private piece of the idle ring, so global accesses are
    // contiguous row segments (see gemm_f16x2.hip) ---------------------------------------------------------------------
    constexpr int EPLD = WTN * 4 + 16;
    constexpr int PIECES = WTN / 4;
    static_assert(NW * 32 * EPLD <= RING_BYTES, "epilogue staging must fit in the ring");
    __builtin_amdgcn_s_barrier();
    char* ep = smemw + wave * (32 * EPLD);

    auto run = [&](auto epi_tag, auto fmt_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr bool F = decltype(fmt_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (F) {
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        float x[4], y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int nx = n0 + wn * WTN + j * 32 + 8 * q + 4 * h + e, ny = nx + 8;
                            x[e] = acc[i][j][4 * q + e] * ws + ((g.bias && nx < N) ? g.bias[nx] : 0.f);
                            y[e] = acc[i][j][4 * q + 4 + e] * ws + ((g.bias && ny < N) ? g.bias[ny] : 0.f);
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
                            float t = acc[i][j][4 * q + e] * ws + b;
                            if constexpr (EPI == SM_EPI_GELU) t = 0.5f * t * (1.0f + fast_erff(t * 0.70710678118654752440f));
                            else if constexpr (EPI == SM_EPI_RELU) t = fmaxf(t, 0.f);
                            vp[e] = t;
                        }
                        *reinterpret_cast<float4*>(ep + r * EPLD + nl * 4) = val;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 32 * PIECES / 64; ++it) {
                const int idx = it * 64 + lane, row = idx / PIECES, pc = idx % PIECES;
                int m = m0 + wm * WTM + i * 32 + row;
                const int n = n0 + wn * WTN + pc * 4;
                if (m < M && n < N) {
                    float4 v = *reinterpret_cast<const float4*>(ep + row * EPLD + pc * 16);
                    if constexpr (EPI == SM_EPI_RESIDUAL) {
                        const float4 rr = *reinterpret_cast<const float4*>(g.R + (int64_t)m * g.ldr + n);
                        v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w;
                    } else if constexpr (EPI == SM_EPI_PATCH) {
                        const int img = m / g.patch_n, p = m - img * g.patch_n;
                        const float4 rr = *reinterpret_cast<const float4*>(g.R + (int64_t)(1 + p) * g.ldr + n);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                        m = img * (g.patch_n + 1) + 1 + p;
                    }
                    *reinterpret_cast<float4*>(C + (int64_t)m * g.ldc + n) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    using T = std::true_type;
    using Fa = std::false_type;
    switch (g.epilogue) {
        case SM_EPI_GELU: out_split ? run(std::integral_constant<int, SM_EPI_GELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_GELU>{}, Fa{}); break;
        case SM_EPI_RELU: out_split ? run(std::integral_constant<int, SM_EPI_RELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_RELU>{}, Fa{}); break;
        case SM_EPI_RESIDUAL: run(std::integral_constant<int, SM_EPI_RESIDUAL>{}, Fa{}); break;
        case SM_EPI_PATCH: run(std::integral_constant<int, SM_EPI_PATCH>{}, Fa{}); break;
        default: out_split ? run(std::integral_constant<int, SM_EPI_BIAS>{}, T{}) : run(std::integral_constant<int, SM_EPI_BIAS>{}, Fa{}); break;
    }
}

#endif  // SM_TUNING

// ---- 16x16x32 MFMA variant -------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_f16.  Why: with three batches in flight the chip sits at its power limit, and in
// an MFMA-dense loop on random operands the 16x16x32 shape sustains 12-14 % more FLOP/s than 32x32x16 at the same cycles
// per FLOP (scripts/mb/mfma_shape.hip: 1.75 vs 1.55 PFLOP/s; MI355X_MICROARCH.md, DVFS give-back item 7).
// A / B operands: lane l holds row (l & 15), k-group (l >> 4) of a 32-k stage - one 16-B hi chunk and one 16-B lo chunk per
// fragment, so a stage is ONE MFMA step.  C: lane l holds output row m = l & 15 and four consecutive n = 4 (l >> 4) + reg.
// LDS image of a 128-B stage row (four k-groups x [hi | lo]): the XOR swizzle of the 32x32 kernel puts two lanes of every
// ds_read_b128 group on one 16-B slot here (the 16 lanes of a group are 8 rows of k-group a and 8 rows of k-group a ^ 1), so
// the pair of k-group kg sits at pair position (kg + 2 * ((row >> 3) & 1)) % 4 and hi / lo swap places on rows with
// (row >> 1) & 1 - conflict-free for all four lane groups and both halves (brute-forced over the 4^8 x 2^8 layouts of this
// family).  The LDS-DMA destination is linear, so the permutation goes on the per-lane SOURCE address, as before.
typedef float f32x4v __attribute__((ext_vector_type(4)));


#ifdef SM_TUNING  // in-kernel stamps (tuning build only; a buffer nothing else reads): prologue / K loop / epilogue of a tile
__device__ unsigned long long g_gemm_stamps[2048 * 4];
__device__ int g_gemm_stamp_filter[3];  // (N, K, M) of the launches that stamp; N = 0: every launch (sm_gemm_stamp_filter)
// each workgroup keeps its stamps in registers and writes the record once, at the end: launches of several streams share the
// buffer, and a record must come from ONE workgroup
#define GEMM_STAMP(i)                                                                                                          \
    do {                                                                                                                       \
        asm volatile("" ::: "memory"); /* no load or store moves across a stamp */                                             \
        stamp_[i] = __builtin_amdgcn_s_memtime();                                                                              \
        asm volatile("" ::: "memory");                                                                                         \
    } while (0)
#define GEMM_STAMP_DECL unsigned long long stamp_[4] = {0, 0, 0, 0}
#define GEMM_STAMP_FLUSH                                                                                                         \
    do {                                                                                                                        \
        if (blockIdx.x < 2048 && tid == 0 && (g_gemm_stamp_filter[0] == 0 || (g.N == g_gemm_stamp_filter[0] && g.K == g_gemm_stamp_filter[1] && g.M == g_gemm_stamp_filter[2]))) \
            for (int i_ = 0; i_ < 4; ++i_) g_gemm_stamps[blockIdx.x * 4 + i_] = stamp_[i_];                                      \
    } while (0)
#else
#define GEMM_STAMP(i) do {} while (0)
#define GEMM_STAMP_DECL do {} while (0)
#define GEMM_STAMP_FLUSH do {} while (0)
#endif

// TERMS = 3: the fp32-grade product (wh*ah + wl*ah + whs*al').  TERMS = 1: "throughput mode"  - only wh*ah,
// plain f16 operands with fp32 accumulation: a DIAGNOSTIC of what the kernel structure reaches without the x3, never the
// metric (the results miss the 1e-4 gate by two orders of magnitude).  Same operand formats: the lo halves are staged and ignored.
// Ring feed.  Source addresses are a wave-uniform base (advanced per K-tile by scalar adds) + one 32-bit per-lane offset per
// piece.  SM_GEMM_LOADH=1 (experiment build, round 3): only the first half of the waves - the older wave(s) of every SIMD - issue
// the LDS-DMA pieces (twice as many each) so that the younger ones start their MFMAs at once.  It took 11 % off the fused
// QKV kernel's projection loop (qkv_attention.hip, shipped there) but nothing off these GEMMs: K loop of the 256 x 128 fc2 tile
// 102.5k cycles against 97.4k, pipeline 21.75k vs 21.75k images/s (profiles/r03_loader_half_ab.log) - four waves per SIMD
// already cover each other's issue time.  Default off.
#ifndef SM_GEMM_LOADH
#define SM_GEMM_LOADH 0
#endif
#ifndef SM_GEMM_LATE_HALF
#define SM_GEMM_LATE_HALF 0
#endif
template <int BM, int BN, int NST, int NWM, int NWN, int WPS, int TERMS = 3>
__global__ __launch_bounds__(NWM * NWN * 64, WPS) void gemm_w16m16_kernel(sm_gemm_args g) {
    constexpr int NW = NWM * NWN, WTM = BM / NWM, WTN = BN / NWN;
    constexpr int TM = WTM / 16, TN = WTN / 16;       // 16x16 tiles per wave
    constexpr int ROWB = 128;
    // waves that feed the ring: all of them, or the first half where a stage's 8-row pieces do not divide among all (the 256 x 192 tile)
    constexpr int NL = ((SM_GEMM_LOADH && NW >= 8) || (BN / 8) % NW != 0 || (BM / 8) % NW != 0) ? NW / 2 : NW;
    constexpr int A_INST = BM / 8 / NL, W_INST = BN / 8 / NL;
    static_assert(A_INST * 8 * NL == BM && W_INST * 8 * NL == BN && TM * 16 * NWM == BM && TN * 16 * NWN == BN && (TM % 2) == 0, "tile split");
    constexpr int NI = A_INST + W_INST;
    constexpr int A_STAGE = BM * ROWB, W_STAGE = BN * ROWB, W_RING = NST * A_STAGE, RING_BYTES = NST * (A_STAGE + W_STAGE);
    extern __shared__ __attribute__((aligned(16))) char smemm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int r16 = lane & 15, kg = lane >> 4;
    int tile_id = blockIdx.x;
    const int ntn = (g.N + BN - 1) / BN;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = tile_id & 7, slot = tile_id >> 3;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int n0 = (tile_id % ntn) * BN, m0 = (tile_id / ntn) * BM;
    const int M = g.M, N = g.N;
    const int split = g.split_k > 1 ? g.split_k : 1;
    const int nk = g.K / 32 / split;
    const int k_begin = split > 1 ? (int)blockIdx.z * nk * 32 : 0;
    const char* A = reinterpret_cast<const char*>(((g.alt_from_n > 0 && n0 >= g.alt_from_n) ? g.A_alt : g.A) + k_begin);
    const char* W = reinterpret_cast<const char*>(g.W + k_begin);
    // DMA: lane -> (row = lane >> 3 of its 8-row piece, slot p = lane & 7); the slot holds chunk (2 kg + x) with
    // kg = ((p >> 1) + 2 * ((row >> 3) & 1)) & 3 (the pair rotation is its own inverse), x = (p & 1) ^ ((row >> 1) & 1)
    const bool loader = wave < NL;
    const int lw = loader ? wave : 0;  // (the offsets of a non-loader are never used)
    unsigned a_off[A_INST], w_off[W_INST];  // byte offsets from A / W (the host checks that M lda 4 and N ldw 4 fit in 32 bits)
    auto src_chunk = [&](int row) { return m16_chunk_of_slot(row, lane & 7); };
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = (lw * A_INST + i) * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_off[i] = (unsigned)gm * (unsigned)g.lda * 4u + src_chunk(row) * 16;
    }
#pragma unroll
    for (int i = 0; i < W_INST; ++i) {
        const int row = (lw * W_INST + i) * 8 + (lane >> 3);
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;
        w_off[i] = (unsigned)gn * (unsigned)g.ldw * 4u + src_chunk(row) * 16;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smemm;
    auto issue_step = [&](int kt) {
        if (!loader) return;  // wave-uniform; a non-loader has no piece in flight, its vmcnt waits pass at once
        const int t = kt + NST - 1;
        const int tt = t < nk ? t : nk - 1, slot = t % NST;
        const char* wb = W + tt * ROWB;  // scalar: the K-tile's column block of both operands
        const char* ab = A + tt * ROWB;
        const unsigned sw = __builtin_amdgcn_readfirstlane(lds_base + W_RING + slot * W_STAGE + wave * W_INST * 1024);
#pragma unroll
        for (int i = 0; i < W_INST; ++i) lds_dma16_s(wb, w_off[i], sw + i * 1024);
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds_base + slot * A_STAGE + wave * A_INST * 1024);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) lds_dma16_s(ab, a_off[i], sa + i * 1024);
    };
    // fragment offsets: tile rows are multiples of 16, so the slot of (row = base + r16, kg) does not depend on the tile
    const int off_hi = r16 * ROWB + m16_slot(r16, kg, 0) * 16, off_lo = r16 * ROWB + m16_slot(r16, kg, 1) * 16;
    const int a_base = wm * WTM * ROWB, w_base = wn * WTN * ROWB;

    f32x4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;
    GEMM_STAMP_DECL;
    GEMM_STAMP(0);
#pragma unroll
    for (int v = -(NST - 1); v < 0; ++v) issue_step(v);
    // LayerNorm folded into this GEMM (g.ln_stats): A holds the RAW residual stream x (F16X2), W the weight times the norm's gain,
    // and LN(x) W^T + b = r (x W'^T - mu c) + b' with c = row sums of W', b' = b + W beta - applied in the epilogue.  Here, under
    // the latency of the first K-tile: (mu r, r) of this tile's rows from the twelve 32-column partials (mean, M2) the producing
    // residual GEMM left per row, merged in segment order, into LDS behind the ring.
    float2* lnrow = reinterpret_cast<float2*>(smemm + RING_BYTES);
    if (g.ln_stats && tid < BM) {
        int m = m0 + tid;
        m = m < M ? m : M - 1;
        const float4* sp = reinterpret_cast<const float4*>(g.ln_stats + (int64_t)m * 24);
        float mean_s[12], m2 = 0.f, msum = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float4 v = sp[q];
            mean_s[2 * q] = v.x; mean_s[2 * q + 1] = v.z;
            m2 += v.y; m2 += v.w;
            msum += v.x; msum += v.z;
        }
        const float mu = msum * (1.0f / 12.0f);
        float dev = 0.f;
#pragma unroll
        for (int q = 0; q < 12; ++q) dev += (mean_s[q] - mu) * (mean_s[q] - mu);
        const float rstd = 1.0f / sqrtf((m2 + 32.0f * dev) * (1.0f / 384.0f) + g.ln_eps);
        lnrow[tid] = make_float2(mu * rstd, rstd);
    }
    const f16x8 down = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                        (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt_w<(NST - 2) * NI>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt == 0) GEMM_STAMP(1);
        // Rings of three have a whole K-tile of slack (the pieces issued now feed tile kt + 2), so with SM_GEMM_LATE_HALF the second
        // half of the waves issues its pieces AFTER its MFMAs: every wave leaves the barrier together, and with all of them in the
        // issue queue of the CU's one address unit first, the matrix pipe idled ~500 cycles at the top of every K-tile
        constexpr bool LATE = SM_GEMM_LATE_HALF && NST >= 3 && NW >= 8;
        const bool late = LATE && wave >= NW / 2;
        if (!late) issue_step(kt);
        const char* sta = smemm + (kt % NST) * A_STAGE + a_base;
        const char* stw = smemm + W_RING + (kt % NST) * W_STAGE + w_base;
        // Register plan: the W fragments of the step stay live (TN x 12 registers), the A fragments come in blocks of at most
        // four row tiles (32 registers) - with TM = 8 all sixteen A fragments do not fit beside 64 accumulator registers in a
        // 128-register budget, and left to itself the compiler re-read some of them from LDS in an order that changed from
        // build to build (a K loop of 42.5k or 51.6k cycles for the same source, scripts/gemm_stamps.py)
        f16x8 wh[TN], wl[TN], whs[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            wh[j] = *reinterpret_cast<const f16x8*>(stw + j * 16 * ROWB + off_hi);
            if constexpr (TERMS == 3) wl[j] = *reinterpret_cast<const f16x8*>(stw + j * 16 * ROWB + off_lo);
        }
        constexpr int IB = TM > 4 ? 4 : TM;
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += IB) {
            f16x8 ah[IB], al[IB];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                ah[ii] = *reinterpret_cast<const f16x8*>(sta + (i0 + ii) * 16 * ROWB + off_hi);
                if constexpr (TERMS == 3) al[ii] = *reinterpret_cast<const f16x8*>(sta + (i0 + ii) * 16 * ROWB + off_lo);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (TERMS == 3) {
                    if (i0 == 0) whs[j] = wh[j] * down;
                }
#pragma unroll
                for (int ii = 0; ii < IB; ++ii) {
                    acc[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[ii], acc[i0 + ii][j], 0, 0, 0);
                    if constexpr (TERMS == 3) {
                        acc[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah[ii], acc[i0 + ii][j], 0, 0, 0);
                        acc[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whs[j], al[ii], acc[i0 + ii][j], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (late) {
            __builtin_amdgcn_sched_barrier(0);
            issue_step(kt);
        }
    }
    wait_vmcnt_w<0>();
    GEMM_STAMP(2);

    float* C = g.C + (split > 1 ? (int64_t)blockIdx.z : 0) * g.strideC;
    const bool out_split = g.patch_n < 0;
    const float ws = g.w_scale;
    // Epilogue staging (the idle ring): 32 rows of this wave's WTN columns.  For 32-column slices (the 256 x 256 and 256 x 128
    // shapes) rows are 128 B with the 16-B piece c of row r at slot c ^ (r & 7) and, for F16X2 output, the two 8-B halves of a piece
    // swapped on rows with bit 3 set: the 8-lane groups of a float4 store, the 16-lane groups of an 8-B store and the four 16-lane
    // groups of the read-back each cover a bank row once.  (Rounds 2-3 padded rows to 144 B: lanes r and r + 8 of every 8-B store met on
    // one bank pair and two lanes of every read-back group shared a slot - the 0.14 conflict ratio of profiles/r03_pmc_sq_counters.txt
    // was this epilogue, not the K loop.)  Other slice widths keep the padded rows.
    constexpr bool SWZ = WTN == 32;
    constexpr int EPLD = SWZ ? 128 : WTN * 4 + 16;
    constexpr int PIECES = WTN / 4;
    static_assert(NW * 32 * EPLD <= RING_BYTES, "epilogue staging must fit in the ring");
    __builtin_amdgcn_s_barrier();
    char* ep = smemm + wave * (32 * EPLD);

    auto run = [&](auto epi_tag, auto fmt_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr bool F = decltype(fmt_tag)::value;
        // The rows added in the epilogue (residual stream / position table) are fetched a 32-row block ahead, all of a
        // block's loads before any of its stores: C may alias R (in-place residual), so left in one loop every load would
        // wait behind the previous store: 16-24 exposed memory latencies per tile (scripts/gemm_stamps.py).
        constexpr int NIT = 32 * PIECES / 64;
        constexpr bool HASR = EPI == SM_EPI_RESIDUAL || EPI == SM_EPI_PATCH;
        constexpr bool AHEAD = WPS <= 2 && NIT <= 8 && TM * TN <= 16;  // a second block of rows in registers only where there is room
        float4 res[HASR ? NIT : 1];
        auto load_res = [&](int ib) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 64 + lane, row = idx / PIECES, pc = idx % PIECES;
                const int m = m0 + wm * WTM + ib * 32 + row, n = n0 + wn * WTN + pc * 4;
                res[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < M && n < N) {
                    if constexpr (EPI == SM_EPI_RESIDUAL) res[it] = *reinterpret_cast<const float4*>(g.R + (int64_t)m * g.ldr + n);
                    else if constexpr (EPI == SM_EPI_PATCH) res[it] = *reinterpret_cast<const float4*>(g.R + (int64_t)(1 + m % g.patch_n) * g.ldr + n);
                }
            }
        };
        if constexpr (HASR) load_res(0);
        // this lane's bias values (columns 16 j + 4 kg .. + 3 of the wave's slice), fetched ONCE and all together: read per
        // element inside the loops below they were TM x TN x 4 guarded scalar loads, each waited for in its own branch
        float brow[TN][4];
        {
            const bool vec = g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + 4 * kg;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (vec && n + 3 < N) {
                    bv = *reinterpret_cast<const float4*>(g.bias + n);
                } else if (g.bias) {
                    if (n < N) bv.x = g.bias[n];
                    if (n + 1 < N) bv.y = g.bias[n + 1];
                    if (n + 2 < N) bv.z = g.bias[n + 2];
                    if (n + 3 < N) bv.w = g.bias[n + 3];
                }
                brow[j][0] = bv.x; brow[j][1] = bv.y; brow[j][2] = bv.z; brow[j][3] = bv.w;
            }
        }
        // folded LayerNorm (consumer side): c[n] = sum_k W'[n][k] of this lane's columns (N % 4 == 0, c 16-B aligned: host-checked)
        constexpr bool CANFOLD = !HASR;
        const bool fold = CANFOLD && g.ln_stats != nullptr;
        float crow[CANFOLD ? TN : 1][4];
        if constexpr (CANFOLD) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + 4 * kg;
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fold && n + 3 < N) cv = *reinterpret_cast<const float4*>(g.ln_c + n);
                crow[j][0] = cv.x; crow[j][1] = cv.y; crow[j][2] = cv.z; crow[j][3] = cv.w;
            }
        }
#pragma unroll
        for (int ib = 0; ib < TM / 2; ++ib) {  // 32 staged rows = two 16-row tiles
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * ib + ii, row = ii * 16 + r16;
                float rs = ws, mur = 0.f;  // fold: t = acc (2^-s r) + (b' - mu r c)
                if constexpr (CANFOLD) {
                    if (fold) {
                        const float2 lr = lnrow[wm * WTM + i * 16 + r16];
                        rs = ws * lr.y;
                        mur = lr.x;
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = j * 16 + 4 * kg, n = n0 + wn * WTN + nl;  // this lane: columns nl .. nl+3 of its row
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (CANFOLD) {  // (wave-uniform branch: the plain path keeps its one fma per element)
                            if (fold) t[e] = acc[i][j][e] * rs + (brow[j][e] - mur * crow[j][e]);
                            else t[e] = acc[i][j][e] * ws + brow[j][e];
                        } else {
                            t[e] = acc[i][j][e] * ws + brow[j][e];
                        }
                        if constexpr (EPI == SM_EPI_RELU) t[e] = fmaxf(t[e], 0.f);
                    }
                    if constexpr (EPI == SM_EPI_GELU) gelu4(t);
                    if constexpr (F) {  // F16X2: elements nl..nl+3 of group nl / 8: hi at 32 G + 8 (kg & 1), lo 16 B further
                        f16x4 hi, lo;
                        split4(t, hi, lo);
                        if constexpr (SWZ) {
                            const int pc = 2 * (nl >> 3), half = ((kg & 1) ^ ((row >> 3) & 1)) * 8;
                            char* p = ep + row * EPLD + half;
                            *reinterpret_cast<f16x4*>(p + ((pc ^ (row & 7)) * 16)) = hi;
                            *reinterpret_cast<f16x4*>(p + (((pc + 1) ^ (row & 7)) * 16)) = lo;
                        } else {
                            char* p = ep + row * EPLD + (nl >> 3) * 32 + (kg & 1) * 8;
                            *reinterpret_cast<f16x4*>(p) = hi;
                            *reinterpret_cast<f16x4*>(p + 16) = lo;
                        }
                    } else {
                        const int pc = SWZ ? ((nl >> 2) ^ (row & 7)) : (nl >> 2);
                        *reinterpret_cast<float4*>(ep + row * EPLD + pc * 16) = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto store_piece = [&](int it, const float4& val) {
                const int idx = it * 64 + lane, row = idx / PIECES, pc = idx % PIECES;
                int m = m0 + wm * WTM + ib * 32 + row;
                const int n = n0 + wn * WTN + pc * 4;
                if (m < M && n < N) {
                    if constexpr (EPI == SM_EPI_PATCH) {
                        const int img = m / g.patch_n, p = m - img * g.patch_n;
                        m = img * (g.patch_n + 1) + 1 + p;
                    }
                    *reinterpret_cast<float4*>(C + (int64_t)m * g.ldc + n) = val;
                    if constexpr (EPI == SM_EPI_RESIDUAL) {
                        if (g.C2) {  // the F16X2 copy of the new residual stream: A operand of the GEMM the next LayerNorm is folded into
                            const float vv[4] = {val.x, val.y, val.z, val.w};
                            store_f16x2_4(g.C2 + (int64_t)m * g.ldc, n, vv);
                        }
                    }
                }
                if constexpr (EPI == SM_EPI_RESIDUAL && WTN == 32) {
                    // ... and that norm's row statistics over this wave's 32 columns: (mean, M2) by two 8-lane butterflies (the
                    // eight lanes pc = 0..7 of a row are consecutive), two-pass, written to slot (row, column segment): fixed
                    // order everywhere, no atomics.  Rows past M join the shuffles (their staging rows hold finite values).
                    if (g.ln_stats_out) {
                        float sm_ = (val.x + val.y) + (val.z + val.w);
                        sm_ += __shfl_xor(sm_, 1, 64); sm_ += __shfl_xor(sm_, 2, 64); sm_ += __shfl_xor(sm_, 4, 64);
                        const float mean = sm_ * (1.0f / 32.0f);
                        const float dx = val.x - mean, dy = val.y - mean, dz = val.z - mean, dw = val.w - mean;
                        float q2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
                        q2 += __shfl_xor(q2, 1, 64); q2 += __shfl_xor(q2, 2, 64); q2 += __shfl_xor(q2, 4, 64);
                        if (pc == 0 && m < M && n < N)
                            *reinterpret_cast<float2*>(g.ln_stats_out + ((int64_t)m * 12 + (n0 + wn * WTN) / 32) * 2) = make_float2(mean, q2);
                    }
                }
            };
            auto staged = [&](int it) {
                const int idx = it * 64 + lane, row = idx / PIECES, pc = idx % PIECES;
                float4 val = *reinterpret_cast<const float4*>(ep + row * EPLD + (SWZ ? (pc ^ (row & 7)) : pc) * 16);
                if constexpr (SWZ && F) {  // rows with bit 3 set hold the halves of a piece swapped
                    if (row & 8) val = make_float4(val.z, val.w, val.x, val.y);
                }
                if constexpr (HASR) { val.x = res[it].x + val.x; val.y = res[it].y + val.y; val.z = res[it].z + val.z; val.w = res[it].w + val.w; }
                return val;
            };
            if constexpr (HASR && AHEAD) {  // registers to spare: the next block's rows fly under this block's stores
                float4 v[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) v[it] = staged(it);
                if (ib + 1 < TM / 2) load_res(ib + 1);
#pragma unroll
                for (int it = 0; it < NIT; ++it) store_piece(it, v[it]);
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) store_piece(it, staged(it));
                if constexpr (HASR) {
                    if (ib + 1 < TM / 2) load_res(ib + 1);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    using T = std::true_type;
    using Fa = std::false_type;
    switch (g.epilogue) {
        case SM_EPI_GELU: out_split ? run(std::integral_constant<int, SM_EPI_GELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_GELU>{}, Fa{}); break;
        case SM_EPI_RELU: out_split ? run(std::integral_constant<int, SM_EPI_RELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_RELU>{}, Fa{}); break;
        case SM_EPI_RESIDUAL: run(std::integral_constant<int, SM_EPI_RESIDUAL>{}, Fa{}); break;
        case SM_EPI_PATCH: run(std::integral_constant<int, SM_EPI_PATCH>{}, Fa{}); break;
        default: out_split ? run(std::integral_constant<int, SM_EPI_BIAS>{}, T{}) : run(std::integral_constant<int, SM_EPI_BIAS>{}, Fa{}); break;
    }
#ifdef SM_TUNING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stamp sees the stores retired (tuning build only)
    GEMM_STAMP(3);
    GEMM_STAMP_FLUSH;
#endif
}

template <int BM, int BN, int NST, int NWM, int NWN, int WPS, int TERMS = 3>
static int launch_gemm_m16_terms(const sm_gemm_args& g, hipStream_t st) {
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.split_k > 1 ? g.split_k : 1);
    constexpr size_t lds = (size_t)NST * (BM + BN) * 128 + (size_t)BM * 8;  // ring + (mu r, r) of the tile's rows (folded LayerNorm)
    if (lds > 64 * 1024) {
        static std::once_flag attr_once;
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w16m16_kernel<BM, BN, NST, NWM, NWN, WPS, TERMS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipGetLastError();
        });
    }
    hipLaunchKernelGGL((gemm_w16m16_kernel<BM, BN, NST, NWM, NWN, WPS, TERMS>), grid, dim3(NWM * NWN * 64), lds, st, g);
    return check_launch("sm_gemm_w16 (16x16x32)");
}
template <int BM, int BN, int NST, int NWM, int NWN, int WPS>
static int launch_gemm_m16(const sm_gemm_args& g, hipStream_t st) {
    if (g.mfma_terms == 1) return launch_gemm_m16_terms<BM, BN, NST, NWM, NWN, WPS, 1>(g, st);
    return launch_gemm_m16_terms<BM, BN, NST, NWM, NWN, WPS, 3>(g, st);
}

#ifdef SM_TUNING  // measured and rejected (fc1 -7 % alone, others +-1 %, nothing in the pipeline): tuning build only
// ---- persistent variant ----------------------------------------------------------------------------------------------------
// What the one-tile-per-workgroup kernel above cannot hide: every workgroup of a launch starts
// together, so all of them wait for their first K-tiles together, run their MFMA loops together and write their C tiles
// together - the epilogue's HBM burst and the prologue's DMA latency are dead time for the matrix cores (a 128 x 128 tile
// with K = 384 lives 12 K-tiles: prologue + epilogue are a third of its life).  Here min(tiles, 512) workgroups stay
// resident (two per CU) and walk the tile list with a stride:
//   * the LAST K-tile iteration of a tile issues the FIRST K-tile of the workgroup's next tile into the ring slot that is
//     already free, so that DMA flies under the epilogue (which is staged in the other slot, unpadded + XOR-swizzled to fit);
//   * the second resident workgroup of every CU starts half a tile late (one s_sleep loop, once per launch), so one
//     workgroup's epilogue runs beside the other's MFMA loop instead of beside its epilogue.
// 128 x 128 tile, eight waves of 64 x 32, 32-k stages, two ring slots (64 KiB); K / 32 must be even (slot parity).
template <int WPS>
__global__ __launch_bounds__(512, WPS) void gemm_w16_persist_kernel(sm_gemm_args g, int n_tiles, int stagger) {
    constexpr int BM = 128, BN = 128, KT = 32, NWM = 2, NWN = 4, NW = 8, WTM = 64, WTN = 32, TM = 2, TN = 1;
    constexpr int ROWB = 128, A_INST = 2, W_INST = 2, NI = 4;
    constexpr int A_STAGE = BM * ROWB, W_STAGE = BN * ROWB, W_RING = 2 * A_STAGE;
    extern __shared__ __attribute__((aligned(16))) char smemp[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int r = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;
    const int ntn = (N + BN - 1) / BN;
    const int nk = g.K / KT;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smemp;
    const f16x8 down = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                        (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};
    const float ws = g.w_scale;
    const bool out_split = g.patch_n < 0;

    // virtual tile id -> (m0, n0): XCD-aware order over the whole list (ids 8 apart share an XCD; gridDim.x % 8 == 0)
    auto tile_origin = [&](int vid, int& m0, int& n0) {
        const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xcd = vid & 7, slot = vid >> 3;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        n0 = (t % ntn) * BN;
        m0 = (t / ntn) * BM;
    };
    struct Src { const char* a[A_INST]; const char* w[W_INST]; };
    auto make_src = [&](int m0, int n0, Src& s) {
        const char* A = reinterpret_cast<const char*>((g.alt_from_n > 0 && n0 >= g.alt_from_n) ? g.A_alt : g.A);
        const char* W = reinterpret_cast<const char*>(g.W);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
            const int row = (wave * A_INST + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            s.a[i] = A + ((int64_t)gm * g.lda) * 4 + c * 16;
        }
#pragma unroll
        for (int i = 0; i < W_INST; ++i) {
            const int row = (wave * W_INST + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int gn = n0 + row;
            gn = gn < N ? gn : N - 1;
            s.w[i] = W + ((int64_t)gn * g.ldw) * 4 + c * 16;
        }
    };
    auto issue = [&](const Src& s, int kt, int slot) {
        const unsigned sw = __builtin_amdgcn_readfirstlane(lds_base + W_RING + slot * W_STAGE + wave * W_INST * 1024);
#pragma unroll
        for (int i = 0; i < W_INST; ++i) lds_dma16(s.w[i] + kt * ROWB, sw + i * 1024);
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds_base + slot * A_STAGE + wave * A_INST * 1024);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) lds_dma16(s.a[i] + kt * ROWB, sa + i * 1024);
    };
    const int swz = (r >> 1) & 7;
    int off_hi[2], off_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        off_hi[s] = ((2 * (2 * s + h)) ^ swz) * 16;
        off_lo[s] = ((2 * (2 * s + h) + 1) ^ swz) * 16;
    }
    const int a_row = (wm * WTM + r) * ROWB;
    const int w_row = (wn * WTN + r) * ROWB;
    // epilogue staging of this wave: 32 rows x 128 B inside ring slot 1 (waves 0-3 in the A half, 4-7 in the W half)
    char* ep = smemp + (wave < 4 ? A_STAGE + wave * 4096 : W_RING + W_STAGE + (wave - 4) * 4096);

    int vid = blockIdx.x;
    if (vid >= n_tiles) return;
    if (stagger > 0 && ((blockIdx.x >> 8) & 1)) {  // the second resident workgroup of a CU: start half a tile late
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(8);  // ~512 clocks each
    }
    int m0, n0;
    tile_origin(vid, m0, n0);
    Src cur, nxt;
    make_src(m0, n0, cur);
    issue(cur, 0, 0);
    while (true) {
        const int vnext = vid + gridDim.x;
        const bool has_next = vnext < n_tiles;
        int m1 = 0, n1 = 0;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][0][v] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt_w<0>();              // K-tile kt has landed (and, at kt = 0, the previous tile's stores are acknowledged)
            __builtin_amdgcn_s_barrier();   // ... for every wave; every wave is done with the slot refilled next (at kt = 0:
            __builtin_amdgcn_sched_barrier(0);  //     with the epilogue staging in slot 1)
            if (kt + 1 < nk) issue(cur, kt + 1, (kt + 1) & 1);
            else if (has_next) {  // nk is even: slot 0 is free during the last K-tile and the epilogue
                tile_origin(vnext, m1, n1);
                make_src(m1, n1, nxt);  // (computed here, not at the tile's start: 16 fewer live registers in the loop)
                issue(nxt, 0, 0);
            }
            const char* sta = smemp + (kt & 1) * A_STAGE;
            const char* stw = smemp + W_RING + (kt & 1) * W_STAGE;
            f16x8 ah[2][TM], al[2][TM], wh[2], wl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                wh[s] = *reinterpret_cast<const f16x8*>(stw + w_row + off_hi[s]);
                wl[s] = *reinterpret_cast<const f16x8*>(stw + w_row + off_lo[s]);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[s][i] = *reinterpret_cast<const f16x8*>(sta + a_row + i * 32 * ROWB + off_hi[s]);
                    al[s][i] = *reinterpret_cast<const f16x8*>(sta + a_row + i * 32 * ROWB + off_lo[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 whs = wh[s] * down;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], ah[s][i], acc[i][0], 0, 0, 0);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], ah[s][i], acc[i][0], 0, 0, 0);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whs, al[s][i], acc[i][0], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // every wave is done reading slot 1 (nk - 1 is odd): it becomes the epilogue's staging

        // ---- epilogue (same arithmetic and order as gemm_w16_kernel; staging rows are 128 B, piece p of row r at p ^ (r & 7))
        auto run = [&](auto epi_tag, auto fmt_tag) {
            constexpr int EPI = decltype(epi_tag)::value;
            constexpr bool F = decltype(fmt_tag)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (F) {
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        float x[4], y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int nx = n0 + wn * WTN + 8 * q + 4 * h + e, ny = nx + 8;
                            x[e] = acc[i][0][4 * q + e] * ws + ((g.bias && nx < N) ? g.bias[nx] : 0.f);
                            y[e] = acc[i][0][4 * q + 4 + e] * ws + ((g.bias && ny < N) ? g.bias[ny] : 0.f);
                            if constexpr (EPI == SM_EPI_RELU) {
                                x[e] = fmaxf(x[e], 0.f);
                                y[e] = fmaxf(y[e], 0.f);
                            }
                        }
                        if constexpr (EPI == SM_EPI_GELU) { gelu4(x); gelu4(y); }
                        pair_groups(x, y);  // this lane now owns the whole 8-element group q + h: pieces 2 (q + h), + 1
                        f16x4 h0, l0, h1, l1;
                        split4(x, h0, l0);
                        split4(y, h1, l1);
                        f16x8 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hi[e] = h0[e]; hi[4 + e] = h1[e]; lo[e] = l0[e]; lo[4 + e] = l1[e]; }
                        const int pc = 2 * (q + h);
                        *reinterpret_cast<f16x8*>(ep + r * 128 + ((pc ^ (r & 7)) * 16)) = hi;
                        *reinterpret_cast<f16x8*>(ep + r * 128 + (((pc + 1) ^ (r & 7)) * 16)) = lo;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wn * WTN + 8 * q + 4 * h;
                        float4 val;
                        float* vp = &val.x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float b = (g.bias && n < N) ? g.bias[n + e] : 0.f;
                            float t = acc[i][0][4 * q + e] * ws + b;
                            if constexpr (EPI == SM_EPI_GELU) t = 0.5f * t * (1.0f + fast_erff(t * 0.70710678118654752440f));
                            else if constexpr (EPI == SM_EPI_RELU) t = fmaxf(t, 0.f);
                            vp[e] = t;
                        }
                        *reinterpret_cast<float4*>(ep + r * 128 + (((2 * q + h) ^ (r & 7)) * 16)) = val;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = it * 64 + lane, row = idx >> 3, pc = idx & 7;
                    int m = m0 + wm * WTM + i * 32 + row;
                    const int n = n0 + wn * WTN + pc * 4;
                    if (m < M && n < N) {
                        float4 v = *reinterpret_cast<const float4*>(ep + row * 128 + ((pc ^ (row & 7)) * 16));
                        if constexpr (EPI == SM_EPI_RESIDUAL) {
                            const float4 rr = *reinterpret_cast<const float4*>(g.R + (int64_t)m * g.ldr + n);
                            v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w;
                        } else if constexpr (EPI == SM_EPI_PATCH) {
                            const int img = m / g.patch_n, p = m - img * g.patch_n;
                            const float4 rr = *reinterpret_cast<const float4*>(g.R + (int64_t)(1 + p) * g.ldr + n);
                            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                            m = img * (g.patch_n + 1) + 1 + p;
                        }
                        *reinterpret_cast<float4*>(g.C + (int64_t)m * g.ldc + n) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        };
        using T = std::true_type;
        using Fa = std::false_type;
        switch (g.epilogue) {
            case SM_EPI_GELU: out_split ? run(std::integral_constant<int, SM_EPI_GELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_GELU>{}, Fa{}); break;
            case SM_EPI_RELU: out_split ? run(std::integral_constant<int, SM_EPI_RELU>{}, T{}) : run(std::integral_constant<int, SM_EPI_RELU>{}, Fa{}); break;
            case SM_EPI_RESIDUAL: run(std::integral_constant<int, SM_EPI_RESIDUAL>{}, Fa{}); break;
            case SM_EPI_PATCH: run(std::integral_constant<int, SM_EPI_PATCH>{}, Fa{}); break;
            default: out_split ? run(std::integral_constant<int, SM_EPI_BIAS>{}, T{}) : run(std::integral_constant<int, SM_EPI_BIAS>{}, Fa{}); break;
        }
        if (!has_next) break;
        vid = vnext;
        m0 = m1; n0 = n1;
        cur = nxt;
    }
    wait_vmcnt_w<0>();
}

static int launch_gemm_w_persist(const sm_gemm_args& g, int stagger, hipStream_t st) {
    const int n_tiles = ((g.N + 127) / 128) * ((g.M + 127) / 128);
    int grid = n_tiles < 512 ? n_tiles : 512;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w16_persist_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  64 * 1024);
        (void)hipGetLastError();
    });
    hipLaunchKernelGGL((gemm_w16_persist_kernel<4>), dim3(grid), dim3(512), 64 * 1024, st, g, n_tiles, stagger);
    return check_launch("sm_gemm_w16 (persistent)");
}

#endif  // SM_TUNING

// fp32 weights (rows, K) -> W16: per group of 8 k, 16 B of wh = f16(w * scale) then 16 B of wl = f16(w * scale - wh)
__global__ __launch_bounds__(256) void split_w16_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                        int64_t ldd, int K, int64_t total_groups, float scale) {
    const int gpr = K / 8;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total_groups; t += (int64_t)gridDim.x * 256) {
        const int64_t row = t / gpr;
        const int gidx = (int)(t - row * gpr);
        const float4 a = *reinterpret_cast<const float4*>(src + row * lds_ + gidx * 8);
        const float4 b = *reinterpret_cast<const float4*>(src + row * lds_ + gidx * 8 + 4);
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = x[e] * scale;  // exact: scale is a power of two
            _Float16 hh;
            float hf;
            asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hh) : "v"(v));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(hh));
            hi[e] = hh;
            lo[e] = (_Float16)(v - hf);
        }
        char* p = reinterpret_cast<char*>(dst + row * ldd) + gidx * 32;
        *reinterpret_cast<f16x8*>(p) = hi;
        *reinterpret_cast<f16x8*>(p + 16) = lo;
    }
}

#ifdef SM_TUNING
template <int BM, int BN, int KT, int NST, int NWM, int NWN, int MINB, int PIPE = 0>
static int launch_gemm_w(const sm_gemm_args& g, hipStream_t st) {
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.split_k > 1 ? g.split_k : 1);
    constexpr size_t lds = (size_t)NST * (BM + BN) * KT * 4;
    if (lds > 64 * 1024) {
        static std::once_flag attr_once;
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w16_kernel<BM, BN, KT, NST, NWM, NWN, MINB, PIPE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipGetLastError();
        });
    }
    hipLaunchKernelGGL((gemm_w16_kernel<BM, BN, KT, NST, NWM, NWN, MINB, PIPE>), grid, dim3(NWM * NWN * 64), lds, st, g);
    return check_launch("sm_gemm_w16");
}

#endif  // SM_TUNING

}  // namespace sm

extern "C" int sm_split_w16(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int32_t K, float scale,
                            void* stream) {
    SM_REQUIRE(src && dst && rows > 0 && K > 0 && K % 8 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= K &&
                   ld_dst >= K,
               "sm_split_w16: bad arguments (K %% 8 == 0, strides %% 4 == 0)");
    int ex = 0;
    SM_REQUIRE(scale > 0.f && frexpf(scale, &ex) == 0.5f, "sm_split_w16: scale must be a power of two");
    const int64_t groups = rows * (K / 8);
    int64_t grid = (groups + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sm::split_w16_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, K,
                       groups, scale);
    return sm::check_launch("sm_split_w16");
}

// variants: 0 = 256x128, 8 waves of 64x64, 16-k stages x 3 (72 KiB: two workgroups per CU)
//           1 = 256x128, 16 waves of 64x32, 32-k stages x 2 (96 KiB: one workgroup per CU)
//           2 = 128x128, 8 waves of 64x32, 32-k stages x 2 (64 KiB: two per CU)        [the f16x2 default shape]
//           3 = 128x128, 4 waves of 64x64, 16-k stages x 3 (48 KiB: three per CU)
//           4 = 64x64, 4 waves of 32x32, 32-k stages x 3 (48 KiB)                        [small decoder GEMMs]
//           6 = 256x128, 8 waves of 64x64, 32-k stages x 2 (96 KiB: one per CU)
//           7 = 128x64, 4 waves of 64x32, 32-k stages x 2 (48 KiB: three per CU)
//           8 = 128x128, 8 waves of 64x32, 16-k stages x 3 (48 KiB: three per CU)
extern "C" int sm_gemm_w16_tile(const sm_gemm_args* g, int out_f16x2, int variant, void* stream) {
    SM_REQUIRE(g && g->A && g->W && g->C, "sm_gemm_w16: null pointer");
    SM_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0 && g->K % 32 == 0 && g->batch <= 1, "sm_gemm_w16: bad shape (K %% 32, batch 1)");
    SM_REQUIRE(g->N % 4 == 0 && g->ldc % 4 == 0 && ((uintptr_t)g->C % 16 == 0), "sm_gemm_w16: N, ldc must be multiples of 4");
    SM_REQUIRE(g->lda % 8 == 0 && g->ldw % 8 == 0 && ((uintptr_t)g->A % 16 == 0) && ((uintptr_t)g->W % 16 == 0),
               "sm_gemm_w16: lda/ldw must be multiples of 8, pointers 16-B aligned");
    SM_REQUIRE(g->epilogue == SM_EPI_BIAS || g->epilogue == SM_EPI_GELU || g->epilogue == SM_EPI_RELU ||
                   g->epilogue == SM_EPI_RESIDUAL || g->epilogue == SM_EPI_PATCH, "sm_gemm_w16: unsupported epilogue");
    int ex = 0;
    SM_REQUIRE(g->w_scale > 0.f && frexpf(g->w_scale, &ex) == 0.5f, "sm_gemm_w16: w_scale must be the weight tensor's 2^-s");
    SM_REQUIRE((uint64_t)g->M * (uint64_t)g->lda * 4 < (1ull << 32) && (uint64_t)g->N * (uint64_t)g->ldw * 4 < (1ull << 32),
               "sm_gemm_w16: operands beyond 4 GiB (the ring's source addresses are 32-bit offsets from A / W)");
    if (g->ln_stats || g->ln_stats_out || g->C2) {
        SM_REQUIRE(variant >= 40 && variant < 50 && variant != 46 && (variant != 43 || !(g->ln_stats_out || g->C2)), "sm_gemm_w16: the LayerNorm fold needs the 16x16x32 kernels with 32-column wave tiles");
        if (g->ln_stats)
            SM_REQUIRE(g->K == SM_EMBED && g->ln_c && ((uintptr_t)g->ln_c % 16) == 0 && ((uintptr_t)g->ln_stats % 16) == 0 && g->ln_eps > 0.f &&
                           g->epilogue != SM_EPI_RESIDUAL && g->epilogue != SM_EPI_PATCH && !(g->split_k > 1) && g->alt_from_n == 0,
                       "sm_gemm_w16: folded LayerNorm needs K = 384, ln_c (16-B aligned), ln_eps and a BIAS / GELU / RELU epilogue");
        if (g->ln_stats_out || g->C2)
            SM_REQUIRE(g->epilogue == SM_EPI_RESIDUAL && g->N == SM_EMBED && g->ldc % 8 == 0 && !(g->split_k > 1) &&
                           (!g->C2 || ((uintptr_t)g->C2 % 32) == 0) && (!g->ln_stats_out || ((uintptr_t)g->ln_stats_out % 8) == 0),
                       "sm_gemm_w16: the F16X2 copy / row statistics come from a RESIDUAL epilogue with N = 384");
    }
    SM_REQUIRE(sm_gemm_w16_variant_name(variant) != nullptr, "sm_gemm_w16_tile: variant %d is not in this build (shipped: 40, 42, 43, 44, 45, 47; "
               "the rejected shapes are in the tuning build)", variant);
    SM_REQUIRE(g->mfma_terms == 0 || g->mfma_terms == 3 || (g->mfma_terms == 1 && variant >= 40 && variant < 50),
               "sm_gemm_w16: mfma_terms must be 0/3 (fp32-grade) or 1 (throughput mode, 16x16x32 kernels only)");
    if (out_f16x2)
        SM_REQUIRE(g->N % 8 == 0 && g->ldc % 8 == 0 && (g->epilogue == SM_EPI_BIAS || g->epilogue == SM_EPI_GELU ||
                                                      g->epilogue == SM_EPI_RELU) && !(g->split_k > 1),
                   "sm_gemm_w16: F16X2 output needs N %% 8 == 0 and a BIAS/GELU/RELU epilogue");
    if (g->epilogue == SM_EPI_RESIDUAL) SM_REQUIRE(g->R && g->ldr % 4 == 0, "sm_gemm_w16: residual needs R, ldr %% 4 == 0");
    if (g->epilogue == SM_EPI_PATCH) SM_REQUIRE(g->R && g->patch_n > 0, "sm_gemm_w16: PATCH needs R/patch_n");
    if (g->alt_from_n > 0) SM_REQUIRE(g->A_alt && g->alt_from_n % 256 == 0, "sm_gemm_w16: bad A_alt (multiple of 256)");
    if (g->split_k > 1)
        SM_REQUIRE(g->epilogue == SM_EPI_BIAS && !g->bias && (g->K / 32) % g->split_k == 0, "sm_gemm_w16: bad split_k");
    sm_gemm_args a = *g;
    if (out_f16x2) a.patch_n = -1;
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        // the shipped v_mfma_f32_16x16x32_f16 kernels
        case 40: return sm::launch_gemm_m16<256, 256, 2, 2, 8, 4>(a, st);   // 16 waves of 128x32 (fc1, all-layer K/V)
        case 42: return sm::launch_gemm_m16<128, 128, 2, 2, 4, 4>(a, st);   // 8 waves of 64x32
        case 43: return sm::launch_gemm_m16<256, 192, 2, 4, 4, 4>(a, st);   // 16 waves of 64x48: N = 1536 / 4608 in 400 / 1200 tiles
        case 44: return sm::launch_gemm_m16<64, 64, 3, 2, 2, 3>(a, st);     // 4 waves of 32x32 (decoder, batch 1)
        case 45: return sm::launch_gemm_m16<128, 64, 2, 2, 2, 3>(a, st);    // 4 waves of 64x32
        case 47: return sm::launch_gemm_m16<256, 128, 3, 4, 4, 4>(a, st);   // 16 waves of 64x32, ring of three (proj, fc2, qkv, patch)
#ifdef SM_TUNING  // measured-and-rejected shapes, kept as comparisons (scripts/gemm_w16_sweep.py, scripts/gemm_stamps.py)
        case 49: return sm::launch_gemm_m16<64, 64, 6, 2, 2, 1>(a, st);     // 64 x 64 behind a ring of six (96 KiB), round 4: batch-1 forward 1.209 -> 1.256 ms
        case 41: return sm::launch_gemm_m16<256, 128, 3, 4, 2, 2>(a, st);   // 8 waves of 64x64, ring of three
        case 46: return sm::launch_gemm_m16<128, 384, 2, 2, 4, 2>(a, st);   // full 384-wide rows: 8 waves of 64x96 (N = 384 GEMMs on 99 CUs)
        case 48: return sm::launch_gemm_m16<256, 128, 2, 4, 4, 4>(a, st);   // as 47 with a ring of two (96 KiB)
        // the v_mfma_f32_32x32x16_f16 family
        case 0: return sm::launch_gemm_w<256, 128, 16, 3, 4, 2, 4>(a, st);
        case 1: return sm::launch_gemm_w<256, 128, 32, 2, 4, 4, 4>(a, st);
        case 2: return sm::launch_gemm_w<128, 128, 32, 2, 2, 4, 4>(a, st);
        case 3: return sm::launch_gemm_w<128, 128, 16, 3, 2, 2, 3>(a, st);
        case 4: return sm::launch_gemm_w<64, 64, 32, 3, 2, 2, 3>(a, st);
        case 6: return sm::launch_gemm_w<256, 128, 32, 2, 4, 2, 2>(a, st);
        case 7: return sm::launch_gemm_w<128, 64, 32, 2, 2, 2, 3>(a, st);
        case 8: return sm::launch_gemm_w<128, 128, 16, 3, 2, 4, 6>(a, st);
        case 10: return sm::launch_gemm_w<128, 128, 16, 3, 2, 2, 3, 1>(a, st);  // software-pipelined K loop (PIPE = 1)
        case 11: return sm::launch_gemm_w<128, 128, 16, 4, 2, 2, 2, 1>(a, st);
        case 12: return sm::launch_gemm_w<128, 128, 16, 4, 2, 4, 4, 1>(a, st);
        case 13: return sm::launch_gemm_w<256, 128, 16, 4, 4, 2, 2, 1>(a, st);
        case 14: return sm::launch_gemm_w<128, 64, 16, 4, 2, 2, 4, 1>(a, st);
        case 15: return sm::launch_gemm_w<128, 128, 32, 3, 2, 2, 2, 1>(a, st);
        case 30: return sm::launch_gemm_w<256, 128, 32, 3, 4, 4, 4>(a, st);   // deep rings, one workgroup per CU
        case 31: return sm::launch_gemm_w<256, 128, 32, 3, 4, 2, 2>(a, st);
        case 32: return sm::launch_gemm_w<256, 256, 32, 2, 2, 8, 4>(a, st);
        case 33: return sm::launch_gemm_w<128, 128, 32, 4, 2, 4, 4>(a, st);
        case 34: return sm::launch_gemm_w<128, 128, 32, 5, 2, 4, 4>(a, st);
        case 35: return sm::launch_gemm_w<256, 128, 16, 6, 4, 2, 2>(a, st);
        case 36: return sm::launch_gemm_w<512, 128, 32, 2, 8, 2, 4>(a, st);
        case 20: case 21: case 22: case 23: case 24: {  // persistent 128x128
            SM_REQUIRE((g->K / 32) % 2 == 0 && !(g->split_k > 1), "sm_gemm_w16: the persistent variant needs an even number of 32-k tiles, no split-K");
            static const int stag_env = getenv("SM_W16_STAGGER") ? atoi(getenv("SM_W16_STAGGER")) : -1;
            const int stag[5] = {0, 8, 16, 32, 64};  // x ~512 clocks of s_sleep: 0, 4k, 8k, 16k, 33k cycles
            return sm::launch_gemm_w_persist(a, stag_env >= 0 ? stag_env : stag[variant - 20], st);
        }
#endif
    }
    sm::set_error("sm_gemm_w16_tile: unknown variant %d", variant);
    return SM_EINVAL;
}

extern "C" const char* sm_gemm_w16_variant_name(int variant) {
    switch (variant) {
        case 40: return "gemm_w16m16_kernel<256, 256, 2, 2, 8, 4, 3>";
        case 42: return "gemm_w16m16_kernel<128, 128, 2, 2, 4, 4, 3>";
        case 43: return "gemm_w16m16_kernel<256, 192, 2, 4, 4, 4, 3>";
        case 44: return "gemm_w16m16_kernel<64, 64, 3, 2, 2, 3, 3>";
        case 45: return "gemm_w16m16_kernel<128, 64, 2, 2, 2, 3, 3>";
        case 47: return "gemm_w16m16_kernel<256, 128, 3, 4, 4, 4, 3>";
#ifdef SM_TUNING
        case 49: return "gemm_w16m16_kernel<64, 64, 6, 2, 2, 1, 3>";
        case 41: return "gemm_w16m16_kernel<256, 128, 3, 4, 2, 2, 3>";
        case 46: return "gemm_w16m16_kernel<128, 384, 2, 2, 4, 2, 3>";
        case 48: return "gemm_w16m16_kernel<256, 128, 2, 4, 4, 4, 3>";
        case 0: return "gemm_w16_kernel<256, 128, 16, 3, 4, 2, 4, 0>";
        case 1: return "gemm_w16_kernel<256, 128, 32, 2, 4, 4, 4, 0>";
        case 2: return "gemm_w16_kernel<128, 128, 32, 2, 2, 4, 4, 0>";
        case 3: return "gemm_w16_kernel<128, 128, 16, 3, 2, 2, 3, 0>";
        case 4: return "gemm_w16_kernel<64, 64, 32, 3, 2, 2, 3, 0>";
        case 6: return "gemm_w16_kernel<256, 128, 32, 2, 4, 2, 2, 0>";
        case 7: return "gemm_w16_kernel<128, 64, 32, 2, 2, 2, 3, 0>";
        case 8: return "gemm_w16_kernel<128, 128, 16, 3, 2, 4, 6, 0>";
        case 10: return "gemm_w16_kernel<128, 128, 16, 3, 2, 2, 3, 1>";
        case 11: return "gemm_w16_kernel<128, 128, 16, 4, 2, 2, 2, 1>";
        case 12: return "gemm_w16_kernel<128, 128, 16, 4, 2, 4, 4, 1>";
        case 13: return "gemm_w16_kernel<256, 128, 16, 4, 4, 2, 2, 1>";
        case 14: return "gemm_w16_kernel<128, 64, 16, 4, 2, 2, 4, 1>";
        case 15: return "gemm_w16_kernel<128, 128, 32, 3, 2, 2, 2, 1>";
        case 20: case 21: case 22: case 23: case 24: return "gemm_w16_persist_kernel<4>";
        case 30: return "gemm_w16_kernel<256, 128, 32, 3, 4, 4, 4, 0>";
        case 31: return "gemm_w16_kernel<256, 128, 32, 3, 4, 2, 2, 0>";
        case 32: return "gemm_w16_kernel<256, 256, 32, 2, 2, 8, 4, 0>";
        case 33: return "gemm_w16_kernel<128, 128, 32, 4, 2, 4, 4, 0>";
        case 34: return "gemm_w16_kernel<128, 128, 32, 5, 2, 4, 4, 0>";
        case 35: return "gemm_w16_kernel<256, 128, 16, 6, 4, 2, 2, 0>";
        case 36: return "gemm_w16_kernel<512, 128, 32, 2, 8, 2, 4, 0>";
#endif
    }
    return nullptr;  // not compiled into this build (the rejected shapes live in the tuning build, build.py --tuning)
}

// Tile choice.  Alone on the GPU every shape from 128 x 64 to 256 x 256 lands within a few per cent of the others
// (scripts/gemm_w16_sweep.py; 128 x 64 is even the fastest for N = 384 because it fills the CUs best).  What ships is
// decided by the quantity the bench measures - three batches in flight, the chip at its power limit (1.9 GHz), other
// streams' kernels filling every idle CU: there the shapes that stage the fewest bytes per MFMA win (profiles/
// r02_pipeline_variant_sweep.log): 256 x 256 tiles (one workgroup of 16 waves per CU) for outputs that are a multiple of
// 256 wide, 256 x 128 with a three-stage ring otherwise - +3...6 % images/s over 128 x 128 / 128 x 64 although a lone
// launch is no faster.  Small problems (decoder, batch 1) keep 128 x 128 / 128 x 64 / 64 x 64 by workgroup count.
extern "C" int sm_gemm_w16_pick(const sm_gemm_args* g) {
    if (!g) return -1;
#ifdef SM_TUNING  // tuning knobs (same results): force a variant for the wide (N > 384) / narrow GEMMs, or the 32x32x16 family
    static const int forced_w = getenv("SM_W16_VARIANT_WIDE") ? atoi(getenv("SM_W16_VARIANT_WIDE")) : -1;
    static const int forced_n = getenv("SM_W16_VARIANT_NARROW") ? atoi(getenv("SM_W16_VARIANT_NARROW")) : -1;
    static const bool m32 = getenv("SM_W16_MFMA") && atoi(getenv("SM_W16_MFMA")) == 32;
#else
    constexpr int forced_w = -1, forced_n = -1;
    constexpr bool m32 = false;
#endif
    const long nb = g->split_k > 1 ? g->split_k : 1;
    const long wg128x64 = (long)((g->M + 127) / 128) * ((g->N + 63) / 64) * nb;
    const long wg128 = (long)((g->M + 127) / 128) * ((g->N + 127) / 128) * nb;
    const long wg256x128 = (long)((g->M + 255) / 256) * ((g->N + 127) / 128) * nb;
    const long wg256 = (long)((g->M + 255) / 256) * ((g->N + 255) / 256) * nb;
    // (Small launches are latency chains - a 64 x 64 tile walks 12 K-tiles behind a ring of three - but a ring of six, five stages in
    // flight before the first MFMA, is SLOWER: the batch-1 forward 1.209 -> 1.256 ms, serving p50 1.20 -> 1.25 ms: issuing 80 KiB of
    // LDS-DMA per workgroup up front costs more address-unit time than the waits it removes.  profiles/r04_deep_ring_ab.log; variant
    // 49 lives in the tuning build.)
    if (wg128x64 < 512) return m32 ? 4 : 44;
    const bool narrow = g->N <= 384;
    if (narrow && forced_n >= 0) return forced_n;
    if (!narrow && forced_w >= 0) return forced_w;
    if (g->alt_from_n == 0 || g->alt_from_n % 256 == 0) {
        // (256 x 192, variant 43: fc1 in 400 tiles = 1.56 rounds of 0.75-size tiles instead of 1.17 rounds in 2.  Measured, three
        // alternations, profiles/r04_tile_256x192_ab.log: the lone launch 79.9 -> 70.7 us (0.086 -> 0.097 of the f16 roof), ONE stream
        // 16.1 k -> 16.7 k images/s (+3.8 %), the three-stream pipeline the metric is quoted on 22.43 k -> 22.20 k (-1.0 %: 17 % more
        // staged bytes per MFMA).  The pick follows the pipeline; an experiment build (-DSM_GEMM_TILE_192) or sm_gemm_w16_tile selects 43.)
#ifdef SM_GEMM_TILE_192
        if (g->N % 192 == 0 && g->N >= 1024 && wg256 >= 128 && !m32) return 43;
#endif
        if (g->N % 256 == 0 && g->N >= 1024 && wg256 >= 128) return m32 ? 32 : 40;  // 256 x 256
        if (wg256x128 >= 128) return m32 ? 31 : 47;                                  // 256 x 128, ring of three, 16 waves
    }
    if (wg128 >= 256) return m32 ? 2 : 42;
    return m32 ? 7 : 45;
}

extern "C" int sm_gemm_w16(const sm_gemm_args* g, int out_f16x2, void* stream) {
    const int v = sm_gemm_w16_pick(g);
    if (v < 0) { sm::set_error("sm_gemm_w16: null arguments"); return SM_EINVAL; }
    return sm_gemm_w16_tile(g, out_f16x2, v, stream);
}

#ifdef SM_TUNING
extern "C" int sm_gemm_stamp_filter(int N, int K, int M) {
    const int f[3] = {N, K, M};
    return hipMemcpyToSymbol(HIP_SYMBOL(sm::g_gemm_stamp_filter), f, sizeof(f)) == hipSuccess ? 0 : 1;
}
extern "C" int sm_gemm_stamps(unsigned long long* host_out, int count) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sm::g_gemm_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif
