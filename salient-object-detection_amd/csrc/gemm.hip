// C = epilogue(A W^T + bias): the F.linear / conv-as-GEMM / bmm replacement of the SelfMask path.
//
// fp32 in, fp32 accumulate on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf
// chain (exact fp32, 64 FLOP/clk/SIMD), which is what the 1e-4 logit-parity gate needs (bf16 operands miss it
// by four orders of magnitude).
//
// Structure (wave64, 256 threads = 4 waves as 2x2, each wave (BM/2)x(BN/2) outputs as TMxTN 32x32 MFMA blocks):
//   * A and W K-tiles (BK = 32 floats = one 128-B line per row) stream HBM/L2 -> LDS with LDS-DMA
//     (global_load_lds_dwordx4: no staging VGPRs, no ds_write), NST buffers deep, retired with a COUNTED
//     s_waitcnt vmcnt(N) + one raw s_barrier per K-tile (never vmcnt(0) in the loop when NST > 2);
//   * the LDS image is linear (an LDS-DMA wave-instruction writes 64 lanes x 16 B contiguously), so bank
//     conflicts are removed by XOR-swizzling the 16-B chunk index with (row>>1)&7 on the per-lane SOURCE
//     address and again on the ds_read_b128 fragment reads (conflict-free for all four 16-lane groups);
//   * k-permutation: within each 8-wide k group lane-half h owns k = 4h..4h+3, so one ds_read_b128 per operand
//     feeds 4 consecutive MFMA steps (A and W use the same permutation: the set of products is unchanged);
//   * two-level summation: the MFMA chain runs over 128 k, then is folded into a second accumulator with VALU
//     adds (a single 1536-long chain measured 5x torch-CPU's error against fp64).
#include "common.h"
#include <mutex>
#include <type_traits>

namespace sm {

constexpr int BK = 32;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS [lds_off, lds_off + 1 KiB).
// Issued from inline asm on purpose: hipcc does not count asm VMEM ops, so it cannot insert its conservative
// `s_waitcnt vmcnt(0)` in front of the next ds_read (it did with the builtin, serialising DMA and MFMA); the
// retire points are the explicit counted waits below.  M0 (the LDS base) is compiler-reserved: save/restore it.
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_off) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_off)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI>
__device__ __forceinline__ void store_out(const sm_gemm_args& g, float* C, int64_t bz, int m, int n, float val) {
    if constexpr (EPI == SM_EPI_GELU) {
        val = 0.5f * val * (1.0f + erff(val * 0.70710678118654752440f));
    } else if constexpr (EPI == SM_EPI_RELU) {
        val = fmaxf(val, 0.f);
    } else if constexpr (EPI == SM_EPI_RESIDUAL) {
        val = (g.R + bz * g.strideR)[(int64_t)m * g.ldr + n] + val;
    } else if constexpr (EPI == SM_EPI_SIGMOID2) {
        (g.C2 + bz * g.strideC)[(int64_t)m * g.ldc + n] = 1.0f / (1.0f + expf(-val));
    } else if constexpr (EPI == SM_EPI_PATCH) {
        const int img = m / g.patch_n, p = m - img * g.patch_n;
        val += g.R[(int64_t)(1 + p) * g.ldr + n];
        C[((int64_t)img * (g.patch_n + 1) + 1 + p) * g.ldc + n] = val;
        return;
    }
    C[(int64_t)m * g.ldc + n] = val;
}

template <int BM, int BN, int NST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(sm_gemm_args g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_INST = BM / 32, W_INST = BN / 32;  // LDS-DMA wave-instructions per wave per K-tile (8 rows each)
    constexpr int NI = A_INST + W_INST;
    constexpr int STAGE = (BM + BN) * BK;              // floats per pipeline stage
    constexpr int FLUSH_KT = 4;                        // fold the MFMA chain every 4 K-tiles (128 k)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    // split-K: blockIdx.z is the K-slice (batch == 1); otherwise it is the batch index
    const int split = g.split_k > 1 ? g.split_k : 1;
    const int64_t bz = split > 1 ? 0 : (int64_t)blockIdx.z;
    const int nk = g.K / BK / split;
    const int k_begin = split > 1 ? (int)blockIdx.z * nk * BK : 0;
    const float* __restrict__ A = ((g.alt_from_n > 0 && n0 >= g.alt_from_n) ? g.A_alt : g.A) + bz * g.strideA + k_begin;
    const float* __restrict__ W = g.W + bz * g.strideW + k_begin;

    // ---- LDS-DMA source pointers: lane L of instruction I fills LDS row (8*(wave*INST+I) + L/8), chunk L%8 --------
    const float* a_src[A_INST];
    const float* w_src[W_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        const int row = (wave * A_INST + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = A + (int64_t)gm * g.lda + c * 4;
    }
#pragma unroll
    for (int i = 0; i < W_INST; ++i) {
        const int row = (wave * W_INST + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;
        w_src[i] = W + (int64_t)gn * g.ldw + c * 4;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;  // LDS byte address of the dynamic region
    auto issue = [&](int kt, int stage) {
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds_base + (stage * STAGE + wave * A_INST * 8 * BK) * 4);
        const unsigned sw = __builtin_amdgcn_readfirstlane(lds_base + (stage * STAGE + BM * BK + wave * W_INST * 8 * BK) * 4);
#pragma unroll
        for (int i = 0; i < A_INST; ++i) dma16(a_src[i] + kt * BK, sa + i * 8 * BK * 4);
#pragma unroll
        for (int i = 0; i < W_INST; ++i) dma16(w_src[i] + kt * BK, sw + i * 8 * BK * 4);
    };

    // ---- fragment read offsets (floats): row r of a 32-row block, chunk (2kb+h) ^ ((r>>1)&7) -----------------------
    const int swz = (r >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) koff[kb] = ((2 * kb + h) ^ swz) * 4;
    const int a_row = (wm * (BM / 2) + r) * BK;
    const int w_row = BM * BK + (wn * (BN / 2) + r) * BK;

    f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[i][j][v] = 0.f; tot[i][j][v] = 0.f; }

#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t < nk ? t : nk - 1, t);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once this wave's DMAs for it are retired (all but the NST-2 younger tiles) and every
        // wave has passed the barrier; the barrier also proves every wave is done reading tile kt-1, whose buffer
        // the next issue overwrites.  Tail iterations re-issue the last tile so the counts stay uniform.
        wait_vmcnt<(NST - 2) * NI>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            const int nt = kt + NST - 1;
            issue(nt < nk ? nt : nk - 1, nt % NST);
        }
        const float* st = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            float4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(st + a_row + i * 32 * BK + koff[kb]);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const float4*>(st + w_row + j * 32 * BK + koff[kb]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float wv = s == 0 ? wf[j].x : s == 1 ? wf[j].y : s == 2 ? wf[j].z : wf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if ((kt & (FLUSH_KT - 1)) == FLUSH_KT - 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { tot[i][j][v] += acc[i][j][v]; acc[i][j][v] = 0.f; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment read of this tile has returned
    }
    wait_vmcnt<0>();  // drain the (redundant) tail DMAs before the LDS is released
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) tot[i][j][v] += acc[i][j][v];

    // ---- epilogue (one specialised loop per epilogue kind; the kind is kernel-uniform) ---------------------------
    float* C = g.C + (split > 1 ? (int64_t)blockIdx.z : bz) * g.strideC;  // may alias R (in-place residual)
    auto run = [&](auto epi_tag) {
        constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + r;
            if (n >= N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + acc_row(v, h);
                    if (m < M) store_out<EPI>(g, C, bz, m, n, tot[i][j][v] + bv);
                }
            }
        }
    };
    switch (g.epilogue) {
        case SM_EPI_GELU: run(std::integral_constant<int, SM_EPI_GELU>{}); break;
        case SM_EPI_RELU: run(std::integral_constant<int, SM_EPI_RELU>{}); break;
        case SM_EPI_RESIDUAL: run(std::integral_constant<int, SM_EPI_RESIDUAL>{}); break;
        case SM_EPI_SIGMOID2: run(std::integral_constant<int, SM_EPI_SIGMOID2>{}); break;
        case SM_EPI_PATCH: run(std::integral_constant<int, SM_EPI_PATCH>{}); break;
        default: run(std::integral_constant<int, SM_EPI_BIAS>{}); break;
    }
}

template <int BM, int BN, int NST>
static int launch_gemm(const sm_gemm_args& g, hipStream_t st) {
    dim3 grid(((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM), 1, g.split_k > 1 ? g.split_k : g.batch);
    constexpr size_t lds = (size_t)NST * (BM + BN) * BK * sizeof(float);
    if (lds > 64 * 1024) {
        static std::once_flag attr_once;  // per instantiation; safe from several host threads
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, NST>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipGetLastError();
        });
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, NST>), grid, dim3(256), lds, st, g);
    return check_launch("sm_gemm_f32");
}

static int validate(const sm_gemm_args* g) {
    SM_REQUIRE(g && g->A && g->W && g->C, "sm_gemm_f32: null pointer");
    SM_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0 && g->batch > 0, "sm_gemm_f32: empty shape M=%d N=%d K=%d batch=%d",
               g->M, g->N, g->K, g->batch);
    SM_REQUIRE(g->K % BK == 0, "sm_gemm_f32: K=%d must be a multiple of %d", g->K, BK);
    SM_REQUIRE(g->lda >= g->K && g->ldw >= g->K && g->ldc >= g->N, "sm_gemm_f32: leading dimension too small");
    SM_REQUIRE(g->lda % 4 == 0 && g->ldw % 4 == 0, "sm_gemm_f32: lda/ldw must be multiples of 4 (16-B loads)");
    SM_REQUIRE(((uintptr_t)g->A % 16 == 0) && ((uintptr_t)g->W % 16 == 0), "sm_gemm_f32: A/W must be 16-B aligned");
    SM_REQUIRE(g->strideA % 4 == 0 && g->strideW % 4 == 0, "sm_gemm_f32: batch strides must be multiples of 4");
    SM_REQUIRE(g->epilogue >= 0 && g->epilogue <= SM_EPI_PATCH, "sm_gemm_f32: bad epilogue %d", g->epilogue);
    if (g->epilogue == SM_EPI_RESIDUAL) SM_REQUIRE(g->R && g->ldr >= g->N, "sm_gemm_f32: residual needs R/ldr");
    if (g->epilogue == SM_EPI_SIGMOID2) SM_REQUIRE(g->C2, "sm_gemm_f32: SIGMOID2 needs C2");
    if (g->epilogue == SM_EPI_PATCH)
        SM_REQUIRE(g->R && g->patch_n > 0 && g->ldr >= g->N && g->batch == 1, "sm_gemm_f32: PATCH needs R/patch_n");
    if (g->alt_from_n > 0)
        SM_REQUIRE(g->A_alt && g->alt_from_n % 128 == 0 && ((uintptr_t)g->A_alt % 16 == 0),
                   "sm_gemm_f32: A_alt needs a pointer and alt_from_n %% 128 == 0");
    if (g->split_k > 1)
        SM_REQUIRE(g->batch == 1 && g->epilogue == SM_EPI_BIAS && !g->bias && (g->K / BK) % g->split_k == 0,
                   "sm_gemm_f32: split_k needs batch 1, no bias/epilogue and K/32 divisible by split_k");
    return SM_OK;
}

}  // namespace sm

extern "C" int sm_gemm_f32_tile(const sm_gemm_args* g, int bm, int bn, void* stream) {
    int rc = sm::validate(g);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    // pipeline depth per tile: every configuration keeps <= 72 KiB of LDS so two workgroups share a CU
    if (bm == 128 && bn == 128) return sm::launch_gemm<128, 128, 2>(*g, st);
    if (bm == 128 && bn == 64) return sm::launch_gemm<128, 64, 3>(*g, st);
    if (bm == 64 && bn == 64) return sm::launch_gemm<64, 64, 4>(*g, st);
    sm::set_error("sm_gemm_f32_tile: unsupported tile %dx%d", bm, bn);
    return SM_EINVAL;
}

extern "C" int sm_gemm_f32_pick_tile(const sm_gemm_args* g, int* bm, int* bn) {
    SM_REQUIRE(g && bm && bn && g->M > 0 && g->N > 0 && g->batch > 0, "sm_gemm_f32_pick_tile: bad arguments");
    // Tile choice = argmax of (CU-quantisation utilisation) x (measured per-tile efficiency).  The kernel is
    // MFMA-bound, so a launch takes ceil(workgroups / 256 CUs) rounds whatever the co-residency; e.g. proj/fc2
    // (M=12608, N=384) is 297 tiles of 128x128 (2 rounds for 1.16 rounds of work) but 1182 of 64x64 (5 for 4.62).
    // Efficiencies from scripts/gemm_sweep.py on MI355X: 128x128 1.00, 128x64 0.91, 64x64 0.84.
    static const int tiles[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    static const double eff[3] = {1.00, 0.91, 0.84};
    int best = 2;
    double best_score = -1.0;
    for (int t = 0; t < 3; ++t) {
        const double wgs = (double)((g->M + tiles[t][0] - 1) / tiles[t][0]) * ((g->N + tiles[t][1] - 1) / tiles[t][1]) *
                           (g->split_k > 1 ? g->split_k : g->batch);
        const double rounds = wgs / 256.0;
        const double util = rounds / (double)(long)(rounds + 0.999999);
        const double score = util * eff[t];
        if (score > best_score + 1e-9) { best_score = score; best = t; }
    }
    *bm = tiles[best][0];
    *bn = tiles[best][1];
    return SM_OK;
}

extern "C" int sm_gemm_f32(const sm_gemm_args* g, void* stream) {
    int rc = sm::validate(g);
    if (rc) return rc;
    int bm, bn;
    rc = sm_gemm_f32_pick_tile(g, &bm, &bn);
    if (rc) return rc;
    return sm_gemm_f32_tile(g, bm, bn, stream);
}
