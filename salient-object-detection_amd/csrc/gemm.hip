// C = epilogue(A W^T + bias): the F.linear / conv-as-GEMM / bmm replacement of the SelfMask path.
//
// fp32 in, fp32 accumulate on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf
// chain (exact fp32, 64 FLOP/clk/SIMD), which is what the 1e-4 logit-parity gate needs (bf16 operands miss it
// by four orders of magnitude).
//
// Tiling (wave64): a 256-thread workgroup = 4 waves as 2x2; each wave owns (BM/2)x(BN/2) outputs as TMxTN
// 32x32 MFMA blocks.  A and W tiles (BK = 32 deep) go global -> registers -> LDS (double buffered, one barrier
// per K-tile) with rows padded to 36 floats so the ds_read_b128 fragment reads are bank-conflict free.
// k-permutation: within each 8-wide k group lane-half h owns k = 4h..4h+3, so one ds_read_b128 per operand
// feeds 4 consecutive MFMA steps (A and W use the same permutation, so the products summed are unchanged).
#include "common.h"

namespace sm {

constexpr int BK = 32;
constexpr int LDS_LD = 36;

template <int BM, int BN, bool ADD>
__global__ __launch_bounds__(256) void gemm_f32_kernel(sm_gemm_args g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + 2 * BM * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int64_t bz = blockIdx.z;

    const float* __restrict__ A = g.A + bz * g.strideA;
    const float* __restrict__ W = g.W + bz * g.strideW;
    const int M = g.M, K = g.K;

    // per-thread staging coordinates (fixed over the K loop)
    const float* a_src[A_CH];
    const float* a2_src[A_CH];
    const float* w_src[W_CH];
    int a_dst[A_CH], w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 4;
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = A + (int64_t)gm * g.lda + kc;
        a2_src[i] = ADD ? g.A_add + (int64_t)(gm % g.a_add_rows) * g.lda2 + kc : nullptr;
        a_dst[i] = row * LDS_LD + kc;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 4;
        int gn = n0 + row;
        gn = gn < g.N ? gn : g.N - 1;
        w_src[i] = W + (int64_t)gn * g.ldw + kc;
        w_dst[i] = row * LDS_LD + kc;
    }

    // Staging registers.  No branch surrounds the loads/stores (a conditional made hipcc keep these arrays in
    // scratch): the last iteration re-loads the final K-tile and stores it to the idle buffer, which nobody reads.
    float4 ra[A_CH], rb[A_CH], rw[W_CH];
#define SM_LOAD_TILES(k0)                                                                           \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                            \
            ra[i] = *reinterpret_cast<const float4*>(a_src[i] + (k0));                              \
        _Pragma("unroll") for (int i = 0; i < W_CH; ++i)                                            \
            rw[i] = *reinterpret_cast<const float4*>(w_src[i] + (k0));                              \
        if constexpr (ADD) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                        \
                rb[i] = *reinterpret_cast<const float4*>(a2_src[i] + (k0));                         \
        }                                                                                           \
    }
#define SM_STORE_TILES(buf)                                                                         \
    {                                                                                               \
        float* as_ = As + (buf) * BM * LDS_LD;                                                      \
        float* ws_ = Ws + (buf) * BN * LDS_LD;                                                      \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                          \
            float4 t_ = ra[i];                                                                      \
            if constexpr (ADD) { t_.x += rb[i].x; t_.y += rb[i].y; t_.z += rb[i].z; t_.w += rb[i].w; } \
            *reinterpret_cast<float4*>(as_ + a_dst[i]) = t_;                                        \
        }                                                                                           \
        _Pragma("unroll") for (int i = 0; i < W_CH; ++i)                                            \
            *reinterpret_cast<float4*>(ws_ + w_dst[i]) = rw[i];                                     \
    }

    // Two-level summation: the MFMA chain (a k-ordered fmaf chain) runs over FLUSH_KT K-tiles (128 k), then is
    // folded into `tot` with VALU adds.  A single 1536-long chain (fc2) measured 5x the error of torch-CPU's
    // blocked sgemm against fp64; with 128-long chains the kernel is at or below the CPU's error.  Cost: 16 v_add
    // per 32x32 block per 128 k (~1.5 % of the MFMA time).
    constexpr int FLUSH_KT = 4;
    f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[i][j][v] = 0.f; tot[i][j][v] = 0.f; }

    const int nk = K / BK;
    SM_LOAD_TILES(0);
    SM_STORE_TILES(0);
    __syncthreads();

    const int a_frag = (wm * (BM / 2) + r) * LDS_LD + 4 * h;
    const int w_frag = (wn * (BN / 2) + r) * LDS_LD + 4 * h;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        SM_LOAD_TILES((kt + 1 < nk ? kt + 1 : kt) * BK);
        const float* as = As + buf * BM * LDS_LD + a_frag;
        const float* ws = Ws + buf * BN * LDS_LD + w_frag;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            float4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + i * 32 * LDS_LD + kb * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const float4*>(ws + j * 32 * LDS_LD + kb * 8);
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
        SM_STORE_TILES(buf ^ 1);
        if ((kt & (FLUSH_KT - 1)) == FLUSH_KT - 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) { tot[i][j][v] += acc[i][j][v]; acc[i][j][v] = 0.f; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) tot[i][j][v] += acc[i][j][v];

    // ---- epilogue -------------------------------------------------------------------------------------------
    float* C = g.C + bz * g.strideC;  // may alias R (in-place residual)
    const int epi = g.epilogue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + r;
        if (n >= g.N) continue;
        const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * (BM / 2) + i * 32 + acc_row(v, h);
                if (m >= M) continue;
                float val = tot[i][j][v] + bv;
                if (epi == SM_EPI_GELU) {
                    val = 0.5f * val * (1.0f + erff(val * 0.70710678118654752440f));
                } else if (epi == SM_EPI_RELU) {
                    val = fmaxf(val, 0.f);
                } else if (epi == SM_EPI_RESIDUAL) {
                    val = (g.R + bz * g.strideR)[(int64_t)m * g.ldr + n] + val;
                } else if (epi == SM_EPI_SIGMOID2) {
                    (g.C2 + bz * g.strideC)[(int64_t)m * g.ldc + n] = 1.0f / (1.0f + expf(-val));
                } else if (epi == SM_EPI_PATCH) {
                    const int img = m / g.patch_n, p = m - img * g.patch_n;
                    val += g.R[(int64_t)(1 + p) * g.ldr + n];
                    C[((int64_t)img * (g.patch_n + 1) + 1 + p) * g.ldc + n] = val;
                    continue;
                }
                C[(int64_t)m * g.ldc + n] = val;
            }
        }
    }
}

template <int BM, int BN>
static int launch_gemm(const sm_gemm_args& g, hipStream_t st) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch);
    const size_t lds = 2 * (BM + BN) * LDS_LD * sizeof(float);
    if (g.a_add_rows > 0)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true>), grid, dim3(256), lds, st, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false>), grid, dim3(256), lds, st, g);
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
    SM_REQUIRE(g->epilogue >= 0 && g->epilogue <= SM_EPI_PATCH, "sm_gemm_f32: bad epilogue %d", g->epilogue);
    if (g->epilogue == SM_EPI_RESIDUAL) SM_REQUIRE(g->R && g->ldr >= g->N, "sm_gemm_f32: residual needs R/ldr");
    if (g->epilogue == SM_EPI_SIGMOID2) SM_REQUIRE(g->C2, "sm_gemm_f32: SIGMOID2 needs C2");
    if (g->epilogue == SM_EPI_PATCH)
        SM_REQUIRE(g->R && g->patch_n > 0 && g->ldr >= g->N && g->batch == 1, "sm_gemm_f32: PATCH needs R/patch_n");
    if (g->a_add_rows > 0)
        SM_REQUIRE(g->A_add && g->lda2 >= g->K && g->lda2 % 4 == 0 && ((uintptr_t)g->A_add % 16 == 0),
                   "sm_gemm_f32: bad A_add");
    return SM_OK;
}

}  // namespace sm

extern "C" int sm_gemm_f32_tile(const sm_gemm_args* g, int bm, int bn, void* stream) {
    int rc = sm::validate(g);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (bm == 128 && bn == 128) return sm::launch_gemm<128, 128>(*g, st);
    if (bm == 128 && bn == 64) return sm::launch_gemm<128, 64>(*g, st);
    if (bm == 64 && bn == 64) return sm::launch_gemm<64, 64>(*g, st);
    sm::set_error("sm_gemm_f32_tile: unsupported tile %dx%d for N=%d", bm, bn, g->N);
    return SM_EINVAL;
}

extern "C" int sm_gemm_f32(const sm_gemm_args* g, void* stream) {
    int rc = sm::validate(g);
    if (rc) return rc;
    // pick the largest tile that still gives every CU (256) at least two workgroups
    const long mt128 = (g->M + 127) / 128;
    const long b = g->batch;
    if (g->N % 128 == 0 && mt128 * (g->N / 128) * b >= 512) return sm_gemm_f32_tile(g, 128, 128, stream);
    if (mt128 * ((g->N + 63) / 64) * b >= 512) return sm_gemm_f32_tile(g, 128, 64, stream);
    return sm_gemm_f32_tile(g, 64, 64, stream);
}
