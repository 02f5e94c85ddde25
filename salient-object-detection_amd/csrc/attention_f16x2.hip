// softmax(scale * Q K^T) V per (batch, head), head_dim 64, on the f16 matrix cores with split operands (fp32-grade).
// Same role as attention.hip (vision_transformer.py:122-130, nn.MultiheadAttention core), used when the projections
// run in the F16X2 mode: Q, K and V arrive already split (the QKV GEMM writes F16X2), so no conversion pass exists.
//
//   S^T = K Q^T    : 3 x v_mfma_f32_32x32x16_f16 per 16 head-dims (hi*hi into `main`; hi*lo + lo*hi into `cross`),
//                    s = (main + cross / 2048) * scale * log2(e); keys on the accumulator rows, the query on the lane:
//                    softmax is register-local + one cross-half shuffle (as in attention.hip);
//   P              : exp2(s - max) split on the fly into hi / lo f16; the accumulator registers 8u..8u+7 of a block ARE
//                    the B operand of 16-key step u (k order 16u + 8(j>>2) + 4h + (j&3));
//   O^T = V^T P^T  : V stays row-major [key][d] in LDS exactly as the GEMM wrote it; the transposed A operand comes
//                    from ds_read_b64_tr_b16 (4 keys x 16 dims per 16-lane group, delivered column-major), whose
//                    4-row blocks match the permuted k order above.
//
// Staging: K and V rows travel as raw 256-B head slices of the F16X2 rows, 64 keys per chunk, by LDS-DMA into a
// two-deep ring (2 x 32 KiB = two workgroups per CU): the DMA of chunk c+1 flies under the MFMAs of chunk c and costs
// neither registers nor ds_writes.  An LDS-DMA writes 64 consecutive 16-B pieces, so rows cannot be padded; instead
// piece p of row r sits at slot p ^ swz(r), swz a bit permutation of r & 15 chosen so that both the b128 key reads
// (16 rows, one piece index) and the transposed value reads (4 consecutive rows x 4 alternate pieces) hit 64 banks.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <mutex>

namespace sm {

constexpr int HAT_CH = 64;                  // keys per ring slot
constexpr int HAT_TENSOR = HAT_CH * 256;    // bytes of K (or V) per slot
constexpr int HAT_SLOT = 2 * HAT_TENSOR;

// ds_read_b64_tr_b16 through the compiler's builtin (round 3): hipcc then counts the read in lgkmcnt and may keep several in
// flight under the MFMAs of the previous group - the inline-asm form needed an s_waitcnt lgkmcnt(0) + sched_barrier before every
// group of MFMAs, i.e. one exposed LDS round trip per 6 MFMAs
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ f16x4 tr_read4(unsigned addr) {
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<__attribute__((address_space(3))) fp16x4_t*>((uintptr_t)addr));
    return __builtin_bit_cast(f16x4, v);
}
// slot permutation of row r: bits (r0, r1, r2, r3) -> XOR mask bits (0, 3, 1, 2)
__device__ __forceinline__ int hat_swz(int r) { return (r & 1) | ((r & 2) << 2) | ((r & 4) >> 1) | ((r & 8) >> 1); }

// ALL (tuning build only; measured slower, see sm_attention_f16x2): every chunk of the keys has its own ring slot (at most four: n_k
// <= 256) and all of them are requested before the first MFMA - meant for launches of at most one workgroup per CU (batch-1 serving, the
// decoder's 20 queries), where a wave has one chunk of compute (~1 us) to hide the next chunk's DMA round trip behind.
template <int NW, bool ALL = false>
__global__ __launch_bounds__(NW * 64, ALL ? 1 : 2) void attention_f16x2_kernel(sm_attn_args a, int groups, int ablate) {
#ifndef SM_TUNING
    ablate = 0;  // the timing-only ablations exist only in the tuning build (build.py --tuning)
#endif
    extern __shared__ __attribute__((aligned(16))) char smema[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smema;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (id % 8), so the `groups` workgroups that
    // share one (batch, head)'s keys and values get ids 8 apart - same XCD, dispatched together - and the second one
    // finds K/V in that XCD's L2 instead of fetching them again from HBM
    int qg, pair;
    {
        const int id = blockIdx.x, pairs = a.heads * a.batch;
        if ((pairs & 7) == 0) {
            const int t = id >> 3;
            qg = t % groups;
            pair = (t / groups) * 8 + (id & 7);
        } else {
            qg = id % groups;
            pair = id / groups;
        }
    }
    const int head = pair % a.heads;
    const int64_t b = pair / a.heads;
    const int q0 = (qg * NW + wave) * 32;
    const bool active = q0 < a.n_q;  // waves past the last query block only help with staging

    const char* Qp = reinterpret_cast<const char*>(a.Q + b * a.sQb + head * SM_HEAD_DIM);
    const char* Kp = reinterpret_cast<const char*>(a.K + b * a.sKb + head * SM_HEAD_DIM);
    const char* Vp = reinterpret_cast<const char*>(a.V + b * a.sVb + head * SM_HEAD_DIM);

    // ring fill: 32 one-KiB pieces per chunk (16 of K, 16 of V; a piece = 4 rows), dealt round-robin over the waves
    const int n_k = a.n_k;
    const int srow = lane >> 4;                       // row of the piece this lane fetches
    auto issue = [&](int chunk, int slot) {
#pragma unroll
        for (int j = 0; j < (32 + NW - 1) / NW; ++j) {
            const int i = wave + NW * j;
            if (i < 32) {
                const int g = i & 15, row = 4 * g + srow;
                int key = chunk * HAT_CH + row;
                key = key < n_k ? key : n_k - 1;      // tail rows repeat the last key (finite data; masked below)
                const int piece = (lane & 15) ^ hat_swz(row);
                const char* src = (i < 16 ? Kp + (int64_t)key * a.sKr * 4 : Vp + (int64_t)key * a.sVr * 4) + piece * 16;
                lds_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + slot * HAT_SLOT + (i >> 4) * HAT_TENSOR + g * 1024));
            }
        }
    };
    const int nch = (n_k + HAT_CH - 1) / HAT_CH;
    if constexpr (ALL) {
        for (int c = 0; c < nch; ++c) issue(c, c);
    } else {
        issue(0, 0);
    }

    // Q fragments (B operand of S^T = K Q^T): lane (r,h), 16-dim step t -> k-group 2t+h: hi chunk, lo chunk
    int qrow = q0 + r;
    qrow = qrow < a.n_q ? qrow : a.n_q - 1;
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const char* p = Qp + (int64_t)qrow * a.sQr * 4 + (2 * t + h) * 32;
        qh[t] = *reinterpret_cast<const f16x8*>(p);
        ql[t] = *reinterpret_cast<const f16x8*>(p + 16);
    }
    const float cs = a.scale * 1.44269504088896340736f;  // scores in log2 units: softmax = one v_exp_f32 per element

    f32x16 om[2], oc[2];
#pragma unroll
    for (int v = 0; v < 16; ++v) { om[0][v] = 0.f; om[1][v] = 0.f; oc[0][v] = 0.f; oc[1][v] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // key fragment offsets inside a row (A operand of S^T): pieces 2(2t+h) [hi] and +1 [lo], permuted by the row
    const int ksw = hat_swz(r & 15);
    int k_hi[4], k_lo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        k_hi[t] = ((2 * (2 * t + h)) ^ ksw) * 16;
        k_lo[t] = ((2 * (2 * t + h) + 1) ^ ksw) * 16;
    }
    // transposed V reads: 16-lane group = (column block cb, key half h); lane 4q+p of a group supplies row q, columns
    // 4p..4p+3 and receives column (lane & 15).  Rows 16u + 4h + q and + 8; pieces 2(4db + 2cb + (p>>1)) [hi], +1 [lo]
    const int cb = (lane >> 4) & 1, li = lane & 15, trq = li >> 2, trp = li & 3;
    const int vrow = 4 * h + trq;
    const int vswA = hat_swz(vrow), vswB = hat_swz(vrow + 8);
    const int vpiece = 4 * cb + 2 * (trp >> 1), vsub = (trp & 1) * 8;

    for (int c = 0; c < nch; ++c) {
        if (!ALL || c == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // chunk c has landed for every wave, and every wave is done with chunk c-1
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!ALL) {
            if (c + 1 < nch && (ablate != 2)) issue(c + 1, (c + 1) & 1);
        }
        if (!active || ablate == 1) continue;
        const int ck = min(HAT_CH, n_k - c * HAT_CH);
        const int nb2 = (ck + 31) >> 5;
        const int slot = ALL ? c : (c & 1);
        const char* Ks = smema + slot * HAT_SLOT;
        const unsigned vs_lds = lds0 + slot * HAT_SLOT + HAT_TENSOR;

        // raw scores t = main + cross / 2048 (the softmax scale cs > 0 is applied inside the exponent's fma)
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int v = 0; v < 16; ++v) s[kb][v] = -INFINITY;
            if (kb < nb2) {
                f32x16 mn, cr;
#pragma unroll
                for (int v = 0; v < 16; ++v) { mn[v] = 0.f; cr[v] = 0.f; }
                const char* kr = Ks + (kb * 32 + r) * 256;
                f16x8 kh[4], kl[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    kh[t] = *reinterpret_cast<const f16x8*>(kr + k_hi[t]);
                    kl[t] = *reinterpret_cast<const f16x8*>(kr + k_lo[t]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    mn = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[t], qh[t], mn, 0, 0, 0);
                    cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[t], ql[t], cr, 0, 0, 0);
                    cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[t], qh[t], cr, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) s[kb][v] = fmaf(cr[v], 1.0f / 2048.0f, mn[v]);
                if ((kb + 1) * 32 > ck) {  // only the last chunk's last block has keys past n_k (wave-uniform branch)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (kb * 32 + acc_row(v, h) >= ck) s[kb][v] = -INFINITY;
                }
            }
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) cmax = fmaxf(cmax, s[kb][v]);
        cmax = halves_max(cmax);
        // Lazy running maximum: the reference point m_run only moves when some query's new maximum exceeds it by more
        // than 2^8 in probability (p <= 256 then - harmless in fp32 and in the f16 split), which after the first chunk
        // is rare; the 64-register rescale of O runs only then (wave-uniform branch).
        const float lim = 8.0f / cs;
        if (__builtin_amdgcn_ballot_w64(cmax > m_run + lim) != 0) {
            const float m_new = fmaxf(m_run, cmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int v = 0; v < 16; ++v) { om[0][v] *= alpha; om[1][v] *= alpha; oc[0][v] *= alpha; oc[1][v] *= alpha; }
        }
        const float moff = -m_run * cs;
        float psum = 0.f;
        f16x8 ph[2][2], pl[2][2];  // [key block][16-key step]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf[j] = __builtin_amdgcn_exp2f(fmaf(s[kb][8 * u + j], cs, moff));  // masked / absent: exp2(-inf) = 0
                    psum += pf[j];
                }
                split8(pf, ph[kb][u], pl[kb][u]);
            }
        }
        l_run += psum;
        // O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb < nb2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned rowA = vs_lds + (kb * 32 + 16 * u + vrow) * 256 + vsub;  // keys 16u + 4h + q; rowB: + 8
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int pc = 8 * db + vpiece;
                        const f16x4 h1 = tr_read4(rowA + ((pc ^ vswA) * 16)), h2 = tr_read4(rowA + 8 * 256 + ((pc ^ vswB) * 16));
                        const f16x4 l1 = tr_read4(rowA + (((pc + 1) ^ vswA) * 16)), l2 = tr_read4(rowA + 8 * 256 + (((pc + 1) ^ vswB) * 16));
                        f16x8 vh, vl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vh[e] = h1[e]; vh[4 + e] = h2[e]; vl[e] = l1[e]; vl[4 + e] = l2[e]; }
                        om[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kb][u], om[db], 0, 0, 0);
                        oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kb][u], oc[db], 0, 0, 0);
                        oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kb][u], oc[db], 0, 0, 0);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the workgroup's LDS allocation
    if (!active) return;

    const float l = halves_sum(l_run);
    const float inv = 1.0f / l;
    if (q0 + r < a.n_q) {
        float* Orow = a.O + b * a.sOb + (int64_t)(q0 + r) * a.sOr;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                float x[4], y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = (om[db][4 * g + e] + oc[db][4 * g + e] * (1.0f / 2048.0f)) * inv;
                    y[e] = (om[db][4 * g + 4 + e] + oc[db][4 * g + 4 + e] * (1.0f / 2048.0f)) * inv;
                }
                if (a.out_f16x2) {  // lanes l / l^32 trade halves: each writes one whole 32-B F16X2 group
                    pair_groups(x, y);
                    store_f16x2_8(Orow, head * SM_HEAD_DIM + db * 32 + 8 * (g + h), x, y);
                } else {
                    const int d = head * SM_HEAD_DIM + db * 32 + 8 * g + 4 * h;
                    *reinterpret_cast<float4*>(Orow + d) = make_float4(x[0], x[1], x[2], x[3]);
                    *reinterpret_cast<float4*>(Orow + d + 8) = make_float4(y[0], y[1], y[2], y[3]);
                }
            }
        }
    }
}

template <int NW, bool ALL>
static int launch_attn_h(const sm_attn_args& a, int groups, hipStream_t st) {
    constexpr int SLOTS = ALL ? 4 : 2;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16x2_kernel<NW, ALL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SLOTS * HAT_SLOT);
        (void)hipGetLastError();
    });
#ifdef SM_TUNING
    static const int ablate = getenv("SM_ATTN_ABLATE") ? atoi(getenv("SM_ATTN_ABLATE")) : 0;  // timing-only: 1 = no compute, 2 = first chunk only
#else
    constexpr int ablate = 0;
#endif
    dim3 grid(groups * a.heads * a.batch);
    hipLaunchKernelGGL((attention_f16x2_kernel<NW, ALL>), grid, dim3(NW * 64), SLOTS * HAT_SLOT, st, a, groups, ablate);
    return check_launch("sm_attention_f16x2");
}

}  // namespace sm

extern "C" int sm_attention_f16x2(const sm_attn_args* a, void* stream) {
    SM_REQUIRE(a && a->Q && a->K && a->V && a->O, "sm_attention_f16x2: null pointer");
    SM_REQUIRE(a->batch > 0 && a->heads > 0 && a->n_q > 0 && a->n_k > 0, "sm_attention_f16x2: empty shape");
    SM_REQUIRE(a->scale > 0.f, "sm_attention_f16x2: scale must be positive");
    SM_REQUIRE(a->sQr % 8 == 0 && a->sKr % 8 == 0 && a->sVr % 8 == 0 && a->sOr % 8 == 0 && a->sQb % 8 == 0 &&
                   a->sKb % 8 == 0 && a->sVb % 8 == 0 && a->sOb % 8 == 0,
               "sm_attention_f16x2: strides must be multiples of 8 elements (F16X2 groups)");
    SM_REQUIRE(((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) % 32 == 0,
               "sm_attention_f16x2: pointers must be 32-B aligned (one F16X2 group)");
    hipStream_t st = (hipStream_t)stream;
    const int nqb = (a->n_q + 31) / 32;
    const int groups = (nqb + 3) / 4;
    const int nw = (nqb + groups - 1) / groups;
    // always four waves: those past the last query block (decoder: a single block) issue their share of the ring's DMA
    (void)nw;
#ifdef SM_TUNING
    // Round 4, measured and rejected: at most one workgroup per CU and at most four key chunks -> all chunks staged at once (128 KiB of
    // LDS), no per-chunk barrier.  Batch-1 forward 1.209 -> 1.250 ms, batch 8 1.43 -> 1.48 ms (profiles/r04_attn_all_ab.log): the first
    // MFMA now waits for 128 KiB of LDS-DMA instead of 32.  Tuning build only (SM_ATTN_ALL=1).
    static const bool all_env = getenv("SM_ATTN_ALL") && atoi(getenv("SM_ATTN_ALL")) == 1;
    if (all_env && (long)groups * a->heads * a->batch <= 256 && a->n_k <= 4 * sm::HAT_CH) return sm::launch_attn_h<4, true>(*a, groups, st);
#endif
    return sm::launch_attn_h<4, false>(*a, groups, st);
}
