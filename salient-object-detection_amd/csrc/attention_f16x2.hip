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
// K/V rows are staged raw (the 256-B head slice of each F16X2 row) with a 272-B row stride (conflict-free b128 reads).
#include "common.h"
#include <math.h>

namespace sm {

constexpr int HAT_KCH = 128;   // keys per LDS chunk (2 x 128 x 272 B = 68 KiB: two workgroups per CU)
constexpr int HAT_LD = 272;    // bytes per staged row

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x4 tr_read4(unsigned addr) {
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attention_f16x2_kernel(sm_attn_args a, int kc_rows) {
    extern __shared__ __attribute__((aligned(16))) char smema[];
    char* Ks = smema;
    char* Vs = smema + kc_rows * HAT_LD;
    const unsigned vs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Vs;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;

    const char* Qp = reinterpret_cast<const char*>(a.Q + b * a.sQb + head * SM_HEAD_DIM);
    const char* Kp = reinterpret_cast<const char*>(a.K + b * a.sKb + head * SM_HEAD_DIM);
    const char* Vp = reinterpret_cast<const char*>(a.V + b * a.sVb + head * SM_HEAD_DIM);

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

    // per-lane pieces of the transposed V reads: 16-lane group g16 = (column block cb, key half = h); lane 4q+p of a
    // group supplies row q, columns 4p..4p+3 and receives column (lane & 15)
    const int cb = (lane >> 4) & 1, li = lane & 15, trq = li >> 2, trp = li & 3;

    for (int c0 = 0; c0 < a.n_k; c0 += HAT_KCH) {
        const int ck = min(HAT_KCH, a.n_k - c0);
        const int nb = (ck + 31) >> 5;
        __syncthreads();
        for (int c = tid; c < nb * 32 * 16; c += NW * 64) {  // 16 chunks of 16 B per row and tensor
            const int row = c >> 4, ch = (c & 15) * 16;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (row < ck) {
                kv = *reinterpret_cast<const float4*>(Kp + (int64_t)(c0 + row) * a.sKr * 4 + ch);
                vv = *reinterpret_cast<const float4*>(Vp + (int64_t)(c0 + row) * a.sVr * 4 + ch);
            }
            *reinterpret_cast<float4*>(Ks + row * HAT_LD + ch) = kv;
            *reinterpret_cast<float4*>(Vs + row * HAT_LD + ch) = vv;
        }
        __syncthreads();

        // sub-chunks of 2 key blocks (64 keys): keeps the score registers at 64 per lane
        for (int sb = 0; sb < nb; sb += 2) {
            const int nb2 = min(2, nb - sb);
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int v = 0; v < 16; ++v) s[kb][v] = -INFINITY;
                if (kb < nb2) {
                    f32x16 mn, cr;
#pragma unroll
                    for (int v = 0; v < 16; ++v) { mn[v] = 0.f; cr[v] = 0.f; }
                    const char* kr = Ks + ((sb + kb) * 32 + r) * HAT_LD + h * 32;
                    f16x8 kh[4], kl[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        kh[t] = *reinterpret_cast<const f16x8*>(kr + t * 64);
                        kl[t] = *reinterpret_cast<const f16x8*>(kr + t * 64 + 16);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        mn = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[t], qh[t], mn, 0, 0, 0);
                        cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[t], ql[t], cr, 0, 0, 0);
                        cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[t], qh[t], cr, 0, 0, 0);
                    }
                    const bool tail = ((sb + kb) + 1) * 32 > ck;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float sv = (mn[v] + cr[v] * (1.0f / 2048.0f)) * cs;
                        if (tail && (sb + kb) * 32 + acc_row(v, h) >= ck) sv = -INFINITY;
                        s[kb][v] = sv;
                    }
                }
            }
            float cmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int v = 0; v < 16; ++v) cmax = fmaxf(cmax, s[kb][v]);
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            const float m_new = fmaxf(m_run, cmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            float psum = 0.f;
            f16x8 ph[2][2], pl[2][2];  // [key block][16-key step]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float p = __builtin_amdgcn_exp2f(s[kb][8 * u + j] - m_new);  // masked / absent: exp2(-inf) = 0
                        psum += p;
                        _Float16 hi, lo;
                        split1(p, hi, lo);
                        ph[kb][u][j] = hi;
                        pl[kb][u][j] = lo;
                    }
                }
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int v = 0; v < 16; ++v) { om[0][v] *= alpha; om[1][v] *= alpha; oc[0][v] *= alpha; oc[1][v] *= alpha; }
            // O^T += V^T P^T
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb < nb2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int row1 = (sb + kb) * 32 + 16 * u + 4 * h + trq;  // keys 16u + 4h + {0..3}; second read: + 8
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const int d0 = db * 32 + cb * 16 + 4 * trp;
                            const unsigned base = vs_lds + row1 * HAT_LD + (d0 >> 3) * 32 + (d0 & 7) * 2;
                            const f16x4 h1 = tr_read4(base), h2 = tr_read4(base + 8 * HAT_LD);
                            const f16x4 l1 = tr_read4(base + 16), l2 = tr_read4(base + 16 + 8 * HAT_LD);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
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
    }

    const float l = l_run + __shfl_xor(l_run, 32, 64);
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

template <int NW>
static int launch_attn_h(const sm_attn_args& a, int nqb, hipStream_t st) {
    const int kc_rows = min(((a.n_k + 31) / 32) * 32, HAT_KCH);
    const size_t lds = (size_t)kc_rows * HAT_LD * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16x2_kernel<NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, HAT_KCH * HAT_LD * 2);
        (void)hipGetLastError();
        attr_set = true;
    }
    dim3 grid((nqb + NW - 1) / NW, a.heads, a.batch);
    hipLaunchKernelGGL((attention_f16x2_kernel<NW>), grid, dim3(NW * 64), lds, st, a, kc_rows);
    return check_launch("sm_attention_f16x2");
}

}  // namespace sm

extern "C" int sm_attention_f16x2(const sm_attn_args* a, void* stream) {
    SM_REQUIRE(a && a->Q && a->K && a->V && a->O, "sm_attention_f16x2: null pointer");
    SM_REQUIRE(a->batch > 0 && a->heads > 0 && a->n_q > 0 && a->n_k > 0, "sm_attention_f16x2: empty shape");
    SM_REQUIRE(a->sQr % 8 == 0 && a->sKr % 8 == 0 && a->sVr % 8 == 0 && a->sOr % 8 == 0 && a->sQb % 8 == 0 &&
                   a->sKb % 8 == 0 && a->sVb % 8 == 0 && a->sOb % 8 == 0,
               "sm_attention_f16x2: strides must be multiples of 8 elements (F16X2 groups)");
    SM_REQUIRE(((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) % 32 == 0,
               "sm_attention_f16x2: pointers must be 32-B aligned (one F16X2 group)");
    hipStream_t st = (hipStream_t)stream;
    const int nqb = (a->n_q + 31) / 32;
    const int groups = (nqb + 3) / 4;
    const int nw = (nqb + groups - 1) / groups;
    switch (nw) {
        case 1: return sm::launch_attn_h<1>(*a, nqb, st);
        case 2: return sm::launch_attn_h<2>(*a, nqb, st);
        case 3: return sm::launch_attn_h<3>(*a, nqb, st);
        default: return sm::launch_attn_h<4>(*a, nqb, st);
    }
}
