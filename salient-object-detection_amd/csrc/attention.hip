// softmax(scale * Q K^T) V per (batch, head), head_dim 64, exact fp32 on the matrix cores.
// Replaces the q@k^T -> softmax -> @v core of Attention.forward (vision_transformer.py:122-130) and of
// nn.MultiheadAttention in the decoder (transformer_decoder.py:273-289).
//
// One wave owns 32 query rows; the NW waves of a workgroup share one (batch, head) and stage its keys/values
// through LDS in chunks of <= 128 keys with a running max / sum (online softmax) across chunks.
//
// Layout trick (wave64, v_mfma_f32_32x32x2_f32): scores are computed TRANSPOSED, S^T = K Q^T, so the accumulator
// has the query on the lane (col = lane & 31) and the keys in the 16 registers x 2 lane halves.  A row softmax
// is then register-local plus ONE cross-half shuffle, and the exponentiated accumulator registers are, as they
// stand, the B operand of the second product O^T = V^T P^T (k index of step s == key held by register s), so P
// never goes through LDS.  The N x N score matrix never touches HBM (the reference materialises it: 59.6 MB
// per layer at B=64).
#include "common.h"
#include <mutex>
#include <math.h>

namespace sm {

constexpr int ATT_KB = 4;           // 32-key blocks per LDS chunk (128 keys: 66 KiB of LDS -> two workgroups per CU)
constexpr int ATT_KCH = ATT_KB * 32;
constexpr int ATT_KLD = 68;         // padded K row: conflict-free ds_read_b128
constexpr int ATT_VLD = 64;         // V rows are read with ds_read_b32 along d: conflict-free unpadded

// __launch_bounds__(.., 2): at most 256 registers per lane so two workgroups (one wave each per SIMD) share a CU
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attention_f32_kernel(sm_attn_args a, int kc_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + kc_rows * ATT_KLD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wave) * 32;

    const float* Qp = a.Q + b * a.sQb + head * SM_HEAD_DIM;
    const float* Kp = a.K + b * a.sKb + head * SM_HEAD_DIM;
    const float* Vp = a.V + b * a.sVb + head * SM_HEAD_DIM;

    // Q fragments: B operand of S^T = K Q^T.  lane (r,h) holds Q[q0+r][8t+4h .. +3], t = 0..7 (same k-permutation
    // as the K reads below); pre-scaled by scale * log2(e) so the scores come out in log2 units and the softmax
    // exponential is ONE v_exp_f32 (exp2) per element instead of expf's 12-instruction range reduction.  The
    // argument error this adds (|y| * 2^-24 relative) is weighted by the probability itself (p * |y| <= 0.53), i.e.
    // < 3e-8 absolute per attention weight: below fp32 resolution of the normalised weights.
    int qrow = q0 + r;
    qrow = qrow < a.n_q ? qrow : a.n_q - 1;
    float4 qf[8];
    const float qscale = a.scale * 1.44269504088896340736f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        qf[t] = *reinterpret_cast<const float4*>(Qp + (int64_t)qrow * a.sQr + 8 * t + 4 * h);
        qf[t].x *= qscale; qf[t].y *= qscale; qf[t].z *= qscale; qf[t].w *= qscale;
    }

    f32x16 o[2];
#pragma unroll
    for (int v = 0; v < 16; ++v) { o[0][v] = 0.f; o[1][v] = 0.f; }
    float m_run = -INFINITY;  // running max of this lane's query row (identical in both lane halves)
    float l_run = 0.f;        // running sum, PARTIAL per lane half (combined once at the end)

    for (int c0 = 0; c0 < a.n_k; c0 += ATT_KCH) {
        const int ck = min(ATT_KCH, a.n_k - c0);
        const int nb = (ck + 31) >> 5;
        __syncthreads();  // previous chunk fully consumed
        for (int c = tid; c < nb * 32 * 16; c += NW * 64) {
            const int row = c >> 4, c4 = (c & 15) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (row < ck) {
                kv = *reinterpret_cast<const float4*>(Kp + (int64_t)(c0 + row) * a.sKr + c4);
                vv = *reinterpret_cast<const float4*>(Vp + (int64_t)(c0 + row) * a.sVr + c4);
            }
            *reinterpret_cast<float4*>(Ks + row * ATT_KLD + c4) = kv;
            *reinterpret_cast<float4*>(Vs + row * ATT_VLD + c4) = vv;
        }
        __syncthreads();

        // ---- S^T = K Q^T for every 32-key block of the chunk ------------------------------------------------
        f32x16 s[ATT_KB];
#pragma unroll
        for (int kb = 0; kb < ATT_KB; ++kb) {
#pragma unroll
            for (int v = 0; v < 16; ++v) s[kb][v] = 0.f;
            if (kb < nb) {
                const float* kr = Ks + (kb * 32 + r) * ATT_KLD + 4 * h;
                float4 kf[8];  // all eight fragment reads of the block in flight before the first MFMA needs one
#pragma unroll
                for (int t = 0; t < 8; ++t) kf[t] = *reinterpret_cast<const float4*>(kr + 8 * t);
                __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (hipcc sinks them otherwise)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t].x, qf[t].x, s[kb], 0, 0, 0);
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t].y, qf[t].y, s[kb], 0, 0, 0);
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t].z, qf[t].z, s[kb], 0, 0, 0);
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t].w, qf[t].w, s[kb], 0, 0, 0);
                }
            }
        }
        // ---- mask the padded keys, chunk max ---------------------------------------------------------------
        float cmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < ATT_KB; ++kb) {
            if (kb < nb) {
                if ((kb + 1) * 32 > ck) {  // only the last block of the last chunk has padded keys (uniform branch)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (kb * 32 + acc_row(v, h) >= ck) s[kb][v] = -INFINITY;
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) cmax = fmaxf(cmax, s[kb][v]);
            }
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));  // the other half holds the other 16 keys of each block
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first chunk: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < ATT_KB; ++kb) {
            if (kb < nb) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][v] - m_new);
                    s[kb][v] = p;
                    psum += p;
                }
            }
        }
        l_run = l_run * alpha + psum;
        if (c0 > 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) { o[0][v] *= alpha; o[1][v] *= alpha; }
        }
        // ---- O^T += V^T P^T: the P registers are the B operand as they stand --------------------------------
#pragma unroll
        for (int kb = 0; kb < ATT_KB; ++kb) {
            if (kb < nb) {
                float v0[16], v1[16];  // the block's 32 V operands are read ahead of its 32 MFMAs
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float* vr = Vs + (kb * 32 + acc_row(v, h)) * ATT_VLD + r;
                    v0[v] = vr[0];
                    v1[v] = vr[32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[v], s[kb][v], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[v], s[kb][v], o[1], 0, 0, 0);
                }
            }
        }
    }

    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    if (q0 + r < a.n_q) {
        float* Op = a.O + b * a.sOb + (int64_t)(q0 + r) * a.sOr + head * SM_HEAD_DIM;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 w;
                w.x = o[db][4 * g + 0] * inv; w.y = o[db][4 * g + 1] * inv;
                w.z = o[db][4 * g + 2] * inv; w.w = o[db][4 * g + 3] * inv;
                if (a.out_f16x2) {
                    const float wv[4] = {w.x, w.y, w.z, w.w};
                    store_f16x2_4(a.O + b * a.sOb + (int64_t)(q0 + r) * a.sOr, head * SM_HEAD_DIM + db * 32 + 8 * g + 4 * h, wv);
                } else {
                    *reinterpret_cast<float4*>(Op + db * 32 + 8 * g + 4 * h) = w;
                }
            }
        }
    }
}

template <int NW>
static int launch_attn(const sm_attn_args& a, int nqb, hipStream_t st) {
    const int kc_rows = min(((a.n_k + 31) / 32) * 32, ATT_KCH);
    const size_t lds = (size_t)kc_rows * (ATT_KLD + ATT_VLD) * sizeof(float);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_kernel<NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ATT_KCH * (ATT_KLD + ATT_VLD) * 4);
        (void)hipGetLastError();
    });
    dim3 grid((nqb + NW - 1) / NW, a.heads, a.batch);
    hipLaunchKernelGGL((attention_f32_kernel<NW>), grid, dim3(NW * 64), lds, st, a, kc_rows);
    return check_launch("sm_attention_f32");
}

}  // namespace sm

extern "C" int sm_attention_f32(const sm_attn_args* a, void* stream) {
    SM_REQUIRE(a && a->Q && a->K && a->V && a->O, "sm_attention_f32: null pointer");
    SM_REQUIRE(a->batch > 0 && a->heads > 0 && a->n_q > 0 && a->n_k > 0, "sm_attention_f32: empty shape");
    SM_REQUIRE(a->sQr % 4 == 0 && a->sKr % 4 == 0 && a->sVr % 4 == 0 && a->sOr % 4 == 0 && a->sQb % 4 == 0 &&
                   a->sKb % 4 == 0 && a->sVb % 4 == 0 && a->sOb % 4 == 0,
               "sm_attention_f32: strides must be multiples of 4 floats (16-B accesses)");
    if (a->out_f16x2) SM_REQUIRE(a->sOr % 8 == 0 && a->sOb % 8 == 0, "sm_attention_f32: F16X2 output needs strides %% 8 == 0");
    SM_REQUIRE(((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) % 16 == 0,
               "sm_attention_f32: pointers must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nqb = (a->n_q + 31) / 32;
    // waves per workgroup: as even a split of the 32-row query blocks as possible, at most 4 (one per SIMD; two
    // workgroups share a CU, so a SIMD alternates one wave's MFMAs with the other's softmax VALU work)
    const int groups = (nqb + 3) / 4;
    const int nw = (nqb + groups - 1) / groups;
    switch (nw) {
        case 1: return sm::launch_attn<1>(*a, nqb, st);
        case 2: return sm::launch_attn<2>(*a, nqb, st);
        case 3: return sm::launch_attn<3>(*a, nqb, st);
        default: return sm::launch_attn<4>(*a, nqb, st);
    }
}
