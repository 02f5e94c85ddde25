// Fused QKV projection + softmax attention of one encoder block (vision_transformer.py:110-131: qkv Linear -> split heads
// -> softmax(q k^T / 8) v), the kernel BASELINE.json's north star names.  One workgroup per (image, head), seven waves,
// wave w owns tokens 32w .. 32w+31 as two 16-token tiles - as projection rows first, as queries afterwards.  The (B*N, 1152)
// QKV tensor of the unfused path (58 MB written and read back per layer at B = 64) never exists: K and V of the head live in
// LDS, Q in registers.  What ships is qkv_attention_m16_kernel<2, TERMS, 4> (v_mfma_f32_16x16x32_f16):
//
// Phase 1 - projection.  Xn (the LayerNorm output, F16X2) and the head's 192 rows of the W16 qkv weight stream through a
// two-slot LDS-DMA ring in 32-k stages (128-B rows, the m16_slot image of gemm_w16.hip applied on the DMA source address);
// waves 0-3 - one per SIMD - issue all 52 pieces of a stage, waves 4-6 only compute.  Per stage a wave issues 72 MFMAs into 24
// single-accumulator 16x16 tiles (gemm_w16.hip's arithmetic: wh*ah + wl*ah + (wh*2^-11)*al'):
//     Q^T, K^T = W X^T   (weights = MFMA A operand): the token sits on the lane, four head-dims in the registers;
//     V       = X W^T    (weights = MFMA B operand): the head-dim sits on the lane, four tokens in the registers.
// Two accumulator tiles along the contraction axis are exactly the eight k-elements lane group kg supplies to one 32-k step of
// the next product, in the same permuted order k = 16 (j >> 2) + 4 kg + (j & 3) on both operands, so nothing is transposed:
// Q stays in registers (B operand of S^T = K Q^T), every lane writes its key's K fragments and its head-dim's V^T fragments to
// LDS as the 16-B pieces the consumer lane reads back with ds_read_b128.
//
// Phase 2 - attention: S^T = K Q^T (3 MFMAs per 32 dims), 32 keys per step, softmax on the lane (maxima / sums across the four
// k-groups by v_permlane16_swap + v_permlane32_swap), lazy running maximum behind ONE wave-uniform branch, P split in
// registers by compiler-visible instructions (it feeds the P V MFMAs directly) = B operand of O^T = V^T P^T.
//
// LDS: one workgroup per CU owns all 160 KiB.  During phase 1 it is the ring (2 x 52 KiB); after the last stage (one barrier)
// the same memory becomes K (224 x 256 B) + V^T (64 x 896 B).  N <= 208 tokens: the ViT-S/16 224^2 headline shape (197);
// larger grids take the unfused path (K and V of one head no longer fit: 785 tokens x 256 B x 2 = 392 KiB).
// The 32x32x16 form of round 2 (qkv_attention_kernel<KT, NST>), the all-waves-load form and the three-slot ring live in the
// tuning build only (measured: DESIGN.md section 5).
#include "common.h"
#include <type_traits>
#include <math.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace sm {

constexpr int QA_WAVES = 7;
constexpr int QA_TOK = QA_WAVES * 32;        // 224 token rows staged per K-stage (rows past N repeat the last token)
constexpr int QA_KROWS = 208;                // keys kept in LDS (13 MFMA steps of 16 keys)
constexpr int QA_K_BYTES = QA_KROWS * 256;   // 53248
constexpr int QA_VLD = 208 * 4 + 16;         // bytes per head-dim row of V^T (+16: b128 reads of 16 rows hit 16 slots)
constexpr int QA_V_BYTES = 64 * QA_VLD;      // 54272
constexpr int QA_LDS = 160 * 1024;           // the whole LDS of a CU (one workgroup per CU)
static_assert(QA_K_BYTES + QA_V_BYTES <= QA_LDS, "K and V^T overlay the ring");

#ifdef SM_TUNING  // the v_mfma_f32_32x32x16_f16 form (round 2, first half): 77-78 us against 65 us for what ships
// KT = k per ring stage: 32 -> a stage row is one full 128-B line (8 rows per 1-KiB LDS-DMA piece), 16 -> 64-B half lines
// (16 rows per piece).  NST = ring stages.  Stage = [224 Xn rows | 192 weight rows] x KT * 4 bytes.
template <int KT, int NST>
__global__ __launch_bounds__(QA_WAVES * 64, 2) void qkv_attention_kernel(sm_qkv_attn_args a) {
    constexpr int ROWB = KT * 4, CH = KT / 4, RPP = 1024 / ROWB, KS = KT / 16;
    constexpr int XT = QA_TOK * ROWB, WT = 192 * ROWB, STAGE = XT + WT;
    constexpr int XP = XT / 1024, NP = STAGE / 1024;                  // Xn pieces / all pieces per stage
    constexpr int PPW = (NP + QA_WAVES - 1) / QA_WAVES;               // pieces per wave, the same immediate in every wave: the
    constexpr int DUMP = NST * STAGE;                                 // surplus ("filler") pieces land in a dump zone behind the ring
    static_assert(DUMP + (PPW * QA_WAVES - NP) * 1024 <= QA_LDS, "ring + dump zone must fit in LDS");
    extern __shared__ __attribute__((aligned(16))) char smq[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smq;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // (image, head) of this workgroup: ids 8 apart share an XCD (round-robin dispatch), so the six heads of an image run
    // on one XCD back to back and five of them find the image's Xn rows in that L2
    int head, b;
    {
        const int id = blockIdx.x, pairs = a.B * SM_HEADS;
        int lin = id;
        if ((pairs & 7) == 0) lin = (id & 7) * (pairs >> 3) + (id >> 3);
        b = lin / SM_HEADS;
        head = lin - b * SM_HEADS;
    }
    const int N = a.N;
    const char* X = reinterpret_cast<const char*>(a.Xn + (int64_t)b * N * a.ldx);
    const char* W = reinterpret_cast<const char*>(a.Wqkv);
    auto swz = [](int row) { return KT == 32 ? (row >> 1) & 7 : (row >> 2) & 3; };

    // ---- ring fill: piece p of a stage = RPP rows; p < XP: Xn rows, else weight rows; dealt round-robin over the waves.
    // Lane l fetches chunk (l % CH) ^ swz(row) of row l / CH.
    const char* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave + QA_WAVES * j;
        const int prow = lane / CH;
        if (p < XP) {
            const int row = p * RPP + prow;
            const int c = (lane % CH) ^ swz(row);
            const int tok = row < N ? row : N - 1;
            src[j] = X + (int64_t)tok * a.ldx * 4 + c * 16;
        } else {
            const int row = ((p < NP ? p : XP) - XP) * RPP + prow;  // 0..191: [Q dims | K dims | V dims] of this head
            const int c = (lane % CH) ^ swz(row);
            const int wrow = (row >> 6) * SM_EMBED + head * SM_HEAD_DIM + (row & 63);
            src[j] = W + (int64_t)wrow * SM_EMBED * 4 + c * 16;
        }
    }
    auto issue = [&](int kt, int slot) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave + QA_WAVES * j;
            const unsigned d = p < NP ? lds0 + slot * STAGE + p * 1024 : lds0 + DUMP + (p - NP) * 1024;  // fillers: dump zone
            lds_dma16(src[j] + kt * ROWB, __builtin_amdgcn_readfirstlane(d));
        }
    };

    // fragment offsets inside a stage row: k16 step s, lane half h -> k-group 2s+h -> chunks 2(2s+h) (hi), +1 (lo)
    const int xrow = wave * 32 + r;
    int x_hi[KS], x_lo[KS], w_hi[KS], w_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        x_hi[s] = xrow * ROWB + (((2 * (2 * s + h)) ^ swz(xrow)) * 16);
        x_lo[s] = xrow * ROWB + (((2 * (2 * s + h) + 1) ^ swz(xrow)) * 16);
        w_hi[s] = XT + r * ROWB + (((2 * (2 * s + h)) ^ swz(r)) * 16);   // weight rows 32 blk + r: swz does not depend on blk
        w_lo[s] = XT + r * ROWB + (((2 * (2 * s + h) + 1) ^ swz(r)) * 16);
    }

    f32x16 acc[6];  // [Q d0-31, Q d32-63, K d0-31, K d32-63, V d0-31, V d32-63]
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const f16x8 down = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                        (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};  // 2^-11

    constexpr int NKT = SM_EMBED / KT;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t, t);
    for (int kt = 0; kt < NKT; ++kt) {
        // stage kt has landed (this wave's pieces): all but the NST - 2 younger stages' pieces are done
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();  // ... for every wave; every wave is done with stage kt-1, whose slot is refilled now
        __builtin_amdgcn_sched_barrier(0);
        {   // stages past the end re-fetch the last one, so that every iteration issues the same number of pieces
            const int t = kt + NST - 1;
            issue(t < NKT ? t : NKT - 1, t % NST);
        }
        const char* st = smq + (kt % NST) * STAGE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(st + x_hi[s]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(st + x_lo[s]);
            f16x8 wh[6], wl[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                wh[i] = *reinterpret_cast<const f16x8*>(st + w_hi[s] + i * 32 * ROWB);
                wl[i] = *reinterpret_cast<const f16x8*>(st + w_lo[s] + i * 32 * ROWB);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // Q^T, K^T: D[dim][token]
                const f16x8 whs = wh[i] * down;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], ah, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], ah, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whs, al, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 4; i < 6; ++i) {  // V: D[token][dim]
                const f16x8 whs = wh[i] * down;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[i], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[i], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, whs, acc[i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every DMA has landed and every wave has read its last stage: the ring becomes K / V^T

    // ---- accumulators -> Q fragments (registers), K and V^T (LDS) ---------------------------------------------------------
    const float ws = a.w_scale;
    const float* bq = a.bias + head * SM_HEAD_DIM;
    const float* bk = bq + SM_EMBED;
    const float* bv = bk + SM_EMBED;
    const int key = wave * 32 + r;  // the token this lane holds as a key (Q^T / K^T blocks) ...
    f16x8 qh[4], ql[4];             // step t = 2 db + s: dims 32 db + 16 s + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 kh8, kl8;
            float qf[8], kf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = db * 32 + acc_row(8 * s + j, h);
                qf[j] = acc[db][8 * s + j] * ws + bq[d];
                kf[j] = acc[2 + db][8 * s + j] * ws + bk[d];
            }
            split8(qf, qh[2 * db + s], ql[2 * db + s]);
            split8(kf, kh8, kl8);
            if (key < QA_KROWS) {  // K row of this key: chunk 2 (4 db + 2 s + h) (+1 = lo), XOR-ed with key & 15
                const int c = 2 * (4 * db + 2 * s + h);
                *reinterpret_cast<f16x8*>(smq + key * 256 + ((c ^ (key & 15)) * 16)) = kh8;
                *reinterpret_cast<f16x8*>(smq + key * 256 + (((c + 1) ^ (key & 15)) * 16)) = kl8;
            }
        }
#pragma unroll
    for (int db = 0; db < 2; ++db) {  // ... and the head-dim it holds in the V blocks: rows = this wave's 32 tokens
        const int d = db * 32 + r;
        const float bias = bv[d];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (wave * 32 + 16 * u < QA_KROWS) {
                f16x8 vh8, vl8;
                float vf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = acc[4 + db][8 * u + j] * ws + bias;
                split8(vf, vh8, vl8);
                char* p = smq + QA_K_BYTES + d * QA_VLD + wave * 128 + u * 64 + h * 32;
                *reinterpret_cast<f16x8*>(p) = vh8;
                *reinterpret_cast<f16x8*>(p + 16) = vl8;
            }
        }
    }
    __syncthreads();  // K and V^T of the head are complete

    // ---- attention: this wave's 32 queries against all keys --------------------------------------------------------------
    const int q0 = wave * 32;
    if (q0 >= N) return;
    const float cs = a.scale * 1.44269504088896340736f;  // scores in log2 units
    const int nsteps16 = (N + 15) >> 4;                  // 16-key MFMA steps that hold at least one real key
    f32x16 om[2], oc[2];
#pragma unroll
    for (int v = 0; v < 16; ++v) { om[0][v] = 0.f; om[1][v] = 0.f; oc[0][v] = 0.f; oc[1][v] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int ksw = r & 15;
    int k_hi[4], k_lo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        k_hi[t] = ((2 * (2 * t + h)) ^ ksw) * 16;
        k_lo[t] = ((2 * (2 * t + h) + 1) ^ ksw) * 16;
    }
    const char* Vt = smq + QA_K_BYTES + r * QA_VLD + h * 32;  // + db * 32 rows, + kb * 128 + u * 64
    const int nch = (N + 63) >> 6;
    for (int c = 0; c < nch; ++c) {
        const int ck = min(64, N - c * 64);
        const int nb2 = (ck + 31) >> 5;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int v = 0; v < 16; ++v) s[kb][v] = -INFINITY;
            if (kb < nb2) {
                f32x16 mn, cr;
#pragma unroll
                for (int v = 0; v < 16; ++v) { mn[v] = 0.f; cr[v] = 0.f; }
                const char* kr = smq + (c * 64 + kb * 32 + r) * 256;
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
                if ((kb + 1) * 32 > ck) {  // keys past N: rows >= 208 alias other LDS data, rows 197..207 repeat a token
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
        const float lim = 8.0f / cs;  // lazy running maximum (attention_f16x2.hip): p <= 2^8 between moves
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
        f16x8 ph[2][2], pl[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf[j] = __builtin_amdgcn_exp2f(fmaf(s[kb][8 * u + j], cs, moff));
                    psum += pf[j];
                }
                split8(pf, ph[kb][u], pl[kb][u]);
            }
        l_run += psum;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if ((c * 2 + kb) * 2 + u < nsteps16) {  // wave-uniform: steps past the last real key hold no V data
                    const char* vp = Vt + (c * 2 + kb) * 128 + u * 64;
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const f16x8 vh = *reinterpret_cast<const f16x8*>(vp + db * 32 * QA_VLD);
                        const f16x8 vl = *reinterpret_cast<const f16x8*>(vp + db * 32 * QA_VLD + 16);
                        om[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kb][u], om[db], 0, 0, 0);
                        oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kb][u], oc[db], 0, 0, 0);
                        oc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kb][u], oc[db], 0, 0, 0);
                    }
                }
            }
    }

    const float l = halves_sum(l_run);
    const float inv = 1.0f / l;
    if (q0 + r < N) {
        float* Orow = a.O + ((int64_t)b * N + q0 + r) * a.ldo;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                float x[4], y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = (om[db][4 * g + e] + oc[db][4 * g + e] * (1.0f / 2048.0f)) * inv;
                    y[e] = (om[db][4 * g + 4 + e] + oc[db][4 * g + 4 + e] * (1.0f / 2048.0f)) * inv;
                }
                if (a.out_f16x2) {
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
#endif  // SM_TUNING

// ---- the kernel on v_mfma_f32_16x16x32_f16 ----------------------------------------------------------------------------------
// Less energy per FLOP at the chip's power limit (gemm_w16.hip, 16x16x32 variant) and 16-row granularity.  The operand
// reuse carries over tile for tile: a 16x16 accumulator holds, per lane (c = lane & 15, kg = lane >> 4), column c and rows
// 4 kg + reg; two such tiles along the contraction axis are the 8 k-elements lane-group kg supplies to one 32-k step of the
// next product, in the order k = 16 (j >> 2) + 4 kg + (j & 3) - the same on both operands, so Q (B operand of S^T = K Q^T)
// stays in registers, K rows and V^T rows go to LDS as the 16-B hi / lo pieces the consumer lane (same c, same kg) reads
// back, and P (the S^T accumulators) is the B operand of O^T = V^T P^T.  A wave owns two 16-token tiles.
typedef float f32x4q __attribute__((ext_vector_type(4)));
constexpr int QM_KROWS = QA_TOK;                 // 224 key rows kept (14 tiles; tile 13 only ever holds repeats of the last token)
constexpr int QM_K_BYTES = QM_KROWS * 256;       // K: [key][dim-step s ^ (key & 1)][slot]: 256 B per key
constexpr int QM_VLD = QA_WAVES * 128;           // V^T: [dim][32-key step][slot]: 896 B per head-dim (odd multiple of 128 B)
constexpr int QM_V_BYTES = 64 * QM_VLD;
static_assert(QM_K_BYTES + QM_V_BYTES <= QA_LDS, "K and V^T overlay the ring");

#ifdef SM_TUNING  // in-kernel stamps (tuning build only; written to a buffer nothing else reads): where a workgroup's life goes
__device__ unsigned long long g_qkv_stamps[1024 * QA_WAVES * 8];
#define QKV_STAMP(i) /* kept in registers, one record per wave; no load or store moves across a stamp */ \
    do { asm volatile("" ::: "memory"); stamp_[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#define QKV_STAMP_DECL unsigned long long stamp_[7] = {0, 0, 0, 0, 0, 0, 0}
#define QKV_STAMP_FLUSH \
    do { if (blockIdx.x < 1024 && lane == 0) for (int i_ = 0; i_ < 7; ++i_) g_qkv_stamps[(blockIdx.x * QA_WAVES + wave) * 8 + i_] = stamp_[i_]; } while (0)
#else
#define QKV_STAMP(i) do {} while (0)
#define QKV_STAMP_DECL do {} while (0)
#define QKV_STAMP_FLUSH do {} while (0)
#endif

// TERMS = 3: fp32-grade products everywhere (three MFMAs each).  TERMS = 1: throughput-mode diagnostic (hi x hi only in the
// projection, in K Q^T and in P V; softmax statistics stay fp32) - see gemm_w16.hip
// LOADERS = waves that issue the LDS-DMA pieces of the ring (the first LOADERS waves, pieces dealt round-robin over them).
// 7: every wave (round 2).  4: waves 0-3 - one per SIMD - feed the ring, waves 4-6 only compute: every wave leaves the stage
// barrier at the same moment, a stage is 52 pieces = ~830 cycles of the CU's one address unit, and with all seven waves queued on
// it the matrix pipe idled at the top of every stage (stage = DMA issue + MFMA instead of their maximum: 3.2k cycles for 2.3k
// of MFMA work).  With the younger wave of each SIMD free of DMA, its MFMAs run while the older one feeds the ring.
template <int NST, int TERMS = 3, int LOADERS = QA_WAVES>
__global__ __launch_bounds__(QA_WAVES * 64, 2) void qkv_attention_m16_kernel(sm_qkv_attn_args a) {
    constexpr int ROWB = 128, XT = QA_TOK * ROWB, WT = 192 * ROWB, STAGE = XT + WT;
    constexpr int XP = XT / 1024, NP = STAGE / 1024, PPW = (NP + LOADERS - 1) / LOADERS, DUMP = NST * STAGE;
    static_assert(DUMP + (PPW * LOADERS - NP) * 1024 <= QA_LDS, "ring + dump zone must fit in LDS");
    extern __shared__ __attribute__((aligned(16))) char smm[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    int head, b;
    {
        const int id = blockIdx.x, pairs = a.B * SM_HEADS;
        int lin = id;
        if ((pairs & 7) == 0) lin = (id & 7) * (pairs >> 3) + (id >> 3);
        b = lin / SM_HEADS;
        head = lin - b * SM_HEADS;
    }
    const int N = a.N;
    const char* X = reinterpret_cast<const char*>(a.Xn + (int64_t)b * N * a.ldx);
    const char* W = reinterpret_cast<const char*>(a.Wqkv);

    const bool loader = wave < LOADERS;
    const char* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wave + LOADERS * j;
        const int prow = lane >> 3;
        if (p < XP) {
            const int row = p * 8 + prow;
            const int tok = row < N ? row : N - 1;
            src[j] = X + (int64_t)tok * a.ldx * 4 + m16_chunk_of_slot(row, lane & 7) * 16;
        } else {
            const int row = ((p < NP ? p : XP) - XP) * 8 + prow;  // 0..191: [Q dims | K dims | V dims] of this head
            const int wrow = (row >> 6) * SM_EMBED + head * SM_HEAD_DIM + (row & 63);
            src[j] = W + (int64_t)wrow * SM_EMBED * 4 + m16_chunk_of_slot(row, lane & 7) * 16;
        }
    }
    auto issue = [&](int kt, int slot) {
        if (!loader) return;  // wave-uniform: the compute-only waves have no piece in flight, their vmcnt waits pass at once
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave + LOADERS * j;
            const unsigned d = p < NP ? lds0 + slot * STAGE + p * 1024 : lds0 + DUMP + (p - NP) * 1024;
            lds_dma16(src[j] + kt * ROWB, __builtin_amdgcn_readfirstlane(d));
        }
    };
    // fragment offsets: tiles start at multiples of 16 rows, so a lane's slot does not depend on the tile
    const int f_hi = c16 * ROWB + m16_slot(c16, kg, 0) * 16, f_lo = c16 * ROWB + m16_slot(c16, kg, 1) * 16;

    f32x4q acc[12][2];  // [Q dims 0-15 .. 48-63 | K ... | V ...][token tile of this wave]
#pragma unroll
    for (int d = 0; d < 12; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[d][t][v] = 0.f;
    const f16x8 down = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                        (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};  // 2^-11
    constexpr int NKT = SM_EMBED / 32;
    QKV_STAMP_DECL;
    QKV_STAMP(0);
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t, t);
    // LayerNorm folded into the projection (a.ln_stats; see gemm_w16.hip): Xn then holds the RAW residual stream, the weights carry
    // the norm's gain, and (mu r, r) per token come from the producer's twelve partials - computed here under the first stage's
    // latency, kept at the top of the LDS (above the ring and above the K / V^T overlay)
    float2* lnrow = reinterpret_cast<float2*>(smm + QA_LDS - QA_TOK * 8);
    static_assert(DUMP + (PPW * LOADERS - NP) * 1024 <= QA_LDS - QA_TOK * 8 && QM_K_BYTES + QM_V_BYTES <= QA_LDS - QA_TOK * 8, "row statistics above everything");
    if (a.ln_stats && tid < QA_TOK) {
        const int tok = tid < N ? tid : N - 1;
        const float4* sp = reinterpret_cast<const float4*>(a.ln_stats + ((int64_t)b * N + tok) * 24);
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
        const float rstd = 1.0f / sqrtf((m2 + 32.0f * dev) * (1.0f / 384.0f) + a.ln_eps);
        lnrow[tid] = make_float2(mu * rstd, rstd);
    }
    // A wave whose second 16-token tile lies wholly beyond N (wave 6 at N = 197: rows 208..223) skips that tile's MFMAs in
    // both phases: its accumulators stay zero, so its K rows / V^T columns hold the bias (finite; those keys are masked).
    const bool two_tiles = wave * 32 + 16 < N;
    auto project = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        for (int kt = 0; kt < NKT; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PPW) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt == 0) QKV_STAMP(1);
            {
                const int t = kt + NST - 1;
                issue(t < NKT ? t : NKT - 1, t % NST);
            }
            const char* st = smm + (kt % NST) * STAGE;
            f16x8 xh[2], xl[2];
    #pragma unroll
            for (int t = 0; t < NT; ++t) {
                xh[t] = *reinterpret_cast<const f16x8*>(st + (wave * 32 + t * 16) * ROWB + f_hi);
                if constexpr (TERMS == 3) xl[t] = *reinterpret_cast<const f16x8*>(st + (wave * 32 + t * 16) * ROWB + f_lo);
            }
    #pragma unroll
            for (int d = 0; d < 12; ++d) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(st + XT + d * 16 * ROWB + f_hi);
                f16x8 wl = wh, whs = wh;
                if constexpr (TERMS == 3) {
                    wl = *reinterpret_cast<const f16x8*>(st + XT + d * 16 * ROWB + f_lo);
                    whs = wh * down;
                }
    #pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (d < 8) {  // Q^T, K^T: D[dim][token]
                        acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[t], acc[d][t], 0, 0, 0);
                        if constexpr (TERMS == 3) {
                            acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[t], acc[d][t], 0, 0, 0);
                            acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whs, xl[t], acc[d][t], 0, 0, 0);
                        }
                    } else {      // V: D[token][dim]
                        acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], wh, acc[d][t], 0, 0, 0);
                        if constexpr (TERMS == 3) {
                            acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], wl, acc[d][t], 0, 0, 0);
                            acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[t], whs, acc[d][t], 0, 0, 0);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    if (two_tiles) project(std::integral_constant<int, 2>{});
    else project(std::integral_constant<int, 1>{});
    QKV_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the ring becomes K / V^T
    QKV_STAMP(3);

    // ---- accumulators -> Q fragments (registers), K and V^T (LDS) ---------------------------------------------------------
    const float ws = a.w_scale;
    const float* bq = a.bias + head * SM_HEAD_DIM;
    const float* bk = bq + SM_EMBED;
    const float* bv = bk + SM_EMBED;
    const bool fold = a.ln_stats != nullptr;  // then: value = acc (2^-s r) + (b' - (mu r) c), r and mu of the TOKEN, c of the output dim
    // (two copies of the conversion behind one wave-uniform branch: the plain path must not pay for the fold's loads and multiplies)
    f16x8 qh[2][2], ql[2][2];  // [query tile][dim step s]: element j = dim 32 s + 16 (j >> 2) + 4 kg + (j & 3)
    auto convert = [&](auto fold_tag) {
        constexpr bool FOLD = decltype(fold_tag)::value;
        const float* cq = FOLD ? a.ln_c + head * SM_HEAD_DIM : nullptr;
        const float* ck = FOLD ? cq + SM_EMBED : nullptr;
        const float* cv = FOLD ? ck + SM_EMBED : nullptr;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = wave * 32 + t * 16 + c16;
            float rs = ws, mur = 0.f;
            if constexpr (FOLD) {
                const float2 lr = lnrow[key];
                rs = ws * lr.y;
                mur = lr.x;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 kh8, kl8;
                float qf[8], kf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int dt = 2 * s + (j >> 2), d = 16 * dt + 4 * kg + (j & 3);
                    if constexpr (FOLD) {
                        qf[j] = acc[dt][t][j & 3] * rs + (bq[d] - mur * cq[d]);
                        kf[j] = acc[4 + dt][t][j & 3] * rs + (bk[d] - mur * ck[d]);
                    } else {
                        qf[j] = acc[dt][t][j & 3] * ws + bq[d];
                        kf[j] = acc[4 + dt][t][j & 3] * ws + bk[d];
                    }
                }
                split8(qf, qh[t][s], ql[t][s]);
                split8(kf, kh8, kl8);
                char* kp = smm + key * 256 + ((s ^ (key & 1)) * 128);
                *reinterpret_cast<f16x8*>(kp + m16_slot(key, kg, 0) * 16) = kh8;
                *reinterpret_cast<f16x8*>(kp + m16_slot(key, kg, 1) * 16) = kl8;
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {  // V^T row of head-dim 16 dt + c16: this wave's 32 tokens = key step `wave`
            const int d = 16 * dt + c16;
            const float bias = bv[d];
            f16x8 vh8, vl8;
            float vf[8];
            if constexpr (FOLD) {
                const float cdim = cv[d];
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // token of element j: 32 wave + 16 (j >> 2) + 4 kg + (j & 3)
                    const float2 lr = lnrow[wave * 32 + 16 * (j >> 2) + 4 * kg + (j & 3)];
                    vf[j] = acc[8 + dt][j >> 2][j & 3] * (ws * lr.y) + (bias - lr.x * cdim);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) vf[j] = acc[8 + dt][j >> 2][j & 3] * ws + bias;
            }
            split8(vf, vh8, vl8);
            char* vp = smm + QM_K_BYTES + d * QM_VLD + wave * 128;
            *reinterpret_cast<f16x8*>(vp + m16_slot(d, kg, 0) * 16) = vh8;
            *reinterpret_cast<f16x8*>(vp + m16_slot(d, kg, 1) * 16) = vl8;
        }
    };
    if (fold) convert(std::true_type{});
    else convert(std::false_type{});
    __syncthreads();
    QKV_STAMP(4);

    // ---- attention: this wave's two 16-query tiles against all keys, 32 keys per step ---------------------------------------
    const int q0 = wave * 32;
    if (q0 >= N) return;
    const float cs = a.scale * 1.44269504088896340736f;
    f32x4q om[4][2], oc[4][2];  // O^T tiles [dim tile][query tile]: lane holds query c16, dims 4 kg + reg
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) { om[dt][t][v] = 0.f; oc[dt][t][v] = 0.f; }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int nsteps = (N + 31) >> 5;
    auto attend = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        // scores of one 32-key step: S^T tile [key tile kt][query tile t] = sum over the two dim steps.  Software-pipelined one
        // step ahead: the MFMAs of step s + 1 are independent of the softmax of step s and run in its shadow (a lone wave used
        // to serialise 768 MFMA cycles and ~1000 VALU cycles per step).
        auto scores = [&](int stp, f32x4q (&sc)[2][2]) {  // NT: query tiles of this wave
    #pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = stp * 32 + kt * 16 + c16;
                f16x8 kh[2], kl[2];
    #pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const char* kp = smm + key * 256 + ((s ^ (key & 1)) * 128);
                    kh[s] = *reinterpret_cast<const f16x8*>(kp + m16_slot(key, kg, 0) * 16);
                    kl[s] = *reinterpret_cast<const f16x8*>(kp + m16_slot(key, kg, 1) * 16);
                }
    #pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4q mn = {0.f, 0.f, 0.f, 0.f}, cr = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        mn = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[s], qh[t][s], mn, 0, 0, 0);
                        if constexpr (TERMS == 3) {
                            cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[s], ql[t][s], cr, 0, 0, 0);
                            cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[s], qh[t][s], cr, 0, 0, 0);
                        }
                    }
                    if constexpr (TERMS == 3) sc[kt][t] = cr * (1.0f / 2048.0f) + mn;
                    else sc[kt][t] = mn;
                }
            }
        };
        f32x4q sc[2][2], sn[2][2];
        scores(0, sc);
        // One step = softmax + P V of 32 keys.  Layout of the body (round 3): the running-maximum test of BOTH query tiles comes
        // first and the (rare) rescale sits behind ONE wave-uniform branch, so that everything after it - the scores of the NEXT
        // step (MFMA, independent of this step's softmax), the exponentials / hi-lo split (VALU) and P V (MFMA) - is a single
        // basic block the scheduler can interleave.  Before, the scores were issued at the top of the body and two branches
        // stood between them and the softmax: the wave ran 24 MFMAs, then ~170 VALU instructions, then 24 MFMAs, each alone.
        // wave-priority experiment (tuning build: SM_QKV_PRIO; bits 1-2 of out_f16x2, 0 in the product): the two waves of a SIMD are
        // arbitrated by age, so wave w < 4 runs its attention steps nearly unimpeded (14.7k cycles) and its partner w + 4 gets the
        // leftovers (22.7k: it finishes alone).  1 = the younger wave at priority 1 throughout, 2 = priorities swapped every step.
        const int prio_mode = (a.out_f16x2 >> 1) & 3;
        if (prio_mode == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
        auto step = [&](int stp, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            if (prio_mode == 2) {
                if (((stp & 1) != 0) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (LAST) {  // only the last step can hold keys >= N (K rows 197..223 repeat the last token)
    #pragma unroll
                for (int kt = 0; kt < 2; ++kt)
    #pragma unroll
                    for (int t = 0; t < NT; ++t)
    #pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (stp * 32 + kt * 16 + 4 * kg + v >= N) sc[kt][t][v] = -INFINITY;
            }
            const float lim = 8.0f / cs;  // lazy running maximum: p <= 2^8 between moves
            float cmax[2];
            bool move = false;
    #pragma unroll
            for (int t = 0; t < NT; ++t) {
                float c = -INFINITY;
    #pragma unroll
                for (int kt = 0; kt < 2; ++kt)
    #pragma unroll
                    for (int v = 0; v < 4; ++v) c = fmaxf(c, sc[kt][t][v]);
                cmax[t] = rows4_max(c);
                move = move || cmax[t] > m_run[t] + lim;
            }
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
    #pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (__builtin_amdgcn_ballot_w64(cmax[t] > m_run[t] + lim) != 0) {
                        const float m_new = fmaxf(m_run[t], cmax[t]);
                        const float alpha = __builtin_amdgcn_exp2f((m_run[t] - m_new) * cs);
                        m_run[t] = m_new;
                        l_run[t] *= alpha;
    #pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
    #pragma unroll
                            for (int v = 0; v < 4; ++v) { om[dt][t][v] *= alpha; oc[dt][t][v] *= alpha; }
                    }
                }
            }
            if constexpr (!LAST) scores(stp + 1, sn);
            f16x8 ph[2], pl[2];
    #pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float moff = -m_run[t] * cs;
                float psum = 0.f, pf[8];
                const f32x4q e0 = sc[0][t] * cs + moff, e1 = sc[1][t] * cs + moff;  // packed fma
    #pragma unroll
                for (int j = 0; j < 8; ++j) {  // element j = key 16 (j >> 2) + 4 kg + (j & 3) of the step
                    pf[j] = __builtin_amdgcn_exp2f(j < 4 ? e0[j & 3] : e1[j & 3]);
                    psum += pf[j];
                }
                split8c(pf, ph[t], pl[t]);  // compiler-visible: P feeds the MFMAs below straight from registers
                l_run[t] += psum;
            }
    #pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = 16 * dt + c16;
                const char* vp = smm + QM_K_BYTES + d * QM_VLD + stp * 128;
                const f16x8 vh = *reinterpret_cast<const f16x8*>(vp + m16_slot(d, kg, 0) * 16);
                const f16x8 vl = *reinterpret_cast<const f16x8*>(vp + m16_slot(d, kg, 1) * 16);
    #pragma unroll
                for (int t = 0; t < NT; ++t) {
                    om[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[t], om[dt][t], 0, 0, 0);
                    if constexpr (TERMS == 3) {
                        oc[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[t], oc[dt][t], 0, 0, 0);
                        oc[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[t], oc[dt][t], 0, 0, 0);
                    }
                }
            }
            if constexpr (!LAST) {
    #pragma unroll
                for (int kt = 0; kt < 2; ++kt)
    #pragma unroll
                    for (int t = 0; t < NT; ++t) sc[kt][t] = sn[kt][t];
            }
        };
        for (int stp = 0; stp + 1 < nsteps; ++stp) step(stp, std::false_type{});
        step(nsteps - 1, std::true_type{});

        QKV_STAMP(5);
    #pragma unroll
        for (int t = 0; t < NT; ++t) {
            float l = l_run[t];
            l = rows4_sum(l);
            const float inv = 1.0f / l;
            const int q = q0 + t * 16 + c16;
            if (q < N) {
                float* Orow = a.O + ((int64_t)b * N + q) * a.ldo;
    #pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    float x[4];
    #pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = (om[dt][t][e] + oc[dt][t][e] * (1.0f / 2048.0f)) * inv;
                    const int d = head * SM_HEAD_DIM + 16 * dt + 4 * kg;  // this lane's four consecutive head-dims
                    if (a.out_f16x2 & 1) store_f16x2_4(Orow, d, x);
                    else *reinterpret_cast<float4*>(Orow + d) = make_float4(x[0], x[1], x[2], x[3]);
                }
            }
        }
    };
    if (two_tiles) attend(std::integral_constant<int, 2>{});
    else attend(std::integral_constant<int, 1>{});
    QKV_STAMP(6);
    QKV_STAMP_FLUSH;
}

}  // namespace sm

// kernel selection.  Product: qkv_attention_m16_kernel<2, 3, 4> (or <2, 1, 4>: the throughput-mode diagnostic).  Tuning build:
// SM_QKV_RING = "m16x2L4" (default) | "m16x2" (every wave feeds the ring: round 2) | "m16x3L4" | "m16x3" | "32x2" | "32x3" |
// "16x2" | "16x6" (the 32x32x16-MFMA kernel, "<k per stage>x<stages>") - same results up to summation order.
static int qkv_mode() {
#ifdef SM_TUNING
    static const char* ring = getenv("SM_QKV_RING");
    return !ring ? 6 : !strcmp(ring, "32x3") ? 1 : !strcmp(ring, "16x2") ? 2 : !strcmp(ring, "16x6") ? 3 : !strcmp(ring, "32x2") ? 0 :
           !strcmp(ring, "m16x3") ? 5 : !strcmp(ring, "m16x2") ? 4 : !strcmp(ring, "m16x3L4") ? 7 : 6;
#else
    return 6;
#endif
}
// name rocprofv3 reports for the selected kernel (labels the in-situ taps of forward.hip)
const char* sm_qkv_attention_kernel_name(int mfma_terms) {
    if (mfma_terms == 1) return "qkv_attention_m16_kernel<2, 1, 4>";  // the one-MFMA diagnostic launches this instantiation whatever the mode
    static const char* names[] = {"qkv_attention_kernel<32, 2>", "qkv_attention_kernel<32, 3>", "qkv_attention_kernel<16, 2>",
                                  "qkv_attention_kernel<16, 6>", "qkv_attention_m16_kernel<2>", "qkv_attention_m16_kernel<3>",
                                  "qkv_attention_m16_kernel<2, 3, 4>", "qkv_attention_m16_kernel<3, 3, 4>"};
    return names[qkv_mode()];
}

extern "C" int sm_qkv_attention_max_tokens(void) { return sm::QA_KROWS; }

extern "C" int sm_qkv_attention_w16(const sm_qkv_attn_args* a, void* stream) {
    SM_REQUIRE(a && a->Xn && a->Wqkv && a->bias && a->O, "sm_qkv_attention_w16: null pointer");
    SM_REQUIRE(a->B > 0 && a->N > 0 && a->N <= sm::QA_KROWS, "sm_qkv_attention_w16: N=%d tokens (1..%d: K and V of a head stay in LDS)",
               a ? a->N : 0, sm::QA_KROWS);
    SM_REQUIRE(a->scale > 0.f, "sm_qkv_attention_w16: scale must be positive");
    SM_REQUIRE(a->mfma_terms == 0 || a->mfma_terms == 1 || a->mfma_terms == 3, "sm_qkv_attention_w16: mfma_terms must be 0/3 or 1");
    if (a->ln_stats)
        SM_REQUIRE(a->ln_c && a->ln_eps > 0.f && ((uintptr_t)a->ln_stats % 16) == 0,
                   "sm_qkv_attention_w16: folded LayerNorm needs ln_c (1152 row sums of the gain-scaled weight), ln_eps and 16-B aligned statistics");
    int ex = 0;
    SM_REQUIRE(a->w_scale > 0.f && frexpf(a->w_scale, &ex) == 0.5f, "sm_qkv_attention_w16: w_scale must be the weight's 2^-s");
    SM_REQUIRE(a->ldx % 8 == 0 && a->ldx >= SM_EMBED && a->ldo % 8 == 0 && a->ldo >= SM_EMBED &&
                   ((uintptr_t)a->Xn | (uintptr_t)a->Wqkv | (uintptr_t)a->O) % 32 == 0,
               "sm_qkv_attention_w16: row strides must be multiples of 8 elements, pointers 32-B aligned");
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        const void* ks[] = {reinterpret_cast<const void*>(&sm::qkv_attention_m16_kernel<2, 3, 4>), reinterpret_cast<const void*>(&sm::qkv_attention_m16_kernel<2, 1, 4>),
#ifdef SM_TUNING
                            reinterpret_cast<const void*>(&sm::qkv_attention_kernel<32, 2>), reinterpret_cast<const void*>(&sm::qkv_attention_kernel<32, 3>),
                            reinterpret_cast<const void*>(&sm::qkv_attention_kernel<16, 2>), reinterpret_cast<const void*>(&sm::qkv_attention_kernel<16, 6>),
                            reinterpret_cast<const void*>(&sm::qkv_attention_m16_kernel<2>),
                            reinterpret_cast<const void*>(&sm::qkv_attention_m16_kernel<3, 3, 4>),
#endif
        };
        for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, sm::QA_LDS);
        (void)hipGetLastError();
    });
    const int mode = qkv_mode();
    const dim3 grid(a->B * SM_HEADS), block(sm::QA_WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
    sm_qkv_attn_args args = *a;
    args.out_f16x2 = a->out_f16x2 ? 1 : 0;
#ifdef SM_TUNING
    static const int prio = getenv("SM_QKV_PRIO") ? atoi(getenv("SM_QKV_PRIO")) & 3 : 0;
    args.out_f16x2 |= prio << 1;
#endif
    a = &args;
    if (a->mfma_terms == 1) hipLaunchKernelGGL((sm::qkv_attention_m16_kernel<2, 1, 4>), grid, block, sm::QA_LDS, st, *a);
    else if (mode == 6) hipLaunchKernelGGL((sm::qkv_attention_m16_kernel<2, 3, 4>), grid, block, sm::QA_LDS, st, *a);
#ifdef SM_TUNING
    else if (mode == 4) hipLaunchKernelGGL((sm::qkv_attention_m16_kernel<2>), grid, block, sm::QA_LDS, st, *a);
    else if (mode == 5) { sm::set_error("sm_qkv_attention_w16: SM_QKV_RING=m16x3 was retired (use m16x3L4)"); return SM_EINVAL; }
    else if (mode == 7) hipLaunchKernelGGL((sm::qkv_attention_m16_kernel<3, 3, 4>), grid, block, sm::QA_LDS, st, *a);
    else if (mode == 1) hipLaunchKernelGGL((sm::qkv_attention_kernel<32, 3>), grid, block, sm::QA_LDS, st, *a);
    else if (mode == 2) hipLaunchKernelGGL((sm::qkv_attention_kernel<16, 2>), grid, block, sm::QA_LDS, st, *a);
    else if (mode == 3) hipLaunchKernelGGL((sm::qkv_attention_kernel<16, 6>), grid, block, sm::QA_LDS, st, *a);
    else hipLaunchKernelGGL((sm::qkv_attention_kernel<32, 2>), grid, block, sm::QA_LDS, st, *a);
#endif
    return sm::check_launch("sm_qkv_attention_w16");
}

#ifdef SM_TUNING
extern "C" int sm_qkv_stamps(unsigned long long* host_out, int count) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sm::g_qkv_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif
