// Evaluator post-processing + the seven saliency metrics, fused on the device.
//
// Reference: evaluator.pyc@L199-228 (last layer -> bilinear up-sample -> crop -> upper-bound query by IoU ->
// arg-max objectness -> _update_meters) and metrics/{iou,f_measure,mae,pixel_acc,s_measure}.py.  The reference does
// ~14 device->host syncs per image and replicates the mask 255x for F-max (metrics/f_measure.py:61-62); here seven
// stream-ordered launches per BATCH leave 16 floats per image:
//   K0  eval_transpose       masks -> [pixel][32 queries] per pass of 32 queries, + the band table of every image
//   K1  eval_query           |p>0.5 & g|, |p>0.5 | g| of every query, ground-truth moments (per-slot partial counts)
//   K1b eval_reduce_query    partials -> per-query counts, IoU, the two selected queries (arg-max objectness, arg-max IoU)
//   K2  eval_sum / K2b eval_adapt   sum p of the selected masks -> adaptive threshold 2 * mean (f_measure.py:76)
//   K3  eval_metrics         one pass per selected mask: threshold counts, the 2 x 256-bin histogram behind F-max (exactly the
//                            255 strict '>' thresholds k/255 of f_measure.py:65), |p - g|, per-quadrant and per-class moments
//   K4  eval_finalize        the 7 values with the reference's fp32 operation order
// K1-K3 walk an image in one of two ways, chosen per image: the BAND walk (below) for up-sampled masks - the evaluator's
// case - and the RASTER walk (2048-pixel chunks, taps staged in LDS) otherwise.  Bound: vector / scalar issue, not HBM (a batch
// of 64 reads ~30 MB: its ground truths three times and the 28 x 28 masks; 169 us against 311 us for the raster walk alone).
#include "common.h"
#include <math.h>

// The integer-count metrics must round exactly like torch's separate fp32 mul / add / div kernels, so this file is
// compiled with contraction off and uses plain operators (HIP's __fmul_rn/__fadd_rn are inline header functions
// whose instructions keep the header's contract=fast flag and DO get fused after inlining: F-measure came out
// 2 ulp off).  Fused ops are written explicitly (__builtin_fmaf) where torch's CPU kernel fuses.
#pragma clang fp contract(off)
#define SM_MUL(a, b) ((a) * (b))
#define SM_ADD(a, b) ((a) + (b))
#define SM_DIV(a, b) ((a) / (b))

namespace sm {

struct UpIdx { int i0, i1; float l0, l1; };

// at::native::area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false) + the bilinear taps
__device__ __forceinline__ UpIdx up_index(int dst, float scale, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    UpIdx u;
    u.i0 = (int)src;
    u.i1 = u.i0 + (u.i0 < in_size - 1 ? 1 : 0);
    u.l1 = src - (float)u.i0;
    u.l0 = 1.0f - u.l1;
    return u;
}

__device__ __forceinline__ float up_sample(const float* __restrict__ m, int mw, const UpIdx& uy, const UpIdx& ux) {
    const float p00 = m[uy.i0 * mw + ux.i0], p01 = m[uy.i0 * mw + ux.i1];
    const float p10 = m[uy.i1 * mw + ux.i0], p11 = m[uy.i1 * mw + ux.i1];
    // Bit-for-bit the arithmetic of torch-CPU's upsample_bilinear2d (ATen UpSampleKernel.cpp, compiled with fma
    // contraction; established by brute force against F.interpolate): along x then y, each level
    // fma(first_tap, w_first, second_tap * w_second).
    const float top = __builtin_fmaf(p00, ux.l0, p01 * ux.l1);
    const float bot = __builtin_fmaf(p10, ux.l0, p11 * ux.l1);
    return __builtin_fmaf(top, uy.l0, bot * uy.l1);
}

template <typename T>
__device__ __forceinline__ T wave_total(T v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int EV_THREADS = 256;
constexpr int EV_PPT = 8;                       // pixels per thread
constexpr int EV_CHUNK = EV_THREADS * EV_PPT;   // pixels per workgroup
constexpr int EV_MAXQ = 32;                     // queries handled per pass of the query kernel
constexpr int EV_NACC = 22;                     // fp64 partial sums per (image, which, chunk)

struct QueryStats { unsigned inter, uni; };
struct GtStats { unsigned long long sum_g, sum_gx, sum_gy; };

// pixel index -> (y, x) without an integer division: float reciprocal + one correction step (idx < 2^24)
__device__ __forceinline__ void split_idx(int idx, int W, float inv_w, int& y, int& x) {
    y = (int)((float)idx * inv_w);
    x = idx - y * W;
    if (x < 0) { --y; x += W; } else if (x >= W) { ++y; x -= W; }
}

constexpr int EV_QS = 32;  // query stride of the transposed mask copy (floats): one 128-B line per low-res pixel

// ---- band walk ----------------------------------------------------------------------------------------------------------
// When the ground truth is larger than the mask (the evaluator's case: a 28 x 28 mask against a 300-400 px GT, or the reference
// mode's x8) the kernels walk an image in BANDS: the GT rows [ytab[i], ytab[i + 1]) whose upper tap is low-res row i.  A work
// unit is (band i, 64 GT columns), one wave each, one lane per column:
//   * a lane's horizontal interpolation top = p[i][j0] lx0 + p[i][j1] lx1, bot = p[i+1][..] is the same for every row of the
//     band: computed once per unit, the per-pixel work shrinks to fma(top, ly0, bot * ly1) - exactly the last step of
//     torch's arithmetic, so the values are bit-identical to the raster walk's;
//   * the row's vertical weights are wave-uniform, the GT bytes of a row are one coalesced 64-byte read, and no pixel needs
//     index arithmetic;
//   * 0/1 counts are ballots + s_bcnt1 on the scalar unit or carry-in adds, not per-lane adds behind wave reductions.
// The raster walk (2048-pixel chunks, taps staged in LDS) stays for images that are not up-sampled at least twice; the walk is
// chosen PER IMAGE (a row of results must not depend on its neighbours in the batch): band_walk().
constexpr int EV_BAND_MIN = 2;  // an image takes the band walk from H >= EV_BAND_MIN * mh (kernel argument band_min; the
                                // tuning build reads SM_EVAL_BAND_MIN, 0 = raster walk for every image)
constexpr int EV_NR = 4;        // GT rows per pass over the queries (query kernel)
constexpr int EV_NW = EV_THREADS / 64;

template <int N> struct IntTag { static constexpr int value = N; };

constexpr int EV_UPW = 4;          // work units per wave: a wave ends with cross-lane reductions (16 fp64 sums in the metrics kernel,
                                   // two counts per query in the query kernel) - a third of the instructions of a one-unit wave
constexpr int EV_BAND_MAX_H = 16383;  // a lane of the query kernel counts rows in 16 bits: EV_UPW units x at most H rows each
static_assert(EV_UPW * EV_BAND_MAX_H < 65536, "per-lane 16-bit row counts of the query kernel");
// `walk` (a kernel argument) = band_min | units per wave << 8: EV_BAND_MIN | EV_UPW << 8 in the product, SM_EVAL_BAND_MIN /
// SM_EVAL_UPW (<= EV_UPW) in the tuning build
__device__ __forceinline__ bool band_walk(const sm_eval_image& im, int mh, int walk) {
    const int band_min = walk & 255;
    return band_min > 0 && im.H >= band_min * mh && im.H <= EV_BAND_MAX_H;
}
__device__ __forceinline__ int band_units(const sm_eval_image& im, int mh) { return mh * ((im.W + 63) >> 6); }
// partial-sum slots an image uses (its workgroups blockIdx.x < slots_of write one each): a workgroup's four waves take the
// units 4 * blockIdx.x + wave, stepping by 4 * slots - `upw` units per wave - or the raster chunk blockIdx.x.
__device__ __forceinline__ int slots_of(const sm_eval_image& im, int mh, int band_min, int nslot) {
    if (!band_walk(im, mh, band_min)) return (im.H * im.W + EV_CHUNK - 1) / EV_CHUNK;
    const int upw = (band_min >> 8) > 0 ? band_min >> 8 : 1;
    const int s = (band_units(im, mh) + EV_NW * upw - 1) / (EV_NW * upw);
    return s < nslot ? s : nslot;
}

// first destination index d in [0, n] whose (clamped) upper source tap is >= i: the map d -> tap is monotone, so the rows of
// tap i are [first_dst(i), first_dst(i + 1)).  Same float expression as up_index (this file: contraction off).
__device__ __forceinline__ int first_dst(int i, float scale, int n, int in_size) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        float src = scale * ((float)mid + 0.5f) - 0.5f;
        src = src < 0.f ? 0.f : src;
        int t = (int)src;
        t = t < in_size - 1 ? t : in_size - 1;
        if (t >= i) hi = mid; else lo = mid + 1;
    }
    return lo;
}

constexpr int EV_GROUP = 16;  // GT rows whose bytes a lane requests together

// one lane's column of a unit: x, whether it exists, its two low-res columns and their weights
struct BandCol { int x, j0, j1; bool in; float lx0, lx1; };
__device__ __forceinline__ BandCol band_col(int seg, int lane, int W, float sx, int mw) {
    BandCol c;
    c.x = seg * 64 + lane;
    c.in = c.x < W;
    const UpIdx ux = up_index(c.in ? c.x : 0, sx, mw);
    c.j0 = ux.i0; c.j1 = ux.i1; c.lx0 = ux.l0; c.lx1 = ux.l1;
    return c;
}

// K0: maskT[b][p][q] = mask_pred[b][q][p] (q padded to 32) so the nq queries of one bilinear tap are one cache line
// (models with more than 32 queries - the reference constructor's default is 100, maskformer.py:13 - take one pass of
// the query kernel per group of 32: blockIdx.z, copy [pass][b][p][32])
// (+ the band table of the image, once: ytab[b][i] = first GT row of band i, ytab[b][mh] = H)
__global__ __launch_bounds__(256) void eval_transpose_kernel(sm_eval_args a, float* __restrict__ maskT, int* __restrict__ ytab) {
    const int b = blockIdx.y, plane = a.mh * a.mw, qbase = blockIdx.z * EV_MAXQ;
    if (blockIdx.x == 0 && blockIdx.z == 0) {
        const sm_eval_image im = a.images[b];
        const float sy = a.scale > 0.f ? 1.0f / a.scale : (float)a.mh / (float)im.H;
        for (int i = threadIdx.x; i <= a.mh; i += 256) ytab[b * (a.mh + 1) + i] = i < a.mh ? first_dst(i, sy, im.H, a.mh) : im.H;
    }
    const float* __restrict__ m0 = a.mask_pred + (int64_t)b * a.mask_stride_b;
    float* out = maskT + ((int64_t)blockIdx.z * a.B + b) * plane * EV_QS;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < plane * EV_QS; t += gridDim.x * 256) {
        const int p = t / EV_QS, q = qbase + (t - p * EV_QS);
        out[t] = q < a.nq ? m0[(int64_t)q * plane + p] : 0.f;
    }
}

// K1: one workgroup per (pixel chunk, image): every thread up-samples ALL queries at its pixels - index math, the GT
// byte and the four tap addresses are shared by the nq queries.  The low-res rows the chunk touches (a 2048-pixel
// chunk spans ~7 image rows = 2-3 low-res rows) are staged in LDS first: per-lane global loads of the taps cost a full
// texture-addresser pass each even when 10-14 neighbouring lanes want the same address (measured: 320 us for the
// batch, TA-bound); ds_read_b128 from LDS is an order of magnitude cheaper.  Integer counts go to global memory with
// integer atomics (deterministic).
constexpr int EV_LDS_ROWS = 8;  // low-res rows staged per chunk (falls back to global loads if the chunk needs more)
constexpr int EV_LDS_BYTES = 60 * 1024;  // staging budget: wide masks stage fewer rows (lds_rows = budget / row bytes); with the
                                         // kernel's ~1.1 KiB of static LDS (red_u, red_g) the workgroup stays under the 64-KiB default limit

// bit r: the lane's GT pixel of row yg + r (r < EV_GROUP, rows < yb): EV_GROUP independent byte loads in flight
__device__ __forceinline__ unsigned band_gt_bits(const unsigned char* __restrict__ gt, int W, const BandCol& c, int yg, int yb) {
    // unconditional loads from clamped addresses, masked afterwards: a predicated load is a basic block of its own and the
    // compiler waits for it before the next one (measured: sixteen serial round trips per group)
    const unsigned char* col = gt + (c.in ? c.x : 0);
    unsigned char v[EV_GROUP];
#pragma unroll
    for (int r = 0; r < EV_GROUP; ++r) v[r] = col[(yg + r < yb ? yg + r : yb - 1) * W];
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < EV_GROUP; ++r) bits |= (v[r] != 0 ? 1u : 0u) << r;
    const int nr = yb - yg;
    return c.in ? (nr >= 32 ? bits : bits & ((1u << nr) - 1u)) : 0u;
}

// K1, band walk (see above).  Per unit: top / bot of every query of this pass for the lane's column (4 x 16-byte loads per 4
// queries, once).  Per GT row and query pair one packed mul and one packed fma, per query one compare and one carry-in add that
// shifts the outcome into a per-lane bit word (word = 2 word + bin): no scalar instruction, no cross-lane traffic per pixel.
// Every 32 rows (and when the unit ends) the words are counted: |bin| += popc(word), |bin & g| += popc(word & gword), packed
// 16 + 16 bits in one register per query (band_walk() bounds the rows a lane can see); the waves reduce the counts across
// lanes once, at their end.  BQ4 = groups of four queries the registers are sized for (5: the shipped 20 queries, 8: a full pass).
template <int BQ4>
__device__ __forceinline__ void query_bands(const sm_eval_args& a, const sm_eval_image& im, const float* __restrict__ mt,
                                            const unsigned char* __restrict__ gt, const int* __restrict__ ytab, int nqp,
                                            bool first_pass, float sy, float sx, int slots, QueryStats* qslot, GtStats* gslot) {
    constexpr int BQ = BQ4 * 4;
    __shared__ unsigned wc[EV_NW][2 * EV_MAXQ + 1];  // per wave: |bin & g|, |bin| per query; [2 * EV_MAXQ] = |g|
    __shared__ unsigned long long wg[EV_NW][2];
    static_assert(BQ <= EV_MAXQ && EV_MAXQ <= 64, "lane q of a wave carries the counts of query q");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nseg = (im.W + 63) >> 6, units = a.mh * nseg;
    const int nq4 = (nqp + 3) >> 2;
    unsigned cnt[BQ];  // this lane: |bin| in the low, |bin & g| in the high 16 bits
#pragma unroll
    for (int q = 0; q < BQ; ++q) cnt[q] = 0;
    unsigned ngw = 0, sgx = 0;
    unsigned long long sgy = 0;  // wave-uniform
    for (int u = blockIdx.x * EV_NW + wv; u < units; u += slots * EV_NW) {
        const int i0 = u / nseg, seg = u - i0 * nseg, i1 = i0 + (i0 < a.mh - 1 ? 1 : 0);
        const int ya = ytab[i0], yb = ytab[i0 + 1];
        if (ya >= yb) continue;
        const BandCol c = band_col(seg, lane, im.W, sx, a.mw);
        f32x2 top[BQ / 2], bot[BQ / 2];  // query pairs: the per-pixel mul and fma are packed (v_pk_mul_f32 / v_pk_fma_f32)
        {
            const float4* __restrict__ t00 = reinterpret_cast<const float4*>(mt + (i0 * a.mw + c.j0) * EV_QS);
            const float4* __restrict__ t01 = reinterpret_cast<const float4*>(mt + (i0 * a.mw + c.j1) * EV_QS);
            const float4* __restrict__ t10 = reinterpret_cast<const float4*>(mt + (i1 * a.mw + c.j0) * EV_QS);
            const float4* __restrict__ t11 = reinterpret_cast<const float4*>(mt + (i1 * a.mw + c.j1) * EV_QS);
#pragma unroll
            for (int q4 = 0; q4 < BQ4; ++q4) {
                if (q4 < nq4) {
                    const float4 p00 = t00[q4], p01 = t01[q4], p10 = t10[q4], p11 = t11[q4];
                    const float v00[4] = {p00.x, p00.y, p00.z, p00.w}, v01[4] = {p01.x, p01.y, p01.z, p01.w};
                    const float v10[4] = {p10.x, p10.y, p10.z, p10.w}, v11[4] = {p11.x, p11.y, p11.z, p11.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // a column past the image's width carries zeros: never above 0.5
                        top[q4 * 2 + (e >> 1)][e & 1] = c.in ? __builtin_fmaf(v00[e], c.lx0, v01[e] * c.lx1) : 0.f;
                        bot[q4 * 2 + (e >> 1)][e & 1] = c.in ? __builtin_fmaf(v10[e], c.lx0, v11[e] * c.lx1) : 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) { top[q4 * 2 + e] = f32x2{0.f, 0.f}; bot[q4 * 2 + e] = f32x2{0.f, 0.f}; }
                }
            }
        }
        unsigned word[BQ], gword = 0, cnt_g = 0;  // bit r of word[q] / gword: bin of query q / GT, r rows ago
        int nbits = 0;
#pragma unroll
        for (int q = 0; q < BQ; ++q) word[q] = 0;
        auto flush = [&]() {
#pragma unroll
            for (int q4 = 0; q4 < BQ4; ++q4) {
                if (q4 < nq4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = q4 * 4 + e;
                        cnt[q] += (unsigned)__builtin_popcount(word[q]) + ((unsigned)__builtin_popcount(word[q] & gword) << 16);
                        word[q] = 0;
                    }
                }
            }
            cnt_g += (unsigned)__builtin_popcount(gword);
            gword = 0;
            nbits = 0;
        };
        auto rows = [&](auto nr_tag, int y0, unsigned gb) {  // gb bit r: this lane's GT pixel of row y0 + r
            constexpr int NR = decltype(nr_tag)::value;
            float ly0[NR], ly1[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int y = y0 + r;
                const UpIdx uy = up_index(y, sy, a.mh);
                ly0[r] = uy.l0; ly1[r] = uy.l1;
                const bool g = ((gb >> r) & 1u) != 0;
                gword = gword + gword + (g ? 1u : 0u);
                const unsigned ng_row = (unsigned)__popcll(__ballot(g));
                ngw += ng_row;
                if (first_pass) sgy += (unsigned long long)ng_row * (unsigned)y;
            }
#pragma unroll
            for (int q4 = 0; q4 < BQ4; ++q4) {
                if (q4 < nq4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int q = q4 * 4 + e;
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            // element e & 1 of the pair's packed fma(top, ly0, bot * ly1): the same roundings as the scalar form
                            const f32x2 v = __builtin_elementwise_fma(top[q >> 1], f32x2{ly0[r], ly0[r]}, bot[q >> 1] * f32x2{ly1[r], ly1[r]});
                            // word = 2 word + (v > 0.5): compare into vcc, add with carry-in.  As inline assembly because the
                            // compiler turns the C form into shift / select / or chains behind ALL the compares of a pass, whose
                            // 64-bit masks then spill the scalar file.
                            // (s_nop 0: a packed-fp32 result needs one wait state before a VALU reads it - hipcc pads its own
                            // instructions, not the ones inside an asm statement.)
                            asm("s_nop 0\n\tv_cmp_lt_f32_e32 vcc, 0.5, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(word[q]) : "v"(v[e & 1]) : "vcc");
                        }
                    }
                }
            }
            nbits += NR;
        };
        // The GT bytes of EV_GROUP rows are requested together, ahead of the passes that use them: a wave is a chain of
        // dependent global loads (band table -> taps, GT -> counts) and with one load per pass its latency, not its
        // instructions, set the kernel's time.
        for (int yg = ya; yg < yb; yg += EV_GROUP) {
            const unsigned gbits = band_gt_bits(gt, im.W, c, yg, yb);
            const int ye = yg + EV_GROUP < yb ? yg + EV_GROUP : yb;
            int y = yg;
            for (; y + EV_NR <= ye; y += EV_NR) {
                rows(IntTag<EV_NR>{}, y, gbits >> (y - yg));
                if (nbits == 32) flush();
            }
            switch (ye - y) {  // only the last group has a remainder; at most 28 bits are in use here
                case 1: rows(IntTag<1>{}, y, gbits >> (y - yg)); break;
                case 2: rows(IntTag<2>{}, y, gbits >> (y - yg)); break;
                case 3: rows(IntTag<3>{}, y, gbits >> (y - yg)); break;
                default: break;
            }
        }
        static_assert(EV_NR == 4 && 32 % EV_NR == 0 && EV_GROUP % EV_NR == 0, "the remainder switch covers 1..3 rows; a word holds whole passes");
        flush();
        if (first_pass) sgx += cnt_g * (unsigned)c.x;
    }
    unsigned acc_i = 0, acc_t = 0;  // lane q: the wave's counts of query q
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        if (q < nqp) {
            const unsigned t = wave_total(cnt[q] & 0xffffu), i = wave_total(cnt[q] >> 16);
            acc_t = lane == q ? t : acc_t;
            acc_i = lane == q ? i : acc_i;
        }
    }
    if (lane < EV_MAXQ) { wc[wv][2 * lane] = acc_i; wc[wv][2 * lane + 1] = acc_t; }
    if (lane == 0) wc[wv][2 * EV_MAXQ] = ngw;
    if (first_pass) {
        const unsigned long long tx = wave_total((unsigned long long)sgx);
        if (lane == 0) { wg[wv][0] = tx; wg[wv][1] = sgy; }
    }
    __syncthreads();
    unsigned ng = 0;
    for (int k = 0; k < EV_NW; ++k) ng += wc[k][2 * EV_MAXQ];
    if (tid < nqp) {
        unsigned in = 0, tot = 0;
        for (int k = 0; k < EV_NW; ++k) { in += wc[k][2 * tid]; tot += wc[k][2 * tid + 1]; }
        QueryStats o; o.inter = in; o.uni = tot + ng - in;  // |bin | g| = |bin| + |g| - |bin & g|
        qslot[tid] = o;
    }
    if (tid == 0 && first_pass) {
        GtStats o; o.sum_g = ng; o.sum_gx = 0; o.sum_gy = 0;
        for (int k = 0; k < EV_NW; ++k) { o.sum_gx += wg[k][0]; o.sum_gy += wg[k][1]; }
        *gslot = o;
    }
}

template <int BQ4>
__global__ __launch_bounds__(EV_THREADS) void eval_query_kernel(sm_eval_args a, const float* __restrict__ maskT_all,
                                                                const int* __restrict__ ytab, QueryStats* qpart, GtStats* gpart,
                                                                int nslot, int lds_rows, int band_min) {
    extern __shared__ __attribute__((aligned(16))) float taps[];  // [rows][mw][EV_QS] (raster walk)
    __shared__ unsigned red_u[EV_THREADS / 64][2 * EV_MAXQ];
    __shared__ unsigned long long red_g[EV_THREADS / 64][3];
    const int b = blockIdx.y;
    const int qbase = blockIdx.z * EV_MAXQ, nqp = min(EV_MAXQ, a.nq - qbase);  // this pass: queries qbase .. qbase+nqp-1
    const float* __restrict__ maskT = maskT_all + (int64_t)blockIdx.z * a.B * a.mh * a.mw * EV_QS;
    const sm_eval_image im = a.images[b];
    const int slots = slots_of(im, a.mh, band_min, nslot);
    if ((int)blockIdx.x >= slots) return;  // the reducers stop at slots_of() too
    const int npx = im.H * im.W;
    const int base = blockIdx.x * EV_CHUNK;
    QueryStats* qslot = qpart + ((int64_t)b * nslot + blockIdx.x) * a.nq + qbase;  // per-slot partials: no atomics (thousands
    GtStats* gslot = gpart + (int64_t)b * nslot + blockIdx.x;                      // of adds on two cache lines serialise in L2)
    const unsigned char* __restrict__ gt = a.gt + im.gt_off;
    const float* __restrict__ mt = maskT + (int64_t)b * a.mh * a.mw * EV_QS;
    const float sy = a.scale > 0.f ? 1.0f / a.scale : (float)a.mh / (float)im.H;
    const float sx = a.scale > 0.f ? 1.0f / a.scale : (float)a.mw / (float)im.W;
    if (band_walk(im, a.mh, band_min)) {
        query_bands<BQ4>(a, im, mt, gt, ytab + b * (a.mh + 1), nqp, qbase == 0, sy, sx, slots, qslot, gslot);
        return;
    }
    const float inv_w = 1.0f / (float)im.W;
    const int nq4 = (nqp + 3) >> 2;
    // low-res rows touched by this chunk
    const int last = min(base + EV_CHUNK, npx) - 1;
    const int r_lo = up_index(base / im.W, sy, a.mh).i0, r_hi = up_index(last / im.W, sy, a.mh).i1;
    const bool staged = (r_hi - r_lo + 1) <= lds_rows;
    if (staged) {
        const int n4 = (r_hi - r_lo + 1) * a.mw * (EV_QS / 4);
        const float4* src = reinterpret_cast<const float4*>(mt + (int64_t)r_lo * a.mw * EV_QS);
        for (int t = threadIdx.x; t < n4; t += EV_THREADS) reinterpret_cast<float4*>(taps)[t] = src[t];
    }
    __syncthreads();
    unsigned inter[EV_MAXQ], uni[EV_MAXQ];
#pragma unroll
    for (int q = 0; q < EV_MAXQ; ++q) { inter[q] = 0; uni[q] = 0; }
    unsigned long long sg = 0, sgx = 0, sgy = 0;
    // one body, two call sites, so each keeps its pointer's address space (ds_read for the staged copy, global
    // loads for the fallback): never select between an LDS and a global pointer at run time
    auto body = [&](const float* tb, int row0) {
        for (int k = 0; k < EV_PPT; ++k) {
            const int idx = base + k * EV_THREADS + threadIdx.x;
            if (idx >= npx) break;
            int y, x;
            split_idx(idx, im.W, inv_w, y, x);
            const UpIdx uy = up_index(y, sy, a.mh), ux = up_index(x, sx, a.mw);
            const unsigned g = gt[idx] != 0;
            if (g) { sg += 1; sgx += (unsigned)x; sgy += (unsigned)y; }
            const float4* t00 = reinterpret_cast<const float4*>(tb + ((uy.i0 - row0) * a.mw + ux.i0) * EV_QS);
            const float4* t01 = reinterpret_cast<const float4*>(tb + ((uy.i0 - row0) * a.mw + ux.i1) * EV_QS);
            const float4* t10 = reinterpret_cast<const float4*>(tb + ((uy.i1 - row0) * a.mw + ux.i0) * EV_QS);
            const float4* t11 = reinterpret_cast<const float4*>(tb + ((uy.i1 - row0) * a.mw + ux.i1) * EV_QS);
#pragma unroll
            for (int q4 = 0; q4 < EV_MAXQ / 4; ++q4) {
                if (q4 < nq4) {
                    const float4 p00 = t00[q4], p01 = t01[q4], p10 = t10[q4], p11 = t11[q4];
                    const float v00[4] = {p00.x, p00.y, p00.z, p00.w}, v01[4] = {p01.x, p01.y, p01.z, p01.w};
                    const float v10[4] = {p10.x, p10.y, p10.z, p10.w}, v11[4] = {p11.x, p11.y, p11.z, p11.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float top = __builtin_fmaf(v00[e], ux.l0, v01[e] * ux.l1);
                        const float bot = __builtin_fmaf(v10[e], ux.l0, v11[e] * ux.l1);
                        const unsigned bin = __builtin_fmaf(top, uy.l0, bot * uy.l1) > 0.5f;
                        inter[q4 * 4 + e] += bin & g;
                        uni[q4 * 4 + e] += bin | g;
                    }
                }
            }
        }
    };
    if (staged) body(taps, r_lo);
    else body(mt, 0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < EV_MAXQ; ++q) {
        if (q < nqp) {
            const unsigned i = wave_total(inter[q]), u = wave_total(uni[q]);
            if (lane == 0) { red_u[wv][2 * q] = i; red_u[wv][2 * q + 1] = u; }
        }
    }
    sg = wave_total(sg); sgx = wave_total(sgx); sgy = wave_total(sgy);
    if (lane == 0) { red_g[wv][0] = sg; red_g[wv][1] = sgx; red_g[wv][2] = sgy; }
    __syncthreads();
    if (threadIdx.x < nqp) {
        const int q = threadIdx.x;
        QueryStats o;
        o.inter = red_u[0][2 * q] + red_u[1][2 * q] + red_u[2][2 * q] + red_u[3][2 * q];
        o.uni = red_u[0][2 * q + 1] + red_u[1][2 * q + 1] + red_u[2][2 * q + 1] + red_u[3][2 * q + 1];
        qslot[q] = o;
    }
    if (threadIdx.x == 0 && qbase == 0) {  // the ground-truth sums do not depend on the query pass
        GtStats o;
        o.sum_g = red_g[0][0] + red_g[1][0] + red_g[2][0] + red_g[3][0];
        o.sum_gx = red_g[0][1] + red_g[1][1] + red_g[2][1] + red_g[3][1];
        o.sum_gy = red_g[0][2] + red_g[1][2] + red_g[2][2] + red_g[3][2];
        *gslot = o;
    }
}

// K1b: sum the per-chunk partials of an image (one thread per query)
__device__ int select_query(const sm_eval_args& a, const QueryStats* qs, int b, int which, bool write_ious);

__global__ __launch_bounds__(1024) void eval_reduce_query_kernel(sm_eval_args a, const QueryStats* qpart, const GtStats* gpart,
                                                               QueryStats* qs, GtStats* gs, int* sel, int nslot, int band_min) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int nchunk = slots_of(a.images[b], a.mh, band_min, nslot);  // the slots this image's walk wrote
    if (t < a.nq) {
        QueryStats o; o.inter = 0; o.uni = 0;
        #pragma unroll 8  // the loads do not depend on the running sum: keep eight in flight (add order unchanged)
        for (int c = 0; c < nchunk; ++c) {
            const QueryStats p = qpart[((int64_t)b * nslot + c) * a.nq + t];
            o.inter += p.inter; o.uni += p.uni;
        }
        qs[b * a.nq + t] = o;
    }
    if (t == (int)blockDim.x - 1) {  // blockDim.x = nq + 1 rounded up to whole waves: the last lane is always spare
        GtStats o; o.sum_g = 0; o.sum_gx = 0; o.sum_gy = 0;
        #pragma unroll 8  // the loads do not depend on the running sum: keep eight in flight (add order unchanged)
        for (int c = 0; c < nchunk; ++c) {
            const GtStats p = gpart[(int64_t)b * nslot + c];
            o.sum_g += p.sum_g; o.sum_gx += p.sum_gx; o.sum_gy += p.sum_gy;
        }
        gs[b] = o;
    }
    // the two selected queries of the image, once, for every later kernel (each of their ~10^4 workgroups used to
    // repeat this 20-load chain on one thread before starting)
    __syncthreads();
    if (t < 2) sel[b * 2 + t] = select_query(a, qs, b, t, true);
}

// K2b: adaptive threshold 2 * mean(p) of the selected masks (f_measure.py:76): fixed-order fp64 sum of K2's partials
__global__ __launch_bounds__(64) void eval_adapt_kernel(sm_eval_args a, const double* part, float* thr_adapt, int nslot, int band_min) {
    const int which = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x != 0) return;
    const sm_eval_image im = a.images[b];
    const int nchunk = slots_of(im, a.mh, band_min, nslot);
    double sp = 0.0;
    const double* p0 = part + (int64_t)(b * 2 + which) * nslot * EV_NACC;
    #pragma unroll 8  // the loads do not depend on the running sum: keep eight in flight (add order unchanged)
    for (int k = 0; k < nchunk; ++k) sp += p0[(int64_t)k * EV_NACC];
    thr_adapt[b * 2 + which] = SM_MUL(2.0f, (float)(sp / (double)(im.H * im.W)));
}

// selection (evaluator.pyc@L216-221): which = 0 arg-max objectness, which = 1 arg-max IoU (first maximum)
__device__ int select_query(const sm_eval_args& a, const QueryStats* qs, int b, int which, bool write_ious) {
    int best = 0;
    if (which == 0) {
        const float* o = a.objectness + (int64_t)b * a.obj_stride_b;
        for (int q = 1; q < a.nq; ++q) if (o[q] > o[best]) best = q;
    } else {
        float bi = -1.f;
        for (int q = 0; q < a.nq; ++q) {
            const QueryStats s = qs[b * a.nq + q];
            const float iou = SM_DIV((float)s.inter, SM_ADD((float)s.uni, 1e-7f));
            if (write_ious && a.ious) a.ious[b * a.nq + q] = iou;
            if (iou > bi) { bi = iou; best = q; }
        }
    }
    return best;
}

// fp32 ratios exactly as torch evaluates them (int64 counts -> float32, python scalars -> float32, no fma)
__device__ __forceinline__ float f_measure_from_counts(unsigned tp, unsigned np, unsigned ng) {
    const float prec = SM_DIV((float)tp, SM_ADD((float)np, 1e-7f));
    const float rec = SM_DIV((float)tp, SM_ADD((float)ng, 1e-7f));
    const float num = SM_MUL(SM_MUL(1.09f, prec), rec);                       // (1 + 0.3**2) * prec * recall
    const float den = SM_ADD(SM_ADD(SM_MUL(0.09f, prec), rec), 1e-7f);     // 0.3**2 * prec + recall + eps
    return SM_DIV(num, den);
}

__device__ float ssim_quadrant(double n, double sp, double sg, double spp, double spg) {
    // metrics/s_measure.py:33-52 with one-pass moments (fp64 sums); empty quadrant -> 0/0 = NaN like torch's mean()
    const float x = (float)(sp / n), y = (float)(sg / n);
    const double dx = (double)x, dy = (double)y;
    const float den = (float)(n - 1.0 + 1e-20);
    const float sx2 = (float)(spp - 2.0 * dx * sp + n * dx * dx) / den;
    const float sy2 = (float)(sg - 2.0 * dy * sg + n * dy * dy) / den;   // g in {0,1}: sum g^2 = sum g
    const float sxy = (float)(spg - dx * sg - dy * sp + n * dx * dy) / den;
    const float alpha = 4.f * x * y * sxy;
    const float beta = (x * x + y * y) * (sx2 + sy2);
    if (alpha != 0.f) return alpha / (beta + 1e-20f);
    if (alpha == 0.f && beta == 0.f) return 1.0f;
    return 0.f;
}

__device__ float object_score(double n, double s, double ss) {
    // metrics/s_measure.py:54-60: x = mean, sigma = UNBIASED std, 2x / (x^2 + 1 + sigma + 1e-20)
    const float x = (float)(s / n);
    const double dx = (double)x;
    const float var = (float)((ss - 2.0 * dx * s + n * dx * dx) / (n - 1.0));
    const float sigma = sqrtf(var < 0.f ? 0.f : var);
    return 2.0f * x / (x * x + 1.0f + sigma + 1e-20f);
}

// K2: sum of the selected mask's probabilities per chunk (needed for the adaptive threshold 2*mean before K3)
__global__ __launch_bounds__(EV_THREADS) void eval_sum_kernel(sm_eval_args a, const int* __restrict__ sel, const int* __restrict__ ytab_all,
                                                              double* part, int nslot, int band_min) {
    __shared__ double red[EV_THREADS / 64];
    const int which = blockIdx.y, b = blockIdx.z, c = blockIdx.x;
    const sm_eval_image im = a.images[b];
    const int slots = slots_of(im, a.mh, band_min, nslot);
    if (c >= slots) return;
    const int npx = im.H * im.W, base = c * EV_CHUNK;
    double* slot = part + ((int64_t)(b * 2 + which) * nslot + c) * EV_NACC;
    const int sel_q = sel[b * 2 + which];
    const float* __restrict__ m = a.mask_pred + (int64_t)b * a.mask_stride_b + (int64_t)sel_q * a.mh * a.mw;
    const float sy = a.scale > 0.f ? 1.0f / a.scale : (float)a.mh / (float)im.H;
    const float sx = a.scale > 0.f ? 1.0f / a.scale : (float)a.mw / (float)im.W;
    if (band_walk(im, a.mh, band_min)) {  // both selected queries in one walk (blockIdx.y = 1 has nothing to do)
        if (which != 0) return;
        __shared__ double red2[EV_NW][2];
        const float* __restrict__ m1 = a.mask_pred + (int64_t)b * a.mask_stride_b + (int64_t)sel[b * 2 + 1] * a.mh * a.mw;
        const int* __restrict__ ytab = ytab_all + b * (a.mh + 1);
        const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nseg = (im.W + 63) >> 6, units = a.mh * nseg;
        double s0 = 0.0, s1 = 0.0;
        for (int u = c * EV_NW + wv; u < units; u += slots * EV_NW) {
            const int i0 = u / nseg, seg = u - i0 * nseg, i1 = i0 + (i0 < a.mh - 1 ? 1 : 0);
            const int ya = ytab[i0], yb = ytab[i0 + 1];
            const BandCol k = band_col(seg, lane, im.W, sx, a.mw);
            if (ya >= yb || !k.in) continue;
            const float top0 = __builtin_fmaf(m[i0 * a.mw + k.j0], k.lx0, m[i0 * a.mw + k.j1] * k.lx1);
            const float bot0 = __builtin_fmaf(m[i1 * a.mw + k.j0], k.lx0, m[i1 * a.mw + k.j1] * k.lx1);
            const float top1 = __builtin_fmaf(m1[i0 * a.mw + k.j0], k.lx0, m1[i0 * a.mw + k.j1] * k.lx1);
            const float bot1 = __builtin_fmaf(m1[i1 * a.mw + k.j0], k.lx0, m1[i1 * a.mw + k.j1] * k.lx1);
            for (int y = ya; y < yb; ++y) {
                const UpIdx uy = up_index(y, sy, a.mh);
                s0 += (double)__builtin_fmaf(top0, uy.l0, bot0 * uy.l1);
                s1 += (double)__builtin_fmaf(top1, uy.l0, bot1 * uy.l1);
            }
        }
        s0 = wave_total(s0); s1 = wave_total(s1);
        if (lane == 0) { red2[wv][0] = s0; red2[wv][1] = s1; }
        __syncthreads();
        if (threadIdx.x < 2)
            (slot + (int64_t)threadIdx.x * nslot * EV_NACC)[0] = (red2[0][threadIdx.x] + red2[1][threadIdx.x]) + (red2[2][threadIdx.x] + red2[3][threadIdx.x]);
        return;
    }
    const float inv_w = 1.0f / (float)im.W;
    double sp = 0.0;
    for (int k = 0; k < EV_PPT; ++k) {
        const int idx = base + k * EV_THREADS + threadIdx.x;
        if (idx >= npx) break;
        int y, x;
        split_idx(idx, im.W, inv_w, y, x);
        sp += (double)up_sample(m, a.mw, up_index(y, sy, a.mh), up_index(x, sx, a.mw));
    }
    sp = wave_total(sp);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sp;
    __syncthreads();
    if (threadIdx.x == 0) slot[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct MetricCounts { unsigned tp5, np5, ng, eq5, tpa, npa; unsigned hist[2][256]; };

// number of F-max thresholds strictly below p: start from floor(p * 255) and correct (the table is ascending, ~k/255)
__device__ __forceinline__ int thresholds_below(const float* thr, float p) {
    int lo = (int)(p * 255.0f);
    lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo);
    while (lo > 0 && !(thr[lo - 1] < p)) --lo;
    while (lo < 255 && thr[lo] < p) ++lo;
    return lo;
}

// The same count without data-dependent loops (each costs an LDS round trip and diverges): p in [0, 1] puts the count within
// floor(p * 255) +- 2 (one unit for the rounding of the product, one for the table's own roundings), so four table entries
// around it decide - the table is ascending: everything before the window is below p, everything after is not.
constexpr int EV_THRP = 264;
__device__ __forceinline__ int thresholds_below_window(const float* thrp, float p) {
    int lo = (int)(p * 255.0f);
    lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo);
    const int c = lo - 2 + (thrp[lo] < p ? 1 : 0) + (thrp[lo + 1] < p ? 1 : 0) + (thrp[lo + 2] < p ? 1 : 0) + (thrp[lo + 3] < p ? 1 : 0);
    return c < 0 ? 0 : c;  // NaN: every comparison false
}

// K3, band walk: the selected query's top / bot per lane once per unit, p = fma(top, ly0, bot * ly1) per pixel; every 0/1 count
// a ballot; the S-measure moments of a unit's rows above / below the centroid row accumulate per lane (the row side is
// wave-uniform: two loops) and are folded into the four quadrants by the lane's column side when the unit ends.
__device__ __forceinline__ void metrics_bands(const sm_eval_args& a, const sm_eval_image& im, const float* __restrict__ m,
                                              const unsigned char* __restrict__ gt, const int* __restrict__ ytab, float sy, float sx,
                                              float thr_adapt, int X, int Y, int slots, unsigned (*hist)[256], const float* thr,
                                              double (*red)[EV_NACC], double* slot, MetricCounts* mc) {
    constexpr int NW = EV_NW;
    __shared__ unsigned wcnt[NW][6], gq[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ float thrp[EV_THRP];  // thrp[i] = thr[i - 2], -inf before, +inf after: thresholds_below_window()
    for (int i = tid; i < EV_THRP; i += EV_THREADS) thrp[i] = i < 2 ? -INFINITY : (i - 2 < 255 ? thr[i - 2] : INFINITY);
    if (tid < 4) gq[tid] = 0;
    __syncthreads();
    const int nseg = (im.W + 63) >> 6, units = a.mh * nseg;
    unsigned tp5 = 0, np5 = 0, ng = 0, eq5 = 0, tpa = 0, npa = 0, g0 = 0, g1 = 0, g2 = 0, g3 = 0;  // wave-uniform (scalar registers)
    double aabs = 0.0, fpp = 0.0, bo = 0.0, boo = 0.0;
    double qp[4] = {0.0, 0.0, 0.0, 0.0}, qpp[4] = {0.0, 0.0, 0.0, 0.0}, qpg[4] = {0.0, 0.0, 0.0, 0.0};
    for (int u = blockIdx.x * NW + wv; u < units; u += slots * NW) {
        const int i0 = u / nseg, seg = u - i0 * nseg, i1 = i0 + (i0 < a.mh - 1 ? 1 : 0);
        const int ya = ytab[i0], yb = ytab[i0 + 1];
        if (ya >= yb) continue;
        const BandCol c = band_col(seg, lane, im.W, sx, a.mw);
        float top = 0.f, bot = 0.f;
        if (c.in) {
            top = __builtin_fmaf(m[i0 * a.mw + c.j0], c.lx0, m[i0 * a.mw + c.j1] * c.lx1);
            bot = __builtin_fmaf(m[i1 * a.mw + c.j0], c.lx0, m[i1 * a.mw + c.j1] * c.lx1);
        }
        const bool right = c.x >= X;
        const unsigned long long vm = __ballot(c.in), rm = __ballot(c.in && right);
        // NR rows of the unit from y0, all on one side of the centroid row: sums into (sp, spp, spg), GT counts left / right.
        // The rows of a pass are independent chains (GT byte, table window, histogram add): their latencies overlap.
        auto rows = [&](auto nr_tag, int y0, unsigned gb, double& sp, double& spp, double& spg, unsigned& gl, unsigned& gr) {
            constexpr int NR = decltype(nr_tag)::value;
            float p[NR];
            bool g[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const UpIdx uy = up_index(y0 + r, sy, a.mh);
                p[r] = __builtin_fmaf(top, uy.l0, bot * uy.l1);  // columns past the width: top = bot = 0 -> p = 0
                g[r] = ((gb >> r) & 1u) != 0;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const unsigned long long gm = __ballot(g[r]);
                const unsigned long long b5 = __ballot(c.in && p[r] > 0.5f), ba = __ballot(c.in && p[r] > thr_adapt);
                tp5 += (unsigned)__popcll(b5 & gm); np5 += (unsigned)__popcll(b5); ng += (unsigned)__popcll(gm);
                eq5 += (unsigned)__popcll(~(b5 ^ gm) & vm); tpa += (unsigned)__popcll(ba & gm); npa += (unsigned)__popcll(ba);
                gr += (unsigned)__popcll(gm & rm); gl += (unsigned)__popcll(gm & ~rm);
                if (c.in) atomicAdd(&hist[g[r] ? 1 : 0][thresholds_below_window(thrp, p[r])], 1u);
                // absent columns carry p = 0, g = 0: they add 0 to every sum below except the background's (masked)
                const float gf = g[r] ? 1.0f : 0.0f;
                const double pd = (double)p[r], pp = pd * pd;
                aabs += (double)fabsf(p[r] - gf);
                sp += pd; spp += pp; spg += g[r] ? pd : 0.0;
                fpp += g[r] ? pp : 0.0;
                const double o = (double)(1.0f - p[r]);
                const bool bg = c.in && !g[r];
                bo += bg ? o : 0.0; boo += bg ? o * o : 0.0;
            }
        };
        auto walk = [&](int y_from, int y_to, double& sp, double& spp, double& spg, unsigned& gl, unsigned& gr) {
            for (int yg = y_from; yg < y_to; yg += EV_GROUP) {  // the GT bytes of EV_GROUP rows are requested together (query_bands)
                const unsigned gbits = band_gt_bits(gt, im.W, c, yg, y_to);
                const int ye = yg + EV_GROUP < y_to ? yg + EV_GROUP : y_to;
                int y = yg;
                for (; y + EV_NR <= ye; y += EV_NR) rows(IntTag<EV_NR>{}, y, gbits >> (y - yg), sp, spp, spg, gl, gr);
                for (; y < ye; ++y) rows(IntTag<1>{}, y, gbits >> (y - yg), sp, spp, spg, gl, gr);
            }
        };
        double tp = 0.0, tpp = 0.0, tpg = 0.0, bp = 0.0, bpp = 0.0, bpg = 0.0;  // top rows (y < Y) / bottom rows (y >= Y)
        const int ysplit = Y < ya ? ya : (Y > yb ? yb : Y);
        walk(ya, ysplit, tp, tpp, tpg, g0, g1);
        walk(ysplit, yb, bp, bpp, bpg, g2, g3);
        qp[0] += right ? 0.0 : tp; qpp[0] += right ? 0.0 : tpp; qpg[0] += right ? 0.0 : tpg;
        qp[1] += right ? tp : 0.0; qpp[1] += right ? tpp : 0.0; qpg[1] += right ? tpg : 0.0;
        qp[2] += right ? 0.0 : bp; qpp[2] += right ? 0.0 : bpp; qpg[2] += right ? 0.0 : bpg;
        qp[3] += right ? bp : 0.0; qpp[3] += right ? bpp : 0.0; qpg[3] += right ? bpg : 0.0;
    }
    // acc layout of the raster walk (EV_NACC): 1 = sum|p-g|; 2+4k.. = quadrant k {sum p, sum g, sum p^2, sum p g}; 18, 19 = fg
    // {sum p, sum p^2}; 20, 21 = bg {sum (1-p), sum (1-p)^2}.  sum g is an integer count (gq), fg sum p the total of sum p g.
    {
        double v;
        v = wave_total(aabs); if (lane == 0) red[wv][1] = v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v = wave_total(qp[k]); if (lane == 0) red[wv][2 + 4 * k] = v;
            v = wave_total(qpp[k]); if (lane == 0) red[wv][4 + 4 * k] = v;
            v = wave_total(qpg[k]); if (lane == 0) red[wv][5 + 4 * k] = v;
        }
        v = wave_total(fpp); if (lane == 0) red[wv][19] = v;
        v = wave_total(bo); if (lane == 0) red[wv][20] = v;
        v = wave_total(boo); if (lane == 0) red[wv][21] = v;
    }
    if (lane == 0) {
        wcnt[wv][0] = tp5; wcnt[wv][1] = np5; wcnt[wv][2] = ng; wcnt[wv][3] = eq5; wcnt[wv][4] = tpa; wcnt[wv][5] = npa;
        atomicAdd(&gq[0], g0); atomicAdd(&gq[1], g1); atomicAdd(&gq[2], g2); atomicAdd(&gq[3], g3);
    }
    __syncthreads();
    if (tid > 0 && tid < EV_NACC) {
        const int k = tid;
        double v;
        if (k >= 2 && k < 18 && ((k - 2) & 3) == 1) v = (double)gq[(k - 2) >> 2];
        else if (k == 18) v = 0.0;  // filled below from the four sums of p g
        else v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (k != 18) slot[k] = v;
    }
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < 4; ++k) t += (red[0][5 + 4 * k] + red[1][5 + 4 * k]) + (red[2][5 + 4 * k] + red[3][5 + 4 * k]);
        slot[18] = t;
    }
    if (tid < 6) (&mc->tp5)[tid] = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
    mc->hist[0][tid] = hist[0][tid];
    mc->hist[1][tid] = hist[1][tid];
}

// K3: everything else in one pass per chunk.  Integer counts and the 2x256-bin histogram use integer atomics;
// the fp64 moments are written as per-chunk partials and summed in a fixed order by K4 (bit-reproducible).
__global__ __launch_bounds__(EV_THREADS) void eval_metrics_kernel(sm_eval_args a, const int* __restrict__ sel, const float* __restrict__ thr_adapt_all,
                                                                  const GtStats* gs, const int* __restrict__ ytab_all, double* part, MetricCounts* cnt,
                                                                  int nslot, int band_min) {
    __shared__ unsigned hist[2][256];
    __shared__ float thr[256];
    __shared__ double red[EV_THREADS / 64][EV_NACC];
    const int which = blockIdx.y, b = blockIdx.z, c = blockIdx.x, tid = threadIdx.x;
    const sm_eval_image im = a.images[b];
    const int slots = slots_of(im, a.mh, band_min, nslot);
    if (c >= slots) return;
    const int npx = im.H * im.W, base = c * EV_CHUNK;
    double* slot = part + ((int64_t)(b * 2 + which) * nslot + c) * EV_NACC;
    MetricCounts* mc = cnt + ((int64_t)(b * 2 + which) * nslot + c);
    const int sel_q = sel[b * 2 + which];
    hist[0][tid] = 0; hist[1][tid] = 0;
    thr[tid] = tid < 255 ? a.thresholds[tid] : INFINITY;
    __syncthreads();
    const float* __restrict__ m = a.mask_pred + (int64_t)b * a.mask_stride_b + (int64_t)sel_q * a.mh * a.mw;
    const unsigned char* __restrict__ gt = a.gt + im.gt_off;
    const float sy = a.scale > 0.f ? 1.0f / a.scale : (float)a.mh / (float)im.H;
    const float sx = a.scale > 0.f ? 1.0f / a.scale : (float)a.mw / (float)im.W;
    const float inv_w = 1.0f / (float)im.W;
    const float thr_adapt = thr_adapt_all[b * 2 + which];
    const GtStats g0 = gs[b];
    // centroid (s_measure.py:13-31): round-half-even of the fp32 quotient of exact integer sums
    const int X = (int)rintf(SM_DIV((float)g0.sum_gx, (float)g0.sum_g)), Y = (int)rintf(SM_DIV((float)g0.sum_gy, (float)g0.sum_g));
    if (band_walk(im, a.mh, band_min)) {
        metrics_bands(a, im, m, gt, ytab_all + b * (a.mh + 1), sy, sx, thr_adapt, X, Y, slots, hist, thr, red, slot, mc);
        return;
    }

    unsigned tp5 = 0, np5 = 0, ng = 0, eq5 = 0, tpa = 0, npa = 0;
    double acc[EV_NACC];
#pragma unroll
    for (int k = 0; k < EV_NACC; ++k) acc[k] = 0.0;
    // acc: 1 = sum|p-g|; 2+4k.. = quadrant k {sum p, sum g, sum p^2, sum p g}; 18,19 = fg {sum p, sum p^2};
    //      20,21 = bg {sum (1-p), sum (1-p)^2}
    for (int k = 0; k < EV_PPT; ++k) {
        const int idx = base + k * EV_THREADS + tid;
        const bool valid = idx < npx;
        if (__ballot(valid) == 0) break;  // wave-uniform exit: the ballots below need every lane here
        int y = 0, x = 0;
        float p = 0.f;
        unsigned g = 0;
        if (valid) {
            split_idx(idx, im.W, inv_w, y, x);
            p = up_sample(m, a.mw, up_index(y, sy, a.mh), up_index(x, sx, a.mw));
            g = gt[idx] != 0;
        }
        const unsigned b5 = valid && p > 0.5f, ba = valid && p > thr_adapt;
        g = valid ? g : 0;
        tp5 += b5 & g; np5 += b5; ng += g; eq5 += (valid && b5 == g); tpa += ba & g; npa += ba;
        const float gf = (float)g;
        if (valid) acc[1] += (double)fabsf(p - gf);
        // number of thresholds strictly below p: start from floor(p*255) and correct (the table is ascending, ~k/255)
        int lo = (int)(p * 255.0f);
        lo = lo < 0 ? 0 : (lo > 255 ? 255 : lo);
        while (lo > 0 && !(thr[lo - 1] < p)) --lo;
        while (lo < 255 && thr[lo] < p) ++lo;
        // LDS histogram split by GT.  (Tried and rejected, measured on the B=64 bench: ballot-counting the saturated bins
        // 0/255 is neutral, peeling one distinct key per round with ballots is 1.5x SLOWER - bilinear ramps put ~50
        // distinct bins into a 64-pixel wave.)
        if (valid) atomicAdd(&hist[g][lo], 1u);
        if (!valid) continue;
        const double pd = (double)p, gd = (double)gf;
        const int quad = (y >= Y ? 2 : 0) + (x >= X ? 1 : 0);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            if (quad == qd) { acc[2 + 4 * qd] += pd; acc[3 + 4 * qd] += gd; acc[4 + 4 * qd] += pd * pd; acc[5 + 4 * qd] += pd * gd; }
        }
        if (g) { acc[18] += pd; acc[19] += pd * pd; } else { const double o = (double)(1.0f - p); acc[20] += o; acc[21] += o * o; }
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int k = 1; k < EV_NACC; ++k) {
        const double v = wave_total(acc[k]);
        if (lane == 0) red[wv][k] = v;
    }
    tp5 = wave_total(tp5); np5 = wave_total(np5); ng = wave_total(ng); eq5 = wave_total(eq5);
    tpa = wave_total(tpa); npa = wave_total(npa);
    __shared__ unsigned redc[EV_THREADS / 64][6];
    if (lane == 0) { redc[wv][0] = tp5; redc[wv][1] = np5; redc[wv][2] = ng; redc[wv][3] = eq5; redc[wv][4] = tpa; redc[wv][5] = npa; }
    __syncthreads();
    // per-chunk partial counts + histogram, plain stores (K4 sums them): no same-line atomic storms
    if (tid > 0 && tid < EV_NACC) slot[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid < 6) (&mc->tp5)[tid] = redc[0][tid] + redc[1][tid] + redc[2][tid] + redc[3][tid];
    mc->hist[0][tid] = hist[0][tid];
    mc->hist[1][tid] = hist[1][tid];
}

// K4: finalise the 7 metrics of (image, which) with the reference's fp32 operation order; 256 threads: fixed-order
// partial sums, suffix sums of the histogram and the 255 F-measures in parallel, the scalar tail on thread 0
__global__ __launch_bounds__(256) void eval_finalize_kernel(sm_eval_args a, const int* __restrict__ sel, const GtStats* gs,
                                                            const double* part, const MetricCounts* cnt, int nslot, int band_min) {
    __shared__ double accs[EV_NACC];
    __shared__ unsigned s_tp[257], s_np[257];
    __shared__ float s_f[256];
    const int which = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const sm_eval_image im = a.images[b];
    const int npx = im.H * im.W;
    const int nchunk = slots_of(im, a.mh, band_min, nslot);  // the slots this image's walk wrote
    const MetricCounts* mcp = cnt + (int64_t)(b * 2 + which) * nslot;
    __shared__ unsigned h0[256], h1[256], cts[6];
    {
        unsigned s0 = 0, s1 = 0;
        #pragma unroll 8  // the loads do not depend on the running sum: keep eight in flight (add order unchanged)
        for (int c = 0; c < nchunk; ++c) { s0 += mcp[c].hist[0][tid]; s1 += mcp[c].hist[1][tid]; }
        h0[tid] = s0; h1[tid] = s1;
        if (tid < 6) {
            unsigned v = 0;
            #pragma unroll 8  // the loads do not depend on the running sum: keep eight in flight (add order unchanged)
            for (int c = 0; c < nchunk; ++c) v += (&mcp[c].tp5)[tid];
            cts[tid] = v;
        }
    }
    __syncthreads();
    if (tid < EV_NACC) {
        const double* p0 = part + (int64_t)(b * 2 + which) * nslot * EV_NACC + tid;
        double sacc = 0.0;
        #pragma unroll 8  // the loads do not depend on the running sum: keep eight in flight (add order unchanged)
        for (int c = 0; c < nchunk; ++c) sacc += p0[(int64_t)c * EV_NACC];  // chunk order: deterministic
        accs[tid] = sacc;
    }
    // suffix sums: s_tp[k] = sum_{b > k} hist1[b], s_np[k] = sum_{b > k} (hist0 + hist1)[b]   (k = tid)
    s_tp[tid] = tid < 255 ? h1[tid + 1] : 0u;
    s_np[tid] = tid < 255 ? h0[tid + 1] + h1[tid + 1] : 0u;
    if (tid == 0) { s_tp[256] = 0; s_np[256] = 0; }
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned t1 = tid + o < 256 ? s_tp[tid + o] : 0u, t2 = tid + o < 256 ? s_np[tid + o] : 0u;
        __syncthreads();
        s_tp[tid] += t1; s_np[tid] += t2;
        __syncthreads();
    }
    const unsigned ng = cts[2];
    s_f[tid] = tid < 255 ? f_measure_from_counts(s_tp[tid], s_np[tid], ng) : -INFINITY;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_f[tid] = fmaxf(s_f[tid], s_f[tid + o]);
        __syncthreads();
    }
    if (tid != 0) return;
    double acc[EV_NACC];
    for (int k = 0; k < EV_NACC; ++k) acc[k] = accs[k];
    const unsigned tp5 = cts[0], np5 = cts[1], eq5 = cts[3], tpa = cts[4], npa = cts[5];
    const GtStats g0 = gs[b];
    const int X = (int)rintf(SM_DIV((float)g0.sum_gx, (float)g0.sum_g)), Y = (int)rintf(SM_DIV((float)g0.sum_gy, (float)g0.sum_g));
    const int q = sel[b * 2 + which];

    float* row = a.rows + (int64_t)b * 16 + which * 7;
    const float N = (float)npx;
    row[0] = SM_DIV((float)tp5, SM_ADD((float)(np5 + ng - tp5), 1e-7f));   // iou.py:28-31
    row[1] = SM_DIV((float)eq5, N);                                             // pixel_acc.py:14
    row[2] = f_measure_from_counts(tp5, np5, ng);                                  // f_measure.py:44-50
    row[3] = s_f[0];                                                               // f_measure.py:52-69
    row[4] = f_measure_from_counts(tpa, npa, ng);                                  // f_measure.py:71-81
    row[5] = (float)(acc[1] / (double)npx);                                        // mae.py:9
    {   // S-measure (s_measure.py:105-124)
        const double n1 = (double)ng, n0 = (double)npx - (double)ng;
        const double sum_p = acc[2] + acc[6] + acc[10] + acc[14];
        float Q;
        if (ng == 0) {
            Q = 1.0f - (float)(sum_p / (double)npx);
        } else if ((int)ng == npx) {
            Q = (float)(sum_p / (double)npx);
        } else {
            const float u = SM_DIV((float)ng, N);
            const float s_obj = u * object_score(n1, acc[18], acc[19]) + (1.0f - u) * object_score(n0, acc[20], acc[21]);
            const float area = (float)npx, Xf = (float)X, Yf = (float)Y;
            const float w1 = Xf * Yf / area, w2 = ((float)im.W - Xf) * Yf / area, w3 = Xf * ((float)im.H - Yf) / area;
            const float w4 = 1.0f - w1 - w2 - w3;
            const double nLT = (double)X * Y, nRT = (double)(im.W - X) * Y, nLB = (double)X * (im.H - Y),
                         nRB = (double)(im.W - X) * (im.H - Y);
            const float s_reg = w1 * ssim_quadrant(nLT, acc[2], acc[3], acc[4], acc[5]) +
                                w2 * ssim_quadrant(nRT, acc[6], acc[7], acc[8], acc[9]) +
                                w3 * ssim_quadrant(nLB, acc[10], acc[11], acc[12], acc[13]) +
                                w4 * ssim_quadrant(nRB, acc[14], acc[15], acc[16], acc[17]);
            Q = 0.5f * s_obj + 0.5f * s_reg;
            if (Q < 0.f) Q = 0.f;
        }
        row[6] = Q;
    }
    a.rows[(int64_t)b * 16 + 14 + which] = (float)q;
}

// ---- refinement glue (BASELINE.json configs[2]): the picked query's mask up-sampled to the solver's resolution as the fp64
// target bilateral_solver_output expects (bilateral_solver.py:181: target cast to np.double), and the solver's binary
// result back as an fp32 one-query "mask_pred" for the metric kernels ----------------------------------------------------
__global__ __launch_bounds__(256) void eval_upsample_selected_kernel(const float* __restrict__ masks, int64_t stride_b,
                                                                    const float* __restrict__ rows, int sel_col,
                                                                    double* __restrict__ out, int mh, int mw, int OH, int OW) {
    const int b = blockIdx.y;
    const int q = (int)rows[(int64_t)b * 16 + sel_col];  // query index written by eval_finalize_kernel (column 14 / 15)
    const float* m = masks + (int64_t)b * stride_b + (int64_t)q * mh * mw;
    const float sy = (float)mh / (float)OH, sx = (float)mw / (float)OW;  // F.interpolate(size=(OH, OW)): in / out
    for (int p = blockIdx.x * 256 + threadIdx.x; p < OH * OW; p += gridDim.x * 256) {
        const int y = p / OW, x = p - y * OW;
        out[(int64_t)b * OH * OW + p] = (double)up_sample(m, mw, up_index(y, sy, mh), up_index(x, sx, mw));
    }
}

__global__ __launch_bounds__(256) void eval_u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (float)src[i];
}

struct EvalWs { QueryStats *qs, *qpart; GtStats *gs, *gpart; MetricCounts* cnt; double* part; float* maskT; int *sel, *ytab; float* thr_adapt; size_t total; };

static EvalWs carve_eval(int B, int nq, int nchunk, int mh, int plane, char* base) {
    EvalWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    w.qs = (QueryStats*)take((size_t)B * nq * sizeof(QueryStats));
    w.gs = (GtStats*)take((size_t)B * sizeof(GtStats));
    w.qpart = (QueryStats*)take((size_t)B * nchunk * nq * sizeof(QueryStats));
    w.gpart = (GtStats*)take((size_t)B * nchunk * sizeof(GtStats));
    w.cnt = (MetricCounts*)take((size_t)B * 2 * nchunk * sizeof(MetricCounts));
    w.part = (double*)take((size_t)B * 2 * nchunk * EV_NACC * sizeof(double));
    w.maskT = (float*)take((size_t)((nq + EV_MAXQ - 1) / EV_MAXQ) * B * plane * EV_QS * sizeof(float));
    w.sel = (int*)take((size_t)B * 2 * sizeof(int));
    w.thr_adapt = (float*)take((size_t)B * 2 * sizeof(float));
    w.ytab = (int*)take((size_t)B * (mh + 1) * sizeof(int));
    w.total = off;
    return w;
}

}  // namespace sm

namespace sm {
// partial-sum slots per image: raster chunks of the largest image or low-res rows, whichever walk needs more
static int eval_slots(int max_pixels, int mh) {
    const int nchunk = (max_pixels + EV_CHUNK - 1) / EV_CHUNK;
    return nchunk > mh ? nchunk : mh;
}
}  // namespace sm

static const int SM_EVAL_MAX_PIXELS = 1 << 22;
static const int SM_EVAL_MAX_QUERIES = 960;  // one thread per query in the reduce kernel (a workgroup is <= 1024 threads)

extern "C" size_t sm_evaluate_workspace_bytes(int32_t B, int32_t nq, int32_t mh, int32_t mw, int32_t max_pixels) {
    if (B <= 0 || nq <= 0 || nq > SM_EVAL_MAX_QUERIES || mh <= 0 || mw <= 0 || max_pixels <= 0 || max_pixels > SM_EVAL_MAX_PIXELS) return 0;
    return sm::carve_eval(B, nq, sm::eval_slots(max_pixels, mh), mh, mh * mw, nullptr).total;
}

extern "C" int sm_upsample_selected_f64(const float* masks, int64_t mask_stride_b, const float* rows, int32_t sel_col, double* out,
                                        int32_t B, int32_t mh, int32_t mw, int32_t OH, int32_t OW, void* stream) {
    SM_REQUIRE(masks && rows && out && B > 0 && mh > 0 && mw > 0 && OH > 0 && OW > 0 && (sel_col == 14 || sel_col == 15),
               "sm_upsample_selected_f64: bad arguments (sel_col 14 = picked query, 15 = upper bound)");
    const int gx = (OH * OW + 255) / 256 < 256 ? (OH * OW + 255) / 256 : 256;
    hipLaunchKernelGGL(sm::eval_upsample_selected_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, masks, mask_stride_b, rows,
                       sel_col, out, mh, mw, OH, OW);
    return sm::check_launch("sm_upsample_selected_f64");
}

extern "C" int sm_mask_u8_to_f32(const uint8_t* src, float* dst, int64_t n, void* stream) {
    SM_REQUIRE(src && dst && n > 0, "sm_mask_u8_to_f32: bad arguments");
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(sm::eval_u8_to_f32_kernel, dim3((int)(g < 2048 ? g : 2048)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return sm::check_launch("sm_mask_u8_to_f32");
}

extern "C" int sm_evaluate_masks_f32(const sm_eval_args* a, void* stream) {
    SM_REQUIRE(a && a->mask_pred && a->objectness && a->gt && a->images && a->thresholds && a->rows && a->workspace,
               "sm_evaluate_masks_f32: null pointer");
    SM_REQUIRE(a->B > 0 && a->nq > 0 && a->nq <= SM_EVAL_MAX_QUERIES && a->mh > 0 && a->mw > 0 && a->scale >= 0.f,
               "sm_evaluate_masks_f32: bad shape (B=%d nq=%d (<= %d) mask %dx%d)", a->B, a->nq, SM_EVAL_MAX_QUERIES, a->mh, a->mw);
    SM_REQUIRE(a->max_pixels > 0 && a->max_pixels <= SM_EVAL_MAX_PIXELS,
               "sm_evaluate_masks_f32: max_pixels=%d (largest H*W of the batch, <= %d)", a->max_pixels, SM_EVAL_MAX_PIXELS);
    SM_REQUIRE(a->workspace_bytes >= sm_evaluate_workspace_bytes(a->B, a->nq, a->mh, a->mw, a->max_pixels) &&
                   ((uintptr_t)a->workspace % 256) == 0,
               "sm_evaluate_masks_f32: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    const int nslot = sm::eval_slots(a->max_pixels, a->mh);
    const sm::EvalWs w = sm::carve_eval(a->B, a->nq, nslot, a->mh, a->mh * a->mw, (char*)a->workspace);
    const int npass = (a->nq + sm::EV_MAXQ - 1) / sm::EV_MAXQ;
    // tap staging of the raster walk: up to EV_LDS_ROWS low-res rows of [mw][32 queries] floats inside a 60-KiB budget; masks too
    // wide for even one row (mw > 480) read their taps from global memory (the kernel's fallback path)
    const size_t row_bytes = (size_t)a->mw * sm::EV_QS * sizeof(float);
    const int lds_rows = (int)(sm::EV_LDS_BYTES / row_bytes) < sm::EV_LDS_ROWS ? (int)(sm::EV_LDS_BYTES / row_bytes) : sm::EV_LDS_ROWS;
    int band_min = sm::EV_BAND_MIN | sm::EV_UPW << 8;  // the kernels' `walk` argument
#ifdef SM_TUNING
    {
        int bm = sm::EV_BAND_MIN, upw = sm::EV_UPW;
        if (const char* e = getenv("SM_EVAL_BAND_MIN")) bm = atoi(e) & 255;  // 0: raster walk for every image
        if (const char* e = getenv("SM_EVAL_UPW")) upw = atoi(e) < 1 ? 1 : (atoi(e) > sm::EV_UPW ? sm::EV_UPW : atoi(e));
        band_min = bm | upw << 8;
    }
#endif
    const int red_threads = ((a->nq + 1 + 63) / 64) * 64;  // one thread per query + a spare last lane for the GT sums
    hipLaunchKernelGGL(sm::eval_transpose_kernel, dim3((a->mh * a->mw * sm::EV_QS + 255) / 256, a->B, npass), dim3(256), 0, st, *a, w.maskT, w.ytab);
    if (a->nq <= 20)  // the band walk's registers sized for the shipped 20 queries (4 waves per SIMD instead of 3)
        hipLaunchKernelGGL(sm::eval_query_kernel<5>, dim3(nslot, a->B, npass), dim3(sm::EV_THREADS), lds_rows * row_bytes, st, *a,
                           w.maskT, w.ytab, w.qpart, w.gpart, nslot, lds_rows, band_min);
    else
        hipLaunchKernelGGL(sm::eval_query_kernel<8>, dim3(nslot, a->B, npass), dim3(sm::EV_THREADS), lds_rows * row_bytes, st, *a,
                           w.maskT, w.ytab, w.qpart, w.gpart, nslot, lds_rows, band_min);
    hipLaunchKernelGGL(sm::eval_reduce_query_kernel, dim3(a->B), dim3(red_threads), 0, st, *a, w.qpart, w.gpart, w.qs, w.gs, w.sel, nslot, band_min);
    hipLaunchKernelGGL(sm::eval_sum_kernel, dim3(nslot, 2, a->B), dim3(sm::EV_THREADS), 0, st, *a, w.sel, w.ytab, w.part, nslot, band_min);
    hipLaunchKernelGGL(sm::eval_adapt_kernel, dim3(2, a->B), dim3(64), 0, st, *a, w.part, w.thr_adapt, nslot, band_min);
    hipLaunchKernelGGL(sm::eval_metrics_kernel, dim3(nslot, 2, a->B), dim3(sm::EV_THREADS), 0, st, *a, w.sel, w.thr_adapt, w.gs,
                       w.ytab, w.part, w.cnt, nslot, band_min);
    hipLaunchKernelGGL(sm::eval_finalize_kernel, dim3(2, a->B), dim3(256), 0, st, *a, w.sel, w.gs, w.part, w.cnt, nslot, band_min);
    return sm::check_launch("sm_evaluate_masks_f32");
}
