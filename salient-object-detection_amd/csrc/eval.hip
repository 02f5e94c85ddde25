// Evaluator post-processing + the seven saliency metrics, fused on the device.
//
// Reference: evaluator.pyc@L199-228 (last layer -> bilinear up-sample -> crop -> upper-bound query by IoU ->
// arg-max objectness -> _update_meters) and metrics/{iou,f_measure,mae,pixel_acc,s_measure}.py.  The reference does
// ~14 device->host syncs per image and replicates the mask 255x for F-max (metrics/f_measure.py:61-62); here two
// launches per BATCH leave 16 floats per image:
//   K1  eval_query_kernel   one workgroup per (image, query): up-sample on the fly, |p>0.5 & g|, |p>0.5 | g|, sum p
//                           (+ ground-truth moments once per image);
//   K2  eval_metrics_kernel one workgroup per (image, {objectness pick, upper bound}): selection, then ONE pass that
//                           gathers threshold counts, a 256-bin histogram split by GT (exactly the 255 strict '>'
//                           thresholds k/255 of f_measure.py:65), |p-g|, per-quadrant and per-class moments for the
//                           S-measure, and finalises the 7 values with the reference's fp32 operation order.
// HBM-bound: every pass reads the (nq, h', w') probability maps (L2-resident) and the GT bytes once.
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
__device__ __forceinline__ T block_sum(T v, T* red) {  // 256 threads; red: >= 4 entries of LDS
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

struct QueryStats { unsigned inter, uni; double sum_p; };
struct GtStats { double sum_g, sum_gx, sum_gy; };

__global__ __launch_bounds__(256) void eval_query_kernel(sm_eval_args a, QueryStats* qs, GtStats* gs) {
    __shared__ double redd[4];
    __shared__ unsigned redu[4];
    const int q = blockIdx.x, b = blockIdx.y;
    const sm_eval_image im = a.images[b];
    const unsigned char* __restrict__ gt = a.gt + im.gt_off;
    const float* __restrict__ m = a.mask_pred + (int64_t)b * a.mask_stride_b + (int64_t)q * a.mh * a.mw;
    const float sy = a.scale > 0.f ? 1.0f / a.scale : (float)a.mh / (float)im.H;
    const float sx = a.scale > 0.f ? 1.0f / a.scale : (float)a.mw / (float)im.W;
    unsigned inter = 0, uni = 0;
    double sp = 0.0, sg = 0.0, sgx = 0.0, sgy = 0.0;
    const int npx = im.H * im.W;
    for (int idx = threadIdx.x; idx < npx; idx += 256) {
        const int y = idx / im.W, x = idx - y * im.W;
        const float p = up_sample(m, a.mw, up_index(y, sy, a.mh), up_index(x, sx, a.mw));
        const unsigned g = gt[idx] != 0, bin = p > 0.5f;
        inter += bin & g;
        uni += bin | g;
        sp += (double)p;
        if (q == 0 && g) { sg += 1.0; sgx += (double)x; sgy += (double)y; }
    }
    inter = block_sum<unsigned>(inter, redu);
    uni = block_sum<unsigned>(uni, redu);
    sp = block_sum<double>(sp, redd);
    if (q == 0) {
        sg = block_sum<double>(sg, redd);
        sgx = block_sum<double>(sgx, redd);
        sgy = block_sum<double>(sgy, redd);
    }
    if (threadIdx.x == 0) {
        QueryStats s; s.inter = inter; s.uni = uni; s.sum_p = sp;
        qs[b * a.nq + q] = s;
        if (q == 0) { GtStats t; t.sum_g = sg; t.sum_gx = sgx; t.sum_gy = sgy; gs[b] = t; }
    }
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

__global__ __launch_bounds__(256) void eval_metrics_kernel(sm_eval_args a, const QueryStats* qs, const GtStats* gs) {
    __shared__ unsigned hist[2][256];
    __shared__ float thr[256];
    __shared__ double redd[4];
    __shared__ unsigned redu[4];
    __shared__ int sel_q;
    const int which = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const sm_eval_image im = a.images[b];
    const int npx = im.H * im.W;

    // ---- selection (evaluator.pyc@L216-221): upper bound = arg-max IoU, pick = arg-max objectness (first max) -------
    if (tid == 0) {
        int best = 0;
        if (which == 0) {
            const float* o = a.objectness + (int64_t)b * a.obj_stride_b;
            for (int q = 1; q < a.nq; ++q) if (o[q] > o[best]) best = q;
        } else {
            float bi = -1.f;
            for (int q = 0; q < a.nq; ++q) {
                const QueryStats s = qs[b * a.nq + q];
                const float iou = SM_DIV((float)s.inter, SM_ADD((float)s.uni, 1e-7f));
                if (a.ious) a.ious[b * a.nq + q] = iou;
                if (iou > bi) { bi = iou; best = q; }
            }
        }
        sel_q = best;
    }
    hist[0][tid] = 0; hist[1][tid] = 0;
    thr[tid] = tid < 255 ? a.thresholds[tid] : INFINITY;
    __syncthreads();
    const int q = sel_q;
    const float* __restrict__ m = a.mask_pred + (int64_t)b * a.mask_stride_b + (int64_t)q * a.mh * a.mw;
    const unsigned char* __restrict__ gt = a.gt + im.gt_off;
    const float sy = a.scale > 0.f ? 1.0f / a.scale : (float)a.mh / (float)im.H;
    const float sx = a.scale > 0.f ? 1.0f / a.scale : (float)a.mw / (float)im.W;
    const float mean_p = (float)(qs[b * a.nq + q].sum_p / (double)npx);
    const float thr_adapt = SM_MUL(2.0f, mean_p);  // f_measure.py:76 (2 * mean; the mean itself is fp64-summed here)
    const GtStats g0 = gs[b];
    // centroid (s_measure.py:13-31): round-half-even of the fp32 quotient; gt all-zero takes the early exit below
    const int X = (int)rintf((float)g0.sum_gx / (float)g0.sum_g), Y = (int)rintf((float)g0.sum_gy / (float)g0.sum_g);

    unsigned tp5 = 0, np5 = 0, ng = 0, eq5 = 0, tpa = 0, npa = 0;
    double sabs = 0.0;
    double qp[4] = {0, 0, 0, 0}, qg[4] = {0, 0, 0, 0}, qpp[4] = {0, 0, 0, 0}, qpg[4] = {0, 0, 0, 0};
    double f_s = 0.0, f_ss = 0.0, b_s = 0.0, b_ss = 0.0;
    for (int idx = tid; idx < npx; idx += 256) {
        const int y = idx / im.W, x = idx - y * im.W;
        const float p = up_sample(m, a.mw, up_index(y, sy, a.mh), up_index(x, sx, a.mw));
        const unsigned g = gt[idx] != 0, b5 = p > 0.5f, ba = p > thr_adapt;
        tp5 += b5 & g; np5 += b5; ng += g; eq5 += (b5 == g); tpa += ba & g; npa += ba;
        const float gf = (float)g;
        sabs += (double)fabsf(p - gf);
        // number of thresholds strictly below p (binary search over the 255 ascending values; thr[255] = +inf)
        int lo = 0, hi = 255;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (thr[mid] < p) lo = mid + 1; else hi = mid; }
        atomicAdd(&hist[g][lo], 1u);
        const double pd = (double)p;
        const int quad = (y >= Y ? 2 : 0) + (x >= X ? 1 : 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (quad == k) { qp[k] += pd; qg[k] += (double)gf; qpp[k] += pd * pd; qpg[k] += pd * (double)gf; }
        }
        if (g) { f_s += pd; f_ss += pd * pd; } else { const double o = (double)(1.0f - p); b_s += o; b_ss += o * o; }
    }
    tp5 = block_sum<unsigned>(tp5, redu); np5 = block_sum<unsigned>(np5, redu); ng = block_sum<unsigned>(ng, redu);
    eq5 = block_sum<unsigned>(eq5, redu); tpa = block_sum<unsigned>(tpa, redu); npa = block_sum<unsigned>(npa, redu);
    sabs = block_sum<double>(sabs, redd);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        qp[k] = block_sum<double>(qp[k], redd); qg[k] = block_sum<double>(qg[k], redd);
        qpp[k] = block_sum<double>(qpp[k], redd); qpg[k] = block_sum<double>(qpg[k], redd);
    }
    f_s = block_sum<double>(f_s, redd); f_ss = block_sum<double>(f_ss, redd);
    b_s = block_sum<double>(b_s, redd); b_ss = block_sum<double>(b_ss, redd);
    __syncthreads();
    if (tid != 0) return;

    float* row = a.rows + (int64_t)b * 16 + which * 7;
    const float N = (float)npx;
    row[0] = SM_DIV((float)tp5, SM_ADD((float)(np5 + ng - tp5), 1e-7f));   // iou.py:28-31
    row[1] = SM_DIV((float)eq5, N);                                             // pixel_acc.py:14
    row[2] = f_measure_from_counts(tp5, np5, ng);                                  // f_measure.py:44-50
    {   // F-max over the 255 thresholds: pixels above threshold k are those in bins > k (suffix sums)
        unsigned tp = 0, np = 0;
        float best = -INFINITY;
        for (int k = 254; k >= 0; --k) {
            tp += hist[1][k + 1]; np += hist[0][k + 1] + hist[1][k + 1];
            best = fmaxf(best, f_measure_from_counts(tp, np, ng));
        }
        row[3] = best;
    }
    row[4] = f_measure_from_counts(tpa, npa, ng);                                  // f_measure.py:71-81
    row[5] = (float)(sabs / (double)npx);                                          // mae.py:9
    {   // S-measure (s_measure.py:105-124)
        const double n1 = (double)ng, n0 = (double)npx - (double)ng;
        const double sum_p = qp[0] + qp[1] + qp[2] + qp[3];
        float Q;
        if (ng == 0) {
            Q = 1.0f - (float)(sum_p / (double)npx);
        } else if ((int)ng == npx) {
            Q = (float)(sum_p / (double)npx);
        } else {
            const float u = SM_DIV((float)ng, N);
            const float s_obj = u * object_score(n1, f_s, f_ss) + (1.0f - u) * object_score(n0, b_s, b_ss);
            const float area = (float)npx, Xf = (float)X, Yf = (float)Y;
            const float w1 = Xf * Yf / area, w2 = ((float)im.W - Xf) * Yf / area, w3 = Xf * ((float)im.H - Yf) / area;
            const float w4 = 1.0f - w1 - w2 - w3;
            const double nLT = (double)X * Y, nRT = (double)(im.W - X) * Y, nLB = (double)X * (im.H - Y),
                         nRB = (double)(im.W - X) * (im.H - Y);
            const float s_reg = w1 * ssim_quadrant(nLT, qp[0], qg[0], qpp[0], qpg[0]) +
                                w2 * ssim_quadrant(nRT, qp[1], qg[1], qpp[1], qpg[1]) +
                                w3 * ssim_quadrant(nLB, qp[2], qg[2], qpp[2], qpg[2]) +
                                w4 * ssim_quadrant(nRB, qp[3], qg[3], qpp[3], qpg[3]);
            Q = 0.5f * s_obj + 0.5f * s_reg;
            if (Q < 0.f) Q = 0.f;
        }
        row[6] = Q;
    }
    a.rows[(int64_t)b * 16 + 14 + which] = (float)q;
}

}  // namespace sm

extern "C" size_t sm_evaluate_workspace_bytes(int32_t B, int32_t nq) {
    if (B <= 0 || nq <= 0) return 0;
    return (size_t)B * nq * sizeof(sm::QueryStats) + (size_t)B * sizeof(sm::GtStats) + 256;
}

extern "C" int sm_evaluate_masks_f32(const sm_eval_args* a, void* stream) {
    SM_REQUIRE(a && a->mask_pred && a->objectness && a->gt && a->images && a->thresholds && a->rows && a->workspace,
               "sm_evaluate_masks_f32: null pointer");
    SM_REQUIRE(a->B > 0 && a->nq > 0 && a->mh > 0 && a->mw > 0 && a->scale >= 0.f, "sm_evaluate_masks_f32: bad shape");
    SM_REQUIRE(a->workspace_bytes >= sm_evaluate_workspace_bytes(a->B, a->nq) && ((uintptr_t)a->workspace % 16) == 0,
               "sm_evaluate_masks_f32: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    sm::QueryStats* qs = (sm::QueryStats*)a->workspace;
    sm::GtStats* gs = (sm::GtStats*)(qs + (size_t)a->B * a->nq);
    hipLaunchKernelGGL(sm::eval_query_kernel, dim3(a->nq, a->B), dim3(256), 0, st, *a, qs, gs);
    int rc = sm::check_launch("sm_evaluate_masks_f32/query");
    if (rc) return rc;
    hipLaunchKernelGGL(sm::eval_metrics_kernel, dim3(2, a->B), dim3(256), 0, st, *a, qs, gs);
    return sm::check_launch("sm_evaluate_masks_f32/metrics");
}
