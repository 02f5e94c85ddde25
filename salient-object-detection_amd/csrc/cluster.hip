// Candidate-mask extraction of the pseudo-mask generator, DINO branch (mask_generator.pyc@L136-200):
//     tokens (layer 12, cls dropped) -> F.interpolate(scale_factor=2, mode="bilinear", align_corners=True)
//     -> clusterer(features, k), k in {2, 3, 4} -> one-hot -> F.interpolate(scale_factor=stride // 2, mode="nearest")[..., :h, :w]
// The reference's `clusterings` module (KMeansClustering / SpectralClustering) exists in NO form in the repository, so the
// clustering itself has nothing to be pinned against: what is here is Lloyd's k-means on the up-sampled features with a
// deterministic farthest-point initialisation (the reference's cluster_type="kmeans" option in spirit; its default "spectral"
// - faiss k-NN affinity + eigen-decomposition - stays absent), restated in oracle/cluster_oracle.py and compared with
// scikit-learn from the same initial centres.  The two interpolations are PyTorch's, pinned against F.interpolate.
#include "common.h"

// torch-CPU arithmetic reproduced as written: no a*b+c -> fma contraction (with it, lambda = scale * o - floor(.) is formed
// from the UNROUNDED product and the interpolation weights move by an ulp of the source coordinate: 3e-6 on the features)
#pragma clang fp contract(off)

namespace sm {

// bilinear, align_corners=True, channels-last: up[(oy, ox)][c] - source coordinate o * (in - 1) / (out - 1) (ATen
// area_pixel_compute_scale / _source_index with align_corners: the scale is formed in fp32)
__global__ __launch_bounds__(256) void upsample_ac_kernel(const float* __restrict__ tok, int64_t strideb, float* __restrict__ up, int gh,
                                                         int gw, int sf, int64_t total4) {
    const int oh = sf * gh, ow = sf * gw, C4 = SM_EMBED / 4;
    const float sy = oh > 1 ? (float)(gh - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(gw - 1) / (float)(ow - 1) : 0.f;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % C4) * 4;
        const int64_t px = t / C4;
        const int ox = (int)(px % ow), oy = (int)((px / ow) % oh), b = (int)(px / ((int64_t)ow * oh));
        const float syf = sy * oy, sxf = sx * ox;
        const int y0 = (int)syf, x0 = (int)sxf;
        const int y1 = y0 + (y0 < gh - 1 ? 1 : 0), x1 = x0 + (x0 < gw - 1 ? 1 : 0);
        const float ly1 = syf - y0, ly0 = 1.f - ly1, lx1 = sxf - x0, lx0 = 1.f - lx1;
        const float* base = tok + (int64_t)b * strideb + c;
        const float4 p00 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * gw + x0) * SM_EMBED);
        const float4 p01 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * gw + x1) * SM_EMBED);
        const float4 p10 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * gw + x0) * SM_EMBED);
        const float4 p11 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * gw + x1) * SM_EMBED);
        float4 o;
        o.x = ly0 * (lx0 * p00.x + lx1 * p01.x) + ly1 * (lx0 * p10.x + lx1 * p11.x);
        o.y = ly0 * (lx0 * p00.y + lx1 * p01.y) + ly1 * (lx0 * p10.y + lx1 * p11.y);
        o.z = ly0 * (lx0 * p00.z + lx1 * p01.z) + ly1 * (lx0 * p10.z + lx1 * p11.z);
        o.w = ly0 * (lx0 * p00.w + lx1 * p01.w) + ly1 * (lx0 * p10.w + lx1 * p11.w);
        *reinterpret_cast<float4*>(up + px * SM_EMBED + c) = o;
    }
}

constexpr int KM_MAXK = 8, KM_THREADS = 1024, KM_WAVES = KM_THREADS / 64;

// squared distance of point p to a centre held in LDS: lane l takes dims l, l + 64, ... (6 each), butterfly sum
__device__ __forceinline__ float km_dist(const float* __restrict__ f, const float* c, int lane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SM_EMBED / 64; ++i) {
        const float d = f[lane + 64 * i] - c[lane + 64 * i];
        s += d * d;
    }
    return wave_sum(s);
}

// One workgroup per image: mean -> farthest-point initial centres (first: the point farthest from the mean; then the point
// farthest from its nearest chosen centre; ties go to the lowest index) -> `iters` Lloyd iterations.  Every sum runs in a fixed
// order (no atomics): the labels are a function of the input alone.  An emptied cluster keeps its previous centre.
__global__ __launch_bounds__(KM_THREADS) void kmeans_kernel(const float* __restrict__ feat_all, int n, int k, int iters,
                                                           int* __restrict__ labels_all, float* __restrict__ centers_all,
                                                           float* __restrict__ mind_all) {
    __shared__ float cen[KM_MAXK][SM_EMBED];
    __shared__ float part[2][KM_MAXK][SM_EMBED];
    __shared__ float cand_v[KM_WAVES];
    __shared__ int cand_i[KM_WAVES];
    __shared__ int cnt_w[KM_WAVES][KM_MAXK];
    __shared__ int cnt[KM_MAXK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* feat = feat_all + (int64_t)blockIdx.x * n * SM_EMBED;
    int* labels = labels_all + (int64_t)blockIdx.x * n;
    float* mind = mind_all + (int64_t)blockIdx.x * n;  // distance of every point to its nearest chosen centre (initialisation)
    const int d = tid % SM_EMBED, half = tid / SM_EMBED;  // threads 0..767: (dimension, half of the points); 768..1023 idle here
    const int p_lo = half == 0 ? 0 : n / 2, p_hi = half == 0 ? n / 2 : n;

    // ---- mean (centre slot KM_MAXK - 1 is scratch until the initialisation is over)
    if (half < 2) {
        float s = 0.f;
        for (int p = p_lo; p < p_hi; ++p) s += feat[(int64_t)p * SM_EMBED + d];
        part[half][0][d] = s;
    }
    __syncthreads();
    if (tid < SM_EMBED) cen[KM_MAXK - 1][tid] = (part[0][0][tid] + part[1][0][tid]) / (float)n;
    __syncthreads();

    // ---- farthest-point initialisation
    for (int j = 0; j < k; ++j) {
        const float* ref = j == 0 ? cen[KM_MAXK - 1] : cen[j - 1];
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int p = wave; p < n; p += KM_WAVES) {
            float dist = km_dist(feat + (int64_t)p * SM_EMBED, ref, lane);
            if (j > 0) {
                if (j > 1) dist = fminf(dist, mind[p]);
                if (lane == 0) mind[p] = dist;
            }
            if (dist > bv) { bv = dist; bi = p; }  // p ascending within the wave: the first maximum stays
        }
        if (lane == 0) { cand_v[wave] = bv; cand_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = cand_v[0];
            int i = cand_i[0];
            for (int w = 1; w < KM_WAVES; ++w)
                if (cand_v[w] > v || (cand_v[w] == v && cand_i[w] < i)) { v = cand_v[w]; i = cand_i[w]; }
            cand_i[0] = i;
        }
        __syncthreads();
        const int pick = min(cand_i[0], n - 1);  // non-finite features: every comparison fails and the sentinel index would leave the array
        if (tid < SM_EMBED) cen[j][tid] = feat[(int64_t)pick * SM_EMBED + tid];
        __syncthreads();
    }

    // ---- Lloyd iterations (the last pass only assigns)
    for (int it = 0; it <= iters; ++it) {
        int my_cnt[KM_MAXK];
#pragma unroll
        for (int c = 0; c < KM_MAXK; ++c) my_cnt[c] = 0;
        for (int p = wave; p < n; p += KM_WAVES) {
            float bv = INFINITY;
            int bc = 0;
            for (int c = 0; c < k; ++c) {
                const float dist = km_dist(feat + (int64_t)p * SM_EMBED, cen[c], lane);
                if (dist < bv) { bv = dist; bc = c; }  // ties: the lowest cluster index
            }
            if (lane == 0) labels[p] = bc;
#pragma unroll
            for (int c = 0; c < KM_MAXK; ++c) my_cnt[c] += (bc == c);
        }
        if (lane == 0)
#pragma unroll
            for (int c = 0; c < KM_MAXK; ++c) cnt_w[wave][c] = my_cnt[c];
        __syncthreads();  // labels (global, this workgroup's own writes) and the per-wave counts are visible
        if (it == iters) break;
        if (tid < k) {
            int s = 0;
            for (int w = 0; w < KM_WAVES; ++w) s += cnt_w[w][tid];
            cnt[tid] = s;
        }
        if (half < 2) {
            float s[KM_MAXK];
#pragma unroll
            for (int c = 0; c < KM_MAXK; ++c) s[c] = 0.f;
            for (int p = p_lo; p < p_hi; ++p) {
                const float v = feat[(int64_t)p * SM_EMBED + d];
                const int l = labels[p];
#pragma unroll
                for (int c = 0; c < KM_MAXK; ++c) s[c] += l == c ? v : 0.f;
            }
#pragma unroll
            for (int c = 0; c < KM_MAXK; ++c) part[half][c][d] = s[c];
        }
        __syncthreads();
        if (tid < SM_EMBED)
            for (int c = 0; c < k; ++c)
                if (cnt[c] > 0) cen[c][tid] = (part[0][c][tid] + part[1][c][tid]) / (float)cnt[c];
        __syncthreads();
    }
    if (centers_all && tid < SM_EMBED)
        for (int c = 0; c < k; ++c) centers_all[((int64_t)blockIdx.x * k + c) * SM_EMBED + tid] = cen[c][tid];
}

// one-hot + nearest up-sample by `s` + crop: masks[c][y][x] = labels[(y / s) * lw + x / s] == c  (F.interpolate(mode="nearest")
// with an integer scale factor: source index floor(dst * (1 / s)))
__global__ __launch_bounds__(256) void labels_to_masks_kernel(const int* __restrict__ labels, int lh, int lw, int s, int k, int H, int W,
                                                             unsigned char* __restrict__ masks) {
    const int64_t total = (int64_t)k * H * W;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % W), y = (int)((t / W) % H), c = (int)(t / ((int64_t)W * H));
        const int ly = min(y / s, lh - 1), lx = min(x / s, lw - 1);
        masks[t] = labels[ly * lw + lx] == c ? 1 : 0;
    }
}

struct LmSizes { int k[8], first[9]; };
// every (image, cluster size) in one launch: mask plane c of image b belongs to cluster size s = the one with first[s] <= c < first[s + 1]
__global__ __launch_bounds__(256) void labels_to_masks_batch_kernel(const int* __restrict__ labels, int n_sizes, LmSizes sz, int lh, int lw, int s,
                                                                   int H, int W, unsigned char* __restrict__ masks) {
    const int total_k = sz.first[n_sizes], plane = blockIdx.y, b = blockIdx.z;
    int si = 0;
    while (si + 1 < n_sizes && plane >= sz.first[si + 1]) ++si;
    const int c = plane - sz.first[si];
    const int* lab = labels + ((int64_t)b * n_sizes + si) * lh * lw;
    unsigned char* out = masks + ((int64_t)b * total_k + plane) * H * W;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < H * W; t += gridDim.x * 256) {
        const int x = t % W, y = t / W;
        out[t] = lab[min(y / s, lh - 1) * lw + min(x / s, lw - 1)] == c ? 1 : 0;
    }
}

}  // namespace sm

extern "C" int sm_labels_to_masks_batch_u8(const int32_t* labels, int32_t B, int32_t n_sizes, const int32_t* cluster_sizes, int32_t lh,
                                           int32_t lw, int32_t scale, int32_t H, int32_t W, uint8_t* masks, void* stream) {
    SM_REQUIRE(labels && masks && cluster_sizes && B > 0 && B <= 65535 && n_sizes >= 1 && n_sizes <= 8 && lh > 0 && lw > 0 && scale >= 1 &&
                   H > 0 && W > 0 && H <= lh * scale && W <= lw * scale,
               "sm_labels_to_masks_batch_u8: bad arguments (1..8 cluster sizes, H <= lh * scale, W <= lw * scale)");
    sm::LmSizes sz = {};
    for (int i = 0; i < n_sizes; ++i) {
        SM_REQUIRE(cluster_sizes[i] >= 1 && cluster_sizes[i] <= 64, "sm_labels_to_masks_batch_u8: cluster size %d", cluster_sizes[i]);
        sz.k[i] = cluster_sizes[i];
        sz.first[i + 1] = sz.first[i] + cluster_sizes[i];
    }
    int gx = (H * W + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(sm::labels_to_masks_batch_kernel, dim3(gx, sz.first[n_sizes], B), dim3(256), 0, (hipStream_t)stream, labels, n_sizes, sz,
                       lh, lw, scale, H, W, masks);
    return sm::check_launch("sm_labels_to_masks_batch_u8");
}

extern "C" int sm_upsample_tokens_aligned_f32(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw,
                                              int32_t scale, void* stream) {
    SM_REQUIRE(tok && up && B > 0 && gh > 0 && gw > 0 && strideb % 4 == 0 && scale >= 1 && scale <= 16,
               "sm_upsample_tokens_aligned_f32: bad arguments");
    const int64_t total4 = (int64_t)B * scale * scale * gh * gw * (SM_EMBED / 4);
    int64_t grid = (total4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(sm::upsample_ac_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, tok, strideb, up, gh, gw, scale, total4);
    return sm::check_launch("sm_upsample_tokens_aligned_f32");
}

extern "C" int sm_kmeans_f32(const float* feat, int32_t B, int32_t n, int32_t k, int32_t iters, int32_t* labels, float* centers,
                             float* workspace, void* stream) {
    SM_REQUIRE(feat && labels && workspace && B > 0 && n >= 1 && k >= 1 && k <= sm::KM_MAXK - 1 && k <= n && iters >= 0,
               "sm_kmeans_f32: bad arguments (1 <= k <= %d, k <= n)", sm::KM_MAXK - 1);
    hipLaunchKernelGGL(sm::kmeans_kernel, dim3(B), dim3(sm::KM_THREADS), 0, (hipStream_t)stream, feat, n, k, iters, labels, centers, workspace);
    return sm::check_launch("sm_kmeans_f32");
}

extern "C" int sm_labels_to_masks_u8(const int32_t* labels, int32_t lh, int32_t lw, int32_t scale, int32_t k, int32_t H, int32_t W,
                                     uint8_t* masks, void* stream) {
    SM_REQUIRE(labels && masks && lh > 0 && lw > 0 && scale >= 1 && k >= 1 && H > 0 && W > 0 && H <= lh * scale && W <= lw * scale,
               "sm_labels_to_masks_u8: bad arguments (H <= lh * scale, W <= lw * scale)");
    const int64_t total = (int64_t)k * H * W;
    int64_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(sm::labels_to_masks_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, labels, lh, lw, scale, k, H, W, masks);
    return sm::check_launch("sm_labels_to_masks_u8");
}
