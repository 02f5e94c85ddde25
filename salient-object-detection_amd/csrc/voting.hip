// Spectral-cluster VOTING step of the pseudo-mask generator (BASELINE.json configs[4]): among the candidate
// masks of one image (27 in the reference: 3 backbones x k = 2, 3, 4 clusters), drop the degenerate ones and keep the mask
// that agrees most with all the others.
//   filter_masks / mask_to_bbox   utils/misc.py:269-314 (same logic as mask_generator.pyc@L40-87)
//   vote_mask                     mask_generator.pyc@L202-230: iou[i][j] = |mi & mj| / (|mi | mj| + 1e-7)
//                                 over the surviving masks, score = row sum, best = argsort(descending)[0]
// The clustering that PRODUCES the candidates (faiss k-NN affinity + eigen-decomposition) is absent from the reference in
// any form and stays out of scope.
//
// Byte / integer work, HBM-bound on reading the masks once: each mask becomes a bitmap (16 pixels per lane, four lanes per word), its
// bounding box and area come from integer atomics (deterministic), and the pairwise intersections are popcounts of ANDed
// bitmap words - M (M+1) / 2 pairs x H W / 64 words, nothing next to the first pass.
#include "common.h"

#pragma clang fp contract(off)

namespace sm {

struct VoteBox { int ymin, ymax, xmin, xmax; unsigned area; };

// 16 pixels per lane: four lanes assemble one 64-pixel bitmap word, a workgroup covers 4096 pixels and issues ONE set of box atomics
// (a wave per word and five atomics per wave made 784 x 5 atomics per 224^2 mask on five addresses: 1.5 ms per 128 images x 9 masks;
// this form: see profiles/r04_kernel_stats_pseudo_masks.csv).  VEC: one 16-byte load per lane (every mask starts 16-byte aligned).
// `sizes` (per image: H_b, W_b; may be null): only the top-left H_b x W_b of an image's H x W planes counts - images of different sizes
// that pad to one token grid share a batch (the candidates of the padding are ignored, the coordinates are those of the image alone).
template <bool VEC>
__global__ __launch_bounds__(256) void vote_pack_kernel(const unsigned char* __restrict__ masks, unsigned long long* __restrict__ bits,
                                                       VoteBox* __restrict__ box, int H, int W, int words, int M,
                                                       const int* __restrict__ sizes) {
    const int m = blockIdx.y;
    const int Hb = sizes ? sizes[2 * blockIdx.z] : H, Wb = sizes ? sizes[2 * blockIdx.z + 1] : W;
    const int64_t npx = (int64_t)H * W;
    masks += (int64_t)blockIdx.z * M * npx;  // blockIdx.z: image of a batch (every per-image array is laid end to end)
    bits += (int64_t)blockIdx.z * M * words;
    box += (int64_t)blockIdx.z * M;
    const unsigned char* src = masks + (int64_t)m * npx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ymin = 1 << 30, ymax = -1, xmin = 1 << 30, xmax = -1;
    unsigned area = 0;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < (int64_t)words * 4; g += (int64_t)gridDim.x * 256) {  // quads stay together
        const int64_t p0 = g * 16;
        unsigned b16 = 0;
        if (VEC && p0 + 16 <= npx) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + p0);
            const unsigned q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // 0x80 in every non-zero byte, then the four flags gathered into bits 24..27 by one multiply (no two products collide)
                const unsigned t = ((q[k] | ((q[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7;
                b16 |= ((t * 0x01020408u) >> 24 & 0xfu) << (4 * k);
            }
        } else {
            for (int k = 0; k < 16; ++k)
                if (p0 + k < npx && src[p0 + k] != 0) b16 |= 1u << k;
        }
        if (b16 && (Hb < H || Wb < W)) {  // drop what lies outside the image's own H_b x W_b
            const int y0 = (int)(p0 / W), x0 = (int)(p0 - (int64_t)y0 * W);
            if (!(y0 < Hb && x0 + 15 < Wb))  // (inside in one piece: the 16 pixels are in row y0, left of W_b)
                for (unsigned r = b16; r; r &= r - 1) {
                    const int k = __ffs(r) - 1;
                    const int64_t p = p0 + k;
                    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
                    if (y >= Hb || x >= Wb) b16 &= ~(1u << k);
                }
        }
        unsigned long long part = (unsigned long long)b16 << (16 * (lane & 3));
        part |= __shfl_xor(part, 1, 64);
        part |= __shfl_xor(part, 2, 64);
        if ((lane & 3) == 0) bits[(int64_t)m * words + (g >> 2)] = part;
        if (b16) {
            const int64_t pf = p0 + __ffs(b16) - 1, pl = p0 + 31 - __clz(b16);
            const int yf = (int)(pf / W), yl = (int)(pl / W);
            ymin = min(ymin, yf); ymax = max(ymax, yl);
            if (yf == yl) {
                xmin = min(xmin, (int)(pf - (int64_t)yf * W)); xmax = max(xmax, (int)(pl - (int64_t)yf * W));
            } else {  // the 16 pixels cross a row end
                for (unsigned r = b16; r; r &= r - 1) {
                    const int64_t p = p0 + __ffs(r) - 1;
                    const int x = (int)(p - (p / W) * W);
                    xmin = min(xmin, x); xmax = max(xmax, x);
                }
            }
            area += __popc(b16);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ymin = min(ymin, __shfl_xor(ymin, o, 64)); ymax = max(ymax, __shfl_xor(ymax, o, 64));
        xmin = min(xmin, __shfl_xor(xmin, o, 64)); xmax = max(xmax, __shfl_xor(xmax, o, 64));
        area += __shfl_xor(area, o, 64);
    }
    __shared__ int red[4][5];
    if (lane == 0) { red[wave][0] = ymin; red[wave][1] = ymax; red[wave][2] = xmin; red[wave][3] = xmax; red[wave][4] = (int)area; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            ymin = min(ymin, red[k][0]); ymax = max(ymax, red[k][1]); xmin = min(xmin, red[k][2]); xmax = max(xmax, red[k][3]);
            area += (unsigned)red[k][4];
        }
        if (area) {
            atomicMin(&box[m].ymin, ymin); atomicMax(&box[m].ymax, ymax);
            atomicMin(&box[m].xmin, xmin); atomicMax(&box[m].xmax, xmax);
            atomicAdd(&box[m].area, area);
        }
    }
}

__global__ __launch_bounds__(256) void vote_init_kernel(VoteBox* box, unsigned* inter, int M) {
    box += (int64_t)blockIdx.z * M;
    inter += (int64_t)blockIdx.z * M * M;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < M) { box[t].ymin = 1 << 30; box[t].ymax = -1; box[t].xmin = 1 << 30; box[t].xmax = -1; box[t].area = 0; }
    if (t < M * M) inter[t] = 0;
}

// one workgroup per (pair (i, j >= i), word chunk)
__global__ __launch_bounds__(256) void vote_pairs_kernel(const unsigned long long* __restrict__ bits, unsigned* __restrict__ inter, int M,
                                                        int words) {
    bits += (int64_t)blockIdx.z * M * words;
    inter += (int64_t)blockIdx.z * M * M;
    int i = 0, rem = blockIdx.y;  // pair index -> (i, j): row i holds M - i pairs
    while (rem >= M - i) { rem -= M - i; ++i; }
    const int j = i + rem;
    const unsigned long long* a = bits + (int64_t)i * words;
    const unsigned long long* b = bits + (int64_t)j * words;
    unsigned c = 0;
    for (int wd = blockIdx.x * 256 + threadIdx.x; wd < words; wd += gridDim.x * 256) c += __popcll(a[wd] & b[wd]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&inter[i * M + j], c);
}

__global__ __launch_bounds__(64) void vote_finalize_kernel(const VoteBox* __restrict__ box, const unsigned* __restrict__ inter, int M, int H,
                                                          int W, int remove_long, int remove_small_large, int* __restrict__ keep,
                                                          float* __restrict__ iou, float* __restrict__ row_sums, int* __restrict__ best,
                                                          const int* __restrict__ sizes) {
    const int t = threadIdx.x;
    if (sizes) { H = sizes[2 * blockIdx.z]; W = sizes[2 * blockIdx.z + 1]; }
    box += (int64_t)blockIdx.z * M; inter += (int64_t)blockIdx.z * M * M; keep += (int64_t)blockIdx.z * M;
    iou += (int64_t)blockIdx.z * M * M; row_sums += (int64_t)blockIdx.z * M; best += blockIdx.z;
    __shared__ int skeep[64];
    if (t < M) {
        const VoteBox b = box[t];
        int k = b.area > 0;  // mask_to_bbox skips masks that predict nothing (misc.py:277-281)
        if (k && remove_long) {
            if (b.ymin == 0 && b.ymax + 1 == H) k = 0;
            else if (b.xmin == 0 && b.xmax + 1 == W) k = 0;
        }
        if (k && remove_small_large) {  // misc.py:302-306: tensor sum (fp32 after promotion) against python doubles
            if ((float)b.area < (float)(0.05 * H * W)) k = 0;
            else if ((double)((b.xmax - b.xmin) * (b.ymax - b.ymin)) > 0.95 * H * W) k = 0;
        }
        skeep[t] = k;
    }
    __syncthreads();
    // "rare case where all predictions are filtered" (misc.py:311-314): torch.stack of the empty list raises, the reference
    // catches it and returns dt_masks UNFILTERED with the identity index map - the vote then runs over all M candidates
    // (empty ones included: their IoU entries are 0 / (0 + 1e-7) = 0)
    int any = 0;
    for (int j = 0; j < M; ++j) any |= skeep[j];
    __syncthreads();
    if (t < M) {
        if (!any) skeep[t] = 1;
        keep[t] = skeep[t];
    }
    __syncthreads();
    if (t < M) {
        float s = 0.f;
        for (int j = 0; j < M; ++j) {
            float v = 0.f;
            if (skeep[t] && skeep[j]) {
                const unsigned in = t <= j ? inter[t * M + j] : inter[j * M + t];
                const unsigned un = box[t].area + box[j].area - in;
                v = (float)in / ((float)un + 1e-7f);
                s = s + v;  // surviving masks in index order
            }
            iou[t * M + j] = v;
        }
        row_sums[t] = skeep[t] ? s : -1.f;
    }
    __syncthreads();
    if (t == 0) {  // first maximum among the survivors (ties: the fixtures are tie-free, torch's argsort is unstable)
        int bi = -1;
        float bs = -1.f;
        for (int i = 0; i < M; ++i)
            if (skeep[i] && row_sums[i] > bs) { bs = row_sums[i]; bi = i; }
        *best = bi;
    }
}

// Run boundaries of a mask in column-major order (the order of COCO's run-length code): one workgroup per image, every lane a
// contiguous stretch of positions q = x * H + y - count the changes, scan the counts over the workgroup, write the positions in place.
__global__ __launch_bounds__(1024) void rle_runs_kernel(const unsigned char* __restrict__ masks, int Hp, int Wp, const int* __restrict__ sizes,
                                                        int* __restrict__ starts, int cap, int* __restrict__ info) {
    // the image's own H x W inside planes of Hp x Wp (sizes null: the whole plane); positions count in the image's H
    const int H = sizes ? sizes[2 * blockIdx.x] : Hp, W = sizes ? sizes[2 * blockIdx.x + 1] : Wp;
    const int64_t npx = (int64_t)H * W;
    const unsigned char* m = masks + blockIdx.x * (int64_t)Hp * Wp;
    int* out = starts + (int64_t)blockIdx.x * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int np = (int)npx, per = (np + 1023) / 1024, q0 = tid * per, q1 = q0 + per < np ? q0 + per : np;
    // the pixel before the stretch (pixel 0 itself for the first stretch: no change there)
    const int qb = q0 > 0 ? q0 - 1 : 0;
    const bool before = q0 < q1 ? m[(int64_t)(qb % H) * Wp + qb / H] != 0 : false;
    auto walk = [&](auto&& on_change) {  // (y, x) advance with q: no division per pixel
        int y = q0 % H, x = q0 / H;
        bool prev = before;
        for (int q = q0; q < q1; ++q) {
            const bool v = m[(int64_t)y * Wp + x] != 0;
            if (v != prev) on_change(q);
            prev = v;
            if (++y == H) { y = 0; ++x; }
        }
    };
    int mine = 0;
    walk([&](int) { ++mine; });
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    __shared__ int wtot[16];
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int at_out = inc - mine, total = 0;
    for (int w = 0; w < 16; ++w) { at_out += w < wave ? wtot[w] : 0; total += wtot[w]; }
    if (tid == 0) { info[2 * blockIdx.x] = total; info[2 * blockIdx.x + 1] = m[0] != 0 ? 1 : 0; }
    walk([&](int q) {
        if (at_out < cap) out[at_out] = q;
        ++at_out;
    });
}

}  // namespace sm

extern "C" int sm_rle_runs_u8(const uint8_t* masks, int32_t B, int32_t H, int32_t W, const int32_t* sizes, int32_t* starts, int32_t cap,
                              int32_t* info, void* stream) {
    SM_REQUIRE(masks && starts && info, "sm_rle_runs_u8: null pointer");
    SM_REQUIRE(B > 0 && H > 0 && W > 0 && cap > 0 && (int64_t)H * W < (int64_t)1 << 31, "sm_rle_runs_u8: %d masks of %dx%d, cap %d", B, H, W, cap);
    hipLaunchKernelGGL(sm::rle_runs_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, masks, H, W, sizes, starts, cap, info);
    return sm::check_launch("sm_rle_runs_u8");
}

extern "C" size_t sm_vote_workspace_bytes(int32_t M, int32_t H, int32_t W) {
    if (M <= 0 || M > 64 || H <= 0 || W <= 0) return 0;
    const size_t words = ((size_t)H * W + 63) / 64;
    return (((size_t)M * words * 8 + 255) & ~(size_t)255) + (((size_t)M * sizeof(sm::VoteBox) + 255) & ~(size_t)255) +
           (((size_t)M * M * 4 + 255) & ~(size_t)255);
}

extern "C" int sm_vote_masks_sized_u8(const uint8_t* masks, int32_t B, int32_t M, int32_t H, int32_t W, const int32_t* sizes, int32_t remove_long,
                                      int32_t remove_small_large, int32_t* keep, float* iou, float* row_sums, int32_t* best, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    SM_REQUIRE(masks && keep && iou && row_sums && best && workspace, "sm_vote_masks_u8: null pointer");
    SM_REQUIRE(B > 0 && B <= 65535 && M > 0 && M <= 64 && H > 0 && W > 0, "sm_vote_masks_u8: %d images, M=%d candidates (1..64), %dx%d", B, M, H, W);
    SM_REQUIRE(workspace_bytes >= (size_t)B * sm_vote_workspace_bytes(M, H, W) && ((uintptr_t)workspace % 256) == 0,
               "sm_vote_masks_u8: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    const int words = (int)(((size_t)H * W + 63) / 64);
    char* w = (char*)workspace;  // [bitmaps of all images | boxes of all images | intersections of all images]
    auto* bits = (unsigned long long*)w;
    w += ((size_t)B * M * words * 8 + 255) & ~(size_t)255;
    auto* box = (sm::VoteBox*)w;
    w += ((size_t)B * M * sizeof(sm::VoteBox) + 255) & ~(size_t)255;
    auto* inter = (unsigned*)w;
    hipLaunchKernelGGL(sm::vote_init_kernel, dim3((M * M + 255) / 256, 1, B), dim3(256), 0, st, box, inter, M);
    const int gx = (words + 63) / 64 < 64 ? (words + 63) / 64 : 64;  // 64 words (4096 pixels) per workgroup and pass
    if (((uintptr_t)masks % 16) == 0 && ((int64_t)H * W) % 16 == 0)
        hipLaunchKernelGGL(sm::vote_pack_kernel<true>, dim3(gx, M, B), dim3(256), 0, st, masks, bits, box, H, W, words, M, sizes);
    else
        hipLaunchKernelGGL(sm::vote_pack_kernel<false>, dim3(gx, M, B), dim3(256), 0, st, masks, bits, box, H, W, words, M, sizes);
    hipLaunchKernelGGL(sm::vote_pairs_kernel, dim3((words + 255) / 256 < 16 ? (words + 255) / 256 : 16, M * (M + 1) / 2, B), dim3(256), 0, st,
                       bits, inter, M, words);
    hipLaunchKernelGGL(sm::vote_finalize_kernel, dim3(1, 1, B), dim3(64), 0, st, box, inter, M, H, W, remove_long, remove_small_large, keep, iou,
                       row_sums, best, sizes);
    return sm::check_launch("sm_vote_masks_u8");
}

extern "C" int sm_vote_masks_batch_u8(const uint8_t* masks, int32_t B, int32_t M, int32_t H, int32_t W, int32_t remove_long,
                                      int32_t remove_small_large, int32_t* keep, float* iou, float* row_sums, int32_t* best, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    return sm_vote_masks_sized_u8(masks, B, M, H, W, nullptr, remove_long, remove_small_large, keep, iou, row_sums, best, workspace,
                                  workspace_bytes, stream);
}

extern "C" int sm_vote_masks_u8(const uint8_t* masks, int32_t M, int32_t H, int32_t W, int32_t remove_long, int32_t remove_small_large,
                                int32_t* keep, float* iou, float* row_sums, int32_t* best, void* workspace, size_t workspace_bytes,
                                void* stream) {
    return sm_vote_masks_batch_u8(masks, 1, M, H, W, remove_long, remove_small_large, keep, iou, row_sums, best, workspace, workspace_bytes,
                                  stream);
}
