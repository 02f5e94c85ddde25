// Fast bilateral solver refinement on the device, fp64 like the reference.
// Reference: bilateral_solver.py:41-193 (BilateralGrid, bistochastize, BilateralSolver.solve, the fill-holes /
// second-largest-component post-processing).
//
// Design notes (HBM / latency-bound, irregular; ~3-15 k vertices for a 224^2..400^2 image):
//   * np.unique over the 5-D hash is replaced by a DENSE OCCUPANCY BITMAP over the (v,u,l,y,x) cell lattice in hash
//     order: vertex id = number of occupied cells before this one = prefix popcount.  No sort, deterministic, and
//     identical to the reference's ascending-unique-hash numbering; neighbour look-ups (the 5 blur matrices) are a
//     bit test + rank instead of np.searchsorted.
//   * splat (S.x) is done per 16x16 spatial cell by one workgroup with a leader scan, so every vertex sums its
//     pixels in ascending pixel order exactly like scipy's CSR mat-vec: bit-reproducible, no float atomics.
//   * bistochastize (10 sweeps) + Jacobi-PCG (<= 25 its, scipy cg semantics) + the A.p mat-vec in ascending
//     column order run inside ONE persistent workgroup per image (block barriers only, no grid sync).
//   * post-processing (threshold, binary_fill_holes, 4-connected label, second-largest label incl. background) is
//     an atomicMin union-find run by one workgroup (labels are ordered by their first raster pixel, as ndimage.label).
#include "common.h"
#include <type_traits>
#include <math.h>

#pragma clang fp contract(off)  // the oracle / reference use separate fp64 mul and add; fused ops are explicit

namespace sm {

constexpr int BS_THREADS = 1024;

struct BsDims {
    int H, W, NX, NY, NL, NC;
    int ss, npx;
    double sl, sc;
    long long ncells;
    int nwords;
};

struct BsWs {  // workspace carve-up
    unsigned* bitmap;    // [nwords]
    unsigned* wordrank;  // [nwords]
    unsigned* cell;      // [npx] cell id per pixel
    int* idx;            // [npx] vertex id per pixel
    unsigned* vcell;     // [maxV]
    int* nbr;            // [10][maxV]  row k = 2*d + s  (s = 0: coord-1, s = 1: coord+1)
    double *m, *ws, *b, *n0, *n1, *diag, *minv, *x, *r, *p, *q;  // [maxV] each
    int* parent;         // [npx]
    unsigned char* bin;  // [npx]
    unsigned* csize;     // [npx] component sizes
    int* scal;           // [16] scalars: 0 = V, 1 = cg iterations, 2 = n components, 3 = chosen
    unsigned long long* keys;  // [4]
    size_t total;
};

// Batched solves: image i of a launch lives at workspace + i * ws_bytes and at pixel offset i * npx of the image /
// target / output arrays; blockIdx.z is the image.  Single-workgroup kernels then run one workgroup PER IMAGE, which
// is what fills the GPU (a lone solve occupies one CU).
struct BsBatch {
    size_t ws_bytes;
};
__device__ __forceinline__ BsWs bs_image_ws(BsWs w, size_t bytes) {
    auto mv = [&](auto*& ptr) { ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<char*>(ptr) + bytes); };
    mv(w.bitmap); mv(w.wordrank); mv(w.cell); mv(w.idx); mv(w.vcell); mv(w.nbr);
    mv(w.m); mv(w.ws); mv(w.b); mv(w.n0); mv(w.n1); mv(w.diag); mv(w.minv); mv(w.x); mv(w.r); mv(w.p); mv(w.q);
    mv(w.parent); mv(w.bin); mv(w.csize); mv(w.scal); mv(w.keys);
    return w;
}

static BsDims make_dims(int H, int W, double ss, double sl, double sc) {
    BsDims d;
    d.H = H; d.W = W; d.ss = (int)ss; d.sl = sl; d.sc = sc; d.npx = H * W;
    d.NX = (W - 1) / d.ss + 1; d.NY = (H - 1) / d.ss + 1;
    d.NL = (int)(255.0 / sl) + 1;
    d.NC = (int)(255.5 / sc) + 1;
    d.ncells = (long long)d.NX * d.NY * d.NL * d.NC * d.NC;
    d.nwords = (int)((d.ncells + 31) / 32);
    return d;
}

static BsWs carve_bs(const BsDims& d, char* base) {
    BsWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    const size_t maxV = (size_t)d.npx;  // at most one vertex per pixel
    w.bitmap = (unsigned*)take((size_t)d.nwords * 4);
    w.wordrank = (unsigned*)take((size_t)d.nwords * 4);
    w.cell = (unsigned*)take((size_t)d.npx * 4);
    w.idx = (int*)take((size_t)d.npx * 4);
    w.vcell = (unsigned*)take(maxV * 4);
    w.nbr = (int*)take(maxV * 4 * 10);
    double** arrs[] = {&w.m, &w.ws, &w.b, &w.n0, &w.n1, &w.diag, &w.minv, &w.x, &w.r, &w.p, &w.q};
    for (auto a : arrs) *a = (double*)take(maxV * 8);
    w.parent = (int*)take((size_t)d.npx * 4);
    w.bin = (unsigned char*)take((size_t)d.npx);
    w.csize = (unsigned*)take((size_t)d.npx * 4);
    w.scal = (int*)take(64);
    w.keys = (unsigned long long*)take(64);
    w.total = off;
    return w;
}

// ---- grid construction ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bs_cells_kernel(const unsigned char* __restrict__ img, BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    img += (size_t)blockIdx.z * d.npx * 3;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.npx) return;
    const int y = p / d.W, x = p - y * d.W;
    const double R = img[p * 3 + 0], G = img[p * 3 + 1], B = img[p * 3 + 2];
    // rgb2yuv (:21-22): numpy's tensordot = dgemm, K = 3, accumulates fma(B, m2, fma(G, m1, R*m0)), then + offset
    const double Y = __builtin_fma(B, 0.114, __builtin_fma(G, 0.587, R * 0.299));
    const double U = __builtin_fma(B, 0.5, __builtin_fma(G, -0.331264, R * -0.168736)) + 128.0;
    const double V = __builtin_fma(B, -0.081312, __builtin_fma(G, -0.418688, R * 0.5)) + 128.0;
    const int cx = x / d.ss, cy = y / d.ss;
    const int cl = (int)(Y / d.sl), cu = (int)(U / d.sc), cv = (int)(V / d.sc);
    const unsigned cell = (unsigned)((((long long)(cv * d.NC + cu) * d.NL + cl) * d.NY + cy) * d.NX + cx);
    w.cell[p] = cell;
    atomicOr(&w.bitmap[cell >> 5], 1u << (cell & 31));
}

// exclusive prefix sum of popcount(bitmap[w]) by one workgroup; scal[0] = number of vertices
__global__ __launch_bounds__(BS_THREADS) void bs_scan_kernel(BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    __shared__ unsigned part[BS_THREADS];
    const int t = threadIdx.x;
    const int per = (d.nwords + BS_THREADS - 1) / BS_THREADS;
    const int lo = t * per, hi = min(lo + per, d.nwords);
    unsigned s = 0;
    for (int i = lo; i < hi; ++i) s += __popc(w.bitmap[i]);
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < BS_THREADS; o <<= 1) {  // Hillis-Steele inclusive scan
        const unsigned v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - s;
    for (int i = lo; i < hi; ++i) { w.wordrank[i] = run; run += __popc(w.bitmap[i]); }
    if (t == BS_THREADS - 1) w.scal[0] = (int)part[t];
}

__device__ __forceinline__ int bs_rank(const BsWs& w, unsigned cell) {
    const unsigned word = w.bitmap[cell >> 5], bit = cell & 31;
    return (int)(w.wordrank[cell >> 5] + __popc(word & ((1u << bit) - 1u)));
}

__global__ __launch_bounds__(256) void bs_vertices_kernel(BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.nwords) return;
    unsigned word = w.bitmap[i], r = w.wordrank[i];
    while (word) {
        const int b = __ffs(word) - 1;
        w.vcell[r++] = (unsigned)i * 32u + b;
        word &= word - 1;
    }
}

__global__ __launch_bounds__(256) void bs_pixel_vertex_kernel(BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < d.npx) w.idx[p] = bs_rank(w, w.cell[p]);
}

// blur matrices (:66-81) as neighbour tables: +-1 along each of the 5 lattice axes, present iff the cell is occupied
__global__ __launch_bounds__(256) void bs_neighbors_kernel(BsDims d, BsWs w, int maxV, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int V = w.scal[0];
    if (v >= V) return;
    const unsigned cell = w.vcell[v];
    const int dimsz[5] = {d.NX, d.NY, d.NL, d.NC, d.NC};
    unsigned rem = cell, stride = 1;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = (int)(rem % (unsigned)dimsz[k]);
        rem /= (unsigned)dimsz[k];
        int lo = -1, hi = -1;
        if (c > 0) {
            const unsigned nc = cell - stride;
            if (w.bitmap[nc >> 5] >> (nc & 31) & 1u) lo = bs_rank(w, nc);
        }
        if (c < dimsz[k] - 1) {
            const unsigned nc = cell + stride;
            if (w.bitmap[nc >> 5] >> (nc & 31) & 1u) hi = bs_rank(w, nc);
        }
        w.nbr[(size_t)(2 * k) * maxV + v] = lo;
        w.nbr[(size_t)(2 * k + 1) * maxV + v] = hi;
        stride *= (unsigned)dimsz[k];
    }
}

// splat of {1, w, t*w} (:87-88, :133-137): one workgroup per ss x ss spatial cell; every vertex lives in exactly one
// such cell, its leader (first pixel in raster order) sums the cell's pixels of that vertex in ascending pixel order.
__global__ void bs_splat_kernel(const double* __restrict__ target, double conf, BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    target += (size_t)blockIdx.z * d.npx;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int n = d.ss * d.ss;
    int* vid = (int*)lds;
    double* tw = (double*)(lds + ((n * 4 + 15) & ~15));
    const int t = threadIdx.x;
    const int ly = t / d.ss, lx = t - ly * d.ss;
    const int y = blockIdx.y * d.ss + ly, x = blockIdx.x * d.ss + lx;
    const bool inside = t < n && y < d.H && x < d.W;
    if (t < n) {
        vid[t] = inside ? w.idx[y * d.W + x] : -1;
        tw[t] = inside ? target[y * d.W + x] * conf : 0.0;
    }
    __syncthreads();
    if (!inside) return;
    const int v = vid[t];
    for (int s = 0; s < t; ++s)
        if (vid[s] == v) return;  // not the leader
    double cnt = 0.0, sw = 0.0, sb = 0.0;
    for (int s = t; s < n; ++s) {
        if (vid[s] == v) { cnt += 1.0; sw += conf; sb += tw[s]; }
    }
    w.m[v] = cnt; w.ws[v] = sw; w.b[v] = sb;
}

// ---- bistochastize + PCG, one persistent workgroup ----------------------------------------------------------------
__device__ double bs_block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < BS_THREADS / 64; ++i) s += red[i];  // fixed order: deterministic
    __syncthreads();
    return s;
}

__device__ __forceinline__ double bs_blur(const double* __restrict__ x, const int* __restrict__ nbr, int maxV, int v) {
    double out = 10.0 * x[v];  // 2 * dim * x  (:97)
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // out = out + blur_k.dot(x): row entries in ascending column order
        double t = 0.0;
        const int lo = nbr[(size_t)(2 * k) * maxV + v], hi = nbr[(size_t)(2 * k + 1) * maxV + v];
        if (lo >= 0) t = t + x[lo];
        if (hi >= 0) t = t + x[hi];
        out = out + t;
    }
    return out;
}

// A.p with A = lam*(Dm - Dn blur(Dn)) + diag(S w) as scipy assembles it: ascending column order = the "-1" neighbours
// from the slowest lattice axis (v) to the fastest (x), the diagonal, then the "+1" neighbours from x to v.
__device__ __forceinline__ double bs_matvec(const double* __restrict__ p, const double* __restrict__ n,
                                            const double* __restrict__ diag, const int* __restrict__ nbr, int maxV,
                                            int v, double lam) {
    double out = 0.0;
    const double nv = n[v];
#pragma unroll
    for (int k = 4; k >= 0; --k) {
        const int j = nbr[(size_t)(2 * k) * maxV + v];
        if (j >= 0) out = out + (-(lam * (nv * n[j]))) * p[j];
    }
    out = out + diag[v] * p[v];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int j = nbr[(size_t)(2 * k + 1) * maxV + v];
        if (j >= 0) out = out + (-(lam * (nv * n[j]))) * p[j];
    }
    return out;
}

__global__ __launch_bounds__(BS_THREADS) void bs_solve_kernel(BsWs w, int maxV, double lam, double diag_min, int maxiter,
                                                             double tol, BsBatch batch) {
    __shared__ double red[BS_THREADS / 64];
    w = bs_image_ws(w, blockIdx.z * batch.ws_bytes);
    const int t = threadIdx.x, V = w.scal[0];
    double *na = w.n0, *nb = w.n1;
    // bistochastize (:107-118): n <- sqrt(n*m / blur(n)) x10 from n = 1, then m <- n * blur(n)
    for (int v = t; v < V; v += BS_THREADS) na[v] = 1.0;
    __syncthreads();
    for (int it = 0; it < 10; ++it) {
        for (int v = t; v < V; v += BS_THREADS) nb[v] = sqrt(na[v] * w.m[v] / bs_blur(na, w.nbr, maxV, v));
        __syncthreads();
        double* tmp = na; na = nb; nb = tmp;
    }
    const double* n = na;
    // A diagonal, Jacobi preconditioner, flat initialisation (:132-142)
    for (int v = t; v < V; v += BS_THREADS) {
        const double mv = n[v] * bs_blur(n, w.nbr, maxV, v);
        const double dg = lam * (mv - n[v] * (10.0 * n[v])) + w.ws[v];
        w.diag[v] = dg;
        w.minv[v] = 1.0 / fmax(dg, diag_min);
        w.x[v] = w.b[v] / w.ws[v];
    }
    __syncthreads();
    // scipy.sparse.linalg.cg(A, b, x0, M, maxiter, rtol=tol, atol=0)
    double bb = 0.0;
    for (int v = t; v < V; v += BS_THREADS) bb += w.b[v] * w.b[v];
    const double bnrm = sqrt(bs_block_sum(bb, red));
    const double atol = tol * bnrm;
    int iters = 0;
    if (bnrm == 0.0) {
        for (int v = t; v < V; v += BS_THREADS) w.x[v] = w.b[v];
    } else {
        for (int v = t; v < V; v += BS_THREADS) w.r[v] = w.b[v] - bs_matvec(w.x, n, w.diag, w.nbr, maxV, v, lam);
        __syncthreads();
        double rho_prev = 0.0;
        for (int it = 0; it < maxiter; ++it) {
            double rr = 0.0;
            for (int v = t; v < V; v += BS_THREADS) rr += w.r[v] * w.r[v];
            if (sqrt(bs_block_sum(rr, red)) < atol) break;
            double rz = 0.0;
            for (int v = t; v < V; v += BS_THREADS) rz += w.r[v] * (w.minv[v] * w.r[v]);
            const double rho = bs_block_sum(rz, red);
            if (it > 0) {
                const double beta = rho / rho_prev;
                for (int v = t; v < V; v += BS_THREADS) w.p[v] = w.p[v] * beta + w.minv[v] * w.r[v];
            } else {
                for (int v = t; v < V; v += BS_THREADS) w.p[v] = w.minv[v] * w.r[v];
            }
            __syncthreads();
            double pq = 0.0;
            for (int v = t; v < V; v += BS_THREADS) {
                const double qv = bs_matvec(w.p, n, w.diag, w.nbr, maxV, v, lam);
                w.q[v] = qv;
                pq += w.p[v] * qv;
            }
            const double alpha = rho / bs_block_sum(pq, red);
            for (int v = t; v < V; v += BS_THREADS) {
                w.x[v] = w.x[v] + alpha * w.p[v];
                w.r[v] = w.r[v] - alpha * w.q[v];
            }
            __syncthreads();
            rho_prev = rho;
            iters = it + 1;
        }
    }
    if (t == 0) w.scal[1] = iters;
}

__global__ __launch_bounds__(256) void bs_slice_kernel(BsDims d, BsWs w, double* __restrict__ soft, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    soft += (size_t)blockIdx.z * d.npx;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < d.npx) soft[p] = w.x[w.idx[p]];  // S^T y (:90-91)
}

// ---- post-processing (:184-192) by one workgroup ----------------------------------------------------------------------
__device__ __forceinline__ int uf_find(int* parent, int i) {
    int p;
    while ((p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != i) i = p;
    return i;
}
__device__ void uf_union(int* parent, int a, int b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }       // a < b: hang b under a (roots = first raster pixel)
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

__device__ void uf_label(int* parent, const unsigned char* bin, int H, int W, bool fg_only) {
    const int npx = H * W, t = threadIdx.x;
    for (int p = t; p < npx; p += BS_THREADS) parent[p] = p;
    __syncthreads();
    for (int p = t; p < npx; p += BS_THREADS) {
        const unsigned char b = bin[p];
        if (fg_only && !b) continue;
        const int x = p % W;
        if (x > 0 && bin[p - 1] == b) uf_union(parent, p, p - 1);
        if (p >= W && bin[p - W] == b) uf_union(parent, p, p - W);
    }
    __syncthreads();
    for (int p = t; p < npx; p += BS_THREADS) parent[p] = uf_find(parent, p);
    __syncthreads();
}

__global__ __launch_bounds__(BS_THREADS) void bs_post_kernel(BsDims d, BsWs w, const double* __restrict__ soft,
                                                            unsigned char* __restrict__ out, BsBatch bb) {
    __shared__ unsigned long long best, second;
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    soft += (size_t)blockIdx.z * d.npx;
    out += (size_t)blockIdx.z * d.npx;
    __shared__ unsigned nfg, ncomp;
    const int npx = d.npx, t = threadIdx.x, H = d.H, W = d.W;
    int* parent = w.parent;
    unsigned char* bin = w.bin;
    unsigned* csize = w.csize;
    for (int p = t; p < npx; p += BS_THREADS) bin[p] = soft[p] > 0.5;
    __syncthreads();
    // binary_fill_holes: background pixels not 4-connected to the image border become foreground
    uf_label(parent, bin, H, W, false);
    for (int p = t; p < npx; p += BS_THREADS) csize[p] = 0;   // reused as "touches the border" flags
    __syncthreads();
    for (int p = t; p < npx; p += BS_THREADS) {
        const int y = p / W, x = p - y * W;
        if (!bin[p] && (y == 0 || x == 0 || y == H - 1 || x == W - 1)) csize[parent[p]] = 1;
    }
    __syncthreads();
    for (int p = t; p < npx; p += BS_THREADS)
        if (!bin[p] && !csize[parent[p]]) bin[p] = 1;
    __syncthreads();
    // ndimage.label (4-connectivity) on the filled mask; component sizes
    uf_label(parent, bin, H, W, true);
    for (int p = t; p < npx; p += BS_THREADS) csize[p] = 0;
    if (t == 0) { best = 0; second = 0; nfg = 0; ncomp = 0; }
    __syncthreads();
    unsigned myfg = 0;
    for (int p = t; p < npx; p += BS_THREADS)
        if (bin[p]) { atomicAdd(&csize[parent[p]], 1u); ++myfg; }
    atomicAdd(&nfg, myfg);
    __syncthreads();
    // nb_pixel = [background, label 1, label 2, ...] (labels in raster order of their first pixel);
    // argsort ascending, take [-2]: key = (size << 32 | order) with order 0 = background, root+1 = component
    const unsigned long long bgkey = ((unsigned long long)(npx - nfg) << 32);
    for (int p = t; p < npx; p += BS_THREADS)
        if (bin[p] && parent[p] == p) {
            atomicMax(&best, ((unsigned long long)csize[p] << 32) | (unsigned)(p + 1));
            atomicAdd(&ncomp, 1u);
        }
    if (t == 0) atomicMax(&best, bgkey);
    __syncthreads();
    for (int p = t; p < npx; p += BS_THREADS)
        if (bin[p] && parent[p] == p) {
            const unsigned long long k = ((unsigned long long)csize[p] << 32) | (unsigned)(p + 1);
            if (k < best) atomicMax(&second, k);
        }
    if (t == 0 && bgkey < best) atomicMax(&second, bgkey);
    __syncthreads();
    const unsigned long long pick = second;
    const bool none = ncomp == 0;                        // IndexError branch (:191-192): all ones
    const bool pick_bg = (unsigned)(pick & 0xffffffffu) == 0;
    const int pick_root = (int)(pick & 0xffffffffu) - 1;
    for (int p = t; p < npx; p += BS_THREADS)
        out[p] = none ? 1 : (pick_bg ? !bin[p] : (bin[p] && parent[p] == pick_root));
    if (t == 0) { w.scal[2] = (int)ncomp; w.scal[3] = none ? -2 : pick_root; }
}

__global__ void bs_info_kernel(BsWs w, int* info, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    info += 4 * blockIdx.z;
    if (threadIdx.x < 4) info[threadIdx.x] = w.scal[threadIdx.x];
}

}  // namespace sm

extern "C" size_t sm_bilateral_workspace_bytes(int32_t H, int32_t W, double sigma_spatial, double sigma_luma,
                                               double sigma_chroma) {
    if (H <= 0 || W <= 0 || sigma_spatial < 1 || sigma_luma <= 0 || sigma_chroma <= 0) return 0;
    const sm::BsDims d = sm::make_dims(H, W, sigma_spatial, sigma_luma, sigma_chroma);
    if (d.ncells > (1ll << 31) || d.ss * d.ss > 1024) return 0;
    return sm::carve_bs(d, nullptr).total;
}

extern "C" int sm_bilateral_solver_batch_f64(const sm_bilateral_args* a, int32_t n_images, void* stream) {
    SM_REQUIRE(a && a->img && a->target && a->soft && a->binary && a->workspace, "sm_bilateral_solver_f64: null pointer");
    SM_REQUIRE(n_images >= 1 && n_images <= 65535, "sm_bilateral_solver_batch_f64: n_images=%d (1..65535)", n_images);
    SM_REQUIRE(a->H > 0 && a->W > 0 && a->sigma_spatial >= 1 && a->sigma_spatial == (int)a->sigma_spatial &&
                   a->sigma_luma > 0 && a->sigma_chroma > 0,
               "sm_bilateral_solver_f64: bad shape / sigmas (sigma_spatial must be a positive integer)");
    const size_t need1 = sm_bilateral_workspace_bytes(a->H, a->W, a->sigma_spatial, a->sigma_luma, a->sigma_chroma);
    SM_REQUIRE(need1 != 0, "sm_bilateral_solver_f64: lattice too large (>2^31 cells) or sigma_spatial > 32");
    const size_t need = need1 * (size_t)n_images;
    if (a->workspace_bytes < need || ((uintptr_t)a->workspace % 256) != 0) {
        sm::set_error("sm_bilateral_solver_f64: workspace %zu B < %zu B needed (or not 256-B aligned)", a->workspace_bytes,
                      need);
        return SM_ENOSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const sm::BsDims d = sm::make_dims(a->H, a->W, a->sigma_spatial, a->sigma_luma, a->sigma_chroma);
    const sm::BsWs w = sm::carve_bs(d, (char*)a->workspace);
    const sm::BsBatch bb = {need1};  // per-image workspaces are laid end to end (need1 is a multiple of 256)
    const unsigned nz = (unsigned)n_images;
    const int maxV = d.npx;
    // the occupancy bitmap is the first region of each image's workspace: one strided memset clears them all
    if (hipMemset2DAsync(w.bitmap, need1, 0, (size_t)d.nwords * 4, (size_t)n_images, st) != hipSuccess) {
        sm::set_error("sm_bilateral_solver_f64: hipMemset2DAsync failed");
        return SM_ELAUNCH;
    }
    const int pb = (d.npx + 255) / 256;
    hipLaunchKernelGGL(sm::bs_cells_kernel, dim3(pb, 1, nz), dim3(256), 0, st, a->img, d, w, bb);
    hipLaunchKernelGGL(sm::bs_scan_kernel, dim3(1, 1, nz), dim3(sm::BS_THREADS), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_vertices_kernel, dim3((d.nwords + 255) / 256, 1, nz), dim3(256), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_pixel_vertex_kernel, dim3(pb, 1, nz), dim3(256), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_neighbors_kernel, dim3(pb, 1, nz), dim3(256), 0, st, d, w, maxV, bb);
    const int n = d.ss * d.ss;
    const int threads = ((n + 63) / 64) * 64;
    const size_t lds = ((n * 4 + 15) & ~15) + (size_t)n * 8;
    hipLaunchKernelGGL(sm::bs_splat_kernel, dim3(d.NX, d.NY, nz), dim3(threads), lds, st, a->target, a->confidence, d, w, bb);
    hipLaunchKernelGGL(sm::bs_solve_kernel, dim3(1, 1, nz), dim3(sm::BS_THREADS), 0, st, w, maxV, a->lam, a->a_diag_min,
                       a->cg_maxiter, a->cg_tol, bb);
    hipLaunchKernelGGL(sm::bs_slice_kernel, dim3(pb, 1, nz), dim3(256), 0, st, d, w, a->soft, bb);
    hipLaunchKernelGGL(sm::bs_post_kernel, dim3(1, 1, nz), dim3(sm::BS_THREADS), 0, st, d, w, a->soft, a->binary, bb);
    if (a->info) hipLaunchKernelGGL(sm::bs_info_kernel, dim3(1, 1, nz), dim3(64), 0, st, w, a->info, bb);
    return sm::check_launch("sm_bilateral_solver_f64");
}

extern "C" int sm_bilateral_solver_f64(const sm_bilateral_args* a, void* stream) {
    return sm_bilateral_solver_batch_f64(a, 1, stream);
}
