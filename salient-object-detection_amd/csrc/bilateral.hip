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
//     column order: ONE LAUNCH PER SWEEP / PER PHASE OF AN ITERATION over (BS_SOLVE_BLOCKS workgroups x images) - the kernel
//     boundary is the grid barrier, every dot product leaves as BS_SOLVE_BLOCKS partial sums per image that the next kernel
//     adds in a fixed order (deterministic, batch-invariant).  Rounds 1-3 ran all of it in one persistent workgroup per image:
//     a batch of 32 images held 32 of 256 CUs for 7.8 ms (profiles/r04_kernel_stats_refine_384_first.csv).  A converged
//     image's workgroups fall through the remaining launches (the test is recomputed from the same partial sums).
//   * post-processing (threshold, binary_fill_holes, 4-connected label, second-largest label incl. background) is an
//     atomicMin union-find over pixel-parallel launches: row runs inside a wave by ballot (no atomics), unions only where a
//     run meets the row above for the first time or crosses a wave boundary, then a flatten pass (labels are ordered by
//     their first raster pixel, as ndimage.label).  One workgroup per image took 17 ms per batch of 32 at 384^2.
#include "common.h"
#include <type_traits>
#include <math.h>

#pragma clang fp contract(off)  // the oracle / reference use separate fp64 mul and add; fused ops are explicit

namespace sm {

constexpr int BS_SOLVE_BLOCKS = 32;   // workgroups per image in the solver's launches
constexpr int BS_SOLVE_THREADS = 256;   // (1024-lane workgroups: p / x launches 9-12 -> 19 us, profiles/r04_bilateral_steps.txt)
constexpr int BS_SCAN_WORDS = 1024;   // bitmap words per workgroup of the rank scan
constexpr int BS_NPART = 6;           // bb, rr[2], rz[2], pq  (rr / rz double-buffered by iteration parity)

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
    int* scal;           // [16] scalars: 0 = V, 1 = cg iterations, 2 = n components, 3 = chosen, 4 = foreground pixels
    unsigned long long* keys;  // [4]: 0 = largest (size << 32 | order) key, 1 = second largest
    unsigned* blocksum;  // [nscan] popcounts per 1024-word block of the bitmap
    double* part;        // [BS_NPART][BS_SOLVE_BLOCKS] partial sums of the solver's dot products
    double* coef;        // [10][maxV] off-diagonal entries of A, row k = 2*d + s like nbr
    size_t total;
};

// Batched solves: image i of a launch lives at workspace + i * ws_bytes and at pixel offset i * npx of the image /
// target / output arrays; blockIdx.z is the image.  Single-workgroup kernels then run one workgroup PER IMAGE, which
// is what fills the GPU (a lone solve occupies one CU).
struct BsBatch {
    size_t ws_bytes;
    int n_images;
};
__device__ __forceinline__ BsWs bs_image_ws(BsWs w, size_t bytes) {
    auto mv = [&](auto*& ptr) { ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<char*>(ptr) + bytes); };
    mv(w.bitmap); mv(w.wordrank); mv(w.cell); mv(w.idx); mv(w.vcell); mv(w.nbr);
    mv(w.m); mv(w.ws); mv(w.b); mv(w.n0); mv(w.n1); mv(w.diag); mv(w.minv); mv(w.x); mv(w.r); mv(w.p); mv(w.q);
    mv(w.parent); mv(w.bin); mv(w.csize); mv(w.scal); mv(w.keys); mv(w.blocksum); mv(w.part); mv(w.coef);
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
    w.blocksum = (unsigned*)take(((size_t)(d.nwords + BS_SCAN_WORDS - 1) / BS_SCAN_WORDS) * 4);
    w.part = (double*)take((size_t)BS_NPART * BS_SOLVE_BLOCKS * 8);
    w.coef = (double*)take(maxV * 8 * 10);
    w.total = off;
    return w;
}

// ---- grid construction ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bs_cells_kernel(const unsigned char* __restrict__ img, BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    img += (size_t)blockIdx.z * d.npx * 3;
    const int p = min((int)(blockIdx.x * 256 + threadIdx.x), d.npx - 1);  // the tail lanes repeat the last pixel (they take part in the shuffle)
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
    // consecutive pixels of a row mostly share their cell: the first lane of a stretch sets the bit for all of them
    const unsigned prev = __shfl_up(cell, 1, 64);
    if ((threadIdx.x & 63) == 0 || prev != cell) atomicOr(&w.bitmap[cell >> 5], 1u << (cell & 31));
}

// exclusive prefix sum of popcount(bitmap[w]) (vertex id = occupied cells before this one); scal[0] = number of vertices.
// Two launches: popcount per block of BS_SCAN_WORDS words, then every block adds the blocks before it and scans its own words.
__global__ __launch_bounds__(256) void bs_scan_partial_kernel(BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    __shared__ unsigned wsum[4];
    const int t = threadIdx.x, base = blockIdx.x * BS_SCAN_WORDS + t * 4;
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < d.nwords) s += __popc(w.bitmap[base + i]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((t & 63) == 0) wsum[t >> 6] = s;
    __syncthreads();
    if (t == 0) w.blocksum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void bs_scan_final_kernel(BsDims d, BsWs w, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    __shared__ unsigned wsum[4], wpre[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, base = blockIdx.x * BS_SCAN_WORDS + t * 4;
    unsigned before = 0;  // vertices in the blocks before this one
    for (int i = t; i < (int)blockIdx.x; i += 256) before += w.blocksum[i];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    unsigned c[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = base + i < d.nwords ? __popc(w.bitmap[base + i]) : 0u; mine += c[i]; }
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    if (lane == 0) wpre[wave] = before;
    __syncthreads();
    unsigned run = wpre[0] + wpre[1] + wpre[2] + wpre[3] + inc - mine;
    for (int k = 0; k < wave; ++k) run += wsum[k];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (base + i < d.nwords) { w.wordrank[base + i] = run; run += c[i]; }
    if (blockIdx.x == gridDim.x - 1 && t == 255) w.scal[0] = (int)run;  // the last thread's running total = all vertices
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
// The leader of a vertex = the smallest pixel index carrying it, found through a small open-addressing table in the LDS
// (atomicCAS on the key, atomicMin on the index: integer atomics, the result does not depend on their order); the sum is then one
// straight pass over the cell by the leaders only.  (Rounds 1-3: every pixel scanned the pixels before it for its vertex id and
// the leaders the ones after it, two loops of dependent LDS reads with an early exit - 0.42 ms per batch of 32 at 384^2.)
__global__ void bs_splat_kernel(const double* __restrict__ target, double conf, BsDims d, BsWs w, int table, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    target += (size_t)blockIdx.z * d.npx;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int n = d.ss * d.ss;
    int* vid = (int*)lds;
    double* tw = (double*)(lds + ((n * 4 + 15) & ~15));
    int* key = (int*)(tw + n);
    int* lead = key + table;
    const int t = threadIdx.x;
    const int ly = t / d.ss, lx = t - ly * d.ss;
    const int y = blockIdx.y * d.ss + ly, x = blockIdx.x * d.ss + lx;
    const bool inside = t < n && y < d.H && x < d.W;
    const int v = inside ? w.idx[y * d.W + x] : -1;
    if (t < n) {
        vid[t] = v;
        tw[t] = inside ? target[y * d.W + x] * conf : 0.0;
    }
    for (int i = t; i < table; i += blockDim.x) { key[i] = -1; lead[i] = 0x7fffffff; }
    __syncthreads();
    int h = 0;
    if (inside) {
        h = (int)(((unsigned)v * 2654435761u) >> 8) & (table - 1);
        for (;;) {
            const int old = atomicCAS(&key[h], -1, v);
            if (old == -1 || old == v) break;
            h = (h + 1) & (table - 1);
        }
        atomicMin(&lead[h], t);
    }
    __syncthreads();
    if (!inside || lead[h] != t) return;  // not the leader
    double cnt = 0.0, sw = 0.0, sb = 0.0;
    for (int s = t; s < n; ++s) {
        if (vid[s] == v) { cnt += 1.0; sw += conf; sb += tw[s]; }
    }
    w.m[v] = cnt; w.ws[v] = sw; w.b[v] = sb;
}

// ---- bistochastize + PCG: one launch per sweep / per phase, BS_SOLVE_BLOCKS workgroups per image -----------------------------
// (Both gathers below fetch ALL neighbour indices, then ALL neighbour values from clamped addresses, and only then add - with the
// reference's skip of an absent neighbour as a select.  Written as `if (j >= 0) t += x[j]` every neighbour was its own basic block with a
// full s_waitcnt: ten dependent memory round trips per vertex.)
__device__ __forceinline__ double bs_blur(const double* __restrict__ x, const int* __restrict__ nbr, int maxV, int v) {
    int j[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) j[k] = nbr[(size_t)k * maxV + v];
    double xv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) xv[k] = x[j[k] >= 0 ? j[k] : v];
    double out = 10.0 * x[v];  // 2 * dim * x  (:97)
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // out = out + blur_k.dot(x): row entries in ascending column order
        double t = 0.0;
        t = j[2 * k] >= 0 ? t + xv[2 * k] : t;
        t = j[2 * k + 1] >= 0 ? t + xv[2 * k + 1] : t;
        out = out + t;
    }
    return out;
}

// A.p with A = lam*(Dm - Dn blur(Dn)) + diag(S w) as scipy assembles it: ascending column order = the "-1" neighbours
// from the slowest lattice axis (v) to the fastest (x), the diagonal, then the "+1" neighbours from x to v.  The off-diagonal
// entries -(lam * (n_v * n_j)) are evaluated once (bs_pcg_setup_kernel -> coef: the same expression, the same bits as forming them
// per product) so that a mat-vec gathers p[j] only.
__device__ __forceinline__ double bs_matvec(const double* __restrict__ p, const double* __restrict__ coef,
                                            const double* __restrict__ diag, const int* __restrict__ nbr, int maxV, int v) {
    int j[10];
    double c[10], pj[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { j[k] = nbr[(size_t)k * maxV + v]; c[k] = coef[(size_t)k * maxV + v]; }
#pragma unroll
    for (int k = 0; k < 10; ++k) pj[k] = p[j[k] >= 0 ? j[k] : v];
    const double dv = diag[v] * p[v];
    double out = 0.0;
#pragma unroll
    for (int k = 4; k >= 0; --k) out = j[2 * k] >= 0 ? out + c[2 * k] * pj[2 * k] : out;
    out = out + dv;
#pragma unroll
    for (int k = 0; k < 5; ++k) out = j[2 * k + 1] >= 0 ? out + c[2 * k + 1] * pj[2 * k + 1] : out;
    return out;
}

// this workgroup's share of a dot product -> part[slot][workgroup of the image]: wave butterfly, then the waves in order
__device__ __forceinline__ void bs_store_partial(double v, double* __restrict__ part, int slot) {
    __shared__ double red[BS_SOLVE_THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();  // red may still be read by a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < BS_SOLVE_THREADS / 64; ++i) t += red[i];  // the waves in index order
        part[slot * BS_SOLVE_BLOCKS + (blockIdx.x >> 3)] = t;
    }
}
// the whole dot product: the workgroups' shares in index order (every thread of every workgroup computes the same bits)
__device__ __forceinline__ double bs_total(const double* __restrict__ part, int slot) {
    double s = 0.0;
#pragma unroll 8
    for (int i = 0; i < BS_SOLVE_BLOCKS; ++i) s += part[slot * BS_SOLVE_BLOCKS + i];
    return s;
}
constexpr int BS_P_BB = 0, BS_P_RR = 1, BS_P_RZ = 3, BS_P_PQ = 5;  // rr, rz: + (iteration & 1)

// Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: all BS_SOLVE_BLOCKS workgroups of an image
// are given ids of ONE residue mod 8, so an image's vectors (n, p, q, r, x, the neighbour table: ~2 MB at 21 k vertices) live in one
// L2 instead of eight.  The grid is (BS_SOLVE_BLOCKS * 8, ceil(images / 8)): linear id L = blockIdx.x + 256 * blockIdx.y.
#define BS_SOLVE_PROLOGUE                                                          \
    const int bs_img = (int)blockIdx.y * 8 + ((int)blockIdx.x & 7);                \
    if (bs_img >= batch.n_images) return;                                          \
    const int bs_blk = (int)blockIdx.x >> 3;                                       \
    w = bs_image_ws(w, bs_img * batch.ws_bytes);                                   \
    const int V = w.scal[0];                                                       \
    const int g0 = bs_blk * BS_SOLVE_THREADS + threadIdx.x;                        \
    constexpr int GS = BS_SOLVE_BLOCKS * BS_SOLVE_THREADS;

// bistochastize (:107-118): n <- sqrt(n*m / blur(n)) x10 from n = 1; sweep `it` reads n{it & 1} and writes the other
__global__ __launch_bounds__(BS_SOLVE_THREADS) void bs_bisto_kernel(BsWs w, int maxV, int it, BsBatch batch) {
    BS_SOLVE_PROLOGUE
    const double* na = (it & 1) ? w.n1 : w.n0;
    double* nb = (it & 1) ? w.n0 : w.n1;
    if (it < 0) {  // n = 1
        for (int v = g0; v < V; v += GS) w.n0[v] = 1.0;
        return;
    }
    for (int v = g0; v < V; v += GS) nb[v] = sqrt(na[v] * w.m[v] / bs_blur(na, w.nbr, maxV, v));
}

// after ten sweeps n lives in n0.  A diagonal, Jacobi preconditioner, flat initialisation (:132-142), |b|^2
__global__ __launch_bounds__(BS_SOLVE_THREADS) void bs_pcg_setup_kernel(BsWs w, int maxV, double lam, double diag_min, BsBatch batch) {
    BS_SOLVE_PROLOGUE
    const double* n = w.n0;
    double bb = 0.0;
    for (int v = g0; v < V; v += GS) {
        const double mv = n[v] * bs_blur(n, w.nbr, maxV, v);  // m <- n * blur(n)
        const double dg = lam * (mv - n[v] * (10.0 * n[v])) + w.ws[v];
        w.diag[v] = dg;
        w.minv[v] = 1.0 / fmax(dg, diag_min);
        w.x[v] = w.b[v] / w.ws[v];
        bb += w.b[v] * w.b[v];
        const double nv = n[v];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int j = w.nbr[(size_t)k * maxV + v];
            const double nj = n[j >= 0 ? j : v];
            w.coef[(size_t)k * maxV + v] = j >= 0 ? -(lam * (nv * nj)) : 0.0;
        }
    }
    bs_store_partial(bb, w.part, BS_P_BB);
    if (g0 == 0) w.scal[1] = 0;
}

// scipy.sparse.linalg.cg(A, b, x0, M, maxiter, rtol=tol, atol=0): r0 = b - A x0 (or x = b when b = 0), |r|^2 and r.M^-1 r
__global__ __launch_bounds__(BS_SOLVE_THREADS) void bs_pcg_r0_kernel(BsWs w, int maxV, double lam, BsBatch batch) {
    BS_SOLVE_PROLOGUE
    const double bnrm = sqrt(bs_total(w.part, BS_P_BB));
    if (bnrm == 0.0) return;  // uniform over the image's workgroups; cg returns b itself: bs_slice_kernel reads it
    double rr = 0.0, rz = 0.0;
    for (int v = g0; v < V; v += GS) {
        const double rv = w.b[v] - bs_matvec(w.x, w.coef, w.diag, w.nbr, maxV, v);
        w.r[v] = rv;
        rr += rv * rv;
        rz += rv * (w.minv[v] * rv);
    }
    bs_store_partial(rr, w.part, BS_P_RR);
    bs_store_partial(rz, w.part, BS_P_RZ);
}

// true while iteration `it` is to run: b != 0, every earlier iteration ran to its end (scal[1] = iterations completed) and
// |r| >= tol |b| at the top of this one (|r|^2 of the previous iteration sits in the slot of this iteration's parity).  Once an
// image stops, its workgroups fall through the remaining launches without touching r, x, the partial sums or the counter.
__device__ __forceinline__ bool bs_running(const BsWs& w, int it, double tol) {
    const double bnrm = sqrt(bs_total(w.part, BS_P_BB));
    if (bnrm == 0.0 || w.scal[1] < it) return false;
    return !(sqrt(bs_total(w.part, BS_P_RR + (it & 1))) < tol * bnrm);
}

// phase a of iteration `it`: p <- M^-1 r (+ beta p)
__global__ __launch_bounds__(BS_SOLVE_THREADS) void bs_pcg_p_kernel(BsWs w, int it, double tol, BsBatch batch) {
    BS_SOLVE_PROLOGUE
    if (!bs_running(w, it, tol)) return;
    if (it > 0) {
        const double beta = bs_total(w.part, BS_P_RZ + (it & 1)) / bs_total(w.part, BS_P_RZ + ((it - 1) & 1));
        for (int v = g0; v < V; v += GS) w.p[v] = w.p[v] * beta + w.minv[v] * w.r[v];
    } else {
        for (int v = g0; v < V; v += GS) w.p[v] = w.minv[v] * w.r[v];
    }
}

// phase b: q <- A p, p.q
__global__ __launch_bounds__(BS_SOLVE_THREADS) void bs_pcg_q_kernel(BsWs w, int maxV, double lam, int it, double tol, BsBatch batch) {
    BS_SOLVE_PROLOGUE
    if (!bs_running(w, it, tol)) return;
    double pq = 0.0;
    for (int v = g0; v < V; v += GS) {
        const double qv = bs_matvec(w.p, w.coef, w.diag, w.nbr, maxV, v);
        w.q[v] = qv;
        pq += w.p[v] * qv;
    }
    bs_store_partial(pq, w.part, BS_P_PQ);
}

// phase c: x <- x + alpha p, r <- r - alpha q, the next iteration's |r|^2 and r.M^-1 r
__global__ __launch_bounds__(BS_SOLVE_THREADS) void bs_pcg_x_kernel(BsWs w, int it, double tol, BsBatch batch) {
    BS_SOLVE_PROLOGUE
    if (!bs_running(w, it, tol)) return;
    const double alpha = bs_total(w.part, BS_P_RZ + (it & 1)) / bs_total(w.part, BS_P_PQ);
    double rr = 0.0, rz = 0.0;
    for (int v = g0; v < V; v += GS) {
        w.x[v] = w.x[v] + alpha * w.p[v];
        const double rv = w.r[v] - alpha * w.q[v];
        w.r[v] = rv;
        rr += rv * rv;
        rz += rv * (w.minv[v] * rv);
    }
    bs_store_partial(rr, w.part, BS_P_RR + ((it + 1) & 1));
    bs_store_partial(rz, w.part, BS_P_RZ + ((it + 1) & 1));
    if (g0 == 0) w.scal[1] = it + 1;  // the other workgroups of this launch only test scal[1] >= it
}

__global__ __launch_bounds__(256) void bs_slice_kernel(BsDims d, BsWs w, double* __restrict__ soft, BsBatch bb) {
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);
    soft += (size_t)blockIdx.z * d.npx;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.npx) return;
    const bool zero_b = bs_total(w.part, BS_P_BB) == 0.0;  // cg returns b itself when |b| = 0
    soft[p] = zero_b ? w.b[w.idx[p]] : w.x[w.idx[p]];      // S^T y (:90-91)
}

// ---- post-processing (:184-192), pixel-parallel launches --------------------------------------------------------------------
__device__ __forceinline__ int uf_find(int* parent, int i) {
    int p;
    while ((p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != i) i = p;
    return i;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
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

#define BS_POST_PROLOGUE                                     \
    w = bs_image_ws(w, blockIdx.z * bb.ws_bytes);           \
    const int p = blockIdx.x * 256 + threadIdx.x;           \
    const int npx = d.npx, W = d.W;                         \
    int* parent = w.parent;                                 \
    unsigned char* bin = w.bin;                             \
    unsigned* csize = w.csize;

__global__ __launch_bounds__(256) void bs_post_threshold_kernel(BsDims d, BsWs w, const double* __restrict__ soft, BsBatch bb) {
    BS_POST_PROLOGUE
    soft += (size_t)blockIdx.z * npx;
    if (p < npx) bin[p] = soft[p] > 0.5;
    if (p == 0) { w.keys[0] = 0; w.keys[1] = 0; w.scal[2] = 0; w.scal[4] = 0; }
}

// parent <- start of the pixel's run of equal values inside its wave's 64 consecutive pixels of the row (ballot, no atomics);
// csize <- 0.  fg_only: background pixels stay singletons.
__global__ __launch_bounds__(256) void bs_post_runs_kernel(BsDims d, BsWs w, int fg_only, BsBatch bb) {
    BS_POST_PROLOGUE
    const int lane = threadIdx.x & 63;
    const bool in = p < npx;
    const int x = in ? p % W : 0;
    const unsigned char b = in ? bin[p] : 0;
    const bool start = !in || lane == 0 || x == 0 || bin[p - 1] != b || (fg_only && !b);
    const unsigned long long starts = __ballot(start);
    if (!in) return;
    const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int first = 63 - __builtin_clzll(upto);  // bit `lane` itself is set when start
    parent[p] = p - (lane - first);
    csize[p] = 0;
}

// unions: a run that continues across a wave boundary, and a pixel under an equal pixel where its run meets that run of the row
// above for the first time (the pixels to the left are already joined through the two runs)
__global__ __launch_bounds__(256) void bs_post_union_kernel(BsDims d, BsWs w, int fg_only, BsBatch bb) {
    BS_POST_PROLOGUE
    if (p >= npx) return;
    const unsigned char b = bin[p];
    if (fg_only && !b) return;
    const int x = p % W;
    const bool left_same = x > 0 && bin[p - 1] == b;
    if (left_same && (threadIdx.x & 63) == 0) uf_union(parent, p, p - 1);
    if (p >= W && bin[p - W] == b && !(left_same && bin[p - W - 1] == b)) uf_union(parent, p, p - W);
}

// parent <- root.  count != 0 (the labelling of the filled mask): also component sizes and the foreground count, one atomic per
// stretch of equal roots inside a wave's 64 pixels - and one per workgroup for waves that lie inside a single component (a
// pixel-wise atomicAdd put all of a component's pixels on one address: 4 ms per batch of 32 at 384^2)
__global__ __launch_bounds__(256) void bs_post_flatten_kernel(BsDims d, BsWs w, int count, BsBatch bb) {
    BS_POST_PROLOGUE
    __shared__ int wroot[4];
    int root = -1;
    if (p < npx) {
        const int r = uf_find(parent, p);
        __hip_atomic_store(&parent[p], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bin[p]) root = r;
    }
    if (!count) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prev = __shfl_up(root, 1, 64);
    const bool start = lane == 0 || root != prev;
    const unsigned long long starts = __ballot(start);
    const bool whole = starts == 1ull && __builtin_amdgcn_readfirstlane(root) >= 0;  // 64 pixels of one component
    if (lane == 0) wroot[wave] = whole ? root : -1;
    if (!whole && root >= 0 && start) {
        const unsigned long long after = lane == 63 ? 0ull : starts >> (lane + 1);
        atomicAdd(&csize[root], (unsigned)(after ? __builtin_ctzll(after) + 1 : 64 - lane));
    }
    const unsigned long long m = __ballot(root >= 0);
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = wroot[a];
            if (r < 0) continue;
            unsigned len = 64;
#pragma unroll
            for (int b2 = a + 1; b2 < 4; ++b2)
                if (wroot[b2] == r) { len += 64; wroot[b2] = -1; }
            atomicAdd(&csize[r], len);
        }
    }
    if (lane == 0 && m) atomicAdd(&w.scal[4], __popcll(m));
}

// binary_fill_holes: background components that touch the image border keep their colour ...
__global__ __launch_bounds__(256) void bs_post_border_kernel(BsDims d, BsWs w, BsBatch bb) {
    BS_POST_PROLOGUE
    if (p >= npx) return;
    const int y = p / W, x = p - y * W;
    if (!bin[p] && (y == 0 || x == 0 || y == d.H - 1 || x == W - 1)) csize[parent[p]] = 1;
}
// ... the others become foreground
__global__ __launch_bounds__(256) void bs_post_fill_kernel(BsDims d, BsWs w, BsBatch bb) {
    BS_POST_PROLOGUE
    if (p < npx && !bin[p] && !csize[parent[p]]) bin[p] = 1;
}

// ndimage.label (4-connectivity) on the filled mask is the second runs / union / flatten round (which also counts the sizes)
// nb_pixel = [background, label 1, label 2, ...] (labels in raster order of their first pixel); argsort ascending, take [-2]:
// key = (size << 32 | order) with order 0 = background, root + 1 = component.  Largest key, then the largest below it.
__global__ __launch_bounds__(256) void bs_post_best_kernel(BsDims d, BsWs w, BsBatch bb) {
    BS_POST_PROLOGUE
    if (p == 0) atomicMax(&w.keys[0], (unsigned long long)(npx - w.scal[4]) << 32);
    if (p < npx && bin[p] && parent[p] == p) {
        atomicMax(&w.keys[0], ((unsigned long long)csize[p] << 32) | (unsigned)(p + 1));
        atomicAdd(&w.scal[2], 1);
    }
}
__global__ __launch_bounds__(256) void bs_post_second_kernel(BsDims d, BsWs w, BsBatch bb) {
    BS_POST_PROLOGUE
    const unsigned long long best = w.keys[0];
    if (p == 0) {
        const unsigned long long bgkey = (unsigned long long)(npx - w.scal[4]) << 32;
        if (bgkey < best) atomicMax(&w.keys[1], bgkey);
    }
    if (p < npx && bin[p] && parent[p] == p) {
        const unsigned long long k = ((unsigned long long)csize[p] << 32) | (unsigned)(p + 1);
        if (k < best) atomicMax(&w.keys[1], k);
    }
}
__global__ __launch_bounds__(256) void bs_post_out_kernel(BsDims d, BsWs w, unsigned char* __restrict__ out, BsBatch bb) {
    BS_POST_PROLOGUE
    out += (size_t)blockIdx.z * npx;
    const unsigned long long pick = w.keys[1];
    const bool none = w.scal[2] == 0;                       // IndexError branch (:191-192): all ones
    const bool pick_bg = (unsigned)(pick & 0xffffffffu) == 0;
    const int pick_root = (int)(pick & 0xffffffffu) - 1;
    if (p < npx) out[p] = none ? 1 : (pick_bg ? !bin[p] : (bin[p] && parent[p] == pick_root));
    if (p == 0) w.scal[3] = none ? -2 : pick_root;
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
    const sm::BsBatch bb = {need1, n_images};  // per-image workspaces are laid end to end (need1 is a multiple of 256)
    const unsigned nz = (unsigned)n_images;
    const int maxV = d.npx;
    // the occupancy bitmap is the first region of each image's workspace: one strided memset clears them all
    if (hipMemset2DAsync(w.bitmap, need1, 0, (size_t)d.nwords * 4, (size_t)n_images, st) != hipSuccess) {
        sm::set_error("sm_bilateral_solver_f64: hipMemset2DAsync failed");
        return SM_ELAUNCH;
    }
    const int pb = (d.npx + 255) / 256;
    const double npxd = (double)d.npx * n_images;
    int tap = sm::tap_begin(stream, "bilateral: lattice build (bs_cells .. bs_splat)", 0.0, npxd * (3 + 5 * 4 + 8 + 8 + 4 * 16 + 2 * (8 + 4)));
    hipLaunchKernelGGL(sm::bs_cells_kernel, dim3(pb, 1, nz), dim3(256), 0, st, a->img, d, w, bb);
    const int nscan = (d.nwords + sm::BS_SCAN_WORDS - 1) / sm::BS_SCAN_WORDS;
    hipLaunchKernelGGL(sm::bs_scan_partial_kernel, dim3(nscan, 1, nz), dim3(256), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_scan_final_kernel, dim3(nscan, 1, nz), dim3(256), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_vertices_kernel, dim3((d.nwords + 255) / 256, 1, nz), dim3(256), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_pixel_vertex_kernel, dim3(pb, 1, nz), dim3(256), 0, st, d, w, bb);
    hipLaunchKernelGGL(sm::bs_neighbors_kernel, dim3(pb, 1, nz), dim3(256), 0, st, d, w, maxV, bb);
    const int n = d.ss * d.ss;
    const int threads = ((n + 63) / 64) * 64;
    int table = 64;
    while (table < 2 * n) table *= 2;
    const size_t lds = ((n * 4 + 15) & ~15) + (size_t)n * 8 + (size_t)table * 8;
    hipLaunchKernelGGL(sm::bs_splat_kernel, dim3(d.NX, d.NY, nz), dim3(threads), lds, st, a->target, a->confidence, d, w, table, bb);
    sm::tap_end(tap);
    tap = sm::tap_begin(stream, "bilateral: bs_bisto_kernel x 11", 0.0, 0.0);
    {
        const dim3 sg(sm::BS_SOLVE_BLOCKS * 8, (nz + 7) / 8, 1), sb(sm::BS_SOLVE_THREADS);
        for (int it = -1; it < 10; ++it) hipLaunchKernelGGL(sm::bs_bisto_kernel, sg, sb, 0, st, w, maxV, it, bb);
        sm::tap_end(tap);
        tap = sm::tap_begin(stream, "bilateral: bs_pcg_* (setup, r0, maxiter x (p, q, x))", 0.0, 0.0);
        hipLaunchKernelGGL(sm::bs_pcg_setup_kernel, sg, sb, 0, st, w, maxV, a->lam, a->a_diag_min, bb);
        hipLaunchKernelGGL(sm::bs_pcg_r0_kernel, sg, sb, 0, st, w, maxV, a->lam, bb);
        for (int it = 0; it < a->cg_maxiter; ++it) {
            hipLaunchKernelGGL(sm::bs_pcg_p_kernel, sg, sb, 0, st, w, it, a->cg_tol, bb);
            hipLaunchKernelGGL(sm::bs_pcg_q_kernel, sg, sb, 0, st, w, maxV, a->lam, it, a->cg_tol, bb);
            hipLaunchKernelGGL(sm::bs_pcg_x_kernel, sg, sb, 0, st, w, it, a->cg_tol, bb);
        }
    }
    sm::tap_end(tap);
    tap = sm::tap_begin(stream, "bilateral: bs_slice + bs_post_*", 0.0, npxd * (8 + 8 + 1 + 2 * 4 * 6));
    hipLaunchKernelGGL(sm::bs_slice_kernel, dim3(pb, 1, nz), dim3(256), 0, st, d, w, a->soft, bb);
    {
        const dim3 pg(pb, 1, nz), pt(256);
        hipLaunchKernelGGL(sm::bs_post_threshold_kernel, pg, pt, 0, st, d, w, a->soft, bb);
        for (int fg_only = 0; fg_only < 2; ++fg_only) {
            hipLaunchKernelGGL(sm::bs_post_runs_kernel, pg, pt, 0, st, d, w, fg_only, bb);
            hipLaunchKernelGGL(sm::bs_post_union_kernel, pg, pt, 0, st, d, w, fg_only, bb);
            hipLaunchKernelGGL(sm::bs_post_flatten_kernel, pg, pt, 0, st, d, w, fg_only, bb);
            if (!fg_only) {
                hipLaunchKernelGGL(sm::bs_post_border_kernel, pg, pt, 0, st, d, w, bb);
                hipLaunchKernelGGL(sm::bs_post_fill_kernel, pg, pt, 0, st, d, w, bb);
            }
        }
        hipLaunchKernelGGL(sm::bs_post_best_kernel, pg, pt, 0, st, d, w, bb);
        hipLaunchKernelGGL(sm::bs_post_second_kernel, pg, pt, 0, st, d, w, bb);
        hipLaunchKernelGGL(sm::bs_post_out_kernel, pg, pt, 0, st, d, w, a->binary, bb);
    }
    sm::tap_end(tap);
    if (a->info) hipLaunchKernelGGL(sm::bs_info_kernel, dim3(1, 1, nz), dim3(64), 0, st, w, a->info, bb);
    return sm::check_launch("sm_bilateral_solver_f64");
}

extern "C" int sm_bilateral_solver_f64(const sm_bilateral_args* a, void* stream) {
    return sm_bilateral_solver_batch_f64(a, 1, stream);
}
