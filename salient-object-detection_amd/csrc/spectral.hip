// Spectral clustering of the up-sampled encoder features: the clusterer the shipped configuration selects for the pseudo-mask
// generator (/root/reference/configs/duts-dino-k234-nq20-224-swav-mocov2-dino-p16-sr10100.yaml:11-12 `k: [2,3,4]`,
// `clustering_mode: "spectral"`; mask_generator.pyc@L30-38,160: `self.clusterer(features, k)`; BASELINE.json configs[4] "faiss k-NN
// affinity + eigendecomp").  The reference's `clusterings` module exists in NO form in its repository: PARITY UNPINNED.  What is
// built is normalised spectral clustering as the paper names it (arXiv 2203.12614 section 3.1), in the exact form scikit-learn's
// SpectralClustering(affinity="precomputed") evaluates - the third-party witness of oracle/cluster_oracle.py:
//   1. Gram matrix G = F F^T of the n x 384 features on the matrix cores (sm_gemm_f16x2: fp32-grade products)
//   2. k-NN: for every point the n_neighbors - 1 nearest others by (|f_i - f_j|^2, j)            [knn_select_kernel]
//   3. W = (C + C^T) / 2 (C = connectivity, self loops dropped): the transposed lists by bitmap    [knn_graph_kernel]
//   4. the kw smallest eigenpairs of L = I - D^-1/2 W D^-1/2 in fp64: Chebyshev-filtered subspace
//      iteration (Zhou & Saad) on a block of 8 vectors, one workgroup per image; embedding rows
//      u_i = v_i / sqrt(d_i)                                                                        [spectral_embed_kernel]
//   5. k-means on the first k embedding columns for every requested k (ONE eigen-solve serves all)  [kmeans_embed_kernel]
// Every sum runs in a fixed order (no floating-point atomics): labels are a function of the input alone.
#include "common.h"

// no fused multiply-adds the source does not spell out: the eigen-solver's memory plans are different instantiations of the same
// expressions and must round them alike (tests: the plans give the same bits), and the oracle is numpy
#pragma clang fp contract(off)

namespace sm {

constexpr int SP_B = 8;          // block of vectors iterated (wanted kw <= 6 + guards)
constexpr int SP_THREADS = 512;  // one workgroup per image (8 waves: 256 registers per lane for the 36-entry Gram accumulations)
constexpr int SP_WAVES = SP_THREADS / 64;
constexpr int SP_MAXM = 32;      // neighbours kept per point (n_neighbors - 1)
constexpr int SP_MAXN = 8192;    // points per image

__device__ __forceinline__ double shfl_d(double v, int src) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __shfl((unsigned)u, src, 64), hi = __shfl((unsigned)(u >> 32), src, 64);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double shfl_xor_d(double v, int o) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __shfl_xor((unsigned)u, o, 64), hi = __shfl_xor((unsigned)(u >> 32), o, 64);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Sum over the 64 lanes, every lane returning the same total: lane ^ 1, ^ 2 by DPP quad permutes, then the other quad of the eight
// (row_half_mirror: the quads are uniform by then), the other half of the row (row_mirror), the neighbouring row and the other half of
// the wave by v_permlane16_swap / v_permlane32_swap of the value with itself - no trip through the LDS (six ds_bpermute round trips
// per value made a 36-value block sum 24 k cycles: 60 % of a Cholesky-QR pass).  Both operands of every add are the same two
// numbers on the lanes that exchange them, so all lanes agree bit for bit.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double sum8_d(double x) {  // over each aligned group of eight lanes
    x += dpp_d<0xB1>(x);   // quad_perm [1, 0, 3, 2]
    x += dpp_d<0x4E>(x);   // quad_perm [2, 3, 0, 1]
    x += dpp_d<0x141>(x);  // row_half_mirror
    return x;
}
__device__ __forceinline__ double wave_sum_d(double x) {
    x = sum8_d(x);
    x += dpp_d<0x140>(x);  // row_mirror
    const unsigned long long u = __double_as_longlong(x);
    {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
        x = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
    }
    const unsigned long long w = __double_as_longlong(x);
    {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)w, (unsigned)w, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(w >> 32), (unsigned)(w >> 32), false, false);
        x = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
    }
    return x;
}

// sum of NV values over the workgroup, in a fixed order (the tree of wave_sum_d inside a wave, then the waves in index order);
// every thread returns with the totals in v.  `scratch` holds (SP_WAVES + 1) * NV doubles.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
    __syncthreads();
    if (tid < NV) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SP_WAVES; ++w) s += scratch[w * NV + tid];
        scratch[SP_WAVES * NV + tid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = scratch[SP_WAVES * NV + i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// 2. k-NN selection: one wave per row of the Gram matrix.  key_j = G_jj - 2 G_ij orders the points like |f_i - f_j|^2 (G_ii is
// common to the row).  Every lane keeps the CAP smallest (key, j) of its strided share in registers (sorted, insertion by
// compare-select), then the wave merges the 64 sorted lists: CAP... m rounds of a 64-lane arg-min over the list heads.
template <int CAP>
__global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ gram_all, const float* __restrict__ sq_all, int n, int m,
                                                         int* __restrict__ idx_all) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* g = gram_all + ((int64_t)blockIdx.y * n + row) * n;
    const float* sq = sq_all + (int64_t)blockIdx.y * n;
    float key[CAP];
    int id[CAP];
#pragma unroll
    for (int p = 0; p < CAP; ++p) { key[p] = INFINITY; id[p] = 0x7fffffff; }
    for (int j = lane; j < n; j += 64) {
        if (j == row) continue;
        const float kj = sq[j] - 2.0f * g[j];
        if (kj < key[CAP - 1]) {  // j ascends within a lane: on equal keys the earlier index stays in front
#pragma unroll
            for (int p = CAP - 1; p > 0; --p) {
                const bool up = kj < key[p - 1];  // the element above moves down
                const bool here = !up && kj < key[p];
                key[p] = up ? key[p - 1] : (here ? kj : key[p]);
                id[p] = up ? id[p - 1] : (here ? j : id[p]);
            }
            if (kj < key[0]) { key[0] = kj; id[0] = j; }
        }
    }
    int* out = idx_all + ((int64_t)blockIdx.y * n + row) * m;
    for (int r = 0; r < m; ++r) {
        float bk = key[0];
        int bi = id[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ok = __shfl_xor(bk, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ok < bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
        }
        // non-finite features leave the lists empty: any valid index keeps the later kernels inside their arrays
        if (lane == 0) out[r] = (unsigned)bi < (unsigned)n ? bi : (row + 1 + r) % n;
        if (id[0] == bi) {  // the owner pops its head (indices are unique across lanes)
#pragma unroll
            for (int p = 0; p < CAP - 1; ++p) { key[p] = key[p + 1]; id[p] = id[p + 1]; }
            key[CAP - 1] = INFINITY;
            id[CAP - 1] = 0x7fffffff;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3. the graph as ONE adjacency list per point: its own m neighbours (nearest first), then the points that list it in ascending
// order - and 1 / sqrt(degree).  One workgroup per image: bit (t, i) of a n x n bitmap (zeroed by a memset on the stream) <- i
// lists t (integer atomics: the RESULT does not depend on their order), then one wave per row turns the set bits into a list.
// A pair listed from both ends appears twice, once per direction: W_ij = (C_ij + C_ji) / 2 is "every entry weighs 1/2", and
// d_t = (m + in_degree_t) / 2.
constexpr int SP_COL_SLACK = 32;  // ints readable past the last list (the gathers fetch 24 entries of a list whatever its length)
__host__ __device__ inline int64_t sp_col_capacity(int n, int m) { return (int64_t)2 * n * m + 4 * (int64_t)n + SP_COL_SLACK; }

__global__ __launch_bounds__(SP_THREADS) void knn_graph_kernel(const int* __restrict__ idx_all, int n, int m,
                                                               unsigned long long* __restrict__ bits_all, int* __restrict__ ptr_all,
                                                               int* __restrict__ len_all, int* __restrict__ col_all,
                                                               double* __restrict__ isd_all) {
    __shared__ int part[SP_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, img = blockIdx.x;
    const int nw = (n + 63) / 64;
    const int* idx = idx_all + (int64_t)img * n * m;
    unsigned long long* bits = bits_all + (int64_t)img * n * nw;
    int* ptr = ptr_all + (int64_t)img * (n + 1);
    int* len = len_all + (int64_t)img * n;
    int* col = col_all + (int64_t)img * sp_col_capacity(n, m);
    double* isd = isd_all + (int64_t)img * n;
    for (int64_t e = tid; e < (int64_t)n * m; e += SP_THREADS) {
        const int i = (int)(e / m), t = idx[e];
        atomicOr(&bits[(int64_t)t * nw + (i >> 6)], 1ull << (i & 63));
    }
    __threadfence();
    __syncthreads();
    // list length (m + in-degree, rounded up to 4 entries) of this thread's contiguous chunk of rows, exclusive scan over the workgroup
    const int per = (n + SP_THREADS - 1) / SP_THREADS, r0 = tid * per, r1 = min(n, r0 + per);
    int mine = 0;
    for (int r = r0; r < r1; ++r) {
        int c = 0;
        for (int w = 0; w < nw; ++w) c += __popcll(__hip_atomic_load(&bits[(int64_t)r * nw + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        len[r] = m + c;
        isd[r] = 1.0 / sqrt(0.5 * (double)(m + c));
        mine += (m + c + 3) & ~3;  // every list starts on a 16-B boundary: the gathers fetch four indices per load
    }
    part[tid] = mine;
    __syncthreads();
    if (wave == 0) {  // the partial sums: SP_WAVES per lane, then a 64-lane inclusive scan
        int s[SP_WAVES], tot = 0;
#pragma unroll
        for (int q = 0; q < SP_WAVES; ++q) { s[q] = tot; tot += part[lane * SP_WAVES + q]; }
        int inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        const int base = inc - tot;
#pragma unroll
        for (int q = 0; q < SP_WAVES; ++q) part[lane * SP_WAVES + q] = base + s[q];
    }
    __syncthreads();
    {
        int run = part[tid];
        for (int r = r0; r < r1; ++r) {
            ptr[r] = run;
            run += (len[r] + 3) & ~3;
        }
        if (r1 == n && r0 < n) ptr[n] = run;
    }
    __syncthreads();
    for (int r = wave; r < n; r += SP_WAVES) {
        int pos = ptr[r];
        if (lane < m) col[pos + lane] = idx[(int64_t)r * m + lane];  // m <= 32
        pos += m;
        for (int w0 = 0; w0 < nw; w0 += 64) {
            const int w = w0 + lane;
            unsigned long long word = w < nw ? __hip_atomic_load(&bits[(int64_t)r * nw + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            const int c = __popcll(word);
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            int at = pos + inc - c;
            while (word) {
                col[at++] = w * 64 + __builtin_ctzll(word);
                word &= word - 1;
            }
            pos += __shfl(inc, 63, 64);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 4. eigen-solver
struct SpGraph {
    const int* ptr;     // (n + 1) adjacency list starts, multiples of 4
    const int* len;     // (n) list lengths (m + in-degree)
    const int* col;     // the lists, each entry of weight 1/2
    const double* isd;  // 1 / sqrt(d)
    int n;
};

// One step of the Chebyshev recurrence on the scaled variables (see the kernel): Xw <- ((I - D^-1 W) Yr - c0 Yr) f1 - f2 Xw, the
// gathered block Yr staged through the LDS in groups of CG columns (n * CG doubles fit the workgroup's share of the 160 KB).
// An item = (row i, column jj): sum over the row's adjacency list (start: a multiple of 4, cnt entries) of ylds[nb * CG + jj] in list
// order.  The neighbour indices are the only global loads on this path and the loop is software-pipelined by hand around them: the
// list start / length / scale of the item after next and the first 24 indices of the next item (six 16-B loads) are in flight while
// the current item reads the LDS - all 24 reads first, ONE wait, then the sum.  (Left to the compiler every neighbour was a read ->
// s_waitcnt lgkmcnt(0) -> add round trip, and an item began with two dependent global latencies: 80 k cycles per step at n = 784,
// scripts/spectral_stamps.py.)  Entries past cnt are read but add 0; longer lists (hubs) finish in a plain loop.
template <int CG, bool LAST>
__device__ __forceinline__ void sp_filter_step(const SpGraph& g, const double* __restrict__ Yr, double* __restrict__ Xw, double* ylds,
                                               double c0, double f1, double f2) {
    const int total = g.n * CG, zero_row = g.n;  // ylds holds n + 1 rows: the last one is zeros, the target of a list's padding
    struct Meta { int s, cnt; double w; };
    auto meta = [&](int t) {
        Meta m = {0, 0, 1.0};
        if (t < total) { const int i = t / CG; m.s = g.ptr[i]; m.cnt = g.len[i]; m.w = g.isd[i]; }
        return m;
    };
    struct Idx { int4 v[6]; };
    auto indices = [&](const Meta& m) {
        Idx x;
        const int4* p = reinterpret_cast<const int4*>(g.col + m.s);
#pragma unroll
        for (int k = 0; k < 6; ++k) x.v[k] = p[k];  // (an item past the end reads the first list: valid memory, never used)
        return x;
    };
#pragma unroll 1
    for (int c = 0; c < SP_B; c += CG) {
        for (int t = threadIdx.x; t < total; t += SP_THREADS) ylds[t] = Yr[(t / CG) * SP_B + c + t % CG];
        if (threadIdx.x < CG) ylds[total + threadIdx.x] = 0.0;
        int t = threadIdx.x;
        Meta m_cur = meta(t), m_next = meta(t + SP_THREADS);
        Idx x_cur = indices(m_cur);
        __syncthreads();
#pragma unroll 1
        for (; t < total; t += SP_THREADS) {
            const Idx x_next = indices(m_next);                 // next item's indices: its list start arrived an iteration ago
            const Meta m_after = meta(t + 2 * SP_THREADS);      // and the list start of the one after
            const int i = t / CG, jj = t % CG, cnt = m_cur.cnt;
            int nb[24];
#pragma unroll
            for (int k = 0; k < 6; ++k) { nb[4 * k] = x_cur.v[k].x; nb[4 * k + 1] = x_cur.v[k].y; nb[4 * k + 2] = x_cur.v[k].z; nb[4 * k + 3] = x_cur.v[k].w; }
            double y[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) y[u] = ylds[(u < cnt ? nb[u] : zero_row) * CG + jj];  // past the list: + 0.0 (exact)
            const double yo = ylds[t], xo = Xw[i * SP_B + c + jj];
            asm volatile("" ::: "memory");  // every load above is issued before the first add below
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 24; ++u) acc += y[u];
            for (int e = 24; e < cnt; ++e) acc += ylds[g.col[m_cur.s + e] * CG + jj];
            const double w = m_cur.w;
            const double ly = yo - (0.5 * w * w) * acc;
            const double xn = (ly - c0 * yo) * f1 - f2 * xo;
            Xw[i * SP_B + c + jj] = LAST ? xn / w : xn;  // back to the symmetric variables on the way out (last step only)
            m_cur = m_next;
            m_next = m_after;
            x_cur = x_next;
        }
        __syncthreads();
    }
}

// And with BOTH blocks of the recurrence in the LDS beside the graph (MODE 2; n = 784, the 28 x 28 x 4 points of a 224^2 image, with
// 10 neighbours: 144 KB of the 150): the whole solve runs without touching memory - a filter step gathers from yl, rewrites xl in
// place (each lane its own elements), and the two exchange roles.  With two waves per SIMD nothing hid the memory round trips of the
// staged forms (SQ counters: 69 % of the wave cycles waiting, 3 % of them on the LDS; ~50 k cycles per step,
// profiles/r04_spectral_sq_counters.txt).  Here the graph is a CSR of 16-bit BYTE offsets of the neighbours' rows, lists padded to a
// multiple of four with the zero row, rows SORTED by list length (longest first): an item = a row and two columns (one 16-B read per
// neighbour), a wave walks its 16 rows in chunks of four neighbours up to the length of its first (longest) row - k-NN graphs have a
// long tail (mean 13, maximum 60-70 entries at n = 784), and both the padding of every list to 24 and the serial walk of the hubs'
// tails through memory (33 k cycles for one 71-entry row: the step's critical path) are gone.  Every row's sum still runs in list
// order (the padding adds + 0.0): the results are bit-identical to the other modes' (tests/test_hip_spectral.py, through the tuning build).
// MODE 3 (826 <= n <= ~2100: the 38 x 50 = 1900 points of a 300 x 400 image) keeps the same graph in the LDS beside CG = 4 or 2 COLUMNS
// of the two blocks: the recurrence never mixes columns, so a filter runs all its steps on one column group after the other, each
// resident like MODE 2 (the block itself stays in memory and is read and written once per group and filter); the lists take what the
// launch's 150 KB leave, and a graph that does not fit (decided on the device, where the list lengths are) falls back to MODE 0's steps.
struct SpResGraph {
    const unsigned short* ent;   // the lists: byte offsets (row * bytes per row), each list a multiple of 4 entries
    const unsigned short* ptr4;  // (n + 1) list start / 4, by sorted position (lengths descending): list p has ptr4[p + 1] - ptr4[p] chunks
    const unsigned short* row;   // (n) the row at a sorted position
    const double* isd;           // (n + 1) 1 / sqrt(d) by row; isd[n] = 0
    unsigned zoff;               // byte offset of the zero row: n * 64
};

// layout: [block A (n + 1) x W | block B (n + 1) x W | isd n + 1 (MODE 2 only) | ptr4 n + 1 | row n | lists (16-B aligned, to the end)],
// W doubles per row.  (The lists hold m + in-degree entries per row - a mutual pair twice, every entry weighing 1/2 - 2 n m in all.)
__host__ __device__ constexpr size_t sp_resident_entries(int n, int m) { return (((size_t)(2 * m + 3) * n) + 7) & ~(size_t)7; }  // sum of cnt <= 2 n m
__host__ __device__ constexpr size_t sp_resident_isd_offset(int n, int w) { return (size_t)2 * (n + 1) * w * 8; }
__host__ __device__ constexpr size_t sp_resident_ent_offset(int n, int w) {  // MODE 3 (w < 8) reads 1 / sqrt(d) from memory, once per item
    return (sp_resident_isd_offset(n, w) + (w == SP_B ? (size_t)(n + 1) * 8 : 0) + (size_t)(2 * n + 1) * 2 + 15) & ~(size_t)15;
}
__host__ __device__ constexpr size_t sp_resident_bytes(int n, int m) {  // MODE 2: room for the longest lists there can be
    return (sp_resident_ent_offset(n, SP_B) + sp_resident_entries(n, m) * 2 + 15) & ~(size_t)15;
}

// the gathers of one chunk: four neighbours' (two-column) values
struct SpChunk { double2 v[4]; };
template <int TC>
__device__ __forceinline__ double2 sp_load_cols(const char* p) {  // TC = 2: two columns (16 B); TC = 1: one, the other reads as 0
    if (TC == 2) return *reinterpret_cast<const double2*>(p);
    double2 v;
    v.x = *reinterpret_cast<const double*>(p);
    v.y = 0.0;
    return v;
}
template <int TC>
__device__ __forceinline__ void sp_store_cols(char* p, double2 v) {
    if (TC == 2) *reinterpret_cast<double2*>(p) = v;
    else *reinterpret_cast<double*>(p) = v.x;
}
template <int TC>
__device__ __forceinline__ SpChunk sp_gather_chunk(const char* col, uint2 q) {
    SpChunk c;
    c.v[0] = sp_load_cols<TC>(col + (q.x & 0xffffu));
    c.v[1] = sp_load_cols<TC>(col + (q.x >> 16));
    c.v[2] = sp_load_cols<TC>(col + (q.y & 0xffffu));
    c.v[3] = sp_load_cols<TC>(col + (q.y >> 16));
    return c;
}

// (Two items per lane at a time - two independent chains of LDS round trips per wave - changed nothing: 17.0 k against 16.7 k cycles
// per step; the step is not bound by one chain's latency.)
template <int W, bool LAST>
__device__ __forceinline__ void sp_filter_step_resident(const SpResGraph& rg, int n, const double* yl, double* xl, double c0, double f1,
                                                        double f2) {
    constexpr int TC = W >= 2 ? 2 : 1, PER_ROW = W / TC;
    const int total = n * PER_ROW;
    const uint2 zz = make_uint2(rg.zoff | (rg.zoff << 16), rg.zoff | (rg.zoff << 16));
#pragma unroll 1
    for (int t = threadIdx.x; t < total; t += SP_THREADS) {
        const int pos = t / PER_ROW, jj = (t % PER_ROW) * TC;
        const int i = rg.row[pos], my = (int)rg.ptr4[pos + 1] - (int)rg.ptr4[pos];
        const uint2* e = reinterpret_cast<const uint2*>(rg.ent + (int)rg.ptr4[pos] * 4);
        const int nch = __builtin_amdgcn_readfirstlane(my);  // the wave's first row is its longest
        const char* ycol = reinterpret_cast<const char*>(yl + jj);
        const double2 yo = sp_load_cols<TC>(reinterpret_cast<const char*>(yl + i * W + jj));
        const double2 xo = sp_load_cols<TC>(reinterpret_cast<const char*>(xl + i * W + jj));
        const double w = rg.isd[i];
        double a0 = 0.0, a1 = 0.0;
        SpChunk cur = sp_gather_chunk<TC>(ycol, 0 < my ? e[0] : zz);
        uint2 qn = 1 < my ? e[1] : zz;
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            SpChunk nxt = cur;
            if (c + 1 < nch) nxt = sp_gather_chunk<TC>(ycol, qn);       // the next chunk's gathers and the index pair after it are in flight
            const uint2 qnn = c + 2 < my ? e[c + 2] : zz;           // while this chunk is summed
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += cur.v[u].x; a1 += cur.v[u].y; }
            cur = nxt;
            qn = qnn;
        }
        const double h = 0.5 * w * w;
        const double l0 = yo.x - h * a0, l1 = yo.y - h * a1;
        double2 xn;
        xn.x = (l0 - c0 * yo.x) * f1 - f2 * xo.x;
        xn.y = (l1 - c0 * yo.y) * f1 - f2 * xo.y;
        if (LAST) { xn.x = xn.x / w; xn.y = xn.y / w; }  // back to the symmetric variables on the way out (last step only)
        sp_store_cols<TC>(reinterpret_cast<char*>(xl + i * W + jj), xn);
    }
    __syncthreads();
}

// the same with every term scaled by isd[nb] (the symmetric form: two loads per neighbour; once per outer iteration)
__device__ __forceinline__ double sp_gather2(const int* __restrict__ col, int s, int t, const double* __restrict__ isd,
                                             const double* __restrict__ Y, int j) {
    double acc = 0.0;
    int e = s;
    for (; e + 4 <= t; e += 4) {
        int nb[4];
        double y[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) nb[u] = col[e + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) { y[u] = Y[(int64_t)nb[u] * SP_B + j]; w[u] = isd[nb[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += w[u] * y[u];
    }
    for (; e < t; ++e) acc += isd[col[e]] * Y[(int64_t)col[e] * SP_B + j];
    return acc;
}

// Out = L In (L = I - D^-1/2 W D^-1/2), block of SP_B columns
__device__ __forceinline__ void sp_apply_sym(const SpGraph& g, const double* __restrict__ In, double* __restrict__ Out) {
#pragma unroll 2
    for (int t = threadIdx.x; t < g.n * SP_B; t += SP_THREADS) {
        const int i = t / SP_B, j = t % SP_B;
        Out[t] = In[t] - 0.5 * g.isd[i] * sp_gather2(g.col, g.ptr[i], g.ptr[i] + g.len[i], g.isd, In, j);
    }
}

// Out = L In with both blocks and the graph in the LDS (MODE 2): the same products isd[nb] * In[nb] summed in list order (the padding
// adds isd[n] * In[n] = 0 * 0)
template <int W>
__device__ __forceinline__ void sp_apply_sym_resident(const SpResGraph& rg, int n, const double* in, double* out) {
    constexpr int TC = W >= 2 ? 2 : 1, PER_ROW = W / TC, ISD_SHIFT = W == 8 ? 3 : W == 4 ? 2 : 1;  // row byte offset -> isd byte offset
    const int total = n * PER_ROW;
    const uint2 zz = make_uint2(rg.zoff | (rg.zoff << 16), rg.zoff | (rg.zoff << 16));
    const char* isdb = reinterpret_cast<const char*>(rg.isd);
#pragma unroll 1
    for (int t = threadIdx.x; t < total; t += SP_THREADS) {
        const int pos = t / PER_ROW, jj = (t % PER_ROW) * TC;
        const int i = rg.row[pos], my = (int)rg.ptr4[pos + 1] - (int)rg.ptr4[pos];
        const uint2* e = reinterpret_cast<const uint2*>(rg.ent + (int)rg.ptr4[pos] * 4);
        const int nch = __builtin_amdgcn_readfirstlane(my);
        const char* col = reinterpret_cast<const char*>(in + jj);
        const double2 self = *reinterpret_cast<const double2*>(in + i * W + jj);
        const double wi = rg.isd[i];
        double a0 = 0.0, a1 = 0.0;
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const uint2 q = c < my ? e[c] : zz;
            const unsigned off[4] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16};  // W * 8 B per row, 8 per isd entry
            double2 y[4];
            double w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                y[u] = *reinterpret_cast<const double2*>(col + off[u]);
                w[u] = *reinterpret_cast<const double*>(isdb + (off[u] >> ISD_SHIFT));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += w[u] * y[u].x; a1 += w[u] * y[u].y; }
        }
        double2 o;
        o.x = self.x - 0.5 * wi * a0;
        o.y = self.y - 0.5 * wi * a1;
        *reinterpret_cast<double2*>(out + i * W + jj) = o;
    }
}

// The same for one group of W columns (MODE 3): `xs` holds isd[row] * In[row][c .. c + W) in the LDS (the product every neighbour
// contributes, rounded as sp_gather2 rounds it), In itself and 1 / sqrt(d) of the row come from memory once per item.
template <int W>
__device__ __forceinline__ void sp_apply_sym_cols(const SpResGraph& rg, int n, const double* xs, const double* __restrict__ In,
                                                  double* __restrict__ Out, int c) {
    constexpr int TC = W >= 2 ? 2 : 1, PER_ROW = W / TC;
    const int total = n * PER_ROW;
    const uint2 zz = make_uint2(rg.zoff | (rg.zoff << 16), rg.zoff | (rg.zoff << 16));
#pragma unroll 1
    for (int t = threadIdx.x; t < total; t += SP_THREADS) {
        const int pos = t / PER_ROW, jj = (t % PER_ROW) * TC;
        const int i = rg.row[pos], my = (int)rg.ptr4[pos + 1] - (int)rg.ptr4[pos];
        const uint2* e = reinterpret_cast<const uint2*>(rg.ent + (int)rg.ptr4[pos] * 4);
        const int nch = __builtin_amdgcn_readfirstlane(my);
        const char* col = reinterpret_cast<const char*>(xs + jj);
        const double2 self = sp_load_cols<TC>(reinterpret_cast<const char*>(In + (int64_t)i * SP_B + c + jj));
        const double wi = rg.isd[i];
        double a0 = 0.0, a1 = 0.0;
        SpChunk cur = sp_gather_chunk<TC>(col, 0 < my ? e[0] : zz);
        uint2 qn = 1 < my ? e[1] : zz;
#pragma unroll 1
        for (int k = 0; k < nch; ++k) {
            SpChunk nxt = cur;
            if (k + 1 < nch) nxt = sp_gather_chunk<TC>(col, qn);
            const uint2 qnn = k + 2 < my ? e[k + 2] : zz;
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += cur.v[u].x; a1 += cur.v[u].y; }
            cur = nxt;
            qn = qnn;
        }
        double2 o;
        o.x = self.x - 0.5 * wi * a0;
        o.y = self.y - 0.5 * wi * a1;
        sp_store_cols<TC>(reinterpret_cast<char*>(Out + (int64_t)i * SP_B + c + jj), o);
    }
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

#ifdef SM_SPECTRAL_STAMPS  // experiment build: finer marks inside the phases, thread 0 (SP_MARK(k): cycles since the previous mark -> dbg[k])
#define SP_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); sh.dbg[k] += now_ - sh.dbg_t; sh.dbg_t = now_; } } while (0)
#else
#define SP_MARK(k)
#endif

struct SpShared {
#ifdef SM_SPECTRAL_STAMPS
    unsigned long long dbg[16], dbg_t;
#endif
    double red[(SP_WAVES + 1) * 36];
    double z[SP_B][SP_B];   // R^-1 or the Ritz rotation
    double th[SP_B];
    double res[SP_B];
};

constexpr int sp_tri(int a, int b) { return a * SP_B - a * (a - 1) / 2 + (b - a); }  // index of (a, b), a <= b, in the packed upper triangle

// Cyclic Jacobi on the symmetric SP_B x SP_B matrix `sym` (packed upper triangle, the same totals in every thread's registers),
// run by the first wave with the matrix in REGISTERS: lane k < SP_B owns row k of the matrix and row k of the rotation product;
// a rotation (p, q) is a column update local to every lane, a row update fed by v_readlane broadcasts of rows p and q, and the
// angle from three broadcast entries (p, q are compile-time constants: the 28 rotations of a sweep are unrolled).  Eigenvalues
// ascending to sh.th, eigenvectors to the columns of sh.z.  (A single thread walking the same matrices in LDS spent a millisecond per
// call on dependent LDS round trips - three quarters of the eigen-solver's time.)
__device__ __forceinline__ void jacobi_eig_wave(const double (&sym)[36], SpShared& sh) {
    // lane k (of every group of eight: the groups repeat each other) owns row k.  A sweep = the 7 rounds of a round-robin schedule,
    // each rotating FOUR disjoint pairs at once: the lanes of a pair work out their rotation side by side (one chain of divisions and
    // square roots per round instead of four - the row-cyclic order paid 28 per sweep, ~1 k cycles each: 105-125 k cycles per call),
    // the column halves of the four rotations are register arithmetic with v_readlane broadcasts of (c, s), the row halves one lane
    // exchange per column.  Rotations on disjoint pairs commute, so this is cyclic Jacobi in another order.
    const int lane = threadIdx.x & 63, k = lane & 7, grp = lane & ~7;
    double a[SP_B], z[SP_B];
#pragma unroll
    for (int j = 0; j < SP_B; ++j) {
        double v = 0.0;
#pragma unroll
        for (int kk = 0; kk < SP_B; ++kk) v = k == kk ? sym[kk <= j ? sp_tri(kk, j) : sp_tri(j, kk)] : v;
        a[j] = v;
        z[j] = k == j ? 1.0 : 0.0;
    }
    double prev_off = 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int j = 0; j < SP_B; ++j) {
            const double sq = a[j] * a[j];
            if (k == j) dia += sq; else off += sq;
        }
        off = sum8_d(off);
        dia = sum8_d(dia);
        off = readlane_d(off, 0);
        dia = readlane_d(dia, 0);
        if (off <= 1e-32 * dia) break;
        // at the rounding floor (off-diagonal norm below 1e-14 of the diagonal's) a sweep that no longer quarters it is the last one
        // that helps: some matrices sit just above 1e-32 and would sweep on to the limit of 30
        if (sweep > 0 && off <= 1e-28 * dia && off >= 0.25 * prev_off) break;
        prev_off = off;
#pragma unroll
        for (int r = 0; r < SP_B - 1; ++r) {
            const int pk = k == 7 ? r : (k == r ? 7 : (2 * r - k + 7) % 7);  // this row's partner in round r
            double dk = a[0], apk = a[0];
#pragma unroll
            for (int j = 1; j < SP_B; ++j) { dk = k == j ? a[j] : dk; apk = pk == j ? a[j] : apk; }
            const double dpar = shfl_d(dk, grp | pk);
            const bool pside = k < pk;
            const double app = pside ? dk : dpar, aqq = pside ? dpar : dk;
            double c = 1.0, sn = 0.0;
            if (apk != 0.0) {
                const double tau = (aqq - app) / (2.0 * apk);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                c = 1.0 / sqrt(1.0 + t * t);
                sn = t * c;
            }
            const int psrc = grp | (pside ? k : pk);  // both lanes of a pair use the rotation its lower row worked out (from A[p][q])
            c = shfl_d(c, psrc);
            sn = shfl_d(sn, psrc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // the column halves
                const int u = i == 0 ? r : (r + i) % 7, v = i == 0 ? 7 : (r - i + 7) % 7;
                const int p = u < v ? u : v, q = u < v ? v : u;
                const double ci = readlane_d(c, p), si = readlane_d(sn, p);
                const double akp = a[p], akq = a[q];
                a[p] = ci * akp - si * akq;
                a[q] = si * akp + ci * akq;
                const double zkp = z[p], zkq = z[q];
                z[p] = ci * zkp - si * zkq;
                z[q] = si * zkp + ci * zkq;
            }
#pragma unroll
            for (int j = 0; j < SP_B; ++j) {  // the row halves
                const double par = shfl_d(a[j], grp | pk);
                a[j] = pside ? c * a[j] - sn * par : sn * par + c * a[j];
            }
        }
    }
    double mine = 0.0;  // the eigenvalue this lane's diagonal entry became
#pragma unroll
    for (int j = 0; j < SP_B; ++j) mine = lane == j ? a[j] : mine;
    int rank = 0;  // ascending order, ties by index
#pragma unroll
    for (int kk = 0; kk < SP_B; ++kk) {
        const double other = readlane_d(mine, kk);
        rank += (other < mine || (other == mine && kk < lane)) ? 1 : 0;
    }
    if (lane < SP_B) sh.th[rank] = mine;
#pragma unroll
    for (int j = 0; j < SP_B; ++j) {
        const int rj = __builtin_amdgcn_readlane(rank, j);
        if (lane < SP_B) sh.z[lane][rj] = z[j];
    }
}

// The 8 x 8 Gram matrices of the block on the fp64 matrix cores: acc[tri(a, b)] = sum over rows of P[r][a] Q[r][b] (SYM: the symmetric
// part, (P[r][a] Q[r][b] + P[r][b] Q[r][a]) / 2), every thread returning with the 36 totals.  v_mfma_f64_16x16x4 takes four rows per
// issue: lane l feeds A[l % 16][l / 16] and B[l / 16][l % 16] - for the Gram of a block both are the block's entry (row l / 16, column
// l % 16; columns 8..15 read as 0) - and returns D[l / 16 + 4 r][l % 16] in its four accumulators r.  A wave walks every eighth group
// of four rows, the eight waves' upper triangles meet in the LDS and are added in wave order.  (Per-lane 36-value accumulators over the
// rows and a 36-value workgroup butterfly had cost 17 k cycles of a 29 k-cycle Cholesky-QR pass.)
typedef double sp_d4 __attribute__((ext_vector_type(4)));
template <bool SYM>
__device__ __forceinline__ void gram_upper_mfma(const double* P, const double* Q, int n, SpShared& sh, double (&acc)[36]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kk = lane >> 4;
    sp_d4 c1 = {0.0, 0.0, 0.0, 0.0}, c2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int g = wave; g < n / 4; g += SP_WAVES) {  // n % 4 == 0
        const int at = (4 * g + kk) * SP_B + col;
        const double p = col < SP_B ? P[at] : 0.0;
        const double q = SYM ? (col < SP_B ? Q[at] : 0.0) : p;
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(p, q, c1, 0, 0, 0);          // D[i][j] += P[r][i] Q[r][j]
        if (SYM) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(q, p, c2, 0, 0, 0);  // D[i][j] += Q[r][i] P[r][j]
    }
    __syncthreads();  // sh.red may still be read from a previous call
    if (col < SP_B) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {  // accumulator r of lane l is D[l / 16 + 4 r][l % 16] (the f64 form's own map): rows 0..7 are r = 0, 1
            const int i = kk + 4 * r;
            if (i <= col) sh.red[wave * 36 + sp_tri(i, col)] = SYM ? 0.5 * (c1[r] + c2[r]) : c1[r];
        }
    }
    __syncthreads();
    if (tid < 36) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SP_WAVES; ++w) s += sh.red[w * 36 + tid];
        sh.red[SP_WAVES * 36 + tid] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = sh.red[SP_WAVES * 36 + i];
}

// X <- X R^-1 with R^T R = X^T X + shift * trace * I (Cholesky QR); row-local after one workgroup reduction
__device__ __forceinline__ void chol_qr_pass(double* __restrict__ X, int n, double shift, SpShared& sh, int* guard) {
    const int tid = threadIdx.x;
    double acc[36];
    gram_upper_mfma<false>(X, X, n, sh, acc);
    SP_MARK(1);
    if (tid < 64) {
        // The factor and its inverse, one COLUMN per lane (lane b = column b; lanes 8.. repeat them): step a of the factorisation needs
        // column a's finished entries - v_readlane broadcasts - and every entry is the same expression, evaluated in the same order, as
        // in a serial walk (which had cost ~30 k of a pass's 42 k cycles on one lane: 64 dependent fp64 divisions and square roots).
        const int b = tid & 7;
        double tr = 0.0;
#pragma unroll
        for (int a = 0; a < SP_B; ++a) tr += acc[sp_tri(a, a)];
        double r[SP_B];  // r[a] = G[a][b], then R[a][b] (a <= b)
#pragma unroll
        for (int a = 0; a < SP_B; ++a) {
            double v = 0.0;
#pragma unroll
            for (int bb = a; bb < SP_B; ++bb) v = b == bb ? acc[sp_tri(a, bb)] : v;
            r[a] = a == b ? v + shift * tr : v;
        }
        int bad = 0;
#pragma unroll
        for (int a = 0; a < SP_B; ++a) {
            double sacc = r[a];
#pragma unroll
            for (int k = 0; k < a; ++k) sacc -= readlane_d(r[k], a) * r[k];  // R[k][a] R[k][b]
            double d = readlane_d(sacc, a);
            if (!(d > 1e-300 * tr)) { d = fmax(1e-30 * tr, 1e-300); bad = 1; }
            const double raa = sqrt(d);
            r[a] = b == a ? raa : sacc / raa;
        }
        if (bad && tid == 0) *guard = 1;
        double iv[SP_B];  // iv[a] = (R^-1)[a][b], zero below the diagonal
        {
            double rbb = r[0];
#pragma unroll
            for (int bb = 1; bb < SP_B; ++bb) rbb = b == bb ? r[bb] : rbb;
            const double inv_bb = 1.0 / rbb;
#pragma unroll
            for (int a = 0; a < SP_B; ++a) iv[a] = a == b ? inv_bb : 0.0;
        }
#pragma unroll
        for (int a = SP_B - 2; a >= 0; --a) {
            double sacc = 0.0;
#pragma unroll
            for (int k = a + 1; k < SP_B; ++k) sacc += readlane_d(r[a], k) * iv[k];  // R[a][k] (R^-1)[k][b]; k > b adds 0
            const double v = -sacc / readlane_d(r[a], a);
            iv[a] = a < b ? v : iv[a];
        }
        if (tid < SP_B)
#pragma unroll
            for (int a = 0; a < SP_B; ++a) sh.z[a][b] = iv[a];
    }
    SP_MARK(2);
    __syncthreads();
#pragma unroll 1
    for (int r = tid; r < n; r += SP_THREADS) {
        double x[SP_B], y[SP_B];
#pragma unroll
        for (int j = 0; j < SP_B; ++j) x[j] = X[(int64_t)r * SP_B + j];
#pragma unroll
        for (int b = 0; b < SP_B; ++b) {
            double sacc = 0.0;
#pragma unroll
            for (int a = 0; a <= b; ++a) sacc += x[a] * sh.z[a][b];
            y[b] = sacc;
        }
#pragma unroll
        for (int j = 0; j < SP_B; ++j) X[(int64_t)r * SP_B + j] = y[j];
    }
    __syncthreads();
    SP_MARK(3);
}

// Rayleigh-Ritz on the orthonormal block Q with LQ = L Q: H = Q^T LQ, H = Z Theta Z^T, Q <- Q Z, LQ <- LQ Z, residual norms
__device__ __forceinline__ void rayleigh_ritz(double* __restrict__ Q, double* __restrict__ LQ, int n, SpShared& sh) {
    const int tid = threadIdx.x;
    double acc[36];
    gram_upper_mfma<true>(Q, LQ, n, sh, acc);  // the symmetric part of Q^T (L Q)
    SP_MARK(5);
    if (tid < 64) jacobi_eig_wave(acc, sh);
    SP_MARK(6);
    __syncthreads();
    double rs[SP_B];
#pragma unroll
    for (int j = 0; j < SP_B; ++j) rs[j] = 0.0;
#pragma unroll 1
    for (int r = tid; r < n; r += SP_THREADS) {
        double q[SP_B], l[SP_B], qz[SP_B], lz[SP_B];
#pragma unroll
        for (int j = 0; j < SP_B; ++j) { q[j] = Q[(int64_t)r * SP_B + j]; l[j] = LQ[(int64_t)r * SP_B + j]; }
#pragma unroll
        for (int b = 0; b < SP_B; ++b) {
            double sacc = 0.0, u = 0.0;
#pragma unroll
            for (int a = 0; a < SP_B; ++a) { sacc += q[a] * sh.z[a][b]; u += l[a] * sh.z[a][b]; }
            qz[b] = sacc;
            lz[b] = u;
        }
#pragma unroll
        for (int j = 0; j < SP_B; ++j) {
            Q[(int64_t)r * SP_B + j] = qz[j];
            LQ[(int64_t)r * SP_B + j] = lz[j];
            const double e = lz[j] - sh.th[j] * qz[j];
            rs[j] += e * e;
        }
    }
    block_sum<SP_B>(rs, sh.red);
    if (tid == 0)
#pragma unroll
        for (int j = 0; j < SP_B; ++j) sh.res[j] = sqrt(rs[j]);
    __syncthreads();
    SP_MARK(7);
}

__device__ __forceinline__ double sp_init_value(int i, int j) {  // splitmix64 of (i, j): uniform in (-1, 1)
    unsigned long long x = ((unsigned long long)i * SP_B + j) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (double)(long long)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// MODE 0: graph in memory, CG columns of the block staged per step; 2: graph and both blocks in the LDS for the whole solve (CG = 8);
// 3: graph and CG columns of both blocks in the LDS for a whole filter, column group after column group
template <int CG, int MODE>
__global__ __launch_bounds__(SP_THREADS) void spectral_embed_kernel(const int* __restrict__ ptr_all, const int* __restrict__ len_all,
                                                                    const int* __restrict__ col_all,
                                                                    const double* __restrict__ isd_all, int n, int m, int kw, int degree,
                                                                    int max_outer, double tol, double* __restrict__ blocks_all,
                                                                    double* __restrict__ eig_all, double* __restrict__ emb_all,
                                                                    double* __restrict__ res_all, int* __restrict__ info_all, unsigned lds_bytes) {
    __shared__ SpShared sh;
    extern __shared__ __attribute__((aligned(16))) double ylds[];  // (n + 1) * CG doubles: the gathered block of a filter step + a zero row
    const int tid = threadIdx.x, img = blockIdx.x;
    SpGraph g;
    g.ptr = ptr_all + (int64_t)img * (n + 1);
    g.len = len_all + (int64_t)img * n;
    g.col = col_all + (int64_t)img * sp_col_capacity(n, m);
    g.isd = isd_all + (int64_t)img * n;
    g.n = n;
    // the Ritz vectors U and L U: in memory, or - MODE 2 - in the LDS for the whole solve (each with a zero row at index n)
    double* U = MODE == 2 ? ylds : blocks_all + (int64_t)img * 2 * n * SP_B;
    double* V = MODE == 2 ? ylds + (size_t)(n + 1) * SP_B : U + (int64_t)n * SP_B;
    int guard = 0, matvecs = 0, outer = 0, converged = 0;
#ifdef SM_SPECTRAL_STAMPS
    if (tid == 0) { for (int k = 0; k < 16; ++k) sh.dbg[k] = 0; sh.dbg_t = __builtin_readcyclecounter(); }
#endif
    SpResGraph rg = {nullptr, nullptr, nullptr, nullptr, 0u};
    constexpr int RW = MODE == 2 ? SP_B : CG;  // doubles per row of the LDS blocks
    bool lds_graph = false;
    double* xa = ylds;                            // MODE 3: the two column-group blocks
    double* xb = ylds + (size_t)(n + 1) * RW;
    if (MODE >= 2) {
        char* base = reinterpret_cast<char*>(ylds);
        double* isdl = reinterpret_cast<double*>(base + sp_resident_isd_offset(n, RW));  // MODE 2 only
        unsigned short* ptr4 = reinterpret_cast<unsigned short*>(isdl + (MODE == 2 ? n + 1 : 0));
        unsigned short* rowl = ptr4 + (n + 1);
        unsigned short* ent = reinterpret_cast<unsigned short*>(base + sp_resident_ent_offset(n, RW));
        const int cap4 = (int)((lds_bytes - sp_resident_ent_offset(n, RW)) / 8);  // room for the lists, in groups of four entries
        if (MODE == 2)
            for (int t = tid; t < n; t += SP_THREADS) isdl[t] = g.isd[t];
        // the rows in the order (length descending, index ascending): a bitonic sort of the keys (0xffff - length) << 16 | row, padded
        // to a power of two (the padding sorts last).  (Ranking every row against all others, 784 x 784 comparisons, had cost 200 k
        // cycles: 6 % of a solve.)
        unsigned* keys = reinterpret_cast<unsigned*>(ylds);  // scratch: the blocks (16 RW (n + 1) bytes >= 4 np) are filled after this
        int np = 2;
        while (np < n) np <<= 1;
        for (int t = tid; t < np; t += SP_THREADS) keys[t] = t < n ? ((0xffffu - (unsigned)g.len[t]) << 16) | (unsigned)t : 0xffffffffu;
        __syncthreads();
#pragma unroll 1
        for (int kk = 2; kk <= np; kk <<= 1)
#pragma unroll 1
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int pr = tid; pr < np / 2; pr += SP_THREADS) {
                    const int lo = ((pr & ~(j - 1)) << 1) | (pr & (j - 1)), hi = lo | j;
                    const unsigned x = keys[lo], y = keys[hi];
                    const bool up = (lo & kk) == 0;
                    if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
                }
                __syncthreads();
            }
        // list starts: an exclusive scan of the padded lengths in that order (a run of positions per lane, wave scan, then the waves)
        {
            const int per = np / SP_THREADS > 2 ? np / SP_THREADS : 2, p0 = tid * per;
            int mine = 0;
            for (int u = 0; u < per; ++u) mine += p0 + u < n ? (int)(0xffffu - (keys[p0 + u] >> 16) + 3) >> 2 : 0;
            int inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o, 64);
                if ((tid & 63) >= o) inc += v;
            }
            int* wtot = reinterpret_cast<int*>(sh.red);
            __syncthreads();
            if ((tid & 63) == 63) wtot[tid >> 6] = inc;
            __syncthreads();
            int run = inc - mine, total4 = 0;
            for (int w = 0; w < SP_WAVES; ++w) { run += w < (tid >> 6) ? wtot[w] : 0; total4 += wtot[w]; }
            lds_graph = total4 <= cap4 && total4 <= 0xffff;  // (MODE 2: always - the host sized the launch for the longest lists)
#ifdef SM_SPECTRAL_STAMPS
            if (tid == 0) { sh.dbg[11] = (unsigned long long)total4; sh.dbg[12] = (unsigned long long)cap4; }
#endif
            if (lds_graph)
                for (int u = 0; u < per; ++u)
                    if (p0 + u < n) {
                        const unsigned key = keys[p0 + u];
                        const int len = (int)(0xffffu - (key >> 16));
                        rowl[p0 + u] = (unsigned short)(key & 0xffffu); ptr4[p0 + u] = (unsigned short)run;
                        run += (len + 3) >> 2;
                        if (p0 + u == n - 1) ptr4[n] = (unsigned short)run;
                    }
        }
        __syncthreads();
        if (lds_graph) {
            for (int t = tid; t < n * 4; t += SP_THREADS) {  // four lanes per list
                const int pos = t >> 2, i = rowl[pos], cnt = g.len[i], c4 = (cnt + 3) & ~3, s0 = g.ptr[i];
                unsigned short* e = ent + (int)ptr4[pos] * 4;
                for (int u = t & 3; u < c4; u += 4) e[u] = (unsigned short)((u < cnt ? g.col[s0 + u] : n) * (RW * 8));
            }
            if (MODE == 2 && tid == 0) isdl[n] = 0.0;  // what a list's padding points at (with the blocks' zero rows)
            rg.ent = ent; rg.ptr4 = ptr4; rg.row = rowl; rg.isd = MODE == 2 ? isdl : g.isd; rg.zoff = (unsigned)n * (RW * 8);
        }
        __syncthreads();
        if (tid < RW) { xa[n * RW + tid] = 0.0; xb[n * RW + tid] = 0.0; }
    }
    SP_MARK(8);  // graph build
#ifdef SM_SPECTRAL_STAMPS  // experiment build: shader-clock cycles per phase instead of the residuals (scripts/spectral_stamps.py)
    unsigned long long t_filter = 0, t_chol = 0, t_apply = 0, t_rr = 0, t0 = 0;
#define SP_T0 t0 = __builtin_readcyclecounter()
#define SP_T(acc) acc += __builtin_readcyclecounter() - t0
#else
#define SP_T0
#define SP_T(acc)
#endif

    for (int t = tid; t < n * SP_B; t += SP_THREADS) {
        const int i = t / SP_B, j = t % SP_B;
        U[t] = j == 0 ? 1.0 / g.isd[i] : sp_init_value(i, j);  // column 0: sqrt(d), the eigenvector of eigenvalue 0
    }
    __syncthreads();
    for (outer = 0;; ++outer) {
        // U holds the block to work on (the start block, then the filtered one): orthonormalise, V = L U, Rayleigh-Ritz
        SP_T0;
        SP_MARK(9);  // (everything between the marked phases)
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) chol_qr_pass(U, n, pass == 0 ? 1e-11 : 0.0, sh, &guard);  // shifted Cholesky QR, three passes
        SP_T(t_chol);
        SP_T0;
        if (MODE == 2) sp_apply_sym_resident<SP_B>(rg, n, U, V);
        else if (MODE == 3 && lds_graph) {
#pragma unroll 1
            for (int c = 0; c < SP_B; c += CG) {  // a column group at a time through the LDS
                for (int t = tid; t < n * CG; t += SP_THREADS) xa[t] = g.isd[t / CG] * U[(t / CG) * SP_B + c + t % CG];
                __syncthreads();
                sp_apply_sym_cols<CG>(rg, n, xa, U, V, c);
                __syncthreads();
            }
        } else sp_apply_sym(g, U, V);
        ++matvecs;
        __syncthreads();
        SP_T(t_apply);
        SP_T0;
        SP_MARK(9);
        rayleigh_ritz(U, V, n, sh);
        SP_T(t_rr);
        double worst = 0.0;
        for (int j = 0; j < kw; ++j) worst = fmax(worst, sh.res[j]);
        if (worst <= tol) { converged = 1; break; }
        if (outer >= max_outer) break;
        // Chebyshev filter of degree `degree` damping [a, 2] (a = the block's largest Ritz value), scaled so that the eigenvalue 0
        // keeps the gain 1 (Zhou & Saad 2007, "Chebyshev-filtered subspace iteration", scaled filter): X_0 = U, X_1 = (L U - c U) s/e,
        // X_{i+1} = (L X_i - c X_i) 2 s'/e - s s' X_{i-1}, written over X_{i-1} in place (it is read at (row, column) itself only).
        // The recurrence runs on X^ = D^-1/2 X: D^-1/2 L D^1/2 = I - D^-1 W, whose mat-vec gathers ONE value per neighbour - from the
        // LDS, where the gathered block is staged once per step (sp_filter_step).
        const double a = fmin(fmax(sh.th[SP_B - 1], 1e-8), 1.9), ub = 2.0;
        const double e = 0.5 * (ub - a), c0 = 0.5 * (ub + a);
        double sig = e / (0.0 - c0);
        const double tau = 2.0 / sig;
        __syncthreads();  // everyone has read sh.res / sh.th
        SP_T0;
        if (MODE == 2) {
            double* xl = U;  // X_0 = D^-1/2 U, in place
            double* yl = V;  // X_1
            const double f = sig / e;
            for (int t = tid; t < n * SP_B; t += SP_THREADS) {
                const double w = rg.isd[t / SP_B], x = xl[t] * w;
                xl[t] = x;
                yl[t] = (yl[t] * w - c0 * x) * f;
            }
            __syncthreads();
            for (int it = 2; it <= degree; ++it) {
                const double sn = 1.0 / (tau - sig);
                const double f1 = 2.0 * sn / e, f2 = sig * sn;
                if (it == degree) sp_filter_step_resident<SP_B, true>(rg, n, yl, xl, c0, f1, f2);
                else sp_filter_step_resident<SP_B, false>(rg, n, yl, xl, c0, f1, f2);
                ++matvecs;
                double* sw = xl;
                xl = yl;
                yl = sw;
                sig = sn;
            }
            U = yl;  // the filtered block
            V = xl;
        } else if (MODE == 3 && lds_graph) {
            const double f = sig / e, sig1 = sig;
#pragma unroll 1
            for (int c = 0; c < SP_B; c += CG) {
                double* xl = xa;  // X_0 = D^-1/2 U, these columns
                double* yl = xb;  // X_1
                for (int t = tid; t < n * CG; t += SP_THREADS) {
                    const int at = (t / CG) * SP_B + c + t % CG;
                    const double w = rg.isd[t / CG], x = U[at] * w;
                    xl[t] = x;
                    yl[t] = (V[at] * w - c0 * x) * f;
                }
                __syncthreads();
                sig = sig1;
                for (int it = 2; it <= degree; ++it) {
                    const double sn = 1.0 / (tau - sig);
                    const double f1 = 2.0 * sn / e, f2 = sig * sn;
                    if (it == degree) sp_filter_step_resident<CG, true>(rg, n, yl, xl, c0, f1, f2);
                    else sp_filter_step_resident<CG, false>(rg, n, yl, xl, c0, f1, f2);
                    double* sw = xl;
                    xl = yl;
                    yl = sw;
                    sig = sn;
                }
                for (int t = tid; t < n * CG; t += SP_THREADS) U[(t / CG) * SP_B + c + t % CG] = yl[t];  // the filtered columns, in place
                __syncthreads();
            }
            matvecs += degree - 1;
        } else {
            double* X = U;
            double* Y = V;
            {
                const double f = sig / e;
                for (int t = tid; t < n * SP_B; t += SP_THREADS) {
                    const double w = g.isd[t / SP_B], x = X[t] * w;
                    X[t] = x;
                    Y[t] = (Y[t] * w - c0 * x) * f;
                }
            }
            __syncthreads();
            for (int it = 2; it <= degree; ++it) {
                const double sn = 1.0 / (tau - sig);
                const double f1 = 2.0 * sn / e, f2 = sig * sn;
                if (it == degree) sp_filter_step<CG, true>(g, Y, X, ylds, c0, f1, f2);
                else sp_filter_step<CG, false>(g, Y, X, ylds, c0, f1, f2);
                ++matvecs;
                double* sw = X;
                X = Y;
                Y = sw;
                sig = sn;
            }
            U = Y;  // the filtered block
            V = X;
        }
        SP_T(t_filter);
    }
    // results: eigenvalues ascending, embedding rows v_i / sqrt(d_i), first kw columns
    double* emb = emb_all + (int64_t)img * n * kw;
    for (int t = tid; t < n * kw; t += SP_THREADS) {
        const int i = t / kw, j = t % kw;
        emb[t] = U[(int64_t)i * SP_B + j] * g.isd[i];
    }
    if (tid < kw) {
        if (eig_all) eig_all[(int64_t)img * kw + tid] = sh.th[tid];
        if (res_all) res_all[(int64_t)img * kw + tid] = sh.res[tid];
    }
#ifdef SM_SPECTRAL_STAMPS
    __syncthreads();
    SP_MARK(10);  // results out
    if (tid == 0 && res_all && kw >= 4) {
        double* r = res_all + (int64_t)img * kw;
        r[0] = (double)t_filter; r[1] = (double)t_chol; r[2] = (double)t_apply; r[3] = (double)t_rr;
        for (int k = 0; k < 16; ++k) emb[k] = (double)sh.dbg[k];  // over the first embedding values
    }
#endif
    if (tid == 0 && info_all) {
        int* info = info_all + (int64_t)img * 4;
        info[0] = outer;
        info[1] = matvecs;
        info[2] = converged;
#ifdef SM_SPECTRAL_STAMPS
        info[3] = guard | (lds_graph ? 0 : 2) | (MODE << 4) | (CG << 8);
#else
        info[3] = guard;
#endif
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 5. k-means on the first K columns of the embedding, fp64, one workgroup per (cluster size, image): farthest-point initial
// centres (first the point farthest from the mean; ties to the lowest index), Lloyd until no label changes (at most max_iter
// updates), an emptied cluster keeps its centre - oracle/cluster_oracle.py kmeans_embedding.

template <int K>
__device__ void kmeans_embed_body(const double* __restrict__ emb, int n, int kw, int max_iter, int* __restrict__ labels, double* red,
                                  double (*cen)[SP_B], double* cand_v, int* cand_i) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the embedding (n x kw doubles, L2-resident) is re-read per pass; the previous label of a point lives in `labels`
    {
        double s[K];
#pragma unroll
        for (int d = 0; d < K; ++d) s[d] = 0.0;
#pragma unroll 1
        for (int p = tid; p < n; p += SP_THREADS) {
            labels[p] = -1;
#pragma unroll
            for (int d = 0; d < K; ++d) s[d] += emb[(int64_t)p * kw + d];
        }
        block_sum<K>(s, red);
        __syncthreads();
        if (tid == 0)
            for (int d = 0; d < K; ++d) cen[SP_B - 1][d] = s[d] / (double)n;  // scratch slot: K <= 6
        __syncthreads();
    }
    for (int j = 0; j < K; ++j) {  // farthest point from the mean, then from the nearest centre chosen so far
        double bv = -1.0;
        int bi = 0;  // a valid index even when every distance is NaN
#pragma unroll 1
        for (int p = tid; p < n; p += SP_THREADS) {
            double x[K];
#pragma unroll
            for (int d = 0; d < K; ++d) x[d] = emb[(int64_t)p * kw + d];
            double dist = INFINITY;
            if (j == 0) {
                dist = 0.0;
#pragma unroll
                for (int d = 0; d < K; ++d) { const double t = x[d] - cen[SP_B - 1][d]; dist += t * t; }
            } else {
                for (int c = 0; c < j; ++c) {
                    double dc = 0.0;
#pragma unroll
                    for (int d = 0; d < K; ++d) { const double t = x[d] - cen[c][d]; dc += t * t; }
                    dist = fmin(dist, dc);
                }
            }
            if (dist > bv) { bv = dist; bi = p; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = shfl_xor_d(bv, o);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { cand_v[wave] = bv; cand_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double v = cand_v[0];
            int i = cand_i[0];
            for (int w = 1; w < SP_WAVES; ++w)
                if (cand_v[w] > v || (cand_v[w] == v && cand_i[w] < i)) { v = cand_v[w]; i = cand_i[w]; }
            i = min(max(i, 0), n - 1);
            for (int d = 0; d < K; ++d) cen[j][d] = emb[(int64_t)i * kw + d];
        }
        __syncthreads();
    }
    for (int it = 0; it <= max_iter; ++it) {
        constexpr int NV = K * K + K + 1;
        double acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.0;
#pragma unroll 1
        for (int p = tid; p < n; p += SP_THREADS) {
            double x[K];
#pragma unroll
            for (int d = 0; d < K; ++d) x[d] = emb[(int64_t)p * kw + d];
            double bv = INFINITY;
            int bc = 0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
                double dist = 0.0;
#pragma unroll
                for (int d = 0; d < K; ++d) { const double t = x[d] - cen[c][d]; dist += t * t; }
                if (dist < bv) { bv = dist; bc = c; }
            }
            if (bc != labels[p]) { acc[NV - 1] += 1.0; labels[p] = bc; }
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const double w = bc == c ? 1.0 : 0.0;
                acc[K * K + c] += w;
#pragma unroll
                for (int d = 0; d < K; ++d) acc[c * K + d] += w * x[d];
            }
        }
        block_sum<NV>(acc, red);
        __syncthreads();
        if (acc[NV - 1] == 0.0 || it == max_iter) break;  // uniform: every thread holds the same totals
        if (tid == 0)
            for (int c = 0; c < K; ++c)
                if (acc[K * K + c] > 0.0)
                    for (int d = 0; d < K; ++d) cen[c][d] = acc[c * K + d] / acc[K * K + c];
        __syncthreads();
    }
}

struct KeSizes { int k[8]; };

__global__ __launch_bounds__(SP_THREADS) void kmeans_embed_kernel(const double* __restrict__ emb_all, int n, int kw, KeSizes sizes, int n_sizes,
                                                                  int max_iter, int* __restrict__ labels_all) {
    __shared__ double red[(SP_WAVES + 1) * 43];
    __shared__ double cen[SP_B][SP_B];
    __shared__ double cand_v[SP_WAVES];
    __shared__ int cand_i[SP_WAVES];
    const int s = blockIdx.x, img = blockIdx.y;
    const double* emb = emb_all + (int64_t)img * n * kw;
    int* labels = labels_all + ((int64_t)img * n_sizes + s) * n;
    switch (sizes.k[s]) {
        case 1:
            for (int p = threadIdx.x; p < n; p += SP_THREADS) labels[p] = 0;
            break;
        case 2: kmeans_embed_body<2>(emb, n, kw, max_iter, labels, red, cen, cand_v, cand_i); break;
        case 3: kmeans_embed_body<3>(emb, n, kw, max_iter, labels, red, cen, cand_v, cand_i); break;
        case 4: kmeans_embed_body<4>(emb, n, kw, max_iter, labels, red, cen, cand_v, cand_i); break;
        case 5: kmeans_embed_body<5>(emb, n, kw, max_iter, labels, red, cen, cand_v, cand_i); break;
        default: kmeans_embed_body<6>(emb, n, kw, max_iter, labels, red, cen, cand_v, cand_i); break;
    }
}

// squared norms = the diagonal of the Gram matrix, compact (the selection reads them once per row of the matrix)
__global__ __launch_bounds__(256) void gram_diag_kernel(const float* __restrict__ gram, int n, int64_t total, float* __restrict__ sq) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < total) sq[t] = gram[(t / n) * (int64_t)n * n + (t % n) * (int64_t)(n + 1)];
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SpLayout {
    size_t fs, gram, sq, idx, bits, inptr, inlen, incol, isd, blocks, emb, total;
};
static SpLayout sp_layout(int B, int n, int m, int kw) {
    SpLayout l;
    size_t o = 0;
    const size_t nw = (n + 63) / 64;
    l.fs = o;     o += al256((size_t)B * n * SM_EMBED * 4);
    l.gram = o;   o += al256((size_t)B * n * n * 4);
    l.sq = o;     o += al256((size_t)B * n * 4);
    l.idx = o;    o += al256((size_t)B * n * m * 4);
    l.bits = o;   o += al256((size_t)B * n * nw * 8);
    l.inptr = o;  o += al256((size_t)B * (n + 1) * 4);
    l.inlen = o;  o += al256((size_t)B * n * 4);
    l.incol = o;  o += al256((size_t)B * sp_col_capacity(n, m) * 4);
    l.isd = o;    o += al256((size_t)B * n * 8);
    l.blocks = o; o += al256((size_t)B * 2 * n * SP_B * 8);
    l.emb = o;    o += al256((size_t)B * n * kw * 8);
    l.total = o;
    return l;
}

}  // namespace sm

extern "C" size_t sm_spectral_workspace_bytes(int32_t B, int32_t n, int32_t n_neighbors, int32_t kw) {
    if (B < 1 || n < 2 * sm::SP_B || n > sm::SP_MAXN || n_neighbors < 2 || n_neighbors - 1 > sm::SP_MAXM || kw < 1 || kw > 6) return 0;
    const int m = n_neighbors - 1 < n - 1 ? n_neighbors - 1 : n - 1;
    return sm::sp_layout(B, n, m, kw).total;
}

extern "C" int sm_spectral_cluster_f32(const sm_spectral_args* a, void* stream) {
    SM_REQUIRE(a && a->features && a->labels && a->cluster_sizes && a->workspace, "sm_spectral_cluster_f32: null pointer");
    SM_REQUIRE(a->B >= 1 && a->n >= 2 * sm::SP_B && a->n <= sm::SP_MAXN && a->n % 4 == 0,
               "sm_spectral_cluster_f32: %d points (16 <= n <= %d, n %% 4 == 0)", a->n, sm::SP_MAXN);
    SM_REQUIRE(a->n_neighbors >= 2 && a->n_neighbors - 1 <= sm::SP_MAXM, "sm_spectral_cluster_f32: n_neighbors %d (2..%d)", a->n_neighbors,
               sm::SP_MAXM + 1);
    SM_REQUIRE(a->n_sizes >= 1 && a->n_sizes <= 8, "sm_spectral_cluster_f32: 1..8 cluster sizes");
    int kw = 0;
    sm::KeSizes sizes = {};
    for (int i = 0; i < a->n_sizes; ++i) {
        SM_REQUIRE(a->cluster_sizes[i] >= 1 && a->cluster_sizes[i] <= 6, "sm_spectral_cluster_f32: cluster size %d (1..6)", a->cluster_sizes[i]);
        sizes.k[i] = a->cluster_sizes[i];
        if (sizes.k[i] > kw) kw = sizes.k[i];
    }
    const int B = a->B, n = a->n, m = a->n_neighbors - 1 < n - 1 ? a->n_neighbors - 1 : n - 1;
    const sm::SpLayout l = sm::sp_layout(B, n, m, kw);
    SM_REQUIRE(a->workspace_bytes >= l.total && ((uintptr_t)a->workspace % 256 == 0),
               "sm_spectral_cluster_f32: workspace of %zu bytes, %zu needed (256-B aligned)", a->workspace_bytes, l.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)a->workspace;
    float* fs = (float*)(ws + l.fs);
    float* gram = (float*)(ws + l.gram);
    int* idx = a->knn ? a->knn : (int*)(ws + l.idx);
    double* emb = a->embedding ? a->embedding : (double*)(ws + l.emb);
    const double nd = (double)n, Bd = (double)B;
    int rc;
    {
        sm::TapGuard tap(stream, "spectral: split_f16x2 + Gram gemm_f16x2", 2.0 * Bd * nd * nd * SM_EMBED, Bd * (nd * SM_EMBED * 8 + nd * nd * 4));
        rc = sm_split_f16x2(a->features, SM_EMBED, fs, SM_EMBED, (int64_t)B * n, SM_EMBED, stream);
        if (rc) return rc;
    sm_gemm_args g = {};
    g.A = fs;
    g.W = fs;
    g.C = gram;
    g.strideA = g.strideW = (int64_t)n * SM_EMBED;
    g.strideC = (int64_t)n * n;
    g.M = g.N = n;
    g.K = SM_EMBED;
    g.lda = g.ldw = SM_EMBED;
    g.ldc = n;
    g.batch = B;
    g.epilogue = SM_EPI_BIAS;
        rc = sm_gemm_f16x2(&g, 0, stream);
        if (rc) return rc;
    }
    float* sq = (float*)(ws + l.sq);
    int tap = sm::tap_begin(stream, "spectral: knn_select + knn_graph", 0.0, Bd * (nd * nd * 4 + 3.0 * nd * m * 4));
    hipLaunchKernelGGL(sm::gram_diag_kernel, dim3((unsigned)(((int64_t)B * n + 255) / 256)), dim3(256), 0, st, gram, n, (int64_t)B * n, sq);
    if (m <= 16)
        hipLaunchKernelGGL(sm::knn_select_kernel<16>, dim3((n + 3) / 4, B), dim3(256), 0, st, gram, sq, n, m, idx);
    else
        hipLaunchKernelGGL(sm::knn_select_kernel<32>, dim3((n + 3) / 4, B), dim3(256), 0, st, gram, sq, n, m, idx);
    if (hipMemsetAsync(ws + l.bits, 0, (size_t)B * n * ((n + 63) / 64) * 8, st) != hipSuccess) {
        sm::set_error("sm_spectral_cluster_f32: hipMemsetAsync failed");
        return SM_ELAUNCH;
    }
    hipLaunchKernelGGL(sm::knn_graph_kernel, dim3(B), dim3(sm::SP_THREADS), 0, st, idx, n, m, (unsigned long long*)(ws + l.bits),
                       (int*)(ws + l.inptr), (int*)(ws + l.inlen), (int*)(ws + l.incol), (double*)(ws + l.isd));
    sm::tap_end(tap);
    const int degree = a->degree > 1 ? a->degree : 24, max_outer = a->max_outer > 0 ? a->max_outer : 60;
    const double tol = a->tol > 0.0 ? a->tol : 1e-9;
    {
        // 150 KB of LDS beside the kernel's static 3.2 KB (160 KB per workgroup on gfx950).  MODE 2 when graph (longest possible lists)
        // and both whole blocks fit; else MODE 3 with the most columns (4, 2 or 1) that leave room for the 2 n m list entries plus one
        // entry of padding per row (lists are padded to fours: the kernel sees the real lengths and falls back to MODE 0's steps when
        // they do not fit); else MODE 0 with as many staged columns as fit.
        constexpr size_t LDS_MAX = 159744;  // 156 KB dynamic + the static 3.2 KB (3.4 in the stamps build) of 160
        int cg = 0, mode = 0;
        if (sm::sp_resident_bytes(n, m) <= LDS_MAX && (size_t)n * 64 <= 65535) { cg = 8; mode = 2; }  // (16-bit byte offsets of the rows)
        for (int c : {4, 2, 1})
            if (!cg && sm::sp_resident_ent_offset(n, c) + (size_t)(2 * m + 1) * n * 2 <= LDS_MAX && (size_t)n * c * 8 <= 65535) { cg = c; mode = 3; }
#ifdef SM_TUNING  // the tuning build can force the graph-in-memory plan (tests: the plans give the same bits)
        if (const char* e = getenv("SM_SPECTRAL_PLAN"))
            if (atoi(e) == 0) { cg = 0; mode = 0; }
#endif
        if (!cg) cg = (size_t)(n + 1) * 8 * 8 <= 153600 ? 8 : (size_t)(n + 1) * 4 * 8 <= 153600 ? 4 : 2;
        const size_t lds = mode == 2 ? sm::sp_resident_bytes(n, m) : mode == 3 ? LDS_MAX : (size_t)(n + 1) * cg * 8;
        auto launch = [&](auto kern) {
            static bool once = false;  // per instantiation (the lambda is instantiated per kernel type)
            if (!once) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
                once = true;
            }
            hipLaunchKernelGGL(kern, dim3(B), dim3(sm::SP_THREADS), lds, st, (const int*)(ws + l.inptr), (const int*)(ws + l.inlen),
                               (const int*)(ws + l.incol),
                               (const double*)(ws + l.isd), n, m, kw, degree, max_outer, tol, (double*)(ws + l.blocks), a->eigenvalues, emb,
                               a->residuals, a->info, (unsigned)lds);
        };
        // (bytes: what ONE block mat-vec moves - the block read and written + the adjacency lists; the count of mat-vecs is data-dependent)
        static const char* const names[4][3] = {{"spectral_embed_kernel<8, 0>", "spectral_embed_kernel<4, 0>", "spectral_embed_kernel<2, 0>"},
                                                {"", "", ""},
                                                {"spectral_embed_kernel<8, 2>", "", ""},
                                                {"spectral_embed_kernel<1, 3>", "spectral_embed_kernel<4, 3>", "spectral_embed_kernel<2, 3>"}};
        sm::TapGuard tap2(stream, names[mode][cg == 8 || cg == 1 ? 0 : cg == 4 ? 1 : 2], 0.0, Bd * (2.0 * nd * 8 * 8 + 2.0 * nd * m * 4));
        if (mode == 2) launch(&sm::spectral_embed_kernel<8, 2>);
        else if (mode == 3) {
            if (cg == 4) launch(&sm::spectral_embed_kernel<4, 3>);
            else if (cg == 2) launch(&sm::spectral_embed_kernel<2, 3>);
            else launch(&sm::spectral_embed_kernel<1, 3>);
        } else {
            if (cg == 8) launch(&sm::spectral_embed_kernel<8, 0>);
            else if (cg == 4) launch(&sm::spectral_embed_kernel<4, 0>);
            else launch(&sm::spectral_embed_kernel<2, 0>);
        }
    }
    sm::TapGuard tap3(stream, "kmeans_embed_kernel", 0.0, Bd * nd * kw * 8 * a->n_sizes);
    hipLaunchKernelGGL(sm::kmeans_embed_kernel, dim3(a->n_sizes, B), dim3(sm::SP_THREADS), 0, st, emb, n, kw, sizes, a->n_sizes,
                       a->kmeans_max_iter > 0 ? a->kmeans_max_iter : 100, a->labels);
    return sm::check_launch("sm_spectral_cluster_f32");
}
