// MaskFormer.forward (maskformer.py:164-251; return_intermediate=True, use_binary_classifier=True) as one
// stream-ordered sequence of the kernels of this library.  No allocation, no synchronisation: the caller owns the
// workspace and the stream, so the whole forward can be captured into a hipGraph.
//
// Two GEMM back ends behind sm_weights.gemm_mode:
//   0  exact-fp32 MFMA (gemm.hip): every buffer fp32;
//   1  split-operand f16 MFMA (gemm_f16x2.hip, fp32-grade, ~2.2x faster): every tensor that only feeds a GEMM is
//      produced directly in the F16X2 format by its producer (LayerNorm, attention, GELU/ReLU epilogues, im2col,
//      up-sample); tensors that are also residuals / outputs exist in fp32 as well.  "(S)" marks them below.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace sm {

// ---- timing taps (sm_forward_timing): event pairs around the heavy launches, on the forward's stream -----------------
struct Tap {
    hipStream_t st;
    hipEvent_t e0, e1;
    std::string name;
    double flops, bytes;
};
static bool g_timing = false;
static std::vector<Tap> g_taps;

struct TapScope {
    hipStream_t st;
    bool on;
    TapScope(hipStream_t st_, const std::string& name, double flops, double bytes) : st(st_), on(g_timing) {
        if (!on) return;
        Tap t;
        t.st = st; t.name = name; t.flops = flops; t.bytes = bytes;
        if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(t.e0, st);
        g_taps.push_back(t);
    }
    ~TapScope() {
        if (on) (void)hipEventRecord(g_taps.back().e1, st);
    }
};

// the same taps for the other launch sequences of the library (spectral.hip, bilateral.hip): begin returns a handle (-1 when timing is off)
int tap_begin(void* stream, const char* name, double flops, double bytes) {
    if (!g_timing) return -1;
    Tap t;
    t.st = (hipStream_t)stream; t.name = name; t.flops = flops; t.bytes = bytes;
    if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return -1;
    (void)hipEventRecord(t.e0, t.st);
    g_taps.push_back(t);
    return (int)g_taps.size() - 1;
}
void tap_end(int handle) {
    if (handle >= 0 && handle < (int)g_taps.size()) (void)hipEventRecord(g_taps[handle].e1, g_taps[handle].st);
}

// Forwards this small (batch 1-2 at 224^2: serving) run the encoder's fc2 - K = 1536 on M/64 x 6 workgroups, a 48-step serial K
// loop of ~20 us - split four ways along K; the LayerNorm launch that follows sums the slices (as the decoder's does).  Only on
// the automatic path (sm_forward_io.attn_path = 0): the Evaluator pins a path so that rows do not depend on the batch size.
static const int64_t SM_SPLIT_FC2_ROWS = 512;

struct Shape {
    int B, H, W, P, gh, gw, n, N, L, nq, sf;  // sf: the pixel decoder's scale_factor (2 as shipped)
    int64_t M, Mp, Md, Mo;  // tokens, patch tokens, decoder rows, objectness rows
};

static Shape make_shape(const sm_weights* w, int B, int H, int W) {
    Shape s;
    s.B = B; s.H = H; s.W = W; s.P = w->patch;
    s.gh = (H + s.P - 1) / s.P; s.gw = (W + s.P - 1) / s.P;
    s.n = s.gh * s.gw; s.N = s.n + 1; s.L = w->n_dec_layers; s.nq = w->n_queries;
    s.sf = w->scale_factor > 0 ? w->scale_factor : 2;
    s.M = (int64_t)B * s.N; s.Mp = (int64_t)B * s.n; s.Md = (int64_t)B * s.nq; s.Mo = s.Md * s.L;
    return s;
}

// workspace carve-up (floats, every region 256-B aligned)
struct Ws {
    float *pos, *X, *Xn, *QKV, *AO, *HID, *TOK, *TOKs, *KV, *UP, *TGT, *TGTs, *TGTQ, *T2, *QK, *Qc, *AOd, *HIDd, *PART, *QD,
        *QDs, *LOG, *O1, *O2, *ST;
    size_t total;
};

static Ws carve(const Shape& s, float* base) {
    Ws w;
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = base ? base + off : nullptr;
        off += (nfloat + 63) & ~(size_t)63;
        return p;
    };
    const size_t D = SM_EMBED;
    w.pos = take((size_t)s.N * D);
    w.X = take(s.M * D);
    w.Xn = take(s.M * D);               // (S)
    w.QKV = take(s.M * 3 * D);
    w.AO = take(s.M * D);               // (S)
    // HID (S) doubles as the im2col buffer (S) (consumed by the patch GEMM before fc1 first writes HID)
    size_t hid = s.M * SM_MLP, cols = (size_t)s.Mp * 3 * s.P * s.P;
    w.HID = take(hid > cols ? hid : cols);
    w.TOK = take(s.Mp * D);
    w.TOKs = take(s.Mp * D);            // F16X2 copy of TOK (A operand of the all-layer K/V GEMM)
    w.KV = take(s.Mp * 2 * D * s.L);    // cross-attention K|V of ALL decoder layers: (B*n, L*768)
    w.UP = take(s.Mp * s.sf * s.sf * D);  // (S)
    w.TGT = take(s.Md * D);
    w.TGTs = take(s.Md * D);            // F16X2 copy of TGT
    w.TGTQ = take(s.Md * D);            // (S) tgt + query_pos
    w.T2 = take(s.Md * D);
    w.QK = take(s.Md * 3 * D);
    w.Qc = take(s.Md * D);
    w.AOd = take(s.Md * D);             // (S)
    w.HIDd = take(s.Md * SM_MLP);       // (S)
    // split-K partials: the decoder's linear2, and the encoder's fc2 on forwards of at most SM_SPLIT_FC2_ROWS token rows
    w.PART = take((s.M <= SM_SPLIT_FC2_ROWS && s.M > s.Md ? s.M : s.Md) * D * 4);
    w.QD = take(s.Mo * D);
    w.QDs = take(s.Mo * D);             // F16X2 copy of QD
    w.LOG = take(s.Mo * s.sf * s.sf * s.n);
    w.O1 = take(s.Mo * D);              // (S)
    w.O2 = take(s.Mo * D);
    w.ST = take(s.M * 24);              // folded LayerNorm: (mean, M2) of the twelve 32-column segments of every token row
    w.total = off * sizeof(float);
    return w;
}

struct Ctx {
    bool S;    // split-operand GEMM mode (activations that only feed GEMMs are F16X2)
    bool W16;  // ... with the weights in the W16 format: weight GEMMs run the single-accumulator kernel (gemm_w16.hip)
    hipStream_t st;
    int terms;  // MFMAs per product in the W16 kernels: 3 (fp32-grade) or 1 (gemm_mode 3: the throughput-mode diagnostic)
};

// `ws` = the weight tensor's 2^-s (W16 mode; 0 = W is not a W16 weight: an activation operand, or another mode)
static int gemm(const Ctx& c, const sm_gemm_args& g, bool out_s = false);
// C = epilogue(A W^T + b); in split mode A and W are F16X2 / W16 and `out_s` asks for an F16X2 C
struct Fold {  // LayerNorm folded into the GEMMs around it (sm_gemm_args.ln_stats / ln_stats_out)
    const float* stats = nullptr;  // consumer: the producer's row statistics ...
    const float* cvec = nullptr;   // ... and the row sums of the gain-scaled weight
    float eps = 0.f;
    float* xs = nullptr;           // producer (RESIDUAL): F16X2 copy of the new residual stream ...
    float* stats_out = nullptr;    // ... and its row statistics
};
static int linear(const Ctx& c, const float* A, int lda, const float* W, float ws, const float* b, float* C, int ldc, int64_t M,
                  int N, int K, int epi, const float* R, int ldr, bool out_s = false, const Fold& f = Fold()) {
    sm_gemm_args g = {};
    g.A = A; g.W = W; g.bias = b; g.C = C; g.R = R;
    g.M = (int)M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldc = ldc; g.ldr = ldr;
    g.batch = 1; g.epilogue = epi; g.w_scale = ws;
    g.ln_stats = f.stats; g.ln_c = f.cvec; g.ln_eps = f.eps; g.C2 = f.xs; g.ln_stats_out = f.stats_out;
    return gemm(c, g, out_s);
}
static bool use_w16(const Ctx& c, const sm_gemm_args& g) { return c.W16 && g.w_scale > 0.f; }
static bool pow2(float s) {
    int ex = 0;
    return s > 0.f && frexpf(s, &ex) == 0.5f;
}
static std::string gemm_name(const Ctx& c, const sm_gemm_args& g) {
    if (!g_timing) return std::string();
    char buf[64];
    int bm = 0, bn = 0, nst = 0;
    if (use_w16(c, g)) {
        const char* nm = sm_gemm_w16_variant_name(sm_gemm_w16_pick(&g));
        std::string s = nm ? nm : "gemm_w16_kernel<?>";
        if (c.terms == 1 && s.size() > 3 && s.compare(s.size() - 3, 3, " 3>") == 0) s.replace(s.size() - 2, 1, "1");  // the TERMS template argument
        return s;
    }
    if (c.S) {
        sm_gemm_f16x2_pick_tile(&g, &bm, &bn, &nst);
        snprintf(buf, sizeof buf, "gemm_f16x2_kernel<%d, %d, %d, 2, %d, %d, 0>", bm, bn, nst, bn == 128 ? 4 : 2, bn == 128 ? 2 : 3);
    } else {
        sm_gemm_f32_pick_tile(&g, &bm, &bn);
        snprintf(buf, sizeof buf, "gemm_f32_kernel<%d, %d, %d>", bm, bn, bm == 128 ? (bn == 128 ? 2 : 3) : 4);
    }
    return buf;
}
static int gemm(const Ctx& c, const sm_gemm_args& g, bool out_s) {
    TapScope tap(c.st, gemm_name(c, g), 2.0 * g.M * g.N * g.K * (g.batch > 0 ? g.batch : 1), 0.0);
    if (use_w16(c, g)) {
        sm_gemm_args gt = g;
        gt.mfma_terms = c.terms;
        return sm_gemm_w16(&gt, out_s ? 1 : 0, c.st);
    }
    return c.S ? sm_gemm_f16x2(&g, out_s ? 1 : 0, c.st) : sm_gemm_f32(&g, c.st);
}
// split mode, N = 384: C = R + (A W^T + b) in place on the residual stream AND the next pre-norm of it (F16X2) in one
// launch on the 64 x 384 full-row tile (SM_EPI_RESIDUAL_LN)
static int linear_residual_ln(const Ctx& c, const float* A, int lda, const float* W, const float* b, float* X, int64_t M, int K,
                              const float* ln_w, const float* ln_b, float eps, float* Xn) {
    sm_gemm_args g = {};
    g.A = A; g.W = W; g.bias = b; g.C = X; g.R = X; g.C2 = Xn;
    g.M = (int)M; g.N = SM_EMBED; g.K = K; g.lda = lda; g.ldw = K; g.ldc = SM_EMBED; g.ldr = SM_EMBED;
    g.batch = 1; g.epilogue = SM_EPI_RESIDUAL_LN; g.ln_gamma = ln_w; g.ln_beta = ln_b; g.ln_eps = eps;
    TapScope tap(c.st, "gemm_f16x2_kernel<64, 384, 2, 2, 4, 1, 0>", 2.0 * g.M * g.N * g.K, 0.0);
    return sm_gemm_f16x2_tile(&g, 0, 64, 384, c.st);
}
// in split mode Q, K and V are F16X2 (written so by the projection GEMMs) and the f16 matrix cores do the work
static int attn(const Ctx& c, sm_attn_args& a) {
    TapScope tap(c.st, c.S ? "attention_f16x2_kernel<4, false>" : "attention_f32_kernel", 4.0 * a.batch * a.heads * a.n_q * (double)a.n_k * SM_HEAD_DIM,
                 8.0 * a.batch * a.heads * SM_HEAD_DIM * ((double)a.n_q + a.n_k));  // Q, O and K, V once, 4 B per element
    a.out_f16x2 = c.S;
    return c.S ? sm_attention_f16x2(&a, c.st) : sm_attention_f32(&a, c.st);
}

struct LnOpt {
    sm_row_map in_map = {0, 0, 0}, out_map = {0, 0, 0};
    float* ys = nullptr;        // F16X2 copy of y
    float* y2 = nullptr;        // y + add (fp32, or F16X2 when y2_s)
    bool y2_s = false;
    const float* add = nullptr;
    int add_rows = 0;
    int n_partials = 0;
    int64_t partial_stride = 0;
    const float* pre_bias = nullptr;
    const float* residual = nullptr;
    float* raw = nullptr;       // the value before normalisation (sm_ln_args.raw)
    // a second norm chained onto this one in the same launch (sm_ln_args.chain_*): the decoder's shared final norm
    const float *chain_w = nullptr, *chain_b = nullptr;
    float *chain_y = nullptr, *chain_ys = nullptr;
    sm_row_map chain_map = {0, 0, 0};
    float chain_eps = 0.f;
};
static int ln(const Ctx& c, const float* x, const float* gw, const float* gb, float* y, int64_t rows, float eps,
              const LnOpt& o = LnOpt()) {
    sm_ln_args a = {};
    a.x = x; a.ldx = SM_EMBED; a.in_map = o.in_map; a.gamma = gw; a.beta = gb; a.y = y; a.ldy = SM_EMBED;
    a.out_map = o.out_map; a.y2 = o.y2; a.ldy2 = SM_EMBED; a.add = o.add; a.add_rows = o.add_rows;
    a.rows = (int)rows; a.eps = eps;
    a.n_partials = o.n_partials; a.partial_stride = o.partial_stride; a.pre_bias = o.pre_bias; a.residual = o.residual;
    a.ys = o.ys; a.y2_f16x2 = o.y2_s ? 1 : 0; a.raw = o.raw;
    a.chain_gamma = o.chain_w; a.chain_beta = o.chain_b; a.chain_y = o.chain_y; a.chain_ys = o.chain_ys; a.chain_ldy = SM_EMBED;
    a.chain_map = o.chain_map; a.chain_eps = o.chain_eps;
    TapScope tap(c.st, "layernorm384_kernel", 0.0, 2.0 * rows * SM_EMBED * 4);
    return sm_layernorm_rows_f32(&a, c.st);
}

// decoder start state in one launch: tgt = 0 (fp32 and F16X2 - zero bytes in both), tgt + query_pos = query_pos broadcast
// over the batch (fp32, or F16X2 in split mode).  A kernel rather than hipMemsetAsync: memset nodes in a captured
// hipGraph misbehaved on replay (graphs.py), and it saves three launches.
template <bool SPLIT>
__global__ __launch_bounds__(256) void decoder_init_kernel(const float* __restrict__ qpos, float* __restrict__ tgt,
                                                          float* __restrict__ tgts, float* __restrict__ tgtq, int rows_per,
                                                          int64_t total4) {
    const int64_t per4 = (int64_t)rows_per * (SM_EMBED / 4);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
        reinterpret_cast<float4*>(tgt)[t] = z;
        reinterpret_cast<float4*>(tgts)[t] = z;
        const float4 q = reinterpret_cast<const float4*>(qpos)[t % per4];
        if constexpr (SPLIT) {
            const int64_t e = t * 4, row = e / SM_EMBED;
            const float v[4] = {q.x, q.y, q.z, q.w};
            store_f16x2_4(tgtq + row * SM_EMBED, e - row * SM_EMBED, v);
        } else {
            reinterpret_cast<float4*>(tgtq)[t] = q;
        }
    }
}

#define TRY(x)                \
    do {                      \
        int _rc = (x);        \
        if (_rc) return _rc;  \
    } while (0)

static int forward(const sm_weights* w, const sm_forward_io* io, float* wsbase, hipStream_t st) {
    const Shape s = make_shape(w, io->B, io->H, io->W);
    Ws ws = carve(s, wsbase);
    const int D = SM_EMBED;
    const Ctx c = {w->gemm_mode >= 1, w->gemm_mode >= 2, st, w->gemm_mode == 3 ? 1 : 3};
    const bool S = c.S;

    // ---- tokens: patch embedding + cls + position (vision_transformer.py:269-281) ----------------------------
    const float* pos = w->pos_embed;
    if (s.n != w->pos_grid * w->pos_grid) {  // interpolate_pos_encoding compares token COUNTS (:386-388)
        TRY(sm_pos_embed_bicubic_f32(w->pos_embed, w->pos_grid, ws.pos, s.gh, s.gw, st));
        pos = ws.pos;
    }
    float* cols = ws.HID;
    TRY(S ? sm_im2col_patches_f16x2(io->x, cols, s.B, s.H, s.W, s.P, st)
          : sm_im2col_patches_f32(io->x, cols, s.B, s.H, s.W, s.P, st));
    TRY(sm_cls_rows_f32(w->cls_token, pos, ws.X, s.B, s.N, st));
    {
        sm_gemm_args g = {};
        g.A = cols; g.W = w->patch_w; g.bias = w->patch_b; g.C = ws.X; g.R = pos;
        g.M = (int)s.Mp; g.N = D; g.K = 3 * s.P * s.P; g.lda = g.K; g.ldw = g.K; g.ldc = D; g.ldr = D;
        g.batch = 1; g.epilogue = SM_EPI_PATCH; g.patch_n = s.n; g.w_scale = w->patch_s;
        TRY(gemm(c, g));
    }

    // ---- 12 pre-norm blocks (vision_transformer.py:164-170) -----------------------------------------------------
    // Optional (split mode): the two pre-norms of a block ride on the GEMM that produces their input (proj -> norm2,
    // fc2 -> the next block's norm1); only the first norm1 is then a launch of its own.
    // SM_FUSED_LN (tuning build only, default 0): 1 = proj and fc2, 2 = proj only.  Measured with three batches in flight:
    // 17.3k images/s against 18.0k unfused - the full-row tile needs 112 KiB of LDS (one workgroup per CU, 197 of them).
#ifdef SM_TUNING
    static const int fused_ln_env = getenv("SM_FUSED_LN") ? atoi(getenv("SM_FUSED_LN")) : 0;
    static const int fused_qkv_env = getenv("SM_FUSED_QKV") ? atoi(getenv("SM_FUSED_QKV")) : 1;
#else
    constexpr int fused_ln_env = 0, fused_qkv_env = 1;
#endif
    const bool fuse_proj = S && !c.W16 && fused_ln_env >= 1, fuse_fc2 = S && !c.W16 && fused_ln_env == 1;
    // W16 mode, token grids of <= 208 (ViT-S/16 at 224^2): the qkv projection and the attention are ONE launch
    // (qkv_attention.hip); SM_FUSED_QKV=0 (tuning build only) keeps the two-launch path (same results to rounding)
    // ... and at least 96 (image, head) workgroups: the fused kernel is one workgroup per pair with a ~30 us life, so a small batch
    // leaves most CUs idle for that long - below B = 16 the two-launch path is faster (B = 1: 1.25 vs 1.38 ms per forward, B = 4:
    // 1.40 vs 1.96, B = 8: 1.53 vs 2.10, B = 16: 1.94 vs 1.93; profiles/r03_fused_vs_unfused_by_batch.log)
    const bool fused_qkv = c.W16 && fused_qkv_env != 0 && s.N <= sm_qkv_attention_max_tokens() && io->attn_path != 2 &&
                           (io->attn_path == 1 || s.B * SM_HEADS >= 96);
    LnOpt xs;
    xs.ys = S ? ws.Xn : nullptr;  // LN output only feeds a GEMM: F16X2 in split mode
#ifdef SM_TUNING  // timing-only ablation (tuning build): what the 24 encoder LayerNorm launches cost the pipeline (results are garbage)
    static const int ablate_ln = getenv("SM_ABLATE_LN") ? atoi(getenv("SM_ABLATE_LN")) : 0;
#else
    constexpr int ablate_ln = 0;
#endif
    // sm_weights.ln_fold: norm2 rides on proj -> fc1 and the next block's norm1 on fc2 -> qkv: the residual epilogues also write the
    // raw stream as F16X2 (into Xn) with its row statistics, the consuming GEMM carries the gain in its weights and applies
    // r (x W'^T - mu c) + b' in its epilogue.  Block 0's norm1 (its input comes from the patch embedding) stays a launch.
    // Measured (profiles/r03_ln_fold_ab.log): the fold wins where the forward is latency-bound (B = 1: 1.24 -> 1.20 ms, B = 8: 1.45 -> 1.34)
    // and loses where it is throughput-bound (B = 64, three streams: 21.55k -> 21.3k images/s: the extra F16X2 stream of the residual
    // epilogues and the statistics cost more than 23 co-resident LayerNorm launches) - so it follows the same batch rule and the same
    // pin as the attention path (sm_forward_io.attn_path: 2 = the small-batch kernels, 1 = the large-batch ones).
    const bool lnf = c.W16 && w->ln_fold != 0 && io->attn_path != 1 && (io->attn_path == 2 || s.B * SM_HEADS < 96);
    if (fuse_fc2) TRY(ln(c, ws.X, w->enc[0].norm1_w, w->enc[0].norm1_b, nullptr, s.M, 1e-6f, xs));
    const bool split_fc2 = c.W16 && S && io->attn_path == 0 && s.M <= SM_SPLIT_FC2_ROWS && !fuse_fc2 && !ablate_ln;
    for (int i = 0; i < SM_ENC_DEPTH; ++i) {
        const sm_enc_layer& e = w->enc[i];
        const bool n1_done = split_fc2 && i > 0;  // the previous block's split fc2 ended in this block's norm1
        const bool f1 = lnf && i > 0 && !n1_done;  // this block's norm1 is folded into its qkv projection
        if (!fuse_fc2 && !f1 && !n1_done && !(ablate_ln && i > 0)) TRY(ln(c, ws.X, e.norm1_w, e.norm1_b, S ? nullptr : ws.Xn, s.M, 1e-6f, xs));
        Fold fq;
        if (f1) { fq.stats = ws.ST; fq.cvec = e.qkv_c; fq.eps = 1e-6f; }
        const float* qkv_w = f1 ? e.qkv_fw : e.qkv_w;
        const float* qkv_b = f1 ? e.qkv_fb : e.qkv_b;
        const float qkv_s = f1 ? e.qkv_fs : e.qkv_s;
        if (fused_qkv) {
            sm_qkv_attn_args q = {};
            q.Xn = ws.Xn; q.Wqkv = qkv_w; q.bias = qkv_b; q.O = ws.AO; q.ldx = D; q.ldo = D;
            q.B = s.B; q.N = s.N; q.w_scale = qkv_s; q.scale = 0.125f; q.out_f16x2 = 1; q.mfma_terms = c.terms;
            q.ln_stats = fq.stats; q.ln_c = fq.cvec; q.ln_eps = fq.eps;
            // algorithmic work of SURVEY.md 8d: 2 N 384 1152 + 4 N^2 384 FLOPs, x in + o out bytes per image
            TapScope tap(c.st, sm_qkv_attention_kernel_name(c.terms), (double)s.B * (2.0 * s.N * D * 3 * D + 4.0 * s.N * s.N * D),
                         2.0 * s.M * D * 4);
            TRY(sm_qkv_attention_w16(&q, c.st));
        } else {
            TRY(linear(c, ws.Xn, D, qkv_w, qkv_s, qkv_b, ws.QKV, 3 * D, s.M, 3 * D, D, SM_EPI_BIAS, nullptr, 0, S, fq));
            sm_attn_args a = {};
            a.Q = ws.QKV; a.K = ws.QKV + D; a.V = ws.QKV + 2 * D; a.O = ws.AO;
            a.sQb = a.sKb = a.sVb = (int64_t)s.N * 3 * D; a.sQr = a.sKr = a.sVr = 3 * D;
            a.sOb = (int64_t)s.N * D; a.sOr = D;
            a.batch = s.B; a.heads = SM_HEADS; a.n_q = s.N; a.n_k = s.N; a.scale = 0.125f;
            TRY(attn(c, a));
        }
        if (fuse_proj) {
            TRY(linear_residual_ln(c, ws.AO, D, e.proj_w, e.proj_b, ws.X, s.M, D, e.norm2_w, e.norm2_b, 1e-6f, ws.Xn));
        } else {
            Fold fp;
            if (lnf) { fp.xs = ws.Xn; fp.stats_out = ws.ST; }
            TRY(linear(c, ws.AO, D, e.proj_w, e.proj_s, e.proj_b, ws.X, D, s.M, D, D, SM_EPI_RESIDUAL, ws.X, D, false, fp));
            if (!lnf && !ablate_ln) TRY(ln(c, ws.X, e.norm2_w, e.norm2_b, S ? nullptr : ws.Xn, s.M, 1e-6f, xs));
        }
        Fold f2;
        if (lnf) { f2.stats = ws.ST; f2.cvec = e.fc1_c; f2.eps = 1e-6f; }
        TRY(linear(c, ws.Xn, D, lnf ? e.fc1_fw : e.fc1_w, lnf ? e.fc1_fs : e.fc1_s, lnf ? e.fc1_fb : e.fc1_b, ws.HID, SM_MLP, s.M, SM_MLP, D,
                   SM_EPI_GELU, nullptr, 0, S, f2));
        if (fuse_fc2 && i + 1 < SM_ENC_DEPTH) {
            const sm_enc_layer& nx = w->enc[i + 1];
            TRY(linear_residual_ln(c, ws.HID, SM_MLP, e.fc2_w, e.fc2_b, ws.X, s.M, SM_MLP, nx.norm1_w, nx.norm1_b, 1e-6f, ws.Xn));
        } else if (split_fc2 && i + 1 < SM_ENC_DEPTH) {
            const sm_enc_layer& nx = w->enc[i + 1];
            sm_gemm_args g = {};
            g.A = ws.HID; g.W = e.fc2_w; g.C = ws.PART; g.M = (int)s.M; g.N = D; g.K = SM_MLP; g.lda = SM_MLP;
            g.ldw = SM_MLP; g.ldc = D; g.batch = 1; g.epilogue = SM_EPI_BIAS; g.split_k = 4; g.strideC = s.M * D;
            g.w_scale = e.fc2_s;
            TRY(gemm(c, g));
            LnOpt o;  // x += fc2 (slices + bias + residual, written back as the stream) and the next block's norm1 of it
            o.ys = ws.Xn; o.n_partials = 4; o.partial_stride = s.M * D; o.pre_bias = e.fc2_b; o.residual = ws.X; o.raw = ws.X;
            TRY(ln(c, ws.PART, nx.norm1_w, nx.norm1_b, nullptr, s.M, 1e-6f, o));
        } else {
            Fold fo;
            if (lnf && i + 1 < SM_ENC_DEPTH) { fo.xs = ws.Xn; fo.stats_out = ws.ST; }
            TRY(linear(c, ws.HID, SM_MLP, e.fc2_w, e.fc2_s, e.fc2_b, ws.X, D, s.M, D, SM_MLP, SM_EPI_RESIDUAL, ws.X, D, false, fo));
        }
    }
    // final norm on the last layer only (the other 11 per-layer norms of :299 are dead work when
    // lateral_connection=False), dropping the cls row on the way (maskformer.py:107-108,177)
    float* tok = io->patch_tokens ? io->patch_tokens : ws.TOK;
    {
        LnOpt o;
        o.in_map = {s.n, s.N, 1};
        o.ys = S ? ws.TOKs : nullptr;
        TRY(ln(c, ws.X, w->enc_norm_w, w->enc_norm_b, tok, s.Mp, 1e-6f, o));
    }
    if (io->encoder_only) return SM_OK;
    const float* tok_a = S ? ws.TOKs : tok;  // GEMM-operand view of the tokens

    // ---- 6 post-norm decoder layers (transformer_decoder.py:260-297, :112-150) ----------------------------------
    float* QD = io->queries ? io->queries : ws.QD;
    const float* qpos = w->query_embed;
    {
        const int64_t total4 = s.Md * D / 4;
        const int grid = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
        if (S)
            hipLaunchKernelGGL(decoder_init_kernel<true>, dim3(grid), dim3(256), 0, st, qpos, ws.TGT, ws.TGTs, ws.TGTQ, s.nq, total4);
        else
            hipLaunchKernelGGL(decoder_init_kernel<false>, dim3(grid), dim3(256), 0, st, qpos, ws.TGT, ws.TGTs, ws.TGTQ, s.nq, total4);
        TRY(check_launch("decoder_init"));
    }
    // cross-attention keys/values of every layer depend only on the encoder memory: one large GEMM
    // (B*n x 384) x (384 x L*768) instead of L small ones on the critical chain
    const int KVW = s.L * 2 * D;
    TRY(linear(c, tok_a, D, w->dec_kv_w, w->dec_kv_s, w->dec_kv_b, ws.KV, KVW, s.Mp, KVW, D, SM_EPI_BIAS, nullptr, 0, S));
    for (int l = 0; l < s.L && w->normalize_before; ++l) {
        // forward_pre (transformer_decoder.py:299-327): tgt2 = norm_k(tgt) feeds the sub-block, tgt += sub-block(tgt2); the
        // residual stream ws.TGT (fp32) is updated in place by the RESIDUAL epilogues and never normalised in the layer
        const sm_dec_layer& d = w->dec[l];
        float* nrm = S ? nullptr : ws.T2;                 // fp32 copy of the normed tgt (fp32 mode only)
        const float* nrm_a = S ? ws.TGTs : ws.T2;         // its GEMM-operand view
        {
            LnOpt o;  // norm1 -> tgt2 (value operand) and tgt2 + query_pos (q = k operand)
            o.ys = S ? ws.TGTs : nullptr; o.y2 = ws.TGTQ; o.y2_s = S; o.add = qpos; o.add_rows = s.nq;
            TRY(ln(c, ws.TGT, d.norm1_w, d.norm1_b, nrm, s.Md, 1e-5f, o));
        }
        {
            sm_gemm_args g = {};
            g.A = ws.TGTQ; g.A_alt = nrm_a; g.alt_from_n = 2 * D; g.W = d.sa_in_w; g.bias = d.sa_in_b; g.C = ws.QK;
            g.M = (int)s.Md; g.N = 3 * D; g.K = D; g.lda = D; g.ldw = D; g.ldc = 3 * D; g.batch = 1; g.epilogue = SM_EPI_BIAS;
            g.w_scale = d.sa_in_s;
            TRY(gemm(c, g, S));
        }
        sm_attn_args a = {};
        a.Q = ws.QK; a.K = ws.QK + D; a.V = ws.QK + 2 * D; a.O = ws.AOd;
        a.sQb = a.sKb = a.sVb = (int64_t)s.nq * 3 * D; a.sQr = a.sKr = a.sVr = 3 * D;
        a.sOb = (int64_t)s.nq * D; a.sOr = D;
        a.batch = s.B; a.heads = SM_HEADS; a.n_q = s.nq; a.n_k = s.nq; a.scale = 0.125f;
        TRY(attn(c, a));
        TRY(linear(c, ws.AOd, D, d.sa_out_w, d.sa_out_s, d.sa_out_b, ws.TGT, D, s.Md, D, D, SM_EPI_RESIDUAL, ws.TGT, D));
        {
            LnOpt o;  // norm2 -> only tgt2 + query_pos is needed (cross-attention query; key = value = memory)
            o.ys = S ? ws.TGTs : nullptr; o.y2 = ws.TGTQ; o.y2_s = S; o.add = qpos; o.add_rows = s.nq;
            TRY(ln(c, ws.TGT, d.norm2_w, d.norm2_b, nrm, s.Md, 1e-5f, o));
        }
        TRY(linear(c, ws.TGTQ, D, d.ca_in_w, d.ca_in_s, d.ca_in_b, ws.Qc, D, s.Md, D, D, SM_EPI_BIAS, nullptr, 0, S));
        a = {};
        a.Q = ws.Qc; a.K = ws.KV + (int64_t)l * 2 * D; a.V = a.K + D; a.O = ws.AOd;
        a.sQb = (int64_t)s.nq * D; a.sQr = D; a.sKb = a.sVb = (int64_t)s.n * KVW; a.sKr = a.sVr = KVW;
        a.sOb = (int64_t)s.nq * D; a.sOr = D;
        a.batch = s.B; a.heads = SM_HEADS; a.n_q = s.nq; a.n_k = s.n; a.scale = 0.125f;
        TRY(attn(c, a));
        TRY(linear(c, ws.AOd, D, d.ca_out_w, d.ca_out_s, d.ca_out_b, ws.TGT, D, s.Md, D, D, SM_EPI_RESIDUAL, ws.TGT, D));
        {
            LnOpt o;  // norm3 -> operand of linear1
            o.ys = S ? ws.TGTs : nullptr;
            TRY(ln(c, ws.TGT, d.norm3_w, d.norm3_b, nrm, s.Md, 1e-5f, o));
        }
        TRY(linear(c, nrm_a, D, d.lin1_w, d.lin1_s, d.lin1_b, ws.HIDd, SM_MLP, s.Md, SM_MLP, D, SM_EPI_RELU, nullptr, 0, S));
        TRY(linear(c, ws.HIDd, SM_MLP, d.lin2_w, d.lin2_s, d.lin2_b, ws.TGT, D, s.Md, D, SM_MLP, SM_EPI_RESIDUAL, ws.TGT, D));
        {
            LnOpt o;  // intermediate.append(self.norm(output)) (:138-139), scattered into (B, L, nq, 384) (+ F16X2 copy)
            o.out_map = {s.nq, s.L * s.nq, l * s.nq};
            o.ys = S ? ws.QDs : nullptr;
            TRY(ln(c, ws.TGT, w->dec_norm_w, w->dec_norm_b, QD, s.Md, 1e-5f, o));
        }
    }
    for (int l = 0; l < s.L && !w->normalize_before; ++l) {
        const sm_dec_layer& d = w->dec[l];
        const float* tgt_a = S ? ws.TGTs : ws.TGT;  // GEMM-operand view of tgt (TGT / T2 swap roles every layer)
        // self-attention: q = k = tgt + query_pos, v = tgt  ->  ONE launch: columns [0,768) read TGTQ, [768,1152) TGT
        {
            sm_gemm_args g = {};
            g.A = ws.TGTQ; g.A_alt = tgt_a; g.alt_from_n = 2 * D; g.W = d.sa_in_w; g.bias = d.sa_in_b; g.C = ws.QK;
            g.M = (int)s.Md; g.N = 3 * D; g.K = D; g.lda = D; g.ldw = D; g.ldc = 3 * D; g.batch = 1; g.epilogue = SM_EPI_BIAS;
            g.w_scale = d.sa_in_s;
            TRY(gemm(c, g, S));
        }
        sm_attn_args a = {};
        a.Q = ws.QK; a.K = ws.QK + D; a.V = ws.QK + 2 * D; a.O = ws.AOd;
        a.sQb = a.sKb = a.sVb = (int64_t)s.nq * 3 * D; a.sQr = a.sKr = a.sVr = 3 * D;
        a.sOb = (int64_t)s.nq * D; a.sOr = D;
        a.batch = s.B; a.heads = SM_HEADS; a.n_q = s.nq; a.n_k = s.nq; a.scale = 0.125f;
        TRY(attn(c, a));
        TRY(linear(c, ws.AOd, D, d.sa_out_w, d.sa_out_s, d.sa_out_b, ws.T2, D, s.Md, D, D, SM_EPI_RESIDUAL, ws.TGT, D));
        {
            LnOpt o;  // norm1 -> tgt (fp32: residual of the next block) + tgt + query_pos (cross-attention query operand)
            o.y2 = ws.TGTQ; o.y2_s = S; o.add = qpos; o.add_rows = s.nq;
            TRY(ln(c, ws.T2, d.norm1_w, d.norm1_b, ws.TGT, s.Md, 1e-5f, o));
        }
        // cross-attention: q = tgt + query_pos, k = v = memory (pos = None)
        TRY(linear(c, ws.TGTQ, D, d.ca_in_w, d.ca_in_s, d.ca_in_b, ws.Qc, D, s.Md, D, D, SM_EPI_BIAS, nullptr, 0, S));
        a = {};
        a.Q = ws.Qc; a.K = ws.KV + (int64_t)l * 2 * D; a.V = a.K + D; a.O = ws.AOd;
        a.sQb = (int64_t)s.nq * D; a.sQr = D; a.sKb = a.sVb = (int64_t)s.n * KVW; a.sKr = a.sVr = KVW;
        a.sOb = (int64_t)s.nq * D; a.sOr = D;
        a.batch = s.B; a.heads = SM_HEADS; a.n_q = s.nq; a.n_k = s.n; a.scale = 0.125f;
        TRY(attn(c, a));
        TRY(linear(c, ws.AOd, D, d.ca_out_w, d.ca_out_s, d.ca_out_b, ws.T2, D, s.Md, D, D, SM_EPI_RESIDUAL, ws.TGT, D));
        {
            LnOpt o;  // norm2 -> tgt (residual of the FFN) (+ F16X2 copy: operand of linear1)
            o.ys = S ? ws.TGTs : nullptr;
            TRY(ln(c, ws.T2, d.norm2_w, d.norm2_b, ws.TGT, s.Md, 1e-5f, o));
        }
        // FFN: linear2 (K = 1536, only M/64 x 6 tiles) is split 4-way along K; norm3 sums the slices + bias + residual
        TRY(linear(c, tgt_a, D, d.lin1_w, d.lin1_s, d.lin1_b, ws.HIDd, SM_MLP, s.Md, SM_MLP, D, SM_EPI_RELU, nullptr, 0, S));
        {
            sm_gemm_args g = {};
            g.A = ws.HIDd; g.W = d.lin2_w; g.C = ws.PART; g.M = (int)s.Md; g.N = D; g.K = SM_MLP; g.lda = SM_MLP;
            g.ldw = SM_MLP; g.ldc = D; g.batch = 1; g.epilogue = SM_EPI_BIAS; g.split_k = 4; g.strideC = s.Md * D;
            g.w_scale = d.lin2_s;
            TRY(gemm(c, g));
        }
        {
            LnOpt o;  // norm3(sum of slices + bias + tgt) -> new tgt (fp32 + F16X2) and tgt + query_pos
            o.ys = S ? ws.TGTs : nullptr;
            o.y2 = ws.TGTQ; o.y2_s = S; o.add = qpos; o.add_rows = s.nq;
            o.n_partials = 4; o.partial_stride = s.Md * D; o.pre_bias = d.lin2_b; o.residual = ws.TGT;
            // ... and, chained in the same launch, the shared final norm on this layer's output, scattered into (B, L, nq, 384)
            // (+ F16X2 copy): transformer_decoder.py:138-139.  Six launches fewer on the critical chain of a forward (serving).
            o.chain_w = w->dec_norm_w; o.chain_b = w->dec_norm_b; o.chain_y = QD; o.chain_ys = S ? ws.QDs : nullptr;
            o.chain_map = {s.nq, s.L * s.nq, l * s.nq}; o.chain_eps = 1e-5f;
            TRY(ln(c, ws.PART, d.norm3_w, d.norm3_b, ws.T2, s.Md, 1e-5f, o));
        }
        { float* t = ws.TGT; ws.TGT = ws.T2; ws.T2 = t; }  // norm3 wrote the new tgt into T2
    }
    const float* qd_a = S ? ws.QDs : QD;

    // ---- heads --------------------------------------------------------------------------------------------------
    TRY(sm_query_mean_f32(QD, io->features, s.B, s.L, s.nq, st));
    if (w->mask_head_ffn) {
        // return_intermediate=True with use_binary_classifier=False (maskformer.py:225): the mask einsum takes
        // ffn(queries), a 384->384->384->384 MLP with ReLU between the layers (MLP.forward :265-268), not the queries
        TRY(linear(c, qd_a, D, w->ffn0_w, w->ffn0_s, w->ffn0_b, ws.O1, D, s.Mo, D, D, SM_EPI_RELU, nullptr, 0, S));
        TRY(linear(c, ws.O1, D, w->ffn1_w, w->ffn1_s, w->ffn1_b, ws.O2, D, s.Mo, D, D, SM_EPI_RELU, nullptr, 0, S));
        TRY(linear(c, ws.O2, D, w->ffn2_w, w->ffn2_s, w->ffn2_b, ws.O1, D, s.Mo, D, D, SM_EPI_BIAS, nullptr, 0, S));
        qd_a = ws.O1;
    }
    // return_intermediate=False (the 3-D path, maskformer.py:219-220): the decoder hands back its last layer only, so only that
    // layer's queries reach the mask einsum; the outputs are then (B, 1, nq, 2gh, 2gw)
    const int Lm = io->last_layer_only ? 1 : s.L;
    const float* qm_a = qd_a + (io->last_layer_only ? (int64_t)(s.L - 1) * s.nq * D : 0);
    if (s.n % 4 == 0) {
        // mask_pred = sigmoid(up(Q . tok^T)): the einsum of maskformer.py:223 commutes with the bilinear x2 of the pixel
        // decoder (:144-162) - both linear - so the GEMM runs on the token grid (N = n instead of 4n) and the (B, 4n,
        // 384) up-sampled feature map is never built
        sm_gemm_args g = {};
        g.A = qm_a; g.W = tok_a; g.C = ws.LOG;
        g.M = Lm * s.nq; g.N = s.n; g.K = D; g.lda = D; g.ldw = D; g.ldc = s.n;
        g.strideA = (int64_t)s.L * s.nq * D; g.strideW = (int64_t)s.n * D; g.strideC = (int64_t)Lm * s.nq * s.n;
        g.batch = s.B; g.epilogue = SM_EPI_BIAS;
        TRY(gemm(c, g));
        TRY(sm_upsample_logits_sigmoid_f32(ws.LOG, io->mask_logits, io->mask_pred, (int64_t)s.B * Lm * s.nq, s.gh, s.gw, s.sf, st));
    } else {  // token counts that are not a multiple of 4 (float4 rows of the GEMM output): the literal order
        TRY(S ? sm_upsample_tokens_f16x2(tok, (int64_t)s.n * D, ws.UP, s.B, s.gh, s.gw, s.sf, st)
              : sm_upsample_tokens_f32(tok, (int64_t)s.n * D, ws.UP, s.B, s.gh, s.gw, s.sf, st));
        // mask_pred[b] = sigmoid(Q[b] (L*nq x 384) . up[b]^T (384 x 4n))   (maskformer.py:223)
        sm_gemm_args g = {};
        g.A = qm_a; g.W = ws.UP; g.C = io->mask_logits ? io->mask_logits : ws.LOG; g.C2 = io->mask_pred;
        const int up_n = s.sf * s.sf * s.n;
        g.M = Lm * s.nq; g.N = up_n; g.K = D; g.lda = D; g.ldw = D; g.ldc = up_n;
        g.strideA = (int64_t)s.L * s.nq * D; g.strideW = (int64_t)up_n * D; g.strideC = (int64_t)Lm * s.nq * up_n;
        g.batch = s.B; g.epilogue = SM_EPI_SIGMOID2;
        TRY(gemm(c, g));
    }
    if (w->mask_head_ffn || w->no_objectness) return SM_OK;  // no objectness on these paths (maskformer.py:246-249, :219-220)
    TRY(linear(c, qd_a, D, w->ffn0_w, w->ffn0_s, w->ffn0_b, ws.O1, D, s.Mo, D, D, SM_EPI_RELU, nullptr, 0, S));
    TRY(linear(c, ws.O1, D, w->ffn1_w, w->ffn1_s, w->ffn1_b, ws.O2, D, s.Mo, D, D, SM_EPI_RELU, nullptr, 0));
    TRY(sm_rowdot_sigmoid_f32(ws.O2, w->ffn2_w, w->ffn2_b, io->objectness, (int)s.Mo, st));
    return SM_OK;
}

static int validate(const sm_weights* w, const sm_forward_io* io) {
    SM_REQUIRE(w && io, "sm_maskformer_forward: null arguments");
    SM_REQUIRE(w->patch == 8 || w->patch == 16, "sm_maskformer_forward: patch=%d (8 or 16)", w->patch);
    SM_REQUIRE(w->gemm_mode >= 0 && w->gemm_mode <= 3, "sm_maskformer_forward: gemm_mode=%d (0..3)", w->gemm_mode);
    SM_REQUIRE(w->n_dec_layers >= 1 && w->n_dec_layers <= SM_MAX_DEC_LAYERS, "sm_maskformer_forward: n_dec_layers=%d",
               w->n_dec_layers);
    SM_REQUIRE(w->n_queries >= 1 && w->pos_grid >= 1, "sm_maskformer_forward: bad n_queries/pos_grid");
    SM_REQUIRE(w->scale_factor >= 0 && w->scale_factor <= 16, "sm_maskformer_forward: scale_factor=%d (1..16; 0 = the shipped 2)", w->scale_factor);
    SM_REQUIRE(w->dec_kv_w && w->dec_kv_b, "sm_maskformer_forward: dec_kv_w/dec_kv_b (packed cross-attention K/V) missing");
    SM_REQUIRE(io->x && io->B > 0 && io->H > 0 && io->W > 0, "sm_maskformer_forward: bad input shape");
    if (w->gemm_mode >= 1) {
        const int gh = (io->H + w->patch - 1) / w->patch, gw = (io->W + w->patch - 1) / w->patch;
        const int sf = w->scale_factor > 0 ? w->scale_factor : 2;
        SM_REQUIRE((gh * gw) % 4 == 0 || (sf * sf * gh * gw) % 4 == 0,
                   "sm_maskformer_forward: the token count or the mask size must be a multiple of 4 (scale_factor %d on a %d x %d grid)", sf, gh, gw);
    }
    if (w->gemm_mode >= 2) {
        // W16 weights carry their 2^-s in the *_s fields; a zero (a caller that filled the pointers but not the scales) would
        // otherwise send W16 bytes through the F16X2 kernel: wrong results, no error
        bool ok = pow2(w->patch_s) && pow2(w->dec_kv_s) && pow2(w->ffn0_s) && pow2(w->ffn1_s) && (!w->mask_head_ffn || pow2(w->ffn2_s));
        for (int i = 0; i < SM_ENC_DEPTH && ok; ++i)
            ok = pow2(w->enc[i].qkv_s) && pow2(w->enc[i].proj_s) && pow2(w->enc[i].fc1_s) && pow2(w->enc[i].fc2_s);
        for (int l = 0; l < w->n_dec_layers && ok; ++l) {
            const sm_dec_layer& d = w->dec[l];
            ok = pow2(d.sa_in_s) && pow2(d.sa_out_s) && pow2(d.ca_in_s) && pow2(d.ca_out_s) && pow2(d.lin1_s) && pow2(d.lin2_s);
        }
        SM_REQUIRE(ok, "sm_maskformer_forward: gemm_mode 2/3 needs every weight's 2^-s (*_s fields) to be a positive power of two");
        if (w->ln_fold) {
            for (int i = 0; i < SM_ENC_DEPTH && ok; ++i) {
                const sm_enc_layer& e = w->enc[i];
                ok = e.fc1_fw && e.fc1_fb && e.fc1_c && pow2(e.fc1_fs) && (i == 0 || (e.qkv_fw && e.qkv_fb && e.qkv_c && pow2(e.qkv_fs)));
            }
            SM_REQUIRE(ok, "sm_maskformer_forward: ln_fold needs the gain-scaled weights, folded biases and row sums (*_fw, *_fb, *_c, *_fs) of every encoder layer");
        }
    }
    if (!io->encoder_only)
        SM_REQUIRE(io->mask_pred && (io->objectness || w->mask_head_ffn || w->no_objectness) && io->features,
                   "sm_maskformer_forward: null output");
    else
        SM_REQUIRE(io->patch_tokens, "sm_maskformer_forward: encoder_only needs patch_tokens");
    SM_REQUIRE(io->attn_path >= 0 && io->attn_path <= 2, "sm_maskformer_forward: attn_path=%d (0 auto, 1 fused, 2 two launches)", io->attn_path);
    SM_REQUIRE(!(io->last_layer_only && w->mask_head_ffn), "sm_maskformer_forward: last_layer_only is the 3-D path (no ffn mask head)");
    return SM_OK;
}

}  // namespace sm

extern "C" size_t sm_forward_workspace_bytes(const sm_weights* w, int32_t B, int32_t H, int32_t W) {
    if (!w || B <= 0 || H <= 0 || W <= 0 || (w->patch != 8 && w->patch != 16)) return 0;
    return sm::carve(sm::make_shape(w, B, H, W), nullptr).total;
}

extern "C" int sm_maskformer_forward(const sm_weights* w, const sm_forward_io* io, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int rc = sm::validate(w, io);
    if (rc) return rc;
    const size_t need = sm_forward_workspace_bytes(w, io->B, io->H, io->W);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 256) != 0) {
        sm::set_error("sm_maskformer_forward: workspace %zu B < %zu B needed (or not 256-B aligned)", workspace_bytes, need);
        return SM_ENOSPACE;
    }
    return sm::forward(w, io, (float*)workspace, (hipStream_t)stream);
}

extern "C" int sm_forward_timing(int enable) {
    for (auto& t : sm::g_taps) {
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    sm::g_taps.clear();
    sm::g_timing = enable != 0;
    return SM_OK;
}

extern "C" int sm_forward_timing_read(sm_kernel_time* out, int max_entries) {
    SM_REQUIRE(out && max_entries > 0, "sm_forward_timing_read: null output");
    int n = 0;
    // calibration: an event pair with nothing between it still measures the command processor's event handling
    double overhead_us = 0.0;
    if (!sm::g_taps.empty()) {
        hipStream_t st = sm::g_taps.front().st;
        (void)hipEventSynchronize(sm::g_taps.back().e1);
        const int reps = 16;
        hipEvent_t ev[2 * reps];
        for (auto& e : ev) (void)hipEventCreate(&e);
        for (auto& e : ev) (void)hipEventRecord(e, st);
        (void)hipEventSynchronize(ev[2 * reps - 1]);
        for (int i = 0; i < reps; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
            overhead_us += ms * 1e3 / reps;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    for (auto& t : sm::g_taps) {
        float ms = 0.f;
        if (hipEventSynchronize(t.e1) != hipSuccess || hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) {
            sm::set_error("sm_forward_timing_read: event query failed");
            return SM_ELAUNCH;
        }
        int k = 0;
        while (k < n && t.name != out[k].name) ++k;
        if (k == n) {
            if (n == max_entries) continue;
            memset(&out[n], 0, sizeof out[n]);
            strncpy(out[n].name, t.name.c_str(), sizeof out[n].name - 1);
            ++n;
        }
        out[k].launches += 1;
        out[k].overhead_us = overhead_us;
        out[k].total_us += ms * 1e3 > overhead_us ? ms * 1e3 - overhead_us : 0.0;
        out[k].flops += t.flops;
        out[k].bytes += t.bytes;
    }
    return n;
}
