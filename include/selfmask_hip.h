/*
 * selfmask_hip.h — C ABI of libselfmask_hip.so: the MI355X (gfx950) implementation of the SelfMask
 * saliency-inference hot path (DINO ViT-S encoder -> MaskFormer decoder -> N_q query masks).
 *
 * The reference (DaniyalMuneer786/Salient-Object-Detection) has no native/FFI layer: the path sits behind plain
 * Python objects and every arithmetic step is a stock PyTorch ATen op.  Each entry point below therefore cites the
 * reference *Python* call site whose ATen op(s) it replaces (file:line relative to the reference root).  The
 * Python binding a maintainer adds is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (PyTorch allocations);
 *   - fp32 everywhere (the reference computes in fp32); row-major, innermost dimension contiguous;
 *   - `stream` is a hipStream_t passed as void*; calls are stream-ordered and never synchronise;
 *   - kernels never allocate; forward() uses a caller-sized workspace (sm_forward_workspace_bytes);
 *   - return 0 on success, a negative SM_E* code otherwise; sm_last_error() gives the message (thread-local);
 *   - stateless and re-entrant: safe from several host threads on distinct streams.
 */
#ifndef SELFMASK_HIP_H
#define SELFMASK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_OK 0
#define SM_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define SM_ELAUNCH (-2)  /* hipLaunchKernel reported an error */
#define SM_ENOSPACE (-3) /* workspace too small */

#define SM_EMBED 384
#define SM_HEADS 6
#define SM_HEAD_DIM 64
#define SM_MLP 1536
#define SM_ENC_DEPTH 12
#define SM_MAX_DEC_LAYERS 8

int sm_version(void);
const char* sm_last_error(void);

/* ---- epilogues of sm_gemm_f32 ------------------------------------------------------------------------------ */
#define SM_EPI_BIAS 0      /* C = A W^T + bias                                   (F.linear)                      */
#define SM_EPI_GELU 1      /* C = gelu_erf(A W^T + bias)                         vision_transformer.py:88-91     */
#define SM_EPI_RELU 2      /* C = relu(A W^T + bias)                             transformer_decoder.py:293      */
#define SM_EPI_RESIDUAL 3  /* C = R + (A W^T + bias)                             vision_transformer.py:168-169   */
#define SM_EPI_SIGMOID2 4  /* C = A W^T (+bias); C2 = sigmoid(C)                 maskformer.py:223               */
#define SM_EPI_PATCH 5     /* patch-embed: row m -> token (m/n)*(n+1)+1+m%n, + pos_embed[1+m%n]   v_t.py:184-188,280 */
#define SM_EPI_RESIDUAL_LN 6 /* C = R + (A W^T + bias); C2 = LayerNorm(C) * ln_gamma + ln_beta in F16X2: the residual add and
                               the NEXT block's pre-norm in one launch (vision_transformer.py:168-169 followed by :165 /
                               :169).  sm_gemm_f16x2 only, N == 384, the 64 x 384 full-row tile                       */

typedef struct sm_gemm_args {
    const float* A;      /* [batch][M][lda]   activations, K contiguous                                         */
    const float* W;      /* [batch][N][ldw]   torch Linear weight layout (out_features x in_features)           */
    const float* bias;   /* [N] or NULL                                                                          */
    float* C;            /* [batch][M][ldc]                                                                      */
    const float* R;      /* residual [batch][M][ldr] (SM_EPI_RESIDUAL; may alias C) or pos_embed (SM_EPI_PATCH)  */
    float* C2;           /* second output (SM_EPI_SIGMOID2) or NULL                                              */
    const float* A_alt;  /* optional second A operand (same M, K, lda): output columns n >= alt_from_n use it.
                            Lets one launch compute [q|k] = (tgt+qpos) W_qk^T and v = tgt W_v^T of the decoder
                            self-attention (transformer_decoder.py:271-276)                                        */
    int64_t strideA, strideW, strideC, strideR; /* batch strides in elements (0 = shared)                        */
    int32_t M, N, K;     /* any M, N >= 1; K % 32 == 0                                                           */
    int32_t lda, ldw, ldc, ldr;
    int32_t batch;       /* >= 1                                                                                 */
    int32_t epilogue;    /* SM_EPI_*                                                                             */
    int32_t alt_from_n;  /* multiple of 128 (0 = A_alt unused)                                                   */
    int32_t split_k;     /* 0/1 = off; S > 1: blockIdx.z = K-slice, slice s writes its raw partial products to
                            C + s*strideC (batch must be 1, epilogue SM_EPI_BIAS with bias == NULL, K/32 % S == 0);
                            the consumer sums the slices in a fixed order (sm_layernorm_rows_f32 partials)          */
    int32_t patch_n;     /* SM_EPI_PATCH: patches per image n                                                    */
    const float* ln_gamma; /* SM_EPI_RESIDUAL_LN: weight / bias (384) and eps of the fused LayerNorm; its F16X2 output */
    const float* ln_beta;  /* goes to C2 (row stride ldc)                                                              */
    float ln_eps;
    float w_scale;         /* sm_gemm_w16 only: 2^-s of the W16 weight tensor (sm_split_w16 scale = 2^s); the accumulator
                              is multiplied by it before the bias                                                       */
    /* sm_gemm_w16 only - LayerNorm folded into the neighbouring GEMMs (vision_transformer.py:164-170: x + proj(...) -> norm2 -> fc1,
     * x + fc2(...) -> next norm1 -> qkv) instead of a launch of its own:
     *   producer (SM_EPI_RESIDUAL, N = 384): C2 = the F16X2 copy of the new residual stream (row stride ldc), ln_stats_out =
     *     (rows, 12, 2) floats: (mean, M2) of every 32-column segment of every row, written in a fixed order;
     *   consumer (BIAS / GELU / RELU, K = 384): A = that raw F16X2 stream, W = the weight times the norm's gain (W16), bias =
     *     b + W beta, ln_c = row sums of W' (N floats), ln_stats = the producer's statistics, ln_eps: the epilogue evaluates
     *     LN(x) W^T + b = r (x W'^T - mu c) + b'. */
    const float* ln_stats;
    const float* ln_c;
    float* ln_stats_out;
    int32_t mfma_terms;    /* sm_gemm_w16 only: 0 or 3 = the fp32-grade product (three f16 MFMAs per 32-k step); 1 = "throughput
                              mode" (SURVEY.md 7.2 (b)): hi x hi only, plain f16 operands, fp32 accumulate - a diagnostic of the
                              kernel structure without the x3, two orders of magnitude outside the 1e-4 gate, never the metric */
} sm_gemm_args;

/* C = epilogue(A W^T): replaces every F.linear / conv-as-GEMM / bmm on the path
 * (vision_transformer.py:113,131,89-93,186; transformer_decoder.py:271-293; maskformer.py:223,265-268). */
int sm_gemm_f32(const sm_gemm_args* args, void* stream);
/* same with the workgroup tile forced (bm x bn in {128x128, 128x64, 64x64}); used by the parity tests to cover
 * every instantiation and by tuning runs */
int sm_gemm_f32_tile(const sm_gemm_args* args, int bm, int bn, void* stream);
/* the tile sm_gemm_f32 would launch for this shape (host-side heuristic, no GPU work) */
int sm_gemm_f32_pick_tile(const sm_gemm_args* args, int* bm, int* bn);

/* ---- fp32-grade GEMM on the f16 matrix cores (split operands) -----------------------------------------------------
 * F16X2 format of a row of K floats (K % 8 == 0): per group of 8 consecutive k, 16 B of hi = f16(x) followed by 16 B
 * of lo = f16((x - hi) * 2048).  Same bytes and strides as the fp32 row (4 B / element).  x*y is evaluated as
 * hi*hi + 2^-11 (hi*lo + lo*hi): three v_mfma_f32_32x32x16_f16 per 16-deep step instead of eight fp32 MFMAs, 22
 * significant bits kept (through the whole network as close to fp64 as the fp32 reference is). */
int sm_split_f16x2(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int32_t K, void* stream);
/* same contract as sm_gemm_f32_tile, but A, A_alt and W hold F16X2 data; out_f16x2 != 0 writes C in F16X2 too
 * (BIAS / GELU / RELU epilogues, N % 8 == 0) so it can feed the next GEMM without a conversion pass */
int sm_gemm_f16x2_tile(const sm_gemm_args* args, int out_f16x2, int bm, int bn, void* stream);
/* tile / pipeline depth chosen per shape */
int sm_gemm_f16x2(const sm_gemm_args* args, int out_f16x2, void* stream);
int sm_gemm_f16x2_pick_tile(const sm_gemm_args* args, int* bm, int* bn, int* nst);

/* ---- weight GEMMs with ONE accumulator (W16 weights) ------------------------------------------------------------
 * W16 format of a weight row: the tensor is scaled by a power of two 2^s chosen by the host so that max|W * 2^s| lies in
 * [2^13, 2^14); per group of 8 k, 16 B of wh = f16(w') then 16 B of wl = f16(w' - wh) - NOT scaled by 2^11, the scaled
 * tensor keeps wl a normal f16.  a*w' = ah*wh + ah*wl + al'*(wh*2^-11) then sums in one fp32 accumulator (A stays in the
 * F16X2 format): half the accumulator registers of sm_gemm_f16x2, which buys 256 x 128 workgroup tiles (fewer staged
 * bytes per MFMA).  Same F.linear call sites as sm_gemm_f32 (vision_transformer.py:113,131,89-93,186;
 * transformer_decoder.py:271-293; maskformer.py:265-268); W must be a weight (batch 1). */
int sm_split_w16(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int32_t K, float scale,
                 void* stream);
/* variant: 0 = 256x128 (8 waves, 16-k stages x3), 1 = 256x128 (16 waves, 32-k x2), 2 = 128x128 (8 waves), 3 = 128x128
 * (4 waves, 16-k x3), 4 = 64x64, 6 = 256x128 (8 waves, 32-k x2), 7 = 128x64 */
int sm_gemm_w16_tile(const sm_gemm_args* args, int out_f16x2, int variant, void* stream);
int sm_gemm_w16(const sm_gemm_args* args, int out_f16x2, void* stream);
int sm_gemm_w16_pick(const sm_gemm_args* args); /* the variant sm_gemm_w16 launches for this shape */
const char* sm_gemm_w16_variant_name(int variant); /* kernel instantiation name as rocprofv3 prints it, or NULL */

/* y = LayerNorm(x) over the last dim (cols == 384): nn.LayerNorm at vision_transformer.py:165,169,299 (eps 1e-6)
 * and transformer_decoder.py:280,290,295,139 (eps 1e-5).  x/y row strides in elements; y may alias x. */
int sm_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                     int32_t rows, int32_t cols, float eps, void* stream);

/* General form.  Grouped row remap on either side: logical row r reads x row (r/gi)*si + oi + r%gi and writes y row
 * (r/go)*so + oo + r%go (group size 0 = identity) - used to drop the cls token while normalising
 * (maskformer.py:107-108: [:, 1:, :]) and to scatter decoder layer l into the (B,L,nq,384) stack
 * (transformer_decoder.py:138-147 + maskformer.py:141).  Optional second output y2[r] = y[r] + add[r % add_rows]:
 * the decoder's "tgt + query_pos" operand (transformer_decoder.py:271,283 with_pos_embed) produced by the LayerNorm
 * that writes tgt, so the following projection GEMM reads it with plain LDS-DMA. */
typedef struct sm_row_map { int32_t group, stride, offset; } sm_row_map;
typedef struct sm_ln_args {
    const float* x;  int64_t ldx;  sm_row_map in_map;   /* with n_partials > 0: slice 0 of a split-K GEMM output    */
    const float *gamma, *beta;
    float* y;        int64_t ldy;  sm_row_map out_map;  /* y may be NULL when only ys is wanted                   */
    float* y2;       int64_t ldy2;          /* NULL = none; indexed by the logical row r */
    const float* add; int32_t add_rows;      /* (add_rows,384), row stride 384 */
    int32_t rows;
    float eps;
    /* optional split-K reduction fused in front of the normalisation (decoder linear2, K = 1536):
     * value = (sum_{s < n_partials} x[s*partial_stride + row]) + pre_bias + residual[row], summed in slice order */
    int32_t n_partials;          /* 0 = off */
    int64_t partial_stride;      /* elements between slices */
    const float* pre_bias;       /* (384) */
    const float* residual;       /* (rows,384), row stride ldx */
    /* F16X2 outputs for the split-operand GEMM (sm_gemm_f16x2): */
    float* ys;                   /* NULL or F16X2 copy of y, same row map / stride as y (y itself may then be NULL)  */
    int32_t y2_f16x2;            /* != 0: y2 is written in F16X2 instead of fp32                                     */
    float* raw;                  /* NULL, or (rows,384) fp32 with x's row map / stride: the value BEFORE normalisation (with
                                    n_partials > 0 the reduced sum + pre_bias + residual: a pre-norm block's new residual
                                    stream; may alias `residual`)                                                       */
    /* optional SECOND LayerNorm chained onto the first, in the same launch: chain_y[map(r)] = LN(y[r]; chain_gamma, chain_beta,
     * chain_eps) (+ its F16X2 copy chain_ys) - the decoder's shared final norm applied to every layer's output
     * (transformer_decoder.py:138-139 `self.norm(output)` right after norm3, :295), same bits as a launch of its own on y */
    const float *chain_gamma, *chain_beta; /* NULL = no chained norm */
    float* chain_y;              /* row stride chain_ldy, rows scattered by chain_map; may be NULL when only chain_ys is wanted */
    float* chain_ys;
    int64_t chain_ldy;
    sm_row_map chain_map;
    float chain_eps;
} sm_ln_args;
int sm_layernorm_rows_f32(const sm_ln_args* args, void* stream);

/* y[b*rows_per + i, :] = src[i, :] for b < B  (decoder init: tgt + query_pos with tgt = 0, maskformer.py:130-135) */
int sm_broadcast_rows_f32(const float* src, float* dst, int32_t rows_per, int32_t B, void* stream);

/* softmax(scale * Q K^T) V per (batch, head), head_dim 64: the q@k^T -> softmax -> @v core of
 * Attention.forward (vision_transformer.py:122-130) and of nn.MultiheadAttention (transformer_decoder.py:273,283).
 * Element (b, row, head, d) of X lives at X + b*strideXb + row*strideXr + head*64 + d. */
typedef struct sm_attn_args {
    const float *Q, *K, *V;
    float* O;
    int64_t sQb, sQr, sKb, sKr, sVb, sVr, sOb, sOr;
    int32_t batch, heads, n_q, n_k;
    float scale;
    int32_t out_f16x2; /* != 0: O is written in the F16X2 split format (it only feeds the next projection GEMM) */
} sm_attn_args;
int sm_attention_f32(const sm_attn_args* args, void* stream);
/* Same operation with Q, K and V given in the F16X2 split format (as sm_gemm_f16x2 writes them with out_f16x2 = 1):
 * three f16 MFMAs per product (hi*hi, hi*lo, lo*hi), probabilities split in registers - fp32-grade results at several
 * times the fp32 MFMA rate. Strides/pointers in float units, multiples of 8 (one F16X2 group), 32-B aligned. */
int sm_attention_f16x2(const sm_attn_args* args, void* stream);

/* Fused QKV projection + attention of one encoder block - Attention.forward up to (excluding) the output projection
 * (vision_transformer.py:113-131: self.qkv(x) -> reshape/permute -> (q @ k^T) * scale -> softmax -> @ v -> merge heads).
 * One workgroup per (image, head): K and V of the head are produced into LDS, Q into registers; the (B*N, 1152) QKV
 * tensor is never written.  Xn = the block's LayerNorm output in F16X2, Wqkv = the (1152, 384) qkv weight in W16 with its
 * 2^-s, bias (1152) fp32; O (B*N, 384) head-major columns, fp32 or F16X2.  N <= sm_qkv_attention_max_tokens() (208:
 * the ViT-S/16 224^2 shape, 197 tokens); larger grids use sm_gemm_w16 + sm_attention_f16x2. */
typedef struct sm_qkv_attn_args {
    const float* Xn;    /* [B*N][ldx] F16X2 */
    const float* Wqkv;  /* [1152][384] W16 (sm_split_w16) */
    const float* bias;  /* [1152] */
    float* O;           /* [B*N][ldo] */
    int64_t ldx, ldo;   /* row strides in elements, multiples of 8 */
    int32_t B, N;
    float w_scale;      /* 2^-s of Wqkv */
    float scale;        /* softmax scale: head_dim ** -0.5 = 0.125 (vision_transformer.py:104) */
    int32_t out_f16x2;
    int32_t mfma_terms; /* 0 or 3: fp32-grade products; 1: throughput-mode diagnostic (hi x hi only), see sm_gemm_args */
    const float* ln_stats; /* non-NULL: LayerNorm (norm1) folded into the projection, as sm_gemm_args.ln_stats: Xn is the RAW F16X2   */
    const float* ln_c;     /*   residual stream, Wqkv carries the gain, bias = b + W beta, ln_c = the 1152 row sums of W'             */
    float ln_eps;
} sm_qkv_attn_args;
int sm_qkv_attention_w16(const sm_qkv_attn_args* args, void* stream);
int sm_qkv_attention_max_tokens(void);

/* im2col of non-overlapping PxP patches with zero padding to a multiple of P (make_input_divisible,
 * vision_transformer.py:260-267; PatchEmbed conv :182-188): img (B,3,H,W) -> cols (B*gh*gw, 3*P*P), k=(c,i,j). */
int sm_im2col_patches_f32(const float* img, float* cols, int32_t B, int32_t H, int32_t W, int32_t P, void* stream);
/* same, cols written in the F16X2 split format */
int sm_im2col_patches_f16x2(const float* img, float* cols, int32_t B, int32_t H, int32_t W, int32_t P, void* stream);

/* tokens[b,0,:] = cls + pos[0]   (vision_transformer.py:276-280) */
int sm_cls_rows_f32(const float* cls, const float* pos, float* tokens, int32_t B, int32_t N, void* stream);

/* bicubic (A=-0.75, align_corners=False) resize of the trained position grid, cls slot copied:
 * interpolate_pos_encoding (vision_transformer.py:377-401).  pos_in (1+g0*g0,384) -> pos_out (1+gh*gw,384). */
int sm_pos_embed_bicubic_f32(const float* pos_in, int32_t g0, float* pos_out, int32_t gh, int32_t gw, void* stream);

/* bilinear x2 (align_corners=False) of the patch-token grid, channels-last in and out:
 * forward_pixel_decoder (maskformer.py:158-161).  tok element (b,p,c) at tok + b*strideb + p*384 + c;
 * up (B, 2gh*2gw, 384). */
int sm_upsample2x_tokens_f32(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw,
                             void* stream);
/* same, up written in the F16X2 split format (it is the W operand of the mask GEMM) */
int sm_upsample2x_tokens_f16x2(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw,
                               void* stream);

/* Bilinear x2 (align_corners=False) of low-resolution mask logits (planes, gh, gw) -> logits (planes, 2gh, 2gw)
 * (optional, may be NULL) and prob = sigmoid(logits).  maskformer.py:144-162,219-225 computes einsum(Q, up(tokens));
 * both maps are linear, so the forward evaluates up(einsum(Q, tokens)) - same taps and weights, a quarter of the FLOPs. */
int sm_upsample2x_logits_sigmoid_f32(const float* low, float* logits, float* prob, int64_t planes, int32_t gh, int32_t gw,
                                     void* stream);
/* the same three with the model's scale_factor (maskformer.py:23,161: F.interpolate(scale_factor=s, mode="bilinear")); s = 2 is
 * what the functions above do, s = 1 the identity (+ sigmoid): outputs are (s gh) x (s gw) */
int sm_upsample_tokens_f32(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw, int32_t scale, void* stream);
int sm_upsample_tokens_f16x2(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw, int32_t scale, void* stream);
int sm_upsample_logits_sigmoid_f32(const float* low, float* logits, float* prob, int64_t planes, int32_t gh, int32_t gw,
                                   int32_t scale, void* stream);

/* objectness = sigmoid(h . w3 + b3) for each row of h (rows,384): last layer of MLP + sigmoid (maskformer.py:231-239) */
int sm_rowdot_sigmoid_f32(const float* h, const float* w, const float* b, float* out, int32_t rows, void* stream);

/* features[b,:] = mean_q queries[b, last_layer, q, :]   (maskformer.py:198-203) */
int sm_query_mean_f32(const float* queries, float* features, int32_t B, int32_t L, int32_t nq, void* stream);

/* ---- input pipeline: decoded uint8 images -> the model's normalised fp32 NCHW input ---------------------
 * Replaces, per image, the host-side tail of the reference's data path: [T.Resize((S, S)) on the PIL image ->]
 * ToTensor -> Normalize(mean, std)  (datasets/base_dataset.py:228-256, duts.py:108-147, app.py:198-205). */
typedef struct sm_pre_image {
    int64_t off;      /* byte offset of this image's (H, W, 3) interleaved uint8 RGB pixels inside `in` */
    int64_t out_off;  /* sm_preprocess_normalize_u8: element offset of this image's (3, H, W) block inside `out` */
    int32_t H, W;
    int32_t coef_x, coef_y; /* sm_preprocess_resize_u8: int32 offsets of this image's horizontal / vertical tables in `coef`:
                               [S][2] (first input index, tap count) followed by [S][ks] fixed-point taps (22 bits) */
    int32_t ksx, ksy;       /* taps per output pixel in those tables */
} sm_pre_image;
/* Pillow-exact BILINEAR resize of B images to S x S (horizontal pass -> uint8 -> vertical pass, 22-bit fixed-point taps
 * computed by the host as Pillow's precompute_coeffs does) fused with ToTensor + Normalize through `lut` (768 floats:
 * lut[c*256 + v] = (v / 255 - mean_c) / std_c evaluated in fp32 by the host).  tmp: B x tmp_stride bytes of scratch
 * (>= max_h * S * 3 each); out (B, 3, S, S) fp32. */
int sm_preprocess_resize_u8(const uint8_t* in, const sm_pre_image* images, const int32_t* coef, const float* lut, uint8_t* tmp,
                            int64_t tmp_stride, float* out, uint8_t* resized_u8, int32_t B, int32_t S, int32_t max_h, void* stream);
/* resized_u8: NULL, or (B, S, S, 3) to also receive the resized uint8 RGB image (the bilateral solver's reference image) */
/* native resolution (the reference's test mode): ToTensor + Normalize only; image b -> out + images[b].out_off as (3, H, W) */
int sm_preprocess_normalize_u8(const uint8_t* in, const sm_pre_image* images, const float* lut, float* out, int32_t B,
                               int32_t max_pixels, void* stream);
/* native resolution, images of ONE token grid batched: out (B, 3, Hp, Wp), image b in the top-left corner, zeros to the right and
 * below - the tensor make_input_divisible (vision_transformer.py:260-267) builds from the normalised image, for every image of
 * the batch at once (H <= Hp, W <= Wp; Hp, Wp = the sizes rounded up to the patch size) */
int sm_preprocess_normalize_pad_u8(const uint8_t* in, const sm_pre_image* images, const float* lut, float* out, int32_t B,
                                   int32_t Hp, int32_t Wp, void* stream);

/* serving selection (SelfMaskInference.predict, app.py:266-284): best[b] = argmax_q objectness[b][q] (first maximum),
 * out[b] = clip(masks[b][best[b]], 0, 1); masks: image b, query q at + b*mask_stride_b + q*hw (the last decoder layer) */
int sm_pick_mask_f32(const float* masks, int64_t mask_stride_b, const float* objectness, int64_t obj_stride_b, float* out,
                     int32_t* best, int32_t B, int32_t nq, int32_t hw, void* stream);

/* ---- evaluator post-processing + metrics ------------------------------------------- */
typedef struct sm_eval_image {
    int64_t gt_off; /* byte offset of this image's ground truth (H*W bytes, 0 / non-zero) inside `gt` */
    int32_t H, W;   /* ground-truth (= output) size */
} sm_eval_image;

typedef struct sm_eval_args {
    const float* mask_pred;      /* last decoder layer's probabilities: image b, query q at + b*mask_stride_b + q*mh*mw */
    int64_t mask_stride_b;
    const float* objectness;     /* last layer: image b, query q at + b*obj_stride_b + q                             */
    int64_t obj_stride_b;
    const uint8_t* gt;           /* concatenated ground-truth masks                                                  */
    const sm_eval_image* images; /* device array [B]                                                                 */
    const float* thresholds;     /* device array [255] = float32 arange(0, 1, 1/255) (metrics/f_measure.py:65)       */
    float* rows;                 /* out [B][16]: 0-6 metrics of the arg-max-objectness mask, 7-13 of the upper-bound
                                    mask, each in the order iou, pixel_acc, f_score, f_max, f_mean, mae, s_measure
                                    (evaluator.pyc@L276 header order); 14 = picked query, 15 = upper-bound query      */
    float* ious;                 /* out [B][nq] per-query IoU against the GT, or NULL                                 */
    void* workspace;             /* sm_evaluate_workspace_bytes(B, nq, mh, mw, max_pixels), 256-B aligned             */
    size_t workspace_bytes;
    int32_t B, nq, mh, mw;       /* nq <= SM_EVAL_MAX_QUERIES (960: passes of 32 queries), any mask width                 */
    int32_t max_pixels;          /* largest H*W among the batch's ground truths (<= 2048*2048): sizes the launch grid */
    float scale;                 /* > 0: reference mode F.interpolate(scale_factor=scale)[..., :H, :W]
                                    (evaluator.pyc@L209-211: 4 for ViT-S/8); 0: resize to (H, W) (batched mode)       */
} sm_eval_args;

/* Per image: bilinear up-sample of the nq query masks (align_corners=False), upper-bound query = arg-max IoU vs GT
 * (evaluator.pyc@L101-134,216), pick = arg-max objectness (@L219-221), then compute_iou / FMeasure (f_measure,
 * f_max over 255 thresholds, f_mean) / compute_mae / compute_pixel_accuracy / SMeasure (metrics/ *.py) for both. */
size_t sm_evaluate_workspace_bytes(int32_t B, int32_t nq, int32_t mh, int32_t mw, int32_t max_pixels);
int sm_evaluate_masks_f32(const sm_eval_args* args, void* stream);

/* Refinement glue (BASELINE.json configs[2]: forward -> picked mask -> bilateral solver -> metrics).
 * sm_upsample_selected_f64: out[b] (OH, OW) fp64 = F.interpolate(masks[b][q_b], size=(OH, OW), mode="bilinear",
 * align_corners=False) with q_b = (int)rows[b][sel_col] as sm_evaluate_masks_f32 wrote it (14: arg-max objectness, 15:
 * upper bound) - the `target` of bilateral_solver_output (bilateral_solver.py:152,181).
 * sm_mask_u8_to_f32: the solver's 0/1 bytes as an fp32 one-query mask for sm_evaluate_masks_f32. */
int sm_upsample_selected_f64(const float* masks, int64_t mask_stride_b, const float* rows, int32_t sel_col, double* out,
                             int32_t B, int32_t mh, int32_t mw, int32_t OH, int32_t OW, void* stream);
int sm_mask_u8_to_f32(const uint8_t* src, float* dst, int64_t n, void* stream);

/* ---- pseudo-mask voting (BASELINE.json configs[4]) -------------------------------------------------------
 * masks (M, H, W) uint8 0/1, M <= 64 candidate masks of ONE image.  filter_masks (utils/misc.py:285-314; mask_to_bbox
 * :269-282): a candidate is dropped when it predicts nothing, when (remove_long) its bounding box spans the full height or
 * the full width, when (remove_small_large) its area is under 5 % of the image or its box over 95 %.  vote_mask
 * (datasets/mask_generator.pyc@L202-230): iou[i][j] = |mi & mj| / (|mi | mj| + 1e-7) over the survivors, score = row sum,
 * best = the highest score.  Outputs (device): keep[M] 0/1, iou (M, M) fp32 (0 for dropped rows / columns), row_sums[M]
 * (-1 for dropped masks), best = index into the ORIGINAL list (-1 if nothing survives). */
size_t sm_vote_workspace_bytes(int32_t M, int32_t H, int32_t W);
int sm_vote_masks_u8(const uint8_t* masks, int32_t M, int32_t H, int32_t W, int32_t remove_long, int32_t remove_small_large,
                     int32_t* keep, float* iou, float* row_sums, int32_t* best, void* workspace, size_t workspace_bytes,
                     void* stream);
/* B images of one size in one launch sequence: masks (B, M, H, W), keep (B, M), iou (B, M, M), row_sums (B, M), best (B),
 * workspace >= B * sm_vote_workspace_bytes(M, H, W).  Image b's results are those of its own sm_vote_masks_u8 call. */
int sm_vote_masks_batch_u8(const uint8_t* masks, int32_t B, int32_t M, int32_t H, int32_t W, int32_t remove_long,
                           int32_t remove_small_large, int32_t* keep, float* iou, float* row_sums, int32_t* best, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Run-length form of B binary masks (B, H, W) uint8 (any non-zero byte = 1) in COCO's order - column-major, pycocotools.mask.encode
 * (mask_generator.pyc@L232-252 encodes every voted mask with it): starts (B, cap) receives, per image and ascending, the column-major
 * positions q = x * H + y (1 <= q < H W) whose pixel differs from the one before; info (B, 2) = {number of such positions (also when it
 * exceeds cap: only the first cap are stored), value of pixel 0}.  The counts of the uncompressed RLE are the differences of
 * [0, starts..., H W], with a leading 0 when pixel 0 is set.  sizes (device, (B, 2), or NULL): image b is the top-left sizes[b] = {H_b, W_b}
 * of its H x W plane, and its positions count in H_b. */
int sm_rle_runs_u8(const uint8_t* masks, int32_t B, int32_t H, int32_t W, const int32_t* sizes, int32_t* starts, int32_t cap, int32_t* info,
                   void* stream);
/* sm_vote_masks_batch_u8 for images of DIFFERENT sizes that share planes of H x W (images padded to one token grid, as
 * make_input_divisible pads each of them): sizes (device, (B, 2)) = {H_b, W_b} per image; candidates count only inside the top-left
 * H_b x W_b, and the filters use H_b, W_b - image b's results are those of sm_vote_masks_u8 on its cropped candidates.  NULL: as above. */
int sm_vote_masks_sized_u8(const uint8_t* masks, int32_t B, int32_t M, int32_t H, int32_t W, const int32_t* sizes, int32_t remove_long,
                           int32_t remove_small_large, int32_t* keep, float* iou, float* row_sums, int32_t* best, void* workspace,
                           size_t workspace_bytes, void* stream);
/* labels (B, n_sizes, lh * lw) int32 of sm_spectral_cluster_f32 (or any clusterer) -> masks (B, sum of cluster_sizes, H, W) uint8 in one
 * launch: sm_labels_to_masks_u8 for every (image, cluster size).  cluster_sizes: HOST array of n_sizes <= 8 entries. */
int sm_labels_to_masks_batch_u8(const int32_t* labels, int32_t B, int32_t n_sizes, const int32_t* cluster_sizes, int32_t lh, int32_t lw,
                                int32_t scale, int32_t H, int32_t W, uint8_t* masks, void* stream);

/* ---- candidate masks of the pseudo-mask generator, DINO branch (mask_generator.pyc@L136-200) --------------------
 * features = F.interpolate(tokens, scale_factor=2, mode="bilinear", align_corners=True): tok (B, gh*gw, 384) with batch stride
 * strideb (elements) -> up (B, (s gh)*(s gw), 384) */
int sm_upsample_tokens_aligned_f32(const float* tok, int64_t strideb, float* up, int32_t B, int32_t gh, int32_t gw, int32_t scale,
                                   void* stream);
/* clusterer(features, k): Lloyd's k-means (farthest-point initial centres, `iters` iterations, every sum in a fixed order) on
 * B sets of n points of 384 floats; labels (B, n) int32, centers (B, k, 384) or NULL, workspace >= B * n floats.  The reference's
 * `clusterings` module is absent from its repository in every form: this is a stated stand-in (its cluster_type="kmeans" option),
 * parity UNPINNED (oracle/cluster_oracle.py restates it; scikit-learn from the same initial centres is the third-party check). */
int sm_kmeans_f32(const float* feat, int32_t B, int32_t n, int32_t k, int32_t iters, int32_t* labels, float* centers, float* workspace,
                  void* stream);
/* to_one_hot (utils/misc.py:10-35) + F.interpolate(scale_factor=scale, mode="nearest")[..., :H, :W]: labels (lh, lw) int32 ->
 * masks (k, H, W) uint8 {0, 1} */
int sm_labels_to_masks_u8(const int32_t* labels, int32_t lh, int32_t lw, int32_t scale, int32_t k, int32_t H, int32_t W, uint8_t* masks,
                          void* stream);

/* clusterer(features, k) with cluster_type="spectral": the choice of the shipped configuration
 * (configs/duts-dino-k234-nq20-224-swav-mocov2-dino-p16-sr10100.yaml:11-12 `k: [2,3,4]`, `clustering_mode: "spectral"`;
 * mask_generator.pyc@L30-38,160; BASELINE.json configs[4] "faiss k-NN affinity + eigendecomp").  The reference's `clusterings`
 * module is absent from its repository in every form - parity UNPINNED.  Built: normalised spectral clustering in the form
 * scikit-learn's SpectralClustering(affinity="precomputed") evaluates (the third-party witness, oracle/cluster_oracle.py):
 * Euclidean k-NN graph (every point + its n_neighbors - 1 nearest others) -> W = (C + C^T)/2 without self loops ->
 * L = I - D^-1/2 W D^-1/2 -> eigenvectors of the max(cluster_sizes) smallest eigenvalues (fp64, Chebyshev-filtered subspace
 * iteration, one workgroup per image) -> rows v_i / sqrt(d_i) -> for every k of cluster_sizes a k-means (farthest-point centres,
 * Lloyd to a fixed point) on the first k columns.  ONE eigen-solve serves every k.  Deterministic (no floating-point atomics). */
typedef struct sm_spectral_args {
    const float* features;        /* (B, n, 384) fp32: the up-sampled tokens; 16 <= n <= 8192, n % 4 == 0                  */
    int32_t* labels;              /* out (B, n_sizes, n) int32                                                               */
    const int32_t* cluster_sizes; /* HOST array of n_sizes entries, each 1..6                                                */
    int32_t* knn;                 /* out (B, n, n_neighbors - 1) int32 neighbour lists, nearest first, or NULL (workspace)   */
    double* eigenvalues;          /* out (B, kw) ascending, kw = max(cluster_sizes), or NULL                                 */
    double* embedding;            /* out (B, n, kw) fp64 or NULL (workspace)                                                 */
    double* residuals;            /* out (B, kw): |L v - lambda v| per returned eigenpair, or NULL                           */
    int32_t* info;                /* out (B, 4) or NULL: outer iterations, block mat-vecs, converged (0/1), Cholesky guard hit */
    void* workspace;              /* sm_spectral_workspace_bytes(B, n, n_neighbors, kw), 256-B aligned                       */
    size_t workspace_bytes;
    double tol;                   /* residual bound of the eigen-solver; <= 0: 1e-9                                          */
    int32_t B, n, n_sizes;
    int32_t n_neighbors;          /* 2..33 (scikit-learn's default: 10)                                                       */
    int32_t degree;               /* Chebyshev filter degree per outer iteration; <= 1: 24                                   */
    int32_t max_outer;            /* <= 0: 60                                                                                 */
    int32_t kmeans_max_iter;      /* <= 0: 100                                                                                */
} sm_spectral_args;
size_t sm_spectral_workspace_bytes(int32_t B, int32_t n, int32_t n_neighbors, int32_t kw); /* 0 = unsupported shape */
int sm_spectral_cluster_f32(const sm_spectral_args* args, void* stream);

/* ---- bilateral-solver refinement ---------------------------------------------------- */
typedef struct sm_bilateral_args {
    const uint8_t* img;    /* (H, W, 3) interleaved RGB = np.array(PIL image)   (bilateral_solver.py:159)            */
    const double* target;  /* (H, W) fp64 soft mask (the reference casts to np.double, :181)                         */
    double* soft;          /* out (H, W) fp64: output_solver (:186)                                                   */
    uint8_t* binary;       /* out (H, W) 0/1: binary_solver (:184-192)                                                */
    int32_t* info;         /* out [4] or NULL: vertices, PCG iterations, labelled components, chosen root (-2 = all ones) */
    void* workspace;       /* sm_bilateral_workspace_bytes(...), 256-B aligned                                        */
    size_t workspace_bytes;
    double sigma_spatial, sigma_luma, sigma_chroma; /* 16, 16, 8 (:155-157); sigma_spatial integral, <= 32            */
    double lam, a_diag_min, cg_tol, confidence;     /* 256, 1e-5, 1e-5, 0.999 (:161,170-175)                          */
    int32_t cg_maxiter;                             /* 25                                                              */
    int32_t H, W;
} sm_bilateral_args;

/* bilateral_solver_output (bilateral_solver.py:152-193): BilateralGrid (hash -> unique -> splat / blur matrices),
 * bistochastize, Jacobi-PCG solve, slice, threshold 0.5, binary_fill_holes, 4-connected label, keep the second
 * largest label (background included). */
size_t sm_bilateral_workspace_bytes(int32_t H, int32_t W, double sigma_spatial, double sigma_luma, double sigma_chroma);
int sm_bilateral_solver_f64(const sm_bilateral_args* args, void* stream);
/* n_images solves of one size in one launch sequence (blockIdx.z = image): img (n,H,W,3), target / soft (n,H,W) f64,
 * binary (n,H,W) u8, info (n,4) or NULL, workspace >= n * sm_bilateral_workspace_bytes(...).  The solver's long kernels
 * use one workgroup per image, so a batch fills the GPU where a single solve occupies one CU. */
int sm_bilateral_solver_batch_f64(const sm_bilateral_args* args, int32_t n_images, void* stream);

/* ---- whole forward --------------------------------------------------------------------------------------------- */
typedef struct sm_enc_layer {
    const float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w,
        *fc2_b;
    /* sm_weights.ln_fold (gemm_mode 2 / 3): the pre-norms folded into the GEMMs they feed (sm_gemm_args.ln_stats): W16 copies of
     * qkv.weight * norm1.weight and fc1.weight * norm2.weight (gain along the input dimension), the folded biases b + W beta, and
     * the row sums c of the gain-scaled weights AS ROUNDED to W16 (so that x W'^T - mu c cancels what the MFMAs summed) */
    const float *qkv_fw, *qkv_fb, *qkv_c, *fc1_fw, *fc1_fb, *fc1_c;
    float qkv_s, proj_s, fc1_s, fc2_s; /* gemm_mode 2: 2^-s of the W16 copies (sm_split_w16); unused otherwise */
    float qkv_fs, fc1_fs;              /* 2^-s of qkv_fw / fc1_fw */
} sm_enc_layer;

typedef struct sm_dec_layer {
    const float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b; /* self_attn.{in_proj_weight,in_proj_bias,out_proj.*}  */
    const float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b; /* multihead_attn.*                                     */
    const float *lin1_w, *lin1_b, *lin2_w, *lin2_b;
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
    float sa_in_s, sa_out_s, ca_in_s, ca_out_s, lin1_s, lin2_s; /* gemm_mode 2: 2^-s of the W16 copies */
} sm_dec_layer;

/* Pointer table over the reference's 267-tensor state_dict; tensors stay owned by the caller. */
typedef struct sm_weights {
    const float* query_embed; /* (nq,384) */
    const float* cls_token;   /* (384)    */
    const float* pos_embed;   /* (1+g0*g0,384) */
    const float* patch_w;     /* (384, 3*P*P) */
    const float* patch_b;
    sm_enc_layer enc[SM_ENC_DEPTH];
    const float *enc_norm_w, *enc_norm_b;
    sm_dec_layer dec[SM_MAX_DEC_LAYERS];
    const float *dec_norm_w, *dec_norm_b;
    const float *ffn0_w, *ffn0_b, *ffn1_w, *ffn1_b, *ffn2_w, *ffn2_b; /* objectness MLP 384->384->384->1 (or the mask head, see mask_head_ffn) */
    const float* dec_kv_w; /* packed by the host from the state_dict: rows [384:1152) of every decoder layer's      */
    const float* dec_kv_b; /*   multihead_attn.in_proj_{weight,bias}, concatenated -> (L*768, 384) / (L*768): the   */
                           /*   cross-attention K/V of ALL layers is one GEMM over the encoder memory               */
    int32_t gemm_mode;     /* (3: as 2 with ONE MFMA per product in the weight GEMMs and the encoder attention - the throughput-mode
                              diagnostic of SURVEY.md 7.2 (b), outside the 1e-4 gate, never the metric)
                              0: exact-fp32 MFMA GEMMs; 1: split-operand f16 GEMMs - every GEMM weight pointer above
                              (patch_w, qkv/proj/fc1/fc2, decoder in/out projections, linear1/2, dec_kv_w, ffn0/1)
                              then holds the F16X2 copy of the tensor (sm_split_f16x2), biases / norms stay fp32;
                              2: as 1 with the weights in the W16 format (sm_split_w16) and their 2^-s in the *_s
                              fields: single-accumulator weight GEMMs (sm_gemm_w16), activations still F16X2         */
    int32_t patch;         /* 8 or 16 */
    int32_t pos_grid;      /* g0: trained grid side (224/patch) */
    int32_t n_queries;
    int32_t n_dec_layers;
    float patch_s, ffn0_s, ffn1_s, dec_kv_s; /* gemm_mode 2: 2^-s of patch_w, ffn0_w, ffn1_w, dec_kv_w */
    float ffn2_s;          /* gemm_mode 2, mask_head_ffn only: 2^-s of ffn2_w */
    int32_t mask_head_ffn; /* 0: use_binary_classifier=True - ffn is the objectness MLP 384->384->384->1, ffn2_w (1,384)
                              fp32 (maskformer.py:55-58,227-239).  1: return_intermediate=True with
                              use_binary_classifier=False - ffn is a 384->384->384->384 MLP applied to the decoder
                              queries BEFORE the mask einsum (maskformer.py:59-66,225); ffn2_w (384,384) is then a GEMM
                              weight in the gemm_mode's format, no objectness is produced (io->objectness may be NULL) */
    int32_t normalize_before; /* 1: TransformerDecoderLayer.forward_pre in every decoder layer (transformer_decoder.py:299-327):
                                 each sub-block normalises its input, the residual stream is only normalised by the shared
                                 decoder.norm; 0 (the shipped config): forward_post (:260-297) */
    int32_t ln_fold;          /* gemm_mode 2 / 3: 1 = the *_fw / *_fb / *_c fields of every encoder layer are filled, and forwards on the
                                 small-batch path (sm_forward_io.attn_path) let the encoder's pre-norms ride on the GEMMs around them:
                                 23 of the 25 encoder LayerNorm launches disappear (norm2 of every block into proj -> fc1, norm1 of
                                 blocks 1..11 into fc2 -> qkv; block 0's norm1 and the final norm stay launches).  Same results to
                                 rounding (other summation order of the statistics) */
    int32_t scale_factor;     /* the pixel decoder's up-sampling factor (maskformer.py:23,161; YAML key scale_factor): 0 or 2 = the
                                 shipped x2, any 1..16 accepted; masks are (scale gh) x (scale gw) */
    int32_t no_objectness;    /* 1: use_binary_classifier=False without the mask head (the 3-D path, maskformer.py:219-220): the
                                 objectness tail is skipped and io->objectness may be NULL; ffn2_w is not read */
} sm_weights;

typedef struct sm_forward_io {
    const float* x;     /* (B,3,H,W) normalised image */
    int32_t B, H, W;
    float* mask_logits; /* (B,L,nq,2gh,2gw) pre-sigmoid einsum (maskformer.py:223) or NULL                       */
    float* mask_pred;   /* (B,L,nq,2gh,2gw) sigmoid                                                              */
    float* objectness;  /* (B,L,nq,1) sigmoid                                                                    */
    float* features;    /* (B,384)                                                                               */
    float* queries;     /* (B,L,nq,384) decoder outputs after decoder.norm, or NULL (debug/parity tap)           */
    float* patch_tokens;/* (B,gh*gw,384) final-LN'd encoder tokens, or NULL (debug/parity tap; encoder_only)     */
    int32_t encoder_only;
    int32_t attn_path;       /* which encoder kernels run (gemm_mode 2/3): 0 = by batch size - from B >= 16 up the LARGE-batch set
                                (fused QKV + attention kernel on token grids <= 208, LayerNorm as launches), below it the SMALL-batch
                                set (qkv GEMM + attention pair, pre-norms folded into the GEMMs around them when sm_weights.ln_fold):
                                the faster one each; 1 = always the large-batch set, 2 = always the small-batch set.  The two sets
                                differ in the last bits (other summation orders): a caller that needs bit-identical results across
                                batch sizes pins one (the Evaluator does) */
    int32_t last_layer_only; /* 1: return_intermediate=False (maskformer.py:219-220) - only the last decoder layer reaches the mask
                                einsum; mask_logits / mask_pred are then (B,1,nq,2gh,2gw).  features / queries keep all L layers */
} sm_forward_io;

/* bytes of workspace MaskFormer.forward needs for this shape */
size_t sm_forward_workspace_bytes(const sm_weights* w, int32_t B, int32_t H, int32_t W);

/* MaskFormer.forward(x) (maskformer.py:164-251) with return_intermediate=True, use_binary_classifier=True. */
int sm_maskformer_forward(const sm_weights* w, const sm_forward_io* io, void* workspace, size_t workspace_bytes,
                          void* stream);

/* In-situ kernel timing of sm_maskformer_forward (measurement aid for bench.py; not part of the reference surface).
 * While enabled, every GEMM / attention / LayerNorm launch of a forward is bracketed by two HIP events recorded on the
 * forward's own stream; sm_forward_timing_read waits for them and returns one entry per kernel family (GEMMs by their
 * workgroup tile, i.e. by kernel instantiation) summed over all forwards issued since the enable.  Diagnostic mode:
 * single caller thread, one stream at a time.  sm_spectral_cluster_f32 and sm_bilateral_solver_f64 are tapped too, per PHASE (several
 * kernels between one event pair: names "spectral: ..." / "bilateral: ..."), except the eigen-solver and the embedding k-means,
 * which are one launch each and carry their kernel names. */
typedef struct {
    char name[64];     /* kernel (template instantiation) name as rocprofv3 prints it, without the argument list */
    int32_t launches;
    double total_us;   /* sum of the launches' GPU durations (event-pair times minus overhead_us each) */
    double overhead_us; /* what an EMPTY event pair measures on the same stream (mean of 16), subtracted per launch */
    double flops;      /* algorithmic FLOPs of those launches (2*M*N*K for GEMMs, 4*nq*nk*64 per head for attention) */
    double bytes;      /* algorithmic bytes (LayerNorm: rows * 384 * 4 * 2) */
} sm_kernel_time;
int sm_forward_timing(int enable);                                   /* 1: clear + start tapping; 0: stop + free */
int sm_forward_timing_read(sm_kernel_time* out, int max_entries);    /* returns the number of entries written, < 0 on error */

#ifdef __cplusplus
}
#endif
#endif /* SELFMASK_HIP_H */
