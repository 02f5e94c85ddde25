// Input pipeline on the device: decoded uint8 HWC images -> the normalised fp32 NCHW batch the model
// takes, i.e. what the reference's data path computes on host cores per image
//     Image.open(p).convert("RGB") [-> T.Resize((S, S)) on the PIL image] -> ToTensor (/255) -> Normalize(mean, std)
// (datasets/base_dataset.py:228-256, duts.py:108-147, app.py:198-205).  JPEG decoding stays on the host (worker pool,
// selfmask_amd/pipeline.py); everything after it runs here.
//
// Resize = Pillow's ImagingResample with the BILINEAR filter (what T.Resize does to a PIL image): a triangle filter
// widened by the down-scale factor (anti-aliasing), coefficients in 22-bit fixed point, horizontal pass first, the
// intermediate image rounded and clipped to uint8, then the vertical pass.  The coefficient tables depend only on
// (input size, output size); the host computes them in double exactly as Pillow's precompute_coeffs /
// normalize_coeffs_8bpc do (pipeline.pil_resize_coeffs) and the kernels apply them in int32 - bit-exact with PIL.
// ToTensor + Normalize of a uint8 value has 256 outcomes per channel: a 768-entry fp32 table computed by the host with
// the reference's own fp32 expressions ((v / 255 - mean) / std), looked up here - bit-exact by construction.
#include "common.h"

namespace sm {

constexpr int PP_PRECISION_BITS = 32 - 8 - 2;  // Pillow: PRECISION_BITS

__device__ __forceinline__ unsigned char pp_clip8(int v) {
    v >>= PP_PRECISION_BITS;  // arithmetic shift, as the C code's table index
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[b][y][xx][c] = clip8(2^21 + sum_x in[b][y][xmin + x][c] * k[xx][x])
__global__ __launch_bounds__(256) void pp_resize_h_kernel(const unsigned char* __restrict__ in, const sm_pre_image* __restrict__ imgs,
                                                         const int* __restrict__ coef, unsigned char* __restrict__ tmp, int S,
                                                         int64_t tmp_stride) {
    const int b = blockIdx.z;
    const sm_pre_image im = imgs[b];
    const int y = blockIdx.y;
    if (y >= im.H) return;
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= S) return;
    const int* cb = coef + im.coef_x;           // [S][2] bounds then [S][ksx] coefficients
    const int xmin = cb[2 * xx], xmax = cb[2 * xx + 1];
    const int* k = cb + 2 * S + xx * im.ksx;
    const unsigned char* row = in + im.off + ((int64_t)y * im.W + xmin) * 3;
    int s0 = 1 << (PP_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int w = k[x];
        s0 += row[3 * x] * w; s1 += row[3 * x + 1] * w; s2 += row[3 * x + 2] * w;
    }
    unsigned char* o = tmp + b * tmp_stride + ((int64_t)y * S + xx) * 3;
    o[0] = pp_clip8(s0); o[1] = pp_clip8(s1); o[2] = pp_clip8(s2);
}

// vertical pass + ToTensor + Normalize: out[b][c][yy][xx] = lut[c][clip8(2^21 + sum_y tmp[b][ymin + y][xx][c] * k[yy][y])]
__global__ __launch_bounds__(256) void pp_resize_v_norm_kernel(const unsigned char* __restrict__ tmp, const sm_pre_image* __restrict__ imgs,
                                                              const int* __restrict__ coef, const float* __restrict__ lut,
                                                              float* __restrict__ out, unsigned char* __restrict__ u8_out, int S,
                                                              int64_t tmp_stride) {
    __shared__ float slut[768];
    for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
    const int b = blockIdx.z, yy = blockIdx.y;
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= S) return;
    const sm_pre_image im = imgs[b];
    const int* cb = coef + im.coef_y;
    const int ymin = cb[2 * yy], ymax = cb[2 * yy + 1];
    const int* k = cb + 2 * S + yy * im.ksy;
    const unsigned char* col = tmp + b * tmp_stride + ((int64_t)ymin * S + xx) * 3;
    int s0 = 1 << (PP_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) {
        const int w = k[y];
        const unsigned char* p = col + (int64_t)y * S * 3;
        s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
    }
    const int64_t plane = (int64_t)S * S;
    float* o = out + (int64_t)b * 3 * plane + (int64_t)yy * S + xx;
    const unsigned char r0 = pp_clip8(s0), r1 = pp_clip8(s1), r2 = pp_clip8(s2);
    o[0] = slut[r0];
    o[plane] = slut[256 + r1];
    o[2 * plane] = slut[512 + r2];
    if (u8_out) {  // the resized RGB image itself (H, W, 3): reference image of the bilateral solver
        unsigned char* u = u8_out + ((int64_t)b * plane + (int64_t)yy * S + xx) * 3;
        u[0] = r0; u[1] = r1; u[2] = r2;
    }
}

// native resolution (the reference's test mode): ToTensor + Normalize only; one image per blockIdx.y, out[b] at out_off
__global__ __launch_bounds__(256) void pp_normalize_kernel(const unsigned char* __restrict__ in, const sm_pre_image* __restrict__ imgs,
                                                          const float* __restrict__ lut, float* __restrict__ out) {
    __shared__ float slut[768];
    for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
    const sm_pre_image im = imgs[blockIdx.y];
    const int64_t npx = (int64_t)im.H * im.W;
    const unsigned char* src = in + im.off;
    float* o = out + im.out_off;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npx; p += (int64_t)gridDim.x * 256) {
        o[p] = slut[src[3 * p]];
        o[npx + p] = slut[256 + src[3 * p + 1]];
        o[2 * npx + p] = slut[512 + src[3 * p + 2]];
    }
}

// native resolution, batched by token grid: images whose sizes round up to the same (Hp, Wp) patch multiple share one batch
// (B, 3, Hp, Wp).  Rows / columns beyond an image's own size hold 0 - exactly what make_input_divisible pads the NORMALISED
// tensor with (vision_transformer.py:260-267: F.pad(x, ..., value=0) on the right and bottom), so image b of the batch is
// bit for bit the tensor the reference's batch-1 forward builds for it.
__global__ __launch_bounds__(256) void pp_normalize_pad_kernel(const unsigned char* __restrict__ in, const sm_pre_image* __restrict__ imgs,
                                                              const float* __restrict__ lut, float* __restrict__ out, int Hp, int Wp) {
    __shared__ float slut[768];
    for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
    const sm_pre_image im = imgs[blockIdx.y];
    const int64_t plane = (int64_t)Hp * Wp;
    const unsigned char* src = in + im.off;
    float* o = out + (int64_t)blockIdx.y * 3 * plane;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < plane; p += (int64_t)gridDim.x * 256) {
        const int y = (int)(p / Wp), x = (int)(p - (int64_t)y * Wp);
        float r = 0.f, g = 0.f, b = 0.f;
        if (y < im.H && x < im.W) {
            const unsigned char* q = src + ((int64_t)y * im.W + x) * 3;
            r = slut[q[0]]; g = slut[256 + q[1]]; b = slut[512 + q[2]];
        }
        o[p] = r;
        o[plane + p] = g;
        o[2 * plane + p] = b;
    }
}

}  // namespace sm

extern "C" int sm_preprocess_resize_u8(const uint8_t* in, const sm_pre_image* images, const int32_t* coef, const float* lut,
                                       uint8_t* tmp, int64_t tmp_stride, float* out, uint8_t* resized_u8, int32_t B, int32_t S,
                                       int32_t max_h, void* stream) {
    SM_REQUIRE(in && images && coef && lut && tmp && out, "sm_preprocess_resize_u8: null pointer");
    SM_REQUIRE(B > 0 && S > 0 && max_h > 0 && tmp_stride >= (int64_t)max_h * S * 3, "sm_preprocess_resize_u8: bad shape / tmp_stride");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sm::pp_resize_h_kernel, dim3((S + 255) / 256, max_h, B), dim3(256), 0, st, in, images, coef, tmp, S, tmp_stride);
    hipLaunchKernelGGL(sm::pp_resize_v_norm_kernel, dim3((S + 255) / 256, S, B), dim3(256), 0, st, tmp, images, coef, lut, out,
                       resized_u8, S, tmp_stride);
    return sm::check_launch("sm_preprocess_resize_u8");
}

extern "C" int sm_preprocess_normalize_u8(const uint8_t* in, const sm_pre_image* images, const float* lut, float* out, int32_t B,
                                          int32_t max_pixels, void* stream) {
    SM_REQUIRE(in && images && lut && out && B > 0 && max_pixels > 0, "sm_preprocess_normalize_u8: bad arguments");
    int gx = (max_pixels + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(sm::pp_normalize_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, in, images, lut, out);
    return sm::check_launch("sm_preprocess_normalize_u8");
}

extern "C" int sm_preprocess_normalize_pad_u8(const uint8_t* in, const sm_pre_image* images, const float* lut, float* out, int32_t B,
                                              int32_t Hp, int32_t Wp, void* stream) {
    SM_REQUIRE(in && images && lut && out && B > 0 && Hp > 0 && Wp > 0, "sm_preprocess_normalize_pad_u8: bad arguments");
    int64_t gx = ((int64_t)Hp * Wp + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(sm::pp_normalize_pad_kernel, dim3((int)gx, B), dim3(256), 0, (hipStream_t)stream, in, images, lut, out, Hp, Wp);
    return sm::check_launch("sm_preprocess_normalize_pad_u8");
}
