// HBM-bound kernels of InpaintGenerator.forward (reference backend/inpaint/video/model/propainter.py:321-378 and
// model/modules/sparse_transformer.py); companions of pp_kernels.hip, declared in pp_kernels.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "pp_kernels.h"

#pragma clang fp contract(off)

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int grid_for(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}
#define GRID_STRIDE(i, total) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)
#define LAUNCH(kernel, total, ...)                                                                                  \
    do {                                                                                                            \
        if ((total) <= 0) return 0;                                                                                 \
        hipLaunchKernelGGL(kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);          \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                            \
    } while (0)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float warp_coord(float pos, int size)
{
    const float d = (float)(size - 1 > 1 ? size - 1 : 1);
    const float g = 2.0f * pos / d - 1.0f;
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}
__device__ __forceinline__ float sample_bilinear(const float* __restrict__ img, int h, int w, float iy, float ix)
{
    const float fy0 = floorf(iy), fx0 = floorf(ix);
    const int y0 = (int)fy0, x0 = (int)fx0;
    const float ay = iy - fy0, ax = ix - fx0;
    const bool yin0 = y0 >= 0 && y0 < h, yin1 = y0 + 1 >= 0 && y0 + 1 < h;
    const bool xin0 = x0 >= 0 && x0 < w, xin1 = x0 + 1 >= 0 && x0 + 1 < w;
    const float nw = (yin0 && xin0) ? img[(int64_t)y0 * w + x0] : 0.f;
    const float ne = (yin0 && xin1) ? img[(int64_t)y0 * w + x0 + 1] : 0.f;
    const float sw = (yin1 && xin0) ? img[(int64_t)(y0 + 1) * w + x0] : 0.f;
    const float se = (yin1 && xin1) ? img[(int64_t)(y0 + 1) * w + x0 + 1] : 0.f;
    return nw * ((1.0f - ax) * (1.0f - ay)) + ne * (ax * (1.0f - ay)) + sw * ((1.0f - ax) * ay) + se * (ax * ay);
}

// ---------------------------------------------------------------------------------------
// EW_PP_IM2COL3: torch.cat([masked_frames, masks_in, masks_updated]) (:332-334) fused with the im2col of the encoder's
// first conv (3x3, stride 2, pad 1, 5 -> 64; :200): row (f, oy, ox), 64 columns, k = (ky*3+kx)*5 + c, 45.. zero.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pp_im2col3(const float* __restrict__ frames, const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2, int n, int H, int W,
             float* __restrict__ out)
{
    const int oh = H / 2, ow = W / 2;
    const int64_t hw = (int64_t)H * W, total = (int64_t)n * oh * ow * 16;
    GRID_STRIDE(i, total) {
        const int q = (int)(i & 15);
        const int64_t m = i >> 4;
        const int ox = (int)(m % ow), oy = (int)((m / ow) % oh), f = (int)(m / ((int64_t)ow * oh));
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            float val = 0.f;
            if (k < 45) {
                const int tap = k / 5, c = k - 5 * tap;
                const int y = 2 * oy - 1 + tap / 3, x = 2 * ox - 1 + tap % 3;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const int64_t at = (int64_t)y * W + x;
                    val = c < 3 ? frames[((int64_t)f * 3 + c) * hw + at] : ((c == 3 ? m1 : m2)[(int64_t)f * hw + at] ? 1.0f : 0.0f);
                }
            }
            v[j] = val;
        }
        *reinterpret_cast<f32x4*>(out + m * 64 + 4 * q) = v;
    }
}

// ---------------------------------------------------------------------------------------
// EW_PP_DS_FLOW: F.interpolate(flows, scale_factor=1/4, mode='bilinear', align_corners=False) / 4.0 (:341-344): the
// sample point of output (y, x) is (4y + 1.5, 4x + 1.5), i.e. the mean of a 2x2 block.  Planar [n][2][H][W] -> [n][2][H/4][W/4]
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pp_ds_flow(const float* __restrict__ src, int n2, int H, int W, float* __restrict__ dst)
{
    const int h = H / 4, w = W / 4;
    const int64_t total = (int64_t)n2 * h * w;
    GRID_STRIDE(i, total) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const int64_t p = i / ((int64_t)w * h);
        const float* s = src + p * H * W + (int64_t)(4 * y + 1) * W + 4 * x + 1;
        // torch: lambda = 0.5 on both axes: (0.5*a + 0.5*b) per row, then 0.5 / 0.5 across rows
        const float top = 0.5f * s[0] + 0.5f * s[1], bot = 0.5f * s[W] + 0.5f * s[W + 1];
        dst[i] = (0.5f * top + 0.5f * bot) / 4.0f;
    }
}

// EW_PP_DS_MASK: F.interpolate(mask, scale_factor=1/4, mode='nearest') (:345-350) of masks_in and masks_updated into
// channels 0, 1 of the per-frame mask slot [h+2*halo][w+2*halo][C] of the propagation buffer
__global__ void __launch_bounds__(256)
k_pp_ds_mask(const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2, int n, int H, int W, float* __restrict__ slots, int halo, int C)
{
    const int h = H / 4, w = W / 4, Wp = w + 2 * halo, Hp = h + 2 * halo;
    const int64_t total = (int64_t)n * h * w;
    GRID_STRIDE(i, total) {
        const int x = (int)(i % w), y = (int)((i / w) % h), f = (int)(i / ((int64_t)w * h));
        const int64_t at = ((int64_t)f * H + 4 * y) * W + 4 * x;
        float* d = slots + (((int64_t)f * Hp + y + halo) * Wp + x + halo) * C;
        d[0] = m1[at] ? 1.0f : 0.0f;
        d[1] = m2[at] ? 1.0f : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------
// EW_PP_FEATPROP_PREP: the non-conv part of one learnable propagation step (:145-153): flow-consistency mask, bilinear
// warp of the propagated feature (flow_warp, NHWC slot -> NHWC slot) and the small condition channels
// misc[0:5] = (flow_x, flow_y, valid, mask_in, mask_updated) of cond = cat[cur, warped, flow, valid, mask]
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pp_featprop_prep(const float* __restrict__ prop, const float* __restrict__ fprop, const float* __restrict__ fcheck,
                   const float* __restrict__ maskSlot, int h, int w, int halo, int C, float* __restrict__ warped, float* __restrict__ misc)
{
    const int64_t hw = (int64_t)h * w, total = hw * (C / 4);
    const int Wp = w + 2 * halo;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i % (C / 4));
        const int64_t p = i / (C / 4);
        const int x = (int)(p % w), y = (int)(p / w);
        const float fx = fprop[p], fy = fprop[hw + p];
        const float ix = warp_coord((float)x + fx, w), iy = warp_coord((float)y + fy, h);
        const float fy0 = floorf(iy), fx0 = floorf(ix);
        const int y0 = (int)fy0, x0 = (int)fx0;
        const float ay = iy - fy0, ax = ix - fx0;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
            const int yy = y0 + (corner >> 1), xx = x0 + (corner & 1);
            const float wgt = ((corner & 1) ? ax : 1.0f - ax) * ((corner >> 1) ? ay : 1.0f - ay);
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(prop + ((int64_t)(yy + halo) * Wp + xx + halo) * C + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = acc[e] + v[e] * wgt;
            }
        }
        const int64_t at = ((int64_t)(y + halo) * Wp + x + halo) * C;
        *reinterpret_cast<f32x4*>(warped + at + 4 * c4) = acc;
        if (c4 == 0) {      // one lane per pixel also writes the small channels
            const float bx = sample_bilinear(fcheck, h, w, iy, ix), by = sample_bilinear(fcheck + hw, h, w, iy, ix);
            const float dx = fx + bx, dy = fy + by;
            const float valid = dx * dx + dy * dy < 0.01f * ((fx * fx + fy * fy) + (bx * bx + by * by)) + 0.5f ? 1.0f : 0.0f;
            float* m = misc + at;
            m[0] = fx; m[1] = fy; m[2] = valid; m[3] = maskSlot[at]; m[4] = maskSlot[at + 1];
        }
    }
}

// ---------------------------------------------------------------------------------------
// EW_PP_DEFORM_COLS: DeformableAlignment.forward (:59-72) up to the contraction: offset = 3*tanh(o[0:288]) + flow.flip(1)
// repeated (dy += flow_y, dx += flow_x), mask = sigmoid(o[288:432]); deform_conv2d columns of x (128 channels, 16 offset
// groups of 8): cols[m][((ci/32)*9 + k)*32 + ci%32].  One thread per (pixel, group, tap).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pp_deform_cols(const float* __restrict__ src, const float* __restrict__ off, int ldOff, const float* __restrict__ flow, float maxMag, int h,
                 int w, int halo, int C, float* __restrict__ cols)
{
    const int64_t hw = (int64_t)h * w, total = hw * 144;
    const int Wp = w + 2 * halo;
    GRID_STRIDE(i, total) {
        const int gk = (int)(i % 144);
        const int64_t m = i / 144;
        const int g = gk / 9, k = gk - 9 * g;
        const int xx = (int)(m % w), y = (int)(m / w);
        const float* o = off + m * ldOff;
        const float dy = maxMag * tanhf(o[g * 18 + 2 * k]) + flow[hw + m], dx = maxMag * tanhf(o[g * 18 + 2 * k + 1]) + flow[m];
        const float mk = sigmoidf_(o[288 + g * 9 + k]);
        const float py = (float)(y - 1 + k / 3) + dy, px = (float)(xx - 1 + k % 3) + dx;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const int y0 = (int)fy0, x0 = (int)fx0;
        const float ly = py - fy0, lx = px - fx0;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
            const int yy = y0 + (corner >> 1), x2 = x0 + (corner & 1);
            const float wgt = ((corner >> 1) ? ly : 1.0f - ly) * ((corner & 1) ? lx : 1.0f - lx);
            if (yy >= 0 && yy < h && x2 >= 0 && x2 < w) {
                const float* p = src + ((int64_t)(yy + halo) * Wp + x2 + halo) * C + g * 8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = acc[j][e] + v[e] * wgt;
                }
            }
        }
        const int ci0 = g * 8;
        float* dst = cols + m * (9 * C) + ((ci0 / 32) * 9 + k) * 32 + (ci0 % 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = acc[j][e] * mk;
            *reinterpret_cast<f32x4*>(dst + 4 * j) = acc[j];
        }
    }
}

// ---------------------------------------------------------------------------------------
// EW_PP_LAYERNORM: nn.LayerNorm(512) (sparse_transformer.py:283-284, eps 1e-5) over tokens [t][fh][fw][C] written to a
// token grid [t][gh][gw][C] (gh >= fh, gw >= fw: the window padding of SparseWindowAttention.forward :166-175 stays zero).
// One wave per token.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pp_layernorm(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int t, int fh, int fw, int C,
               int gh, int gw, float* __restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t ntok = (int64_t)t * fh * fw;
    if (tok >= ntok) return;
    const int xx = (int)(tok % fw), yy = (int)((tok / fw) % fh), f = (int)(tok / ((int64_t)fw * fh));
    const float* s = x + tok * C;
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += s[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)C;
    float var = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = s[c] - mean; var += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var / (float)C + 1e-5f);
    float* d = y + (((int64_t)f * gh + yy) * gw + xx) * C;
    for (int c = lane; c < C; c += 64) d[c] = (s[c] - mean) * rstd * gamma[c] + beta[c];
}
// the same with the token held in registers (C = 64 * J): one read of the row -- J independent loads in flight per lane -- instead of
// three dependent passes; a lane owns the same channels and adds them in the same order, so the result is the plain kernel's bit for bit
template <int J>
__global__ void __launch_bounds__(256)
k_pp_layernorm_reg(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int t, int fh, int fw,
                   int gh, int gw, float* __restrict__ y)
{
    constexpr int C = 64 * J;
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t ntok = (int64_t)t * fh * fw;
    if (tok >= ntok) return;
    const int xx = (int)(tok % fw), yy = (int)((tok / fw) % fh), f = (int)(tok / ((int64_t)fw * fh));
    const float* s = x + tok * C;
    float v[J], g[J], b[J];
#pragma unroll
    for (int j = 0; j < J; ++j) v[j] = s[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < J; ++j) { g[j] = gamma[lane + 64 * j]; b[j] = beta[lane + 64 * j]; }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) sum += v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) { const float d = v[j] - mean; var += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var / (float)C + 1e-5f);
    float* d = y + (((int64_t)f * gh + yy) * gw + xx) * C;
#pragma unroll
    for (int j = 0; j < J; ++j) d[lane + 64 * j] = (v[j] - mean) * rstd * g[j] + b[j];
}

// EW_PP_POOL: pool_layer, a depthwise Conv2d(C, C, 4, stride 4) (:134-138,219-222) over the (padded) token grid
// [t][gh][gw][C] -> [t][ph][pw][C]
__global__ void __launch_bounds__(256)
k_pp_pool(const float* __restrict__ y, const float* __restrict__ wgt /*[C][16]*/, const float* __restrict__ bias, int t, int gh, int gw, int C,
          int ph, int pw, float* __restrict__ out)
{
    const int64_t total = (int64_t)t * ph * pw * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int px = (int)((i / C) % pw), py = (int)((i / ((int64_t)C * pw)) % ph), f = (int)(i / ((int64_t)C * pw * ph));
        float acc = bias[c];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            acc += y[(((int64_t)f * gh + 4 * py + k / 4) * gw + 4 * px + k % 4) * C + c] * wgt[c * 16 + k];
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// EW_PP_FOLD: F.fold(kernel 7, stride 3, padding 3) of token vectors [t*fh*fw][ld] into an NHWC map [t][h+2*halo][w+2*halo][C];
// normalize = 1 divides by the fold of ones (FusionFeedForward.forward :82-96), 0 is SoftComp's plain fold (:59-64).
// The vectors are TAP-MAJOR: element (ky*7 + kx)*C + ch (F.fold's own order is ch*49 + tap; the producing GEMM's weight rows are
// permuted at pack time, PpModel::pack / patch_perm): the C channels of a pixel are neighbours in the token row as they are in the
// map, so a wave's loads are runs of C floats instead of single floats 196 bytes apart.
// ---------------------------------------------------------------------------------------
// gelu != 0 (engine-level fusion, flow_engine.hip): the fold that feeds EW_PP_UNFOLD_GELU applies the GELU itself, once per map element --
// the unfold that follows copies a map element into up to nine token rows, and k_pp_unfold_gelu evaluated erff for every copy (152 M
// per launch at the 1080p strip against 28 M map elements; 61 ms per 68-frame batch, profiles/r06_third_call.log).  Same value, same
// function: the token rows are bit-identical.
__global__ void __launch_bounds__(256)
k_pp_fold(const float* __restrict__ vec, int ld, int t, int fh, int fw, int h, int w, int C, int halo, int normalize, int gelu, float* __restrict__ out)
{
    const int64_t total = (int64_t)t * h * w * C;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % w), y = (int)((i / ((int64_t)C * w)) % h), f = (int)(i / ((int64_t)C * w * h));
        float acc = 0.f;
        int cnt = 0;
        // patches (ty, tx) with 0 <= y + 3 - 3*ty <= 6
        for (int ty = (y + 3 - 6 + 2) / 3 > 0 ? (y + 3 - 6 + 2) / 3 : 0; ty < fh && 3 * ty <= y + 3; ++ty) {
            const int ky = y + 3 - 3 * ty;
            for (int tx = (x + 3 - 6 + 2) / 3 > 0 ? (x + 3 - 6 + 2) / 3 : 0; tx < fw && 3 * tx <= x + 3; ++tx) {
                const int kx = x + 3 - 3 * tx;
                acc += vec[(((int64_t)f * fh + ty) * fw + tx) * ld + (ky * 7 + kx) * C + c];
                ++cnt;
            }
        }
        float v = normalize ? acc / (float)cnt : acc;
        if (gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        out[(((int64_t)f * Hp + y + halo) * Wp + x + halo) * C + c] = v;
    }
}

// four channels per thread (C and ld multiples of 4, 16-byte aligned tensors): every load and the store are float4, a quarter of the threads
// and of the index arithmetic; per element the same patches in the same order -- the plain kernel's result bit for bit
__global__ void __launch_bounds__(256)
k_pp_fold4(const float* __restrict__ vec, int ld, int t, int fh, int fw, int h, int w, int C, int halo, int normalize, int gelu, float* __restrict__ out)
{
    const int C4 = C / 4;
    const int64_t total = (int64_t)t * h * w * C4;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(i, total) {
        const int c = 4 * (int)(i % C4);
        const int x = (int)((i / C4) % w), y = (int)((i / ((int64_t)C4 * w)) % h), f = (int)(i / ((int64_t)C4 * w * h));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int cnt = 0;
        for (int ty = (y + 3 - 6 + 2) / 3 > 0 ? (y + 3 - 6 + 2) / 3 : 0; ty < fh && 3 * ty <= y + 3; ++ty) {
            const int ky = y + 3 - 3 * ty;
            for (int tx = (x + 3 - 6 + 2) / 3 > 0 ? (x + 3 - 6 + 2) / 3 : 0; tx < fw && 3 * tx <= x + 3; ++tx) {
                const int kx = x + 3 - 3 * tx;
                const f32x4 u = *reinterpret_cast<const f32x4*>(vec + (((int64_t)f * fh + ty) * fw + tx) * ld + (ky * 7 + kx) * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += u[e];
                ++cnt;
            }
        }
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = normalize ? acc[e] / (float)cnt : acc[e];
            if (gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            r[e] = v;
        }
        *reinterpret_cast<f32x4*>(out + (((int64_t)f * Hp + y + halo) * Wp + x + halo) * C + c) = r;
    }
}

// EW_PP_UNFOLD_GELU: F.unfold of the normalised map followed by fc2's nn.GELU() (exact erf form) (:98-103), tap-major like
// EW_PP_FOLD (fc2's K positions are permuted to match):
// out[(f,ty,tx)][(ky*7 + kx)*C + ch] = gelu(map[f][3ty-3+ky][3tx-3+kx][ch]) (zero padding), columns C*49..ld-1 zero
__global__ void __launch_bounds__(256)
k_pp_unfold_gelu(const float* __restrict__ map, int t, int fh, int fw, int h, int w, int C, int ld, float* __restrict__ out)
{
    const int64_t total = (int64_t)t * fh * fw * ld;
    GRID_STRIDE(i, total) {
        const int e = (int)(i % ld);
        const int64_t tok = i / ld;
        float v = 0.f;
        if (e < C * 49) {
            const int tap = e / C, c = e - C * tap;
            const int tx = (int)(tok % fw), ty = (int)((tok / fw) % fh), f = (int)(tok / ((int64_t)fw * fh));
            const int y = 3 * ty - 3 + tap / 7, x = 3 * tx - 3 + tap % 7;
            if (y >= 0 && y < h && x >= 0 && x < w) {
                const float u = map[(((int64_t)f * h + y) * w + x) * C + c];
                v = 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
            }
        }
        out[i] = v;
    }
}

// the unfold alone (the map already went through the GELU in k_pp_fold): four floats per thread.  C and ld multiples of 4: a float4
// never straddles two taps, every address is 16-byte aligned
__global__ void __launch_bounds__(256)
k_pp_unfold_copy4(const float* __restrict__ map, int t, int fh, int fw, int h, int w, int C, int ld, float* __restrict__ out)
{
    const int ld4 = ld / 4;
    const int64_t total = (int64_t)t * fh * fw * ld4;
    GRID_STRIDE(i, total) {
        const int e = 4 * (int)(i % ld4);
        const int64_t tok = i / ld4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e < C * 49) {
            const int tap = e / C, c = e - C * tap;
            const int tx = (int)(tok % fw), ty = (int)((tok / fw) % fh), f = (int)(tok / ((int64_t)fw * fh));
            const int y = 3 * ty - 3 + tap / 7, x = 3 * tx - 3 + tap % 7;
            if (y >= 0 && y < h && x >= 0 && x < w) v = *reinterpret_cast<const f32x4*>(map + (((int64_t)f * h + y) * w + x) * C + c);
        }
        *reinterpret_cast<f32x4*>(out + 4 * i) = v;
    }
}

// EW_PP_TANH_OUT: torch.tanh(decoder(...)) (:375-376): NHWC [n][H][W][ld] (columns 0..2) -> planar [n][3][H][W]
__global__ void __launch_bounds__(256) k_pp_tanh_out(const float* __restrict__ y, int ld, int n, int H, int W, float* __restrict__ out)
{
    const int64_t hw = (int64_t)H * W, total = (int64_t)n * hw;
    GRID_STRIDE(i, total) {
        const int64_t f = i / hw, p = i - f * hw;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(f * 3 + c) * hw + p] = tanhf(y[i * ld + c]);
    }
}

extern "C" int vsr_pp_launch_im2col3(const float* frames, const uint8_t* m1, const uint8_t* m2, int n, int H, int W, float* out, void* stream)
{
    LAUNCH(k_pp_im2col3, (int64_t)n * (H / 2) * (W / 2) * 16, frames, m1, m2, n, H, W, out);
}
extern "C" int vsr_pp_launch_ds_flow(const float* src, int n2, int H, int W, float* dst, void* stream)
{
    LAUNCH(k_pp_ds_flow, (int64_t)n2 * (H / 4) * (W / 4), src, n2, H, W, dst);
}
extern "C" int vsr_pp_launch_ds_mask(const uint8_t* m1, const uint8_t* m2, int n, int H, int W, float* slots, int halo, int C, void* stream)
{
    LAUNCH(k_pp_ds_mask, (int64_t)n * (H / 4) * (W / 4), m1, m2, n, H, W, slots, halo, C);
}
extern "C" int vsr_pp_launch_featprop_prep(const float* prop, const float* fprop, const float* fcheck, const float* maskSlot, int h, int w,
                                           int halo, int C, float* warped, float* misc, void* stream)
{
    LAUNCH(k_pp_featprop_prep, (int64_t)h * w * (C / 4), prop, fprop, fcheck, maskSlot, h, w, halo, C, warped, misc);
}
extern "C" int vsr_pp_launch_deform_cols(const float* src, const float* off, int ldOff, const float* flow, float maxMag, int h, int w, int halo,
                                         int C, float* cols, void* stream)
{
    LAUNCH(k_pp_deform_cols, (int64_t)h * w * 144, src, off, ldOff, flow, maxMag, h, w, halo, C, cols);
}
extern "C" int vsr_pp_launch_layernorm(const float* x, const float* gamma, const float* beta, int t, int fh, int fw, int C, int gh, int gw,
                                       float* y, void* stream)
{
    const int64_t ntok = (int64_t)t * fh * fw;
    if (ntok <= 0) return 0;
    if (C == 512) hipLaunchKernelGGL((k_pp_layernorm_reg<8>), dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, t, fh, fw, gh, gw, y);
    else
    hipLaunchKernelGGL(k_pp_layernorm, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, t, fh, fw, C, gh, gw, y);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int vsr_pp_launch_pool(const float* y, const float* wgt, const float* bias, int t, int gh, int gw, int C, int ph, int pw, float* out,
                                  void* stream)
{
    LAUNCH(k_pp_pool, (int64_t)t * ph * pw * C, y, wgt, bias, t, gh, gw, C, ph, pw, out);
}
extern "C" int vsr_pp_launch_fold(const float* vec, int ld, int t, int fh, int fw, int h, int w, int C, int halo, int normalize, float* out,
                                  void* stream)
{
    if (C % 4 == 0 && ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(vec) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        LAUNCH(k_pp_fold4, (int64_t)t * h * w * (C / 4), vec, ld, t, fh, fw, h, w, C, halo, normalize, 0, out);
    LAUNCH(k_pp_fold, (int64_t)t * h * w * C, vec, ld, t, fh, fw, h, w, C, halo, normalize, 0, out);
}
extern "C" int vsr_pp_launch_unfold_gelu(const float* map, int t, int fh, int fw, int h, int w, int C, int ld, float* out, void* stream)
{
    LAUNCH(k_pp_unfold_gelu, (int64_t)t * fh * fw * ld, map, t, fh, fw, h, w, C, ld, out);
}
// the pair EW_PP_FOLD -> EW_PP_UNFOLD_GELU with the GELU moved into the fold (see k_pp_fold): the two launches of the fused form
extern "C" int vsr_pp_launch_fold_gelu(const float* vec, int ld, int t, int fh, int fw, int h, int w, int C, int halo, int normalize, float* out,
                                       void* stream)
{
    if (C % 4 == 0 && ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(vec) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        LAUNCH(k_pp_fold4, (int64_t)t * h * w * (C / 4), vec, ld, t, fh, fw, h, w, C, halo, normalize, 1, out);
    LAUNCH(k_pp_fold, (int64_t)t * h * w * C, vec, ld, t, fh, fw, h, w, C, halo, normalize, 1, out);
}
extern "C" int vsr_pp_launch_unfold_plain(const float* map, int t, int fh, int fw, int h, int w, int C, int ld, float* out, void* stream)
{
    if (C % 4 || ld % 4) return -2;         // (the caller keeps the unfused pair for such shapes)
    LAUNCH(k_pp_unfold_copy4, (int64_t)t * fh * fw * (ld / 4), map, t, fh, fw, h, w, C, ld, out);
}
extern "C" int vsr_pp_launch_tanh_out(const float* y, int ld, int n, int H, int W, float* out, void* stream)
{
    LAUNCH(k_pp_tanh_out, (int64_t)n * H * W, y, ld, n, H, W, out);
}
