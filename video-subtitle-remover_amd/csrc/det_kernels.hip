// Kernels of the text detector's forward pass (PP-OCRv5 detection program, reference call site
// backend/tools/subtitle_detect.py:41-58): one launcher per operator type of the PaddlePaddle
// inference program (backend/models/V5/*/inference.json), NCHW fp32 as in the program.  The host runner
// (backend/tools/ocr_det.py) walks the program and calls these through the C-ABI (include/vsr_hip.h).
//
// The networks are small mobile / HGNet-style CNNs (1.2 M and 22 M parameters) dominated by 1x1 and depthwise convs on
// maps of <= 480x272: a direct convolution with the weights on the scalar path and pixels across lanes keeps every
// global access coalesced; the op is launch- and bandwidth-bound, not a matrix-core problem.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/vsr_hip.h"

#pragma clang fp contract(off)

#define GRID_STRIDE(i, total) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)
static inline int grid_for(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (int)g;
}
#define DONE() return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP

__device__ __forceinline__ float det_act(float v, int act)
{
    if (act == 1) return fmaxf(v, 0.f);                                        // relu
    if (act == 2) return v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f;         // hardswish
    return v;
}

// conv2d, groups = 1: one thread = one output pixel x CO output channels (blockIdx.y picks the channel block, so the
// weight addresses are wave-uniform); zero padding by predicate
template <int CO>
__global__ void __launch_bounds__(256)
k_det_conv(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int Cin, int H, int W, int Cout, int kh,
           int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, int act, float* __restrict__ out)
{
    const int n = blockIdx.z, co0 = blockIdx.y * CO;
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (int64_t)Ho * Wo) return;
    const int oy = (int)(pix / Wo), ox = (int)(pix - (int64_t)oy * Wo);
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = (bias != nullptr && co0 + j < Cout) ? bias[co0 + j] : 0.f;
    const float* xn = x + (int64_t)n * Cin * H * W;
    for (int ci = 0; ci < Cin; ++ci)
        for (int ky = 0; ky < kh; ++ky) {
            const int y = oy * sh - pt + ky;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int xx = ox * sw - pl + kx;
                if (xx < 0 || xx >= W) continue;
                const float v = xn[((int64_t)ci * H + y) * W + xx];
#pragma unroll
                for (int j = 0; j < CO; ++j)
                    if (co0 + j < Cout) acc[j] += v * w[(((int64_t)(co0 + j) * Cin + ci) * kh + ky) * kw + kx];
            }
        }
#pragma unroll
    for (int j = 0; j < CO; ++j)
        if (co0 + j < Cout) out[(((int64_t)n * Cout + co0 + j) * Ho + oy) * Wo + ox] = det_act(acc[j], act);
}

// depthwise_conv2d (groups = channels, multiplier 1)
__global__ void __launch_bounds__(256)
k_det_dwconv(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int N, int C, int H, int W, int kh, int kw,
             int sh, int sw, int pt, int pl, int Ho, int Wo, int act, float* __restrict__ out)
{
    const int64_t total = (int64_t)N * C * Ho * Wo;
    GRID_STRIDE(i, total) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const int64_t nc = i / ((int64_t)Wo * Ho);
        const int c = (int)(nc % C);
        const float* xp = x + nc * H * W;
        float acc = bias != nullptr ? bias[c] : 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const int y = oy * sh - pt + ky;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int xx = ox * sw - pl + kx;
                if (xx < 0 || xx >= W) continue;
                acc += xp[(int64_t)y * W + xx] * w[((int64_t)c * kh + ky) * kw + kx];
            }
        }
        out[i] = det_act(acc, act);
    }
}

// conv2d_transpose, kernel 2x2, stride 2, no padding; weight [Cin][Cout/groups][2][2], groups 1 or Cin (depthwise)
__global__ void __launch_bounds__(256)
k_det_deconv2(const float* __restrict__ x, const float* __restrict__ w, int N, int Cin, int H, int W, int Cout, int depthwise,
              float* __restrict__ out)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)N * Cout * Ho * Wo;
    GRID_STRIDE(i, total) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const int co = (int)((i / ((int64_t)Wo * Ho)) % Cout), n = (int)(i / ((int64_t)Wo * Ho * Cout));
        const int y = oy >> 1, xx = ox >> 1, tap = (oy & 1) * 2 + (ox & 1);
        float acc = 0.f;
        if (depthwise) {
            acc = x[(((int64_t)n * Cin + co) * H + y) * W + xx] * w[(int64_t)co * 4 + tap];
        } else {
            for (int ci = 0; ci < Cin; ++ci) acc += x[(((int64_t)n * Cin + ci) * H + y) * W + xx] * w[((int64_t)ci * Cout + co) * 4 + tap];
        }
        out[i] = acc;
    }
}

// elementwise add (op 0) / multiply (op 1): b is the same shape (bmode 0), per channel [C] (1) or per (n, c) [N*C] (2)
__global__ void __launch_bounds__(256)
k_det_binary(const float* __restrict__ a, const float* __restrict__ b, int op, int64_t total, int C, int64_t HW, int bmode, float* __restrict__ out)
{
    GRID_STRIDE(i, total) {
        const int64_t nc = i / HW;
        const float bv = bmode == 0 ? b[i] : (bmode == 1 ? b[nc % C] : (bmode == 2 ? b[nc] : b[0]));   // 3: one scalar
        out[i] = op == 0 ? a[i] + bv : a[i] * bv;
    }
}

// relu (0), hardswish (1), hardsigmoid(slope p0, offset p1) (2), sigmoid (3), scale: x*p0 + p1 (4)
__global__ void __launch_bounds__(256) k_det_unary(const float* __restrict__ x, int64_t total, int kind, float p0, float p1, float* __restrict__ out)
{
    GRID_STRIDE(i, total) {
        const float v = x[i];
        float r;
        if (kind == 0) r = fmaxf(v, 0.f);
        else if (kind == 1) r = v * fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f;
        else if (kind == 2) r = fminf(fmaxf(v * p0 + p1, 0.0f), 1.0f);
        else if (kind == 3) r = 1.0f / (1.0f + expf(-v));
        else r = v * p0 + p1;
        out[i] = r;
    }
}

// y = x * scale[c] + shift[c]: inference batch_norm with (gamma / sqrt(var + eps), beta - mean * that) from the host
__global__ void __launch_bounds__(256)
k_det_affine(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, int64_t total, int C, int64_t HW,
             float* __restrict__ out)
{
    GRID_STRIDE(i, total) {
        const int c = (int)((i / HW) % C);
        out[i] = x[i] * scale[c] + shift[c];
    }
}

// adaptive average pool to 1x1: one wave per (n, c) plane
__global__ void __launch_bounds__(256) k_det_gap(const float* __restrict__ x, int64_t planes, int64_t HW, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= planes) return;
    float s = 0.f;
    for (int64_t i = lane; i < HW; i += 64) s += x[p * HW + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[p] = s / (float)HW;
}

// max pool k x k, stride s, padding (pt, pl) with -inf, output Ho x Wo
__global__ void __launch_bounds__(256)
k_det_maxpool(const float* __restrict__ x, int64_t planes, int H, int W, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo,
              float* __restrict__ out)
{
    const int64_t total = planes * Ho * Wo;
    GRID_STRIDE(i, total) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const int64_t p = i / ((int64_t)Wo * Ho);
        float m = -INFINITY;
        for (int ky = 0; ky < kh; ++ky) {
            const int y = oy * sh - pt + ky;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int xx = ox * sw - pl + kx;
                if (xx < 0 || xx >= W) continue;
                m = fmaxf(m, x[(p * H + y) * W + xx]);
            }
        }
        out[i] = m;
    }
}

// nearest_interp, integer scale s, align_corners = False: out[y][x] = in[y / s][x / s]
__global__ void __launch_bounds__(256) k_det_nearest(const float* __restrict__ x, int64_t planes, int H, int W, int s, float* __restrict__ out)
{
    const int Ho = H * s, Wo = W * s;
    const int64_t total = planes * Ho * Wo;
    GRID_STRIDE(i, total) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const int64_t p = i / ((int64_t)Wo * Ho);
        out[i] = x[(p * H + oy / s) * W + ox / s];
    }
}

// DecodeImage(BGR) + NormalizeImage(scale 1/255, mean, std, hwc) + ToCHWImage (inference.yml PreProcess): u8 HWC -> fp32 CHW
__global__ void __launch_bounds__(256) k_det_normalize(const uint8_t* __restrict__ img, int H, int W, float* __restrict__ out)
{
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t hw = (int64_t)H * W;
    GRID_STRIDE(i, hw) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c * hw + i] = ((float)img[i * 3 + c] * (1.0f / 255.0f) - mean[c]) / stdv[c];
    }
}

// ---- layout changes around the convolutions that run on the gather-GEMM (dense convs of the server program: 268 GFLOP per
// 960x544 frame, 54 % of it in 9x9 kernels).  NCHW plane -> zero-padded NHWC [Hp][Wp][Cp] (Cp a multiple of 32: the K chunks of the
// GEMM tables) and GEMM output [pixels][Np] -> NCHW, both through a 32x32 LDS tile so that reads and writes are coalesced ----
__global__ void __launch_bounds__(256)
k_det_nchw_to_nhwc(const float* __restrict__ x, int C, int H, int W, int pt, int pl, int Hp, int Wp, int Cp, float* __restrict__ out)
{
    x += (int64_t)blockIdx.z * C * H * W;                            // image of the batch
    out += (int64_t)blockIdx.z * Hp * Wp * Cp;
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t q0 = (int64_t)blockIdx.x * 32, total = (int64_t)Hp * Wp;
    const int c0 = blockIdx.y * 32;
    const int64_t q = q0 + tx;
    const int yp = (int)(q / Wp), xp = (int)(q - (int64_t)yp * Wp);
    const int y = yp - pt, xx = xp - pl;
    const bool in = q < total && y >= 0 && y < H && xx >= 0 && xx < W;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        t[j][tx] = (in && c < C) ? x[((int64_t)c * H + y) * W + xx] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t qq = q0 + j;
        if (qq < total) out[qq * Cp + c0 + tx] = t[tx][j];
    }
}

// out[c][p] = act(in[p][c] * scale[c] + shift[c]) (scale == nullptr: no affine); act: 0 none, 1 relu, 2 hardswish
__global__ void __launch_bounds__(256)
k_det_nhwc_to_nchw(const float* __restrict__ in, int C, int64_t P, int Np, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                   float* __restrict__ out)
{
    in += (int64_t)blockIdx.z * P * Np;                              // image of the batch: P pixels each
    out += (int64_t)blockIdx.z * C * P;
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int64_t p = p0 + j;
        const int c = c0 + tx;
        t[j][tx] = (p < P && c < Np) ? in[p * Np + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const int64_t p = p0 + tx;
        if (c < C && p < P) {
            float v = t[tx][j];
            if (scale) v = v * scale[c] + shift[c];
            out[(int64_t)c * P + p] = det_act(v, act);
        }
    }
}

// ---- NHWC-resident plan (round 6; backend/tools/ocr_det_nhwc.py).  Activations stay in zero-haloed NHWC buffers between the GEMM convs:
// a "view" is (pointer to channel c0 of interior pixel (0, 0) of image 0, floats per image, floats per row, floats per pixel); the halo
// around the interior and the channels a slice is padded with are zero and nothing writes them, so the kernels below need no border
// predicates and the concat of the HGNet blocks is the producers writing their channel slices of one buffer. ----

// NCHW [n][C][H][W] -> interior of a view, channels [0, Cw) (Cw a multiple of 32; zeros at and beyond C)
__global__ void __launch_bounds__(256)
k_det_to_view(const float* __restrict__ x, int C, int H, int W, int Cw, float* __restrict__ out, int64_t imgStride, int64_t rowStride, int Cs)
{
    x += (int64_t)blockIdx.z * C * H * W;
    out += (int64_t)blockIdx.z * imgStride;
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t q0 = (int64_t)blockIdx.x * 32, total = (int64_t)H * W;
    const int c0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        t[j][tx] = (q0 + tx < total && c < C) ? x[(int64_t)c * total + q0 + tx] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t q = q0 + j;
        if (q < total) {
            const int y = (int)(q / W), xx = (int)(q - (int64_t)y * W);
            out[y * rowStride + (int64_t)xx * Cs + c0 + tx] = t[tx][j];
        }
    }
}

// interior of a view, channels [0, C) -> NCHW planes out[img * outImgStride + c * H * W + pixel]
__global__ void __launch_bounds__(256)
k_det_from_view(const float* __restrict__ in, int64_t imgStride, int64_t rowStride, int Cs, int C, int H, int W, float* __restrict__ out,
                int64_t outImgStride)
{
    in += (int64_t)blockIdx.z * imgStride;
    out += (int64_t)blockIdx.z * outImgStride;
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 32, total = (int64_t)H * W;
    const int c0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int64_t p = p0 + j;
        const int c = c0 + tx;
        float v = 0.f;
        if (p < total && c < C) {
            const int y = (int)(p / W), xx = (int)(p - (int64_t)y * W);
            v = in[y * rowStride + (int64_t)xx * Cs + c];
        }
        t[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const int64_t p = p0 + tx;
        if (c < C && p < total) out[(int64_t)c * total + p] = t[tx][j];
    }
}

// depthwise conv on views, 4 channels per thread: out = act((sum_taps x * w[tap][c]) * scale[c] + shift[c]); the taps outside the image
// are the halo's zeros (pt, pl <= halo), weights tap-major [kh * kw][C]; the sum runs over (ky, kx) as k_det_dwconv's
__global__ void __launch_bounds__(256)
k_det_dwconv_view(const float* __restrict__ in, int64_t inImg, int64_t inRow, int inCs, const float* __restrict__ w, const float* __restrict__ scale,
                  const float* __restrict__ shift, int N, int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, int act,
                  float* __restrict__ out, int64_t outImg, int64_t outRow, int outCs)
{
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4) * 4;
        const int64_t pix = i / C4;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((int64_t)Wo * Ho));
        const float* xp = in + n * inImg + (int64_t)(oy * sh - pt) * inRow + (int64_t)(ox * sw - pl) * inCs + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx) {
                const float4 v = *reinterpret_cast<const float4*>(xp + ky * inRow + (int64_t)kx * inCs);
                const float4 ww = *reinterpret_cast<const float4*>(w + (int64_t)(ky * kw + kx) * C + c);
                a0 += v.x * ww.x; a1 += v.y * ww.y; a2 += v.z * ww.z; a3 += v.w * ww.w;
            }
        if (scale != nullptr) {
            const float4 s = *reinterpret_cast<const float4*>(scale + c), t = *reinterpret_cast<const float4*>(shift + c);
            a0 = a0 * s.x + t.x; a1 = a1 * s.y + t.y; a2 = a2 * s.z + t.z; a3 = a3 * s.w + t.w;
        }
        float4 o;
        o.x = det_act(a0, act); o.y = det_act(a1, act); o.z = det_act(a2, act); o.w = det_act(a3, act);
        *reinterpret_cast<float4*>(out + n * outImg + oy * outRow + (int64_t)ox * outCs + c) = o;
    }
}

// nearest_interp (integer scale s) view -> view, 4 channels per thread
__global__ void __launch_bounds__(256)
k_det_nearest_view(const float* __restrict__ in, int64_t inImg, int64_t inRow, int inCs, int N, int C, int Ho, int Wo, int s, float* __restrict__ out,
                   int64_t outImg, int64_t outRow, int outCs)
{
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4) * 4;
        const int64_t pix = i / C4;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((int64_t)Wo * Ho));
        *reinterpret_cast<float4*>(out + n * outImg + oy * outRow + (int64_t)ox * outCs + c) =
            *reinterpret_cast<const float4*>(in + n * inImg + (int64_t)(oy / s) * inRow + (int64_t)(ox / s) * inCs + c);
    }
}

// the kh x kw neighbourhoods of a few-channel NCHW map as ONE 32-float chunk per pixel of a view: channel (c * kh + ky) * kw + kx of pixel
// (y, x) = x[c][y + ky - pt][x + kx - pl] (zero outside the image), zeros beyond C * kh * kw -- the conv over concat(probability map, features)
// of the DB head then spends one K chunk on the map instead of one per tap
__global__ void __launch_bounds__(256)
k_det_im2col_view(const float* __restrict__ x, int N, int C, int H, int W, int kh, int kw, int pt, int pl, float* __restrict__ out, int64_t outImg,
                  int64_t outRow, int outCs)
{
    const int64_t total = (int64_t)N * H * W * 8;
    const int taps = C * kh * kw;
    GRID_STRIDE(i, total) {
        const int q = (int)(i & 7);
        const int64_t pix = i >> 3;
        const int xx = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((int64_t)W * H));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = q * 4 + e;
            float r = 0.f;
            if (j < taps) {
                const int c = j / (kh * kw), t = j - c * kh * kw;
                const int sy = y + t / kw - pt, sx = xx + t % kw - pl;
                if (sy >= 0 && sy < H && sx >= 0 && sx < W) r = x[(((int64_t)n * C + c) * H + sy) * W + sx];
            }
            v[e] = r;
        }
        *reinterpret_cast<float4*>(out + n * outImg + y * outRow + (int64_t)xx * outCs + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__device__ __forceinline__ float det_act4(float v, int act)
{
    if (act == 3) return 1.0f / (1.0f + expf(-v));                             // sigmoid
    return det_act(v, act);
}

// NO (1 or 4) dot products per pixel of a view over its C channels (C % 4 == 0): the 1x1 conv to one channel (NO = 1: out[n][H][W]) and
// the 2x2 / stride 2 transposed conv to one channel (NO = 4, taps (dy, dx): out[n][2H][2W]) of the DB head; 16 lanes per pixel, w [NO][C]
template <int NO>
__global__ void __launch_bounds__(256)
k_det_dots_view(const float* __restrict__ in, int64_t inImg, int64_t inRow, int inCs, int N, int C, int H, int W, const float* __restrict__ w,
                const float* __restrict__ bias, int act, float* __restrict__ out)
{
    const int sub = threadIdx.x & 15;
    const int64_t total = (int64_t)N * H * W;
    const int64_t step = (int64_t)gridDim.x * (blockDim.x >> 4);
    const int64_t rounds = (total + step - 1) / step;
    for (int64_t r = 0; r < rounds; ++r) {
        const int64_t pix = r * step + (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
        const bool ok = pix < total;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((int64_t)W * H));
        float acc[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[j] = 0.f;
        if (ok) {
            const float* xp = in + n * inImg + y * inRow + (int64_t)x * inCs;
            for (int c = sub * 4; c < C; c += 64) {
                const float4 v = *reinterpret_cast<const float4*>(xp + c);
#pragma unroll
                for (int j = 0; j < NO; ++j) {
                    const float4 ww = *reinterpret_cast<const float4*>(w + (int64_t)j * C + c);
                    acc[j] += v.x * ww.x; acc[j] += v.y * ww.y; acc[j] += v.z * ww.z; acc[j] += v.w * ww.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NO; ++j)
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
        if (ok && sub == 0) {
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                const float v = det_act4(acc[j] + (bias != nullptr ? bias[0] : 0.f), act);
                if (NO == 1) out[pix] = v;
                else out[((int64_t)n * 2 * H + 2 * y + (j >> 1)) * 2 * W + 2 * x + (j & 1)] = v;
            }
        }
    }
}

// ---- DBPostProcess, device part (inference.yml PostProcess; paddleocr DBPostProcess.boxes_from_bitmap): bitmap = prob > thresh,
// 8-connected components (what cv2.findContours' outer contours enclose), per-component area and bounding box.  Union-find
// labelling (one merge pass over the four forward neighbours with atomicMin on the parent links, then path flattening): the
// label of a component is the raster index of its first pixel, which is also the order scipy.ndimage.label / findContours visit.
__device__ __forceinline__ int ccl_find(volatile int* L, int i)
{
    int p = L[i];
    while (p != i) { i = p; p = L[i]; }
    return i;
}

__device__ __forceinline__ void ccl_unite(int* L, int a, int b)
{
    for (;;) {
        a = ccl_find(L, a);
        b = ccl_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // a > b: hang a under b
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;                                               // somebody moved a meanwhile: unite its new parent with b
    }
}

// whole waves walk the map together (64 consecutive pixels per wave and step; 256 and the grid stride are multiples of 64, so a pixel's lane
// is i & 63): the loop bound is on the wave's first pixel
#define WAVE_STRIDE(i, total) \
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i - (int64_t)(threadIdx.x & 63) < (total); i += (int64_t)gridDim.x * blockDim.x)

// labels start at the first pixel of the pixel's horizontal RUN inside its wave's 64 pixels (ballot arithmetic; a run also starts at
// x == 0): the horizontal unions of a run -- all but the ones across a wave seam, which k_ccl_merge makes -- are never executed.  A solid
// subtitle box of 60 x 670 pixels was 160 000 unions on chains that grow while they are walked (0.5 ms); it is 60 + 600 now.
__global__ void __launch_bounds__(256) k_ccl_init(const float* __restrict__ prob, float thresh, int64_t total, int W, int* __restrict__ L, int* __restrict__ st)
{
    const int lane = threadIdx.x & 63;
    WAVE_STRIDE(i, total) {
        const bool valid = i < total;
        const bool fg = valid && prob[i] > thresh;
        const unsigned long long fgm = __ballot(fg), rowStart = __ballot(valid && (i % W) == 0);
        const unsigned long long starts = fgm & (~(fgm << 1) | rowStart | 1ull);
        if (!valid) continue;
        const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        L[i] = fg ? (int)(i - lane + (63 - __clzll((long long)upto))) : -1;
        st[i * 5 + 0] = 0;                                     // area
        st[i * 5 + 1] = 0x7fffffff; st[i * 5 + 2] = -1;        // x min / max
        st[i * 5 + 3] = 0x7fffffff; st[i * 5 + 4] = -1;        // y min / max
    }
}

// the unions that are not implied by others (runs are connected: k_ccl_init + the seam union below):
//   down        (y+1, x)   unless (y, x-1) and (y+1, x-1) are foreground: (y, x-1) ~ (y+1, x-1) by this rule at x-1, the runs do the rest
//   down-left   (y+1, x-1) unless (y+1, x) is foreground (down + the lower run) or (y, x-1) is (its down union + this run)
//   down-right  (y+1, x+1) unless (y+1, x) is foreground or (y, x+1) is
// the partition is the transitive closure of the unions made, whatever their order
__global__ void __launch_bounds__(256) k_ccl_merge(int H, int W, int* __restrict__ L)
{
    const int64_t total = (int64_t)H * W;
    GRID_STRIDE(i, total) {
        if (L[i] < 0) continue;
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const bool l = x > 0 && L[i - 1] >= 0, r = x + 1 < W && L[i + 1] >= 0;
        if (l && (i & 63) == 0) ccl_unite(L, (int)i - 1, (int)i);                   // a run across the seam of two waves of k_ccl_init
        if (y + 1 < H) {
            const int64_t d = i + W;
            const bool dl = x > 0 && L[d - 1] >= 0, dd = L[d] >= 0, dr = x + 1 < W && L[d + 1] >= 0;
            if (dd && !(l && dl)) ccl_unite(L, (int)i, (int)d);
            if (dl && !dd && !l) ccl_unite(L, (int)i, (int)(d - 1));
            if (dr && !dd && !r) ccl_unite(L, (int)i, (int)(d + 1));
        }
    }
}

// every pixel takes its root; area and bounding box per root with atomics -- ONE set per wave when all its foreground pixels have the same
// root (the pixels of a wave are 64 neighbours of a row: inside a subtitle box that is every wave; 200 000 atomics on five addresses were
// 1 ms per map), per pixel otherwise
__global__ void __launch_bounds__(256) k_ccl_flatten_stats(int H, int W, int* __restrict__ L, int* __restrict__ st)
{
    const int64_t total = (int64_t)H * W;
    WAVE_STRIDE(i, total) {
        const bool fg = i < total && L[i] >= 0;
        int r = -1, x = 0, y = 0;
        if (fg) {
            r = ccl_find(L, (int)i);
            L[i] = r;                                          // roots keep L[r] == r: concurrent finds stay correct
            y = (int)(i / W); x = (int)(i - (int64_t)y * W);
        }
        const unsigned long long m = __ballot(fg);
        if (m == 0) continue;
        const int first = __ffsll((long long)m) - 1;
        const int r0 = __shfl(r, first);
        if (__ballot(fg && r != r0) == 0) {
            int x0 = fg ? x : 0x7fffffff, x1 = fg ? x : -1, y0 = fg ? y : 0x7fffffff, y1 = fg ? y : -1;
            for (int off = 32; off > 0; off >>= 1) {
                x0 = min(x0, __shfl_xor(x0, off)); x1 = max(x1, __shfl_xor(x1, off));
                y0 = min(y0, __shfl_xor(y0, off)); y1 = max(y1, __shfl_xor(y1, off));
            }
            if ((int)(threadIdx.x & 63) == first) {
                atomicAdd(&st[(int64_t)r0 * 5 + 0], __popcll(m));
                atomicMin(&st[(int64_t)r0 * 5 + 1], x0); atomicMax(&st[(int64_t)r0 * 5 + 2], x1);
                atomicMin(&st[(int64_t)r0 * 5 + 3], y0); atomicMax(&st[(int64_t)r0 * 5 + 4], y1);
            }
        } else if (fg) {
            atomicAdd(&st[(int64_t)r * 5 + 0], 1);
            atomicMin(&st[(int64_t)r * 5 + 1], x); atomicMax(&st[(int64_t)r * 5 + 2], x);
            atomicMin(&st[(int64_t)r * 5 + 3], y); atomicMax(&st[(int64_t)r * 5 + 4], y);
        }
    }
}

__global__ void __launch_bounds__(256) k_ccl_compact(int64_t total, const int* __restrict__ L, const int* __restrict__ st, int* __restrict__ comps, int cap,
                                                     int* __restrict__ count)
{
    GRID_STRIDE(i, total) {
        if (L[i] != (int)i) continue;
        const int k = atomicAdd(count, 1);
        if (k < cap) {
            comps[k * 6 + 0] = (int)i;
#pragma unroll
            for (int j = 0; j < 5; ++j) comps[k * 6 + 1 + j] = st[i * 5 + j];
        }
    }
}

// ---- DBPostProcess, polygon part on the device: the steps of PaddleX's DBPostProcess.boxes_from_bitmap (the numpy statement of the
// same steps is ocr_det.box_from_border; the reference they are held to is oracle/db_postprocess.py) for maps WITHOUT HOLES, where
// the borders cv2.findContours(RETR_LIST) returns are exactly the outer borders of the 8-connected components (k_db_quads counts
// the holes by the Euler number; a map with a hole, more than `cap` components or a component taller than 256 rows goes to the
// host).  A minimum-area rectangle only depends on the convex hull, and the hull's vertices are among the leftmost / rightmost
// pixel of every row of the component: k_db_ext collects those per (component, row) with atomics, k_db_boxes (one workgroup per
// component) sorts the <= 2 * rows points by rank, builds the hull with Andrew's monotone chain (integer cross products), runs the
// calipers with one thread per hull edge, scores the rectangle over cv2.fillPoly's raster of its integer-truncated corners (outline
// as 8-connected lines -- membership by the closed form of the line iterator's error recurrence -- plus the 16.16 fixed-point scan
// lines), grows it the way ClipperLib's rounded offset does (integer-truncated input, arc chords, vertices rounded half away from
// zero), fits the rectangle of the grown polygon and rescales it (round half to even).  All geometry in fp64 without contraction,
// the rectangle corners pass through float32 as cv2.boxPoints returns them.
#define DB_MAXROWS 256
#define DB_MAXPTS (2 * DB_MAXROWS)
#define DB_REC 16                                                  // ints per output record: flag, 8 corner coordinates, score bits
#define DB_HDR 4                                                   // out[0] components found, out[1] holes, out[2..3] reserved

__global__ void __launch_bounds__(256) k_db_slots(const int* __restrict__ comps, const int* __restrict__ count, int cap, int* __restrict__ st)
{
    const int n = *count < cap ? *count : cap;
    GRID_STRIDE(k, (int64_t)n) st[(int64_t)comps[k * 6] * 5 + 0] = (int)k;      // root pixel -> slot (the area now lives in comps)
}

__global__ void __launch_bounds__(256) k_db_ext_init(const int* __restrict__ count, int cap, int H, int* __restrict__ ext)
{
    const int n = *count < cap ? *count : cap;
    GRID_STRIDE(i, (int64_t)n * H) { ext[2 * i] = 0x7fffffff; ext[2 * i + 1] = -1; }
}

__global__ void __launch_bounds__(256) k_db_ext(int H, int W, const int* __restrict__ L, const int* __restrict__ st, const int* __restrict__ count, int cap,
                                                int* __restrict__ ext)
{
    if (*count > cap) return;                                      // the host labels a map with that many components itself
    const int64_t total = (int64_t)H * W;
    GRID_STRIDE(i, total) {
        const int r = L[i];
        if (r < 0) continue;
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const bool l = x > 0 && L[i - 1] >= 0, rr = x + 1 < W && L[i + 1] >= 0;      // a foreground neighbour is in the same component:
        if (l && rr) continue;                                                       // only the ends of a run can be a row's extremes
        const int k = st[(int64_t)r * 5 + 0];
        if (!l) atomicMin(&ext[((int64_t)k * H + y) * 2], x);
        if (!rr) atomicMax(&ext[((int64_t)k * H + y) * 2 + 1], x);
    }
}

// Bit-quad counts of the thresholded map (Gray): over all 2x2 windows of the zero-framed image, Q1 = windows with one foreground
// pixel, Q3 = with three, QD = the two diagonal pairs; Euler number (8-connected foreground) = (Q1 - Q3 - 2 QD) / 4 = components - holes.
// quads = count + 1 (three ints, zeroed by the launcher).
__global__ void __launch_bounds__(256) k_db_quads(int H, int W, const int* __restrict__ L, int* __restrict__ quads)
{
    const int64_t total = (int64_t)(H + 1) * (W + 1);
    int q1 = 0, q3 = 0, qd = 0;
    GRID_STRIDE(i, total) {
        const int y = (int)(i / (W + 1)), x = (int)(i - (int64_t)y * (W + 1));
        auto fg = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W && L[(int64_t)yy * W + xx] >= 0) ? 1 : 0; };
        const int a = fg(y - 1, x - 1), b = fg(y - 1, x), c = fg(y, x - 1), d = fg(y, x);
        const int n = a + b + c + d;
        q1 += n == 1; q3 += n == 3; qd += (n == 2 && a == d);
    }
    for (int off = 32; off > 0; off >>= 1) { q1 += __shfl_xor(q1, off); q3 += __shfl_xor(q3, off); qd += __shfl_xor(qd, off); }
    if ((threadIdx.x & 63) == 0) { if (q1) atomicAdd(quads, q1); if (q3) atomicAdd(quads + 1, q3); if (qd) atomicAdd(quads + 2, qd); }
}

struct DbShared {
    int px[DB_MAXPTS], py[DB_MAXPTS], sx[DB_MAXPTS], sy[DB_MAXPTS], hx[2 * DB_MAXPTS + 2], hy[2 * DB_MAXPTS + 2];   // the two chains share hx / hy
    int npts, nh, state;
    double rect[4][2], wh[2];          // minimum-area rectangle (cyclic corner order) and its sides
    float box[4][2];                   // the same, ordered top-left, top-right, bottom-right, bottom-left, as float32
    double redA[256]; int redI[256];
    long long ex[4], eslope[4]; int eya[4], eyb[4], lx0[4], ly0[4], lx1[4], ly1[4];   // edge table of the fill rule
};

// minimum-area rectangle of the npts distinct integer points in (px, py): rank sort, monotone chain, calipers over every hull edge
// (first minimum in chain order).  nh <= 2 afterwards: a point or a segment (no rectangle).  Called by the whole workgroup.
__device__ void db_min_area_rect(DbShared& S, int tid)
{
#pragma clang fp contract(off)
    const int npts = S.npts;
    for (int i = tid; i < npts; i += 256) {                        // rank sort by (x, y); the points are distinct
        int r = 0;
        for (int j = 0; j < npts; ++j) r += (S.px[j] < S.px[i] || (S.px[j] == S.px[i] && S.py[j] < S.py[i])) ? 1 : 0;
        S.sx[r] = S.px[i]; S.sy[r] = S.py[i];
    }
    __syncthreads();
    if (tid == 0) {                                                // monotone chain: lower[:-1] + upper[:-1]
        if (npts <= 2) {
            S.nh = npts;
            for (int i = 0; i < npts; ++i) { S.hx[i] = S.sx[i]; S.hy[i] = S.sy[i]; }
        } else {
            auto cross = [](long long ox, long long oy, long long ax, long long ay, long long bx, long long by) {
                return (ax - ox) * (by - oy) - (ay - oy) * (bx - ox);
            };
            int n = 0;
            for (int i = 0; i < npts; ++i) {
                while (n >= 2 && cross(S.hx[n - 2], S.hy[n - 2], S.hx[n - 1], S.hy[n - 1], S.sx[i], S.sy[i]) <= 0) --n;
                S.hx[n] = S.sx[i]; S.hy[n] = S.sy[i]; ++n;
            }
            const int base = n - 1;                                // drop lower's last point, upper starts there
            n = base;
            for (int i = npts - 1; i >= 0; --i) {
                while (n - base >= 2 && cross(S.hx[n - 2], S.hy[n - 2], S.hx[n - 1], S.hy[n - 1], S.sx[i], S.sy[i]) <= 0) --n;
                S.hx[n] = S.sx[i]; S.hy[n] = S.sy[i]; ++n;
            }
            S.nh = n - 1;                                          // drop upper's last point (= lower's first)
        }
    }
    __syncthreads();
    const int h_ = S.nh;
    if (h_ <= 2) return;
    auto project = [&](int i, double& u0, double& u1, double& umin, double& umax, double& vmin, double& vmax) {
        const int j1 = i + 1 == h_ ? 0 : i + 1;
        const double ex = (double)(S.hx[j1] - S.hx[i]), ey = (double)(S.hy[j1] - S.hy[i]);
        const double nrm = sqrt(ex * ex + ey * ey);               // integer operands: the sum is exact, the root correctly rounded
        u0 = ex / nrm; u1 = ey / nrm;
        const double v0 = -u1, v1 = u0;
        umin = 1e300; umax = -1e300; vmin = 1e300; vmax = -1e300;
        for (int j = 0; j < h_; ++j) {
            const double pu = (double)S.hx[j] * u0 + (double)S.hy[j] * u1, pv = (double)S.hx[j] * v0 + (double)S.hy[j] * v1;
            umin = fmin(umin, pu); umax = fmax(umax, pu); vmin = fmin(vmin, pv); vmax = fmax(vmax, pv);
        }
    };
    double bestA = 1e300; int bestI = 0x7fffffff;
    for (int i = tid; i < h_; i += 256) {
        double u0, u1, umin, umax, vmin, vmax;
        project(i, u0, u1, umin, umax, vmin, vmax);
        const double a = (umax - umin) * (vmax - vmin);
        if (a < bestA) { bestA = a; bestI = i; }
    }
    S.redA[tid] = bestA; S.redI[tid] = bestI;
    __syncthreads();
    if (tid == 0) {
        for (int t = 1; t < 256; ++t)
            if (S.redA[t] < bestA || (S.redA[t] == bestA && S.redI[t] < bestI)) { bestA = S.redA[t]; bestI = S.redI[t]; }
        double u0, u1, umin, umax, vmin, vmax;
        project(bestI, u0, u1, umin, umax, vmin, vmax);
        const double v0 = -u1, v1 = u0;
        S.wh[0] = umax - umin; S.wh[1] = vmax - vmin;
        const double c[4][2] = {{umin * u0 + vmin * v0, umin * u1 + vmin * v1}, {umax * u0 + vmin * v0, umax * u1 + vmin * v1},
                                {umax * u0 + vmax * v0, umax * u1 + vmax * v1}, {umin * u0 + vmax * v0, umin * u1 + vmax * v1}};
        for (int i = 0; i < 4; ++i) { S.rect[i][0] = c[i][0]; S.rect[i][1] = c[i][1]; }
    }
    __syncthreads();
}

// get_mini_boxes' ordering on the float32 corners (cv2.boxPoints): stable sort by x, of the left pair the upper corner first, of the
// right pair the upper corner second -> top-left, top-right, bottom-right, bottom-left
__device__ void db_order_box(const double (*c)[2], float (*o)[2])
{
    float f[4][2];
    for (int i = 0; i < 4; ++i) { f[i][0] = (float)c[i][0]; f[i][1] = (float)c[i][1]; }
    int idx[4] = {0, 1, 2, 3};
    for (int i = 1; i < 4; ++i)
        for (int j = i; j > 0 && f[idx[j]][0] < f[idx[j - 1]][0]; --j) { const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
    int i1, i4, i2, i3;
    if (f[idx[1]][1] > f[idx[0]][1]) { i1 = idx[0]; i4 = idx[1]; } else { i1 = idx[1]; i4 = idx[0]; }
    if (f[idx[3]][1] > f[idx[2]][1]) { i2 = idx[2]; i3 = idx[3]; } else { i2 = idx[3]; i3 = idx[2]; }
    const int r[4] = {i1, i2, i3, i4};
    for (int i = 0; i < 4; ++i) { o[i][0] = f[r[i]][0]; o[i][1] = f[r[i]][1]; }
}

__device__ __forceinline__ long long db_away(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

__global__ void __launch_bounds__(256) k_db_boxes(const float* __restrict__ prob, int H, int W, const int* __restrict__ comps, const int* __restrict__ count, int cap,
                                                  const int* __restrict__ ext, int src_h, int src_w, float box_thresh, float unclip_ratio, int min_size,
                                                  int* __restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ DbShared S;
    const int tid = threadIdx.x, k = blockIdx.x;
    if (k == 0 && tid == 0) {
        out[0] = *count;
        out[1] = *count - (count[1] - count[2] - 2 * count[3]) / 4;       // holes = components - Euler number
    }
    if (*count > cap || k >= *count) return;
    int* o = out + DB_HDR + (int64_t)k * DB_REC;
    const int y0 = comps[k * 6 + 4], y1 = comps[k * 6 + 5];
    if (y1 - y0 + 1 > DB_MAXROWS) { if (tid == 0) o[0] = -1; return; }      // a tall component: the host does this map
    if (tid == 0) {
        int m = 0;
        for (int y = y0; y <= y1; ++y) {
            const int xl = ext[((int64_t)k * H + y) * 2], xr = ext[((int64_t)k * H + y) * 2 + 1];
            if (xr < 0) continue;
            S.px[m] = xl; S.py[m] = y; ++m;
            if (xr != xl) { S.px[m] = xr; S.py[m] = y; ++m; }
        }
        S.npts = m;
        S.state = 1;
    }
    __syncthreads();
    db_min_area_rect(S, tid);
    if (S.nh <= 2) { if (tid == 0) o[0] = 0; return; }            // a point or a segment: one side is 0 < min_size
    int xa = 0, ya = 0, bw = 0, bh = 0;
    if (tid == 0) {
        if (fmin(S.wh[0], S.wh[1]) < (double)min_size) S.state = 0;
        db_order_box(S.rect, S.box);
        // box_score_fast: bounding rows / columns, corners relative to them in float32, truncated; cv2.fillPoly's edge table
        float mnx = S.box[0][0], mxx = mnx, mny = S.box[0][1], mxy = mny;
        for (int i = 1; i < 4; ++i) { mnx = fminf(mnx, S.box[i][0]); mxx = fmaxf(mxx, S.box[i][0]); mny = fminf(mny, S.box[i][1]); mxy = fmaxf(mxy, S.box[i][1]); }
        const int xa_ = (int)fmin(fmax(floor((double)mnx), 0.0), (double)(W - 1)), xb_ = (int)fmin(fmax(ceil((double)mxx), 0.0), (double)(W - 1));
        const int ya_ = (int)fmin(fmax(floor((double)mny), 0.0), (double)(H - 1)), yb_ = (int)fmin(fmax(ceil((double)mxy), 0.0), (double)(H - 1));
        S.redI[0] = xa_; S.redI[1] = ya_; S.redI[2] = xb_ - xa_ + 1; S.redI[3] = yb_ - ya_ + 1;
        int qx[4], qy[4];
        for (int i = 0; i < 4; ++i) { qx[i] = (int)(S.box[i][0] - (float)xa_); qy[i] = (int)(S.box[i][1] - (float)ya_); }
        for (int e = 0; e < 4; ++e) {
            const int p = (e + 3) & 3;
            const int x0 = qx[p], yy0 = qy[p], x1 = qx[e], yy1 = qy[e];
            S.lx0[e] = x0; S.ly0[e] = yy0; S.lx1[e] = x1; S.ly1[e] = yy1;
            if (yy0 == yy1) { S.eya[e] = 1; S.eyb[e] = 0; S.ex[e] = 0; S.eslope[e] = 0; continue; }       // horizontal: outline only
            S.eslope[e] = (((long long)(x1 - x0)) << 16) / (long long)(yy1 - yy0);                       // C division: toward zero
            if (yy0 < yy1) { S.eya[e] = yy0; S.eyb[e] = yy1; S.ex[e] = ((long long)x0) << 16; }
            else { S.eya[e] = yy1; S.eyb[e] = yy0; S.ex[e] = ((long long)x1) << 16; }
        }
    }
    __syncthreads();
    if (!S.state) { if (tid == 0) o[0] = 0; return; }
    xa = S.redI[0]; ya = S.redI[1]; bw = S.redI[2]; bh = S.redI[3];
    __syncthreads();
    double sum = 0.0; int cnt = 0;
    const int npx = bw * bh;
    for (int t = tid; t < npx; t += 256) {
        const int ly = t / bw, lx = t - ly * bw;
        bool in = false;
        long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int x0 = S.lx0[e], yy0 = S.ly0[e], x1 = S.lx1[e], yy1 = S.ly1[e];
            if (x1 < x0) { int tt = x0; x0 = x1; x1 = tt; tt = yy0; yy0 = yy1; yy1 = tt; }                // the line runs left to right
            const int dx = x1 - x0, dy = yy1 >= yy0 ? yy1 - yy0 : yy0 - yy1, sy = yy1 >= yy0 ? 1 : -1;
            if (dy > dx) {
                const int kk = (ly - yy0) * sy;
                if (kk >= 0 && kk <= dy && lx == x0 + max(0, (2 * dx * kk + dy - 1) / (2 * dy))) in = true;
            } else {
                const int kk = lx - x0;
                if (kk >= 0 && kk <= dx && ly == yy0 + sy * (dx ? max(0, (2 * dy * kk + dx - 1) / (2 * dx)) : 0)) in = true;
            }
            if (ly >= S.eya[e] && ly < S.eyb[e]) {
                const long long xf = S.ex[e] + (long long)(ly - S.eya[e]) * S.eslope[e];
                lo = xf < lo ? xf : lo; hi = xf > hi ? xf : hi;
            }
        }
        if (hi >= lo && (long long)lx >= (lo >> 16) && (long long)lx <= (hi >> 16)) in = true;
        if (in) { sum += (double)prob[(int64_t)(ya + ly) * W + xa + lx]; ++cnt; }
    }
    S.redA[tid] = sum; S.redI[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        for (int t = 1; t < 256; ++t) { sum += S.redA[t]; cnt += S.redI[t]; }
        const double score = cnt ? sum / cnt : 0.0;
        o[9] = __float_as_int((float)score);
        if ((double)box_thresh > score) S.state = 0;
        if (S.state) {
            // unclip: cv2.contourArea / cv2.arcLength (float32 segment lengths) of the float32 box, then ClipperLib's rounded offset
            double ar = 0.0, len = 0.0;
            for (int i = 0; i < 4; ++i) {
                const int j = (i + 1) & 3;
                ar += (double)S.box[i][0] * (double)S.box[j][1] - (double)S.box[j][0] * (double)S.box[i][1];
                const float ddx = S.box[i][0] - S.box[(i + 3) & 3][0], ddy = S.box[i][1] - S.box[(i + 3) & 3][1];
                len += (double)__fsqrt_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)));
            }
            ar = fabs(ar) * 0.5;
            int n = 0;
            long long qx[4], qy[4];
            for (int i = 0; i < 4; ++i) {                          // integer-truncated, consecutive duplicates dropped
                const long long x = (long long)S.box[i][0], y = (long long)S.box[i][1];
                if (n && qx[n - 1] == x && qy[n - 1] == y) continue;
                qx[n] = x; qy[n] = y; ++n;
            }
            if (n > 1 && qx[0] == qx[n - 1] && qy[0] == qy[n - 1]) --n;
            int m = 0;
            if (ar > 0.0 && len > 0.0 && n >= 3) {
                const double delta = ar * (double)unclip_ratio / len;
                long long twice = 0;
                for (int i = 0; i < n; ++i) { const int p = (i + n - 1) % n; twice += (qx[p] + qx[i]) * (qy[p] - qy[i]); }
                if (twice > 0)                                    // Clipper's Area() < 0: reversed, the normals then point outwards
                    for (int i = 0; i < n / 2; ++i) { long long t = qx[i]; qx[i] = qx[n - 1 - i]; qx[n - 1 - i] = t; t = qy[i]; qy[i] = qy[n - 1 - i]; qy[n - 1 - i] = t; }
                const double PI = 3.14159265358979323846;
                const double tol = fmin(0.25, delta * 0.25);
                const double steps = fmin(PI / acos(1 - tol / delta), delta * PI);
                const double rot_s = sin(2 * PI / steps), rot_c = cos(2 * PI / steps), per_rad = steps / (2 * PI);
                double nx[4], ny[4];
                for (int i = 0; i < n; ++i) {
                    const int j = (i + 1) % n;
                    const double ex = (double)(qx[j] - qx[i]), ey = (double)(qy[j] - qy[i]);
                    const double inv = 1.0 / sqrt(ex * ex + ey * ey);
                    nx[i] = ey * inv; ny[i] = -ex * inv;
                }
                auto put = [&](long long px_, long long py_, double ax, double ay) {
                    if (m < DB_MAXPTS) { S.px[m] = (int)db_away((double)px_ + ax * delta); S.py[m] = (int)db_away((double)py_ + ay * delta); }
                    ++m;
                };
                for (int i = 0; i < n; ++i) {
                    const int p = (i + n - 1) % n;
                    const double ax = nx[p], ay = ny[p], bx = nx[i], by = ny[i];
                    double sin_t = ax * by - bx * ay;
                    const double cos_t = ax * bx + ay * by;
                    if (fabs(sin_t * delta) < 1.0 && cos_t > 0) { put(qx[i], qy[i], ax, ay); continue; }
                    if (fabs(sin_t * delta) >= 1.0) sin_t = fmin(fmax(sin_t, -1.0), 1.0);
                    if (sin_t * delta < 0) {
                        put(qx[i], qy[i], ax, ay);
                        if (m < DB_MAXPTS) { S.px[m] = (int)qx[i]; S.py[m] = (int)qy[i]; }
                        ++m;
                        put(qx[i], qy[i], bx, by);
                        continue;
                    }
                    const double ang = atan2(sin_t, cos_t);
                    long long ns = db_away(per_rad * fabs(ang));
                    if (ns < 1) ns = 1;
                    double cx = ax, cy = ay;
                    for (long long t = 0; t < ns; ++t) {
                        put(qx[i], qy[i], cx, cy);
                        const double c2 = cx;
                        cx = cx * rot_c - rot_s * cy;
                        cy = c2 * rot_s + cy * rot_c;
                    }
                    put(qx[i], qy[i], bx, by);
                }
            }
            if (m == 0 || m > DB_MAXPTS) S.state = m > DB_MAXPTS ? -1 : 0;
            else {
                // the rectangle fit wants distinct points: drop duplicates (rounded arc chords may coincide)
                int u = 0;
                for (int i = 0; i < m; ++i) {
                    bool dup = false;
                    for (int j = 0; j < u && !dup; ++j) dup = S.px[j] == S.px[i] && S.py[j] == S.py[i];
                    if (!dup) { S.px[u] = S.px[i]; S.py[u] = S.py[i]; ++u; }
                }
                S.npts = u;
            }
        }
    }
    __syncthreads();
    if (S.state != 1) { if (tid == 0) o[0] = S.state; return; }
    db_min_area_rect(S, tid);
    if (tid != 0) return;
    if (S.nh <= 2 || fmin(S.wh[0], S.wh[1]) < (double)(min_size + 2)) { o[0] = 0; return; }
    float big[4][2];
    db_order_box(S.rect, big);
    const double wsc = (double)src_w / (double)W, hsc = (double)src_h / (double)H;
    for (int i = 0; i < 4; ++i) {
        o[1 + 2 * i] = (int)fmin(fmax(rint((double)big[i][0] * wsc), 0.0), (double)src_w);
        o[2 + 2 * i] = (int)fmin(fmax(rint((double)big[i][1] * hsc), 0.0), (double)src_h);
    }
    o[0] = 1;
}

extern "C" {

int vsr_det_launch_conv2d(const float* x, const float* w, const float* bias, int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                          int pt, int pl, int Ho, int Wo, int depthwise, int act, float* out, void* stream)
{
    if (N <= 0 || Ho <= 0 || Wo <= 0) return 0;
    if (depthwise) {
        if (Cin != Cout) return VSR_ERR_ARG;
        const int64_t total = (int64_t)N * Cout * Ho * Wo;
        hipLaunchKernelGGL(k_det_dwconv, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, w, bias, N, Cout, H, W, kh, kw, sh, sw, pt, pl,
                           Ho, Wo, act, out);
        DONE();
    }
    const dim3 grid((unsigned)(((int64_t)Ho * Wo + 255) / 256), (unsigned)((Cout + 7) / 8), (unsigned)N);
    hipLaunchKernelGGL(k_det_conv<8>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, Cin, H, W, Cout, kh, kw, sh, sw, pt, pl, Ho, Wo, act, out);
    DONE();
}

int vsr_det_launch_deconv2x2(const float* x, const float* w, int N, int Cin, int H, int W, int Cout, int depthwise, float* out, void* stream)
{
    const int64_t total = (int64_t)N * Cout * 4 * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_deconv2, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, w, N, Cin, H, W, Cout, depthwise, out);
    DONE();
}

int vsr_det_launch_binary(const float* a, const float* b, int op, int64_t total, int C, int64_t HW, int bmode, float* out, void* stream)
{
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_binary, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b, op, total, C, HW, bmode, out);
    DONE();
}

int vsr_det_launch_unary(const float* x, int64_t total, int kind, float p0, float p1, float* out, void* stream)
{
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_unary, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, total, kind, p0, p1, out);
    DONE();
}

int vsr_det_launch_affine(const float* x, const float* scale, const float* shift, int64_t total, int C, int64_t HW, float* out, void* stream)
{
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_affine, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, total, C, HW, out);
    DONE();
}

int vsr_det_launch_gap(const float* x, int64_t planes, int64_t HW, float* out, void* stream)
{
    if (planes <= 0) return 0;
    hipLaunchKernelGGL(k_det_gap, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, planes, HW, out);
    DONE();
}

int vsr_det_launch_maxpool(const float* x, int64_t planes, int H, int W, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, float* out,
                           void* stream)
{
    const int64_t total = planes * Ho * Wo;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_maxpool, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, planes, H, W, kh, kw, sh, sw, pt, pl, Ho, Wo, out);
    DONE();
}

int vsr_det_launch_nearest(const float* x, int64_t planes, int H, int W, int s, float* out, void* stream)
{
    const int64_t total = planes * H * W * s * s;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_nearest, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, planes, H, W, s, out);
    DONE();
}

int vsr_det_launch_nchw_to_nhwc(const float* x, int n, int C, int H, int W, int pt, int pl, int Hp, int Wp, int Cp, float* out, void* stream)
{
    if (!x || !out || n <= 0 || n > 65535 || C <= 0 || Cp % 32 || Cp < C || Hp < H + pt || Wp < W + pl) return VSR_ERR_ARG;
    const int64_t total = (int64_t)Hp * Wp;
    hipLaunchKernelGGL(k_det_nchw_to_nhwc, dim3((unsigned)((total + 31) / 32), (unsigned)(Cp / 32), (unsigned)n), dim3(256), 0, (hipStream_t)stream, x, C,
                       H, W, pt, pl, Hp, Wp, Cp, out);
    DONE();
}

int vsr_det_launch_nhwc_to_nchw(const float* in, int n, int C, int64_t P, int Np, const float* scale, const float* shift, int act, float* out,
                                void* stream)
{
    if (!in || !out || n <= 0 || n > 65535 || C <= 0 || Np < C || P <= 0 || (scale && !shift)) return VSR_ERR_ARG;
    hipLaunchKernelGGL(k_det_nhwc_to_nchw, dim3((unsigned)((P + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)n), dim3(256), 0, (hipStream_t)stream, in,
                       C, P, Np, scale, shift, act, out);
    DONE();
}

int vsr_det_launch_to_view(const float* x, int n, int C, int H, int W, int Cw, float* out, int64_t img_stride, int64_t row_stride, int Cs, void* stream)
{
    if (!x || !out || n <= 0 || n > 65535 || C <= 0 || Cw % 32 || Cw < C || Cs < Cw) return VSR_ERR_ARG;
    const int64_t total = (int64_t)H * W;
    hipLaunchKernelGGL(k_det_to_view, dim3((unsigned)((total + 31) / 32), (unsigned)(Cw / 32), (unsigned)n), dim3(256), 0, (hipStream_t)stream, x, C, H, W,
                       Cw, out, img_stride, row_stride, Cs);
    DONE();
}

int vsr_det_launch_from_view(const float* in, int64_t img_stride, int64_t row_stride, int Cs, int n, int C, int H, int W, float* out,
                             int64_t out_img_stride, void* stream)
{
    if (!in || !out || n <= 0 || n > 65535 || C <= 0 || Cs < C) return VSR_ERR_ARG;
    const int64_t total = (int64_t)H * W;
    hipLaunchKernelGGL(k_det_from_view, dim3((unsigned)((total + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)n), dim3(256), 0, (hipStream_t)stream, in,
                       img_stride, row_stride, Cs, C, H, W, out, out_img_stride);
    DONE();
}

int vsr_det_launch_dwconv_view(const float* in, int64_t in_img, int64_t in_row, int in_cs, const float* w, const float* scale, const float* shift, int N,
                               int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, int act, float* out, int64_t out_img,
                               int64_t out_row, int out_cs, void* stream)
{
    if (!in || !out || !w || C <= 0 || C % 4 || in_cs % 4 || out_cs % 4 || (scale && !shift) || (((uintptr_t)in | (uintptr_t)out | (uintptr_t)w) & 15))
        return VSR_ERR_ARG;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_dwconv_view, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_img, in_row, in_cs, w, scale, shift, N, C, kh, kw,
                       sh, sw, pt, pl, Ho, Wo, act, out, out_img, out_row, out_cs);
    DONE();
}

int vsr_det_launch_nearest_view(const float* in, int64_t in_img, int64_t in_row, int in_cs, int N, int C, int Ho, int Wo, int s, float* out,
                                int64_t out_img, int64_t out_row, int out_cs, void* stream)
{
    if (!in || !out || C <= 0 || C % 4 || s < 1 || in_cs % 4 || out_cs % 4 || (((uintptr_t)in | (uintptr_t)out) & 15)) return VSR_ERR_ARG;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_nearest_view, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_img, in_row, in_cs, N, C, Ho, Wo, s, out,
                       out_img, out_row, out_cs);
    DONE();
}

int vsr_det_launch_im2col_view(const float* x, int N, int C, int H, int W, int kh, int kw, int pt, int pl, float* out, int64_t out_img, int64_t out_row,
                               int out_cs, void* stream)
{
    if (!x || !out || C <= 0 || kh <= 0 || kw <= 0 || C * kh * kw > 32 || out_cs % 4 || ((uintptr_t)out & 15)) return VSR_ERR_ARG;
    const int64_t total = (int64_t)N * H * W * 8;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_im2col_view, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, kh, kw, pt, pl, out, out_img, out_row, out_cs);
    DONE();
}

int vsr_det_launch_dots_view(const float* in, int64_t in_img, int64_t in_row, int in_cs, int N, int C, int H, int W, const float* w, const float* bias,
                             int n_out, int act, float* out, void* stream)
{
    if (!in || !out || !w || C <= 0 || C % 4 || in_cs % 4 || (n_out != 1 && n_out != 4) || (((uintptr_t)in | (uintptr_t)w) & 15)) return VSR_ERR_ARG;
    const int64_t total = (int64_t)N * H * W;
    if (total <= 0) return 0;
    const int grid = grid_for(total * 16);
    if (n_out == 1)
        hipLaunchKernelGGL(k_det_dots_view<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_img, in_row, in_cs, N, C, H, W, w, bias, act, out);
    else
        hipLaunchKernelGGL(k_det_dots_view<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_img, in_row, in_cs, N, C, H, W, w, bias, act, out);
    DONE();
}

// labels int32 [H*W] (component = raster index of its first pixel, -1 background), stats int32 [H*W*5] scratch, comps int32
// [cap][6] = (label, area, xmin, xmax, ymin, ymax) in no particular order, count = number of components found (may exceed cap)
int vsr_det_launch_ccl(const float* prob, int H, int W, float thresh, int32_t* labels, int32_t* stats, int32_t* comps, int cap, int32_t* count,
                       void* stream)
{
    const int64_t total = (int64_t)H * W;
    if (!prob || !labels || !stats || !comps || !count || total <= 0 || total >= 0x7fffffff / 5 || cap <= 0) return VSR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(int32_t), s) != hipSuccess) return VSR_ERR_HIP;
    hipLaunchKernelGGL(k_ccl_init, dim3(grid_for(total)), dim3(256), 0, s, prob, thresh, total, W, labels, stats);
    hipLaunchKernelGGL(k_ccl_merge, dim3(grid_for(total)), dim3(256), 0, s, H, W, labels);
    hipLaunchKernelGGL(k_ccl_flatten_stats, dim3(grid_for(total)), dim3(256), 0, s, H, W, labels, stats);
    hipLaunchKernelGGL(k_ccl_compact, dim3(grid_for(total)), dim3(256), 0, s, total, labels, stats, comps, cap, count);
    DONE();
}

// The whole DBPostProcess of one probability map on the device: vsr_det_launch_ccl's labelling, the hole count, then the polygon
// work per component.  count int32 [4] (components, then the three bit-quad counts); ext int32 [cap][H][2] scratch; out int32
// [4 + cap * 16]: out[0] = components found, out[1] = holes (background regions enclosed by foreground: their borders are contours
// too, such a map is the host's), then per component slot k (same order as comps) a record (flag, x0, y0, x1, y1, x2, y2, x3, y3,
// score bits): flag 1 = box (source-image pixels, ordered top-left, top-right, bottom-right, bottom-left), 0 = rejected (size /
// score), -1 = not processed (taller than 256 rows, or an offset polygon beyond the point buffer).  More than cap components: only
// out[0] is valid.
int vsr_det_launch_db_boxes(const float* prob, int H, int W, float thresh, int src_h, int src_w, float box_thresh, float unclip_ratio, int min_size,
                            int32_t* labels, int32_t* stats, int32_t* comps, int32_t* count, int32_t* ext, int32_t* out, int cap, void* stream)
{
    if (!ext || !out || cap <= 0 || cap > 65535) return VSR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, 4 * sizeof(int32_t), s) != hipSuccess) return VSR_ERR_HIP;
    const int rc = vsr_det_launch_ccl(prob, H, W, thresh, labels, stats, comps, cap, count, stream);
    if (rc != 0) return rc;
    const int64_t total = (int64_t)H * W;
    hipLaunchKernelGGL(k_db_quads, dim3(grid_for((int64_t)(H + 1) * (W + 1))), dim3(256), 0, s, H, W, labels, count + 1);
    hipLaunchKernelGGL(k_db_slots, dim3(grid_for(cap)), dim3(256), 0, s, comps, count, cap, stats);
    hipLaunchKernelGGL(k_db_ext_init, dim3(grid_for((int64_t)cap * H)), dim3(256), 0, s, count, cap, H, ext);
    hipLaunchKernelGGL(k_db_ext, dim3(grid_for(total)), dim3(256), 0, s, H, W, labels, stats, count, cap, ext);
    hipLaunchKernelGGL(k_db_boxes, dim3(cap), dim3(256), 0, s, prob, H, W, comps, count, cap, ext, src_h, src_w, box_thresh, unclip_ratio, min_size, out);
    DONE();
}

// channel concat = one strided block copy per part: `rows` images, `width` bytes of the part per image (recorded like any other
// launch of the forward)
int vsr_det_launch_copy(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t width, int64_t rows, void* stream)
{
    if (!src || !dst || width < 0 || rows < 0 || src_pitch < width || dst_pitch < width) return VSR_ERR_ARG;
    if (width == 0 || rows == 0) return 0;
    return hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)width, (size_t)rows, hipMemcpyDeviceToDevice,
                            (hipStream_t)stream) == hipSuccess ? 0 : VSR_ERR_HIP;
}

int vsr_det_launch_normalize(const uint8_t* img, int H, int W, float* out, void* stream)
{
    const int64_t total = (int64_t)H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_det_normalize, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, H, W, out);
    DONE();
}

} // extern "C"
