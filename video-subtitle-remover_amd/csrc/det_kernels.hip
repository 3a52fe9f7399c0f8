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

__global__ void __launch_bounds__(256) k_ccl_init(const float* __restrict__ prob, float thresh, int64_t total, int* __restrict__ L, int* __restrict__ st)
{
    GRID_STRIDE(i, total) {
        L[i] = prob[i] > thresh ? (int)i : -1;
        st[i * 5 + 0] = 0;                                     // area
        st[i * 5 + 1] = 0x7fffffff; st[i * 5 + 2] = -1;        // x min / max
        st[i * 5 + 3] = 0x7fffffff; st[i * 5 + 4] = -1;        // y min / max
    }
}

__global__ void __launch_bounds__(256) k_ccl_merge(int H, int W, int* __restrict__ L)
{
    const int64_t total = (int64_t)H * W;
    GRID_STRIDE(i, total) {
        if (L[i] < 0) continue;
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        if (x + 1 < W && L[i + 1] >= 0) ccl_unite(L, (int)i, (int)i + 1);
        if (y + 1 < H) {
            const int64_t d = i + W;
            if (x > 0 && L[d - 1] >= 0) ccl_unite(L, (int)i, (int)(d - 1));
            if (L[d] >= 0) ccl_unite(L, (int)i, (int)d);
            if (x + 1 < W && L[d + 1] >= 0) ccl_unite(L, (int)i, (int)(d + 1));
        }
    }
}

__global__ void __launch_bounds__(256) k_ccl_flatten_stats(int H, int W, int* __restrict__ L, int* __restrict__ st)
{
    const int64_t total = (int64_t)H * W;
    GRID_STRIDE(i, total) {
        if (L[i] < 0) continue;
        const int r = ccl_find(L, (int)i);
        L[i] = r;                                              // roots keep L[r] == r: concurrent finds stay correct
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        atomicAdd(&st[(int64_t)r * 5 + 0], 1);
        atomicMin(&st[(int64_t)r * 5 + 1], x); atomicMax(&st[(int64_t)r * 5 + 2], x);
        atomicMin(&st[(int64_t)r * 5 + 3], y); atomicMax(&st[(int64_t)r * 5 + 4], y);
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

// labels int32 [H*W] (component = raster index of its first pixel, -1 background), stats int32 [H*W*5] scratch, comps int32
// [cap][6] = (label, area, xmin, xmax, ymin, ymax) in no particular order, count = number of components found (may exceed cap)
int vsr_det_launch_ccl(const float* prob, int H, int W, float thresh, int32_t* labels, int32_t* stats, int32_t* comps, int cap, int32_t* count,
                       void* stream)
{
    const int64_t total = (int64_t)H * W;
    if (!prob || !labels || !stats || !comps || !count || total <= 0 || total >= 0x7fffffff / 5 || cap <= 0) return VSR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(int32_t), s) != hipSuccess) return VSR_ERR_HIP;
    hipLaunchKernelGGL(k_ccl_init, dim3(grid_for(total)), dim3(256), 0, s, prob, thresh, total, labels, stats);
    hipLaunchKernelGGL(k_ccl_merge, dim3(grid_for(total)), dim3(256), 0, s, H, W, labels);
    hipLaunchKernelGGL(k_ccl_flatten_stats, dim3(grid_for(total)), dim3(256), 0, s, H, W, labels, stats);
    hipLaunchKernelGGL(k_ccl_compact, dim3(grid_for(total)), dim3(256), 0, s, total, labels, stats, comps, cap, count);
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
