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

// ---- DBPostProcess, polygon part on the device (paddleocr DBPostProcess: get_mini_boxes, box_score_fast, unclip; the numpy
// statement of the same steps is ocr_det._box_of_component).  A minimum-area rectangle only depends on the convex hull, and the
// hull's vertices are among the leftmost / rightmost pixel of every row of the component: k_db_ext collects those per
// (component, row) with atomics, k_db_boxes (one wave per component) sorts the <= 2 * rows points by rank, builds the hull with
// Andrew's monotone chain (integer cross products), runs the calipers with one lane per hull edge, scores the rectangle by the
// mean probability of the pixels inside it (lanes over its bounding box) and offsets / rescales it.  All geometry in fp64.
#define DB_MAXROWS 256
#define DB_MAXPTS (2 * DB_MAXROWS)
#define DB_REC 16                                                  // ints per output record: flag, 8 corner coordinates, score bits

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
        const int k = st[(int64_t)r * 5 + 0];
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        atomicMin(&ext[((int64_t)k * H + y) * 2], x);
        atomicMax(&ext[((int64_t)k * H + y) * 2 + 1], x);
    }
}

__device__ __forceinline__ void db_order_box(const double (*c)[2], double (*o)[2])
{   // get_mini_boxes: stable sort by x, the left pair and the right pair by y -> top-left, top-right, bottom-right, bottom-left
    int idx[4] = {0, 1, 2, 3};
    for (int i = 1; i < 4; ++i)
        for (int j = i; j > 0 && c[idx[j]][0] < c[idx[j - 1]][0]; --j) { const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
    int a = idx[0], b = idx[1], d = idx[2], e = idx[3];
    if (c[b][1] < c[a][1]) { const int t = a; a = b; b = t; }
    if (c[e][1] < c[d][1]) { const int t = d; d = e; e = t; }
    const int r[4] = {a, d, e, b};
    for (int i = 0; i < 4; ++i) { o[i][0] = c[r[i]][0]; o[i][1] = c[r[i]][1]; }
}

__global__ void __launch_bounds__(64) k_db_boxes(const float* __restrict__ prob, int H, int W, const int* __restrict__ comps, const int* __restrict__ count, int cap,
                                                 const int* __restrict__ ext, int src_h, int src_w, float box_thresh, float unclip_ratio, int min_size,
                                                 int* __restrict__ out)
{
    __shared__ int px[DB_MAXPTS], py[DB_MAXPTS], sx[DB_MAXPTS], sy[DB_MAXPTS], hxI[2 * DB_MAXPTS + 2], hyI[2 * DB_MAXPTS + 2];   // the two chains share hxI / hyI
    __shared__ int npts, nh;
    __shared__ double box[4][2], wh[2];
    __shared__ int state;                                         // 1 = still a candidate
    const int lane = threadIdx.x, k = blockIdx.x;
    if (k == 0 && lane == 0) out[0] = *count;
    if (*count > cap || k >= *count) return;
    int* o = out + 1 + (int64_t)k * DB_REC;
    const int y0 = comps[k * 6 + 4], y1 = comps[k * 6 + 5];
    if (y1 - y0 + 1 > DB_MAXROWS) { if (lane == 0) o[0] = -1; return; }      // a tall component: the host does this map
    if (lane == 0) {
        int m = 0;
        for (int y = y0; y <= y1; ++y) {
            const int xl = ext[((int64_t)k * H + y) * 2], xr = ext[((int64_t)k * H + y) * 2 + 1];
            if (xr < 0) continue;
            px[m] = xl; py[m] = y; ++m;
            if (xr != xl) { px[m] = xr; py[m] = y; ++m; }
        }
        npts = m;
        state = 1;
    }
    __syncthreads();
    for (int i = lane; i < npts; i += 64) {                        // rank sort by (x, y); the points are distinct
        int r = 0;
        for (int j = 0; j < npts; ++j) r += (px[j] < px[i] || (px[j] == px[i] && py[j] < py[i])) ? 1 : 0;
        sx[r] = px[i]; sy[r] = py[i];
    }
    __syncthreads();
    if (lane == 0) {                                               // monotone chain: lower[:-1] + upper[:-1]
        if (npts <= 2) {
            nh = npts;
            for (int i = 0; i < npts; ++i) { hxI[i] = sx[i]; hyI[i] = sy[i]; }
        } else {
            auto cross = [](long long ox, long long oy, long long ax, long long ay, long long bx, long long by) {
                return (ax - ox) * (by - oy) - (ay - oy) * (bx - ox);
            };
            int n = 0;
            for (int i = 0; i < npts; ++i) {
                while (n >= 2 && cross(hxI[n - 2], hyI[n - 2], hxI[n - 1], hyI[n - 1], sx[i], sy[i]) <= 0) --n;
                hxI[n] = sx[i]; hyI[n] = sy[i]; ++n;
            }
            const int lo = n - 1;                                  // drop lower's last point, upper starts there
            n = lo;
            const int base = n;
            for (int i = npts - 1; i >= 0; --i) {
                while (n - base >= 2 && cross(hxI[n - 2], hyI[n - 2], hxI[n - 1], hyI[n - 1], sx[i], sy[i]) <= 0) --n;
                hxI[n] = sx[i]; hyI[n] = sy[i]; ++n;
            }
            nh = n - 1;                                            // drop upper's last point (= lower's first)
        }
    }
    __syncthreads();
    const int h_ = nh;
    if (h_ <= 2) { if (lane == 0) o[0] = 0; return; }              // a point or a segment: one side is 0 < min_size
    // calipers: lane i projects the hull on the direction of edge i and on its normal
    double bestA = 1e300; int bestI = 0x7fffffff;
    for (int i = lane; i < h_; i += 64) {
        const int j1 = i + 1 == h_ ? 0 : i + 1;
        const double ex = (double)(hxI[j1] - hxI[i]), ey = (double)(hyI[j1] - hyI[i]);
        const double nrm = hypot(ex, ey);
        const double u0 = ex / nrm, u1 = ey / nrm, v0 = -u1, v1 = u0;
        double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
        for (int j = 0; j < h_; ++j) {
            const double pu = (double)hxI[j] * u0 + (double)hyI[j] * u1, pv = (double)hxI[j] * v0 + (double)hyI[j] * v1;
            umin = fmin(umin, pu); umax = fmax(umax, pu); vmin = fmin(vmin, pv); vmax = fmax(vmax, pv);
        }
        const double a = (umax - umin) * (vmax - vmin);
        if (a < bestA) { bestA = a; bestI = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {                       // first minimum over the edges
        const double a2 = __shfl_xor(bestA, off);
        const int i2 = __shfl_xor(bestI, off);
        if (a2 < bestA || (a2 == bestA && i2 < bestI)) { bestA = a2; bestI = i2; }
    }
    if (lane == 0) {
        const int i = bestI, j1 = i + 1 == h_ ? 0 : i + 1;
        const double ex = (double)(hxI[j1] - hxI[i]), ey = (double)(hyI[j1] - hyI[i]);
        const double nrm = hypot(ex, ey);
        const double u0 = ex / nrm, u1 = ey / nrm, v0 = -u1, v1 = u0;
        double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
        for (int j = 0; j < h_; ++j) {
            const double pu = (double)hxI[j] * u0 + (double)hyI[j] * u1, pv = (double)hxI[j] * v0 + (double)hyI[j] * v1;
            umin = fmin(umin, pu); umax = fmax(umax, pu); vmin = fmin(vmin, pv); vmax = fmax(vmax, pv);
        }
        const double w = umax - umin, h = vmax - vmin;
        wh[0] = w; wh[1] = h;
        if (fmin(w, h) < (double)min_size) state = 0;
        const double c[4][2] = {{umin * u0 + vmin * v0, umin * u1 + vmin * v1}, {umax * u0 + vmin * v0, umax * u1 + vmin * v1},
                                {umax * u0 + vmax * v0, umax * u1 + vmax * v1}, {umin * u0 + vmax * v0, umin * u1 + vmax * v1}};
        db_order_box(c, box);
    }
    __syncthreads();
    if (!state) { if (lane == 0) o[0] = 0; return; }
    // box_score_fast: mean probability over the pixels inside the rectangle
    double bx[4], by[4];
    for (int i = 0; i < 4; ++i) { bx[i] = box[i][0]; by[i] = box[i][1]; }
    const double mnx = fmin(fmin(bx[0], bx[1]), fmin(bx[2], bx[3])), mxx = fmax(fmax(bx[0], bx[1]), fmax(bx[2], bx[3]));
    const double mny = fmin(fmin(by[0], by[1]), fmin(by[2], by[3])), mxy = fmax(fmax(by[0], by[1]), fmax(by[2], by[3]));
    const int xa = (int)fmin(fmax(floor(mnx), 0.0), (double)(W - 1)), xb = (int)fmin(fmax(ceil(mxx), 0.0), (double)(W - 1));
    const int ya = (int)fmin(fmax(floor(mny), 0.0), (double)(H - 1)), yb = (int)fmin(fmax(ceil(mxy), 0.0), (double)(H - 1));
    const int bw = xb - xa + 1, npx = bw * (yb - ya + 1);
    double sum = 0.0; int cnt = 0;
    for (int t = lane; t < npx; t += 64) {
        const int y = ya + t / bw, x = xa + t % bw;
        bool in = true;
        for (int i = 0; i < 4; ++i) {
            const int q = (i + 1) & 3;
            in = in && ((bx[q] - bx[i]) * ((double)y - by[i]) - (by[q] - by[i]) * ((double)x - bx[i]) >= -1e-6);
        }
        if (in) { sum += (double)prob[(int64_t)y * W + x]; ++cnt; }
    }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); cnt += __shfl_xor(cnt, off); }
    if (lane != 0) return;
    const float score = cnt ? (float)(sum / cnt) : 0.f;
    if ((double)score < (double)box_thresh) { o[0] = 0; return; }
    // unclip: offset the rectangle by area * ratio / perimeter, then rescale to the source image
    const double w = wh[0], h = wh[1];
    const double d = (w * h) * (double)unclip_ratio / (2 * (w + h));
    const double cx = (bx[0] + bx[1] + bx[2] + bx[3]) / 4, cy = (by[0] + by[1] + by[2] + by[3]) / 4;
    const double e1x = bx[1] - bx[0], e1y = by[1] - by[0], e3x = bx[3] - bx[0], e3y = by[3] - by[0];
    const double l1 = hypot(e1x, e1y), l3 = hypot(e3x, e3y);
    const double ux = e1x / fmax(l1, 1e-9), uy = e1y / fmax(l1, 1e-9), vx = e3x / fmax(l3, 1e-9), vy = e3y / fmax(l3, 1e-9);
    const double hw = l1 / 2 + d, hh = l3 / 2 + d;
    if (fmin(2 * hw, 2 * hh) < (double)(min_size + 2)) { o[0] = 0; return; }
    double big[4][2] = {{cx - hw * ux - hh * vx, cy - hw * uy - hh * vy}, {cx + hw * ux - hh * vx, cy + hw * uy - hh * vy},
                        {cx + hw * ux + hh * vx, cy + hw * uy + hh * vy}, {cx - hw * ux + hh * vx, cy - hw * uy + hh * vy}};
    for (int i = 0; i < 4; ++i) {
        big[i][0] = fmin(fmax(rint(big[i][0] / (double)W * (double)src_w), 0.0), (double)src_w);
        big[i][1] = fmin(fmax(rint(big[i][1] / (double)H * (double)src_h), 0.0), (double)src_h);
    }
    double ob[4][2];
    db_order_box(big, ob);
    for (int i = 0; i < 4; ++i) { o[1 + 2 * i] = (int)ob[i][0]; o[2 + 2 * i] = (int)ob[i][1]; }
    o[9] = __float_as_int(score);
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

// The whole DBPostProcess of one probability map on the device: vsr_det_launch_ccl's labelling, then the polygon work per
// component.  ext int32 [cap][H][2] scratch; out int32 [1 + cap * 16]: out[0] = number of components found, then per component
// slot k (same order as comps) a record (flag, x0, y0, x1, y1, x2, y2, x3, y3, score bits): flag 1 = box (source-image pixels,
// ordered top-left, top-right, bottom-right, bottom-left), 0 = rejected (size / score), -1 = taller than 256 rows (not
// processed).  More than cap components: only out[0] is valid.
int vsr_det_launch_db_boxes(const float* prob, int H, int W, float thresh, int src_h, int src_w, float box_thresh, float unclip_ratio, int min_size,
                            int32_t* labels, int32_t* stats, int32_t* comps, int32_t* count, int32_t* ext, int32_t* out, int cap, void* stream)
{
    if (!ext || !out || cap <= 0 || cap > 65535) return VSR_ERR_ARG;
    const int rc = vsr_det_launch_ccl(prob, H, W, thresh, labels, stats, comps, cap, count, stream);
    if (rc != 0) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)H * W;
    hipLaunchKernelGGL(k_db_slots, dim3(grid_for(cap)), dim3(256), 0, s, comps, count, cap, stats);
    hipLaunchKernelGGL(k_db_ext_init, dim3(grid_for((int64_t)cap * H)), dim3(256), 0, s, count, cap, H, ext);
    hipLaunchKernelGGL(k_db_ext, dim3(grid_for(total)), dim3(256), 0, s, H, W, labels, stats, count, cap, ext);
    hipLaunchKernelGGL(k_db_boxes, dim3(cap), dim3(64), 0, s, prob, H, W, comps, count, cap, ext, src_h, src_w, box_thresh, unclip_ratio, min_size, out);
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
