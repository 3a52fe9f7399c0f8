// Colour conversion of the raw-container frame transport: planar 8-bit YCbCr <-> packed BGR.
// In the reference this is libswscale's work on both sides of the pipe: cv2.VideoCapture.read() hands out BGR frames
// (backend/inpaint/sttn_auto_inpaint.py:254-262, backend/main.py:171-176) and FFmpegVideoWriter feeds bgr24 frames to an encoder
// that converts them to yuv420p (backend/tools/video_io.py:54-81).  Here the *.y4m reader / writer (backend/tools/video_io.py)
// keep the planes as they are on disk, and these kernels do the BT.601 integer conversion on the GPU: a 1080p frame costs the
// host 47 ms in numpy -- seven times the inpainting itself -- and 10 us here.  Integer work, HBM-bound, bit-exact against the
// numpy statement of the same matrices (video_io._yuv_to_bgr / _bgr_to_yuv; tests/test_gpu_io.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vsr_hip.h"

#define DONE() return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP

__device__ __forceinline__ int clip_u8(int v) { return min(max(v, 0), 255); }

// 16.16 fixed-point BT.601 (the constants of libswscale / OpenCV): studio swing unless `full`
__device__ __forceinline__ void yuv2bgr_px(int y, int u, int v, bool full, int& b, int& g, int& r)
{
    u -= 128;
    v -= 128;
    if (full) {
        const int c = y << 16;
        r = (c + 91881 * v + 32768) >> 16;
        g = (c - 22554 * u - 46802 * v + 32768) >> 16;
        b = (c + 116130 * u + 32768) >> 16;
    } else {
        const int c = 76309 * (y - 16);
        r = (c + 104597 * v + 32768) >> 16;
        g = (c - 25675 * u - 53279 * v + 32768) >> 16;
        b = (c + 132201 * u + 32768) >> 16;
    }
    b = clip_u8(b);
    g = clip_u8(g);
    r = clip_u8(r);
}

__device__ __forceinline__ void bgr2yuv_px(int b, int g, int r, bool full, int& y, int& u, int& v)
{
    if (full) {
        y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
        u = ((-11059 * r - 21709 * g + 32768 * b + 32768) >> 16) + 128;
        v = ((32768 * r - 27439 * g - 5329 * b + 32768) >> 16) + 128;
    } else {
        y = ((16829 * r + 33039 * g + 6416 * b + 32768) >> 16) + 16;
        u = ((-9714 * r - 19070 * g + 28784 * b + 32768) >> 16) + 128;
        v = ((28784 * r - 24103 * g - 4681 * b + 32768) >> 16) + 128;
    }
    y = clip_u8(y);
    u = clip_u8(u);
    v = clip_u8(v);
}

// One thread = 4 horizontally adjacent pixels: one 32-bit luma load, 12 output bytes as three 32-bit stores when the row is
// aligned.  Chroma is replicated (nearest), sx / sy = log2 of the sub-sampling; cw == 0: no chroma planes (mono).
// src = frames of [Y: H*W][U: ch*cw][V: ch*cw] contiguous, `frameBytes` apart; dst BGR [n][H][W][3].
__global__ void __launch_bounds__(256) k_io_yuv_to_bgr(const uint8_t* __restrict__ src, int64_t frameBytes, int H, int W, int cw, int ch,
                                                       int sx, int sy, int full, uint8_t* __restrict__ dst, int nframes)
{
    const int wq = (W + 3) >> 2;
    const int64_t total = (int64_t)nframes * H * wq;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int xq = (int)(t % wq);
        const int yy = (int)((t / wq) % H);
        const int f = (int)(t / ((int64_t)wq * H));
        const uint8_t* Y = src + f * frameBytes + (int64_t)yy * W;
        const uint8_t* U = src + f * frameBytes + (int64_t)H * W + (int64_t)min(yy >> sy, max(ch - 1, 0)) * cw;
        const uint8_t* V = U + (int64_t)ch * cw;
        uint8_t* o = dst + ((int64_t)f * H + yy) * W * 3;
        const int x0 = xq * 4;
        uint8_t out[12];
        const int n = min(4, W - x0);
        for (int j = 0; j < n; ++j) {
            const int x = x0 + j;
            int u = 128, v = 128;
            if (cw > 0) {
                const int xc = min(x >> sx, cw - 1);
                u = U[xc];
                v = V[xc];
            }
            int b, g, r;
            yuv2bgr_px(Y[x], u, v, full != 0, b, g, r);
            if (cw == 0) g = r = b;                          // mono: the reader repeats the blue channel (video_io.Y4mVideo.read)
            out[3 * j] = (uint8_t)b;
            out[3 * j + 1] = (uint8_t)g;
            out[3 * j + 2] = (uint8_t)r;
        }
        uint8_t* p = o + (int64_t)x0 * 3;
        if (n == 4 && (((uintptr_t)p) & 3) == 0) {
            uint32_t* p32 = reinterpret_cast<uint32_t*>(p);
            p32[0] = out[0] | (out[1] << 8) | (out[2] << 16) | ((uint32_t)out[3] << 24);
            p32[1] = out[4] | (out[5] << 8) | (out[6] << 16) | ((uint32_t)out[7] << 24);
            p32[2] = out[8] | (out[9] << 8) | (out[10] << 16) | ((uint32_t)out[11] << 24);
        } else {
            for (int j = 0; j < 3 * n; ++j) p[j] = out[j];
        }
    }
}

// BGR [n][H][W][3] -> planar frames [Y: H*W][U][V], chroma 4:4:4 (sub == 0) or 4:2:0 (sub == 1: the rounded mean of the 2x2 block of
// per-pixel u8 chroma values, edge pixels repeated for odd sizes -- video_io.Y4mWriter).  One thread = one 2x2 block.
__global__ void __launch_bounds__(256) k_io_bgr_to_yuv(const uint8_t* __restrict__ src, int H, int W, int sub, int full, uint8_t* __restrict__ dst,
                                                       int64_t frameBytes, int nframes)
{
    const int bw = (W + 1) >> 1, bh = (H + 1) >> 1;
    const int cw = sub ? bw : W, chh = sub ? bh : H;
    const int64_t total = (int64_t)nframes * bh * bw;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int bx = (int)(t % bw);
        const int by = (int)((t / bw) % bh);
        const int f = (int)(t / ((int64_t)bw * bh));
        const uint8_t* in = src + (int64_t)f * H * W * 3;
        uint8_t* Y = dst + f * frameBytes;
        uint8_t* U = Y + (int64_t)H * W;
        uint8_t* V = U + (int64_t)chh * cw;
        int us = 0, vs = 0;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int y = by * 2 + dy, x = bx * 2 + dx;
                const int yc = min(y, H - 1), xc = min(x, W - 1);       // edge padding of the sub-sampler
                const uint8_t* p = in + ((int64_t)yc * W + xc) * 3;
                int yv, u, v;
                bgr2yuv_px(p[0], p[1], p[2], full != 0, yv, u, v);
                us += u;
                vs += v;
                if (y < H && x < W) {
                    Y[(int64_t)y * W + x] = (uint8_t)yv;
                    if (!sub) {
                        U[(int64_t)y * W + x] = (uint8_t)u;
                        V[(int64_t)y * W + x] = (uint8_t)v;
                    }
                }
            }
        if (sub) {
            U[(int64_t)by * cw + bx] = (uint8_t)((us + 2) >> 2);
            V[(int64_t)by * cw + bx] = (uint8_t)((vs + 2) >> 2);
        }
    }
}

static inline int grid_for(int64_t total)
{
    const int64_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

extern "C" int vsr_io_yuv_to_bgr(const uint8_t* planes_dev, int64_t frame_bytes, int H, int W, int chroma_w, int chroma_h, int full_range,
                                 uint8_t* bgr_dev, int nframes, void* stream)
{
    if (planes_dev == nullptr || bgr_dev == nullptr || H <= 0 || W <= 0 || nframes < 0) return VSR_ERR_ARG;
    if (nframes == 0) return 0;
    int sx = 0, sy = 0;
    if (chroma_w > 0) {
        if (chroma_w == W) sx = 0; else if (chroma_w == (W + 1) / 2) sx = 1; else return VSR_ERR_ARG;
        if (chroma_h == H) sy = 0; else if (chroma_h == (H + 1) / 2) sy = 1; else return VSR_ERR_ARG;
    }
    if (frame_bytes < (int64_t)H * W + 2 * (int64_t)chroma_w * chroma_h) return VSR_ERR_ARG;
    const int64_t total = (int64_t)nframes * H * ((W + 3) / 4);
    hipLaunchKernelGGL(k_io_yuv_to_bgr, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, planes_dev, frame_bytes, H, W, chroma_w,
                       chroma_h, sx, sy, full_range, bgr_dev, nframes);
    DONE();
}

extern "C" int vsr_io_bgr_to_yuv(const uint8_t* bgr_dev, int H, int W, int subsample_420, int full_range, uint8_t* planes_dev,
                                 int64_t frame_bytes, int nframes, void* stream)
{
    if (planes_dev == nullptr || bgr_dev == nullptr || H <= 0 || W <= 0 || nframes < 0) return VSR_ERR_ARG;
    if (nframes == 0) return 0;
    const int64_t cw = subsample_420 ? (W + 1) / 2 : W, ch = subsample_420 ? (H + 1) / 2 : H;
    if (frame_bytes < (int64_t)H * W + 2 * cw * ch) return VSR_ERR_ARG;
    const int64_t total = (int64_t)nframes * ((H + 1) / 2) * ((W + 1) / 2);
    hipLaunchKernelGGL(k_io_bgr_to_yuv, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, bgr_dev, H, W, subsample_420, full_range,
                       planes_dev, frame_bytes, nframes);
    DONE();
}
