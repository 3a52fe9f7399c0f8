// HBM-bound kernels of the optical-flow stages: RAFT (plan ops of kind OP_EW, raft_plan.h) and, further down, the recurrent
// flow completion (rfc_plan.h).  Each one cites the reference lines it
// stands for (backend/inpaint/video/raft/*).  Plain expressions under "fp contract(off)": the sampling coordinates
// follow torch's op order (normalise to [-1,1], un-normalise) so that the CPU oracle and this file round alike.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flow_kernels.h"

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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// EW_IM2COL7_U8: to_tensors()(frames) * 2 - 1 (propainter_inpaint.py:214) fused with the im2col of the stem conv
// (7x7, stride 2, pad 3; extractor.py:135): row (f, oy, ox), 160 columns, k = (ky*7+kx)*3 + c_rgb, 147..159 zero.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_im2col7_u8(const uint8_t* __restrict__ img, int n, int H, int W, int bgr, float* __restrict__ out)
{
    const int oh = H / 2, ow = W / 2;
    const int64_t total = (int64_t)n * oh * ow * 40;
    GRID_STRIDE(i, total) {
        const int q = (int)(i % 40);
        const int64_t m = i / 40;
        const int ox = (int)(m % ow), oy = (int)((m / ow) % oh), f = (int)(m / ((int64_t)ow * oh));
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            float val = 0.f;
            if (k < 147) {
                const int tap = k / 3, c = k - 3 * tap;
                const int ky = tap / 7, kx = tap - 7 * ky;
                const int y = 2 * oy - 3 + ky, x = 2 * ox - 3 + kx;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    const uint8_t u = img[(((int64_t)f * H + y) * W + x) * 3 + (bgr ? 2 - c : c)];
                    val = ((float)u / 255.0f) * 2.0f - 1.0f;
                }
            }
            v[j] = val;
        }
        *reinterpret_cast<f32x4*>(out + m * 160 + 4 * q) = v;
    }
}

// ---------------------------------------------------------------------------------------
// EW_INORM_STATS: nn.InstanceNorm2d (extractor.py:31-35,128-129; eps 1e-5, biased variance, no affine): per (frame,
// channel) mean and 1/sqrt(var+eps) over the interior of an NHWC activation.  One workgroup per (frame, 32 channels,
// pixel slice); fp64 accumulation, slices combined with fp64 atomics into acc[f][C][2], finished by k_inorm_finish.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_inorm_partial(const float* __restrict__ x, int H, int W, int C, int halo, int slices, double* __restrict__ acc)
{
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int chunk = blockIdx.x % (C / 32);
    const int slice = (blockIdx.x / (C / 32)) % slices;
    const int f = blockIdx.x / ((C / 32) * slices);
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    const int c = chunk * 32 + cl;
    const int64_t npix = (int64_t)H * W;
    const int64_t p0 = npix * slice / slices, p1 = npix * (slice + 1) / slices;
    double s = 0.0, ss = 0.0;
    for (int64_t p = p0 + pl; p < p1; p += 8) {
        const int y = (int)(p / W), xx = (int)(p - (int64_t)y * W);
        const float v = x[(((int64_t)f * Hp + y + halo) * Wp + xx + halo) * C + c];
        s += (double)v;
        ss += (double)v * (double)v;
    }
    __shared__ double red[2][8][32];
    red[0][pl][cl] = s;
    red[1][pl][cl] = ss;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
        atomicAdd(acc + ((int64_t)f * C + c) * 2, s);
        atomicAdd(acc + ((int64_t)f * C + c) * 2 + 1, ss);
    }
}
// the same sums with four channels per thread: a workgroup owns (frame, pixel slice) over ALL channels, a thread one float4 of a pixel
// (C / 4 threads per pixel, 256 / (C / 4) pixels per pass), so the loads are 16 bytes wide and a wave covers whole pixels instead of
// 128-byte eighths of them; fp64 partial sums meet in LDS, then one fp64 atomic per (channel, moment) and workgroup as before
// (k_raft_inorm_partial ran at 1.3 TB/s: 17.7 ms of a 68-frame batch, profiles/r06_propainter_f32_raft_kernel_stats.csv)
__global__ void __launch_bounds__(256)
k_raft_inorm_partial4(const float* __restrict__ x, int H, int W, int C, int halo, int slices, double* __restrict__ acc)
{
    const int C4 = C >> 2;
    const int ppi = 256 / C4;                                  // pixels per pass
    const int cl = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int slice = blockIdx.x % slices, f = blockIdx.x / slices;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    const int64_t npix = (int64_t)H * W;
    const int64_t p0 = npix * slice / slices, p1 = npix * (slice + 1) / slices;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    if (pl < ppi) {
#pragma unroll 4
        for (int64_t p = p0 + pl; p < p1; p += ppi) {
            const int y = (int)(p / W), xx = (int)(p - (int64_t)y * W);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)f * Hp + y + halo) * Wp + xx + halo) * C + 4 * cl);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; ss[e] += (double)v[e] * (double)v[e]; }
        }
    }
    __shared__ double red[8][256];
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[e][threadIdx.x] = s[e]; red[4 + e][threadIdx.x] = ss[e]; }
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < ppi; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += red[e][k * C4 + cl]; ss[e] += red[4 + e][k * C4 + cl]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(acc + ((int64_t)f * C + 4 * cl + e) * 2, s[e]);
            atomicAdd(acc + ((int64_t)f * C + 4 * cl + e) * 2 + 1, ss[e]);
        }
    }
}
__global__ void __launch_bounds__(256)
k_raft_inorm_finish(const double* __restrict__ acc, int n, int C, int64_t npix, float* __restrict__ stats)
{
    const int64_t total = (int64_t)n * C;
    GRID_STRIDE(i, total) {
        const double mean = acc[2 * i] / (double)npix;
        double var = acc[2 * i + 1] / (double)npix - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// EW_INORM_APPLY: y = (x - mean) * rstd, optional ReLU, optional "+ residual, ReLU" (ResidualBlock.forward,
// extractor.py:48-58), in place on the interior of x
__global__ void __launch_bounds__(256)
k_raft_inorm_apply(float* __restrict__ x, int n, int H, int W, int C, int halo, const float* __restrict__ stats, int relu,
                   const float* __restrict__ res, int resHalo)
{
    const int C4 = C / 4;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo, Wr = W + 2 * resHalo, Hr = H + 2 * resHalo;
    const int64_t total = (int64_t)n * H * W * C4;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i % C4);
        const int xx = (int)((i / C4) % W), y = (int)((i / ((int64_t)C4 * W)) % H), f = (int)(i / ((int64_t)C4 * W * H));
        float* p = x + (((int64_t)f * Hp + y + halo) * Wp + xx + halo) * C + 4 * c4;
        f32x4 v = *reinterpret_cast<f32x4*>(p);
        const float* st = stats + ((int64_t)f * C + 4 * c4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = (v[j] - st[2 * j]) * st[2 * j + 1];
            if (relu) t = fmaxf(t, 0.f);
            v[j] = t;
        }
        if (res != nullptr) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(res + (((int64_t)f * Hr + y + resHalo) * Wr + xx + resHalo) * C + 4 * c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j] + r[j], 0.f);
        }
        *reinterpret_cast<f32x4*>(p) = v;
    }
}

// ---------------------------------------------------------------------------------------
// EW_CTX_SPLIT: net, inp = split(cnet(image1)); net = tanh(net); inp = relu(inp) (raft.py:112-116), written into the
// recurrent-state buffer of every pair-direction (channels 0..127 = h, 128..255 = inp)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_ctx_split(const float* __restrict__ cmap, const int32_t* __restrict__ frameOf, int pairs, int h, int w, int halo, int Chx,
                 float* __restrict__ hxr)
{
    const int64_t total = (int64_t)pairs * h * w * 64;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i & 63);
        const int64_t m = i >> 6;
        const int xx = (int)(m % w), y = (int)((m / w) % h), p = (int)(m / ((int64_t)w * h));
        f32x4 v = *reinterpret_cast<const f32x4*>(cmap + (((int64_t)frameOf[p] * h + y) * w + xx) * 256 + 4 * c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = c4 < 32 ? tanhf(v[j]) : fmaxf(v[j], 0.f);
        *reinterpret_cast<f32x4*>(hxr + (((int64_t)p * Hp + y + halo) * Wp + xx + halo) * Chx + 4 * c4) = v;
    }
}

// ---------------------------------------------------------------------------------------
// EW_FLOW_UPDATE: coords1 = coords0 (init) or coords1 + delta_flow (raft.py:118,133); flow = coords1 - coords0 goes to
// the plain flow buffer (im2col source, upsampling) and to channels chFlow, chFlow+1 of the recurrent state
// (torch.cat([out, flow]), update.py:98)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_flow_update(const float* __restrict__ delta, int ldDelta, float* __restrict__ coords, float* __restrict__ flow,
                   float* __restrict__ hxr, int pairs, int h, int w, int init, int halo, int Chx, int chFlow)
{
    const int64_t total = (int64_t)pairs * h * w;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(m, total) {
        const int xx = (int)(m % w), y = (int)((m / w) % h), p = (int)(m / ((int64_t)w * h));
        float cx = (float)xx, cy = (float)y;
        if (!init) {
            cx = coords[2 * m] + delta[m * ldDelta];
            cy = coords[2 * m + 1] + delta[m * ldDelta + 1];
        }
        coords[2 * m] = cx;
        coords[2 * m + 1] = cy;
        const float fx = cx - (float)xx, fy = cy - (float)y;
        flow[2 * m] = fx;
        flow[2 * m + 1] = fy;
        float* s = hxr + (((int64_t)p * Hp + y + halo) * Wp + xx + halo) * Chx + chFlow;
        s[0] = fx;
        s[1] = fy;
    }
}

// EW_IM2COL7_FLOW: im2col of convf1 (7x7, pad 3, 2 -> 128; update.py:85): 128 columns, k = (ky*7+kx)*2 + c, 98.. zero
__global__ void __launch_bounds__(256)
k_raft_im2col7_flow(const float* __restrict__ flow, int pairs, int h, int w, float* __restrict__ out)
{
    const int64_t total = (int64_t)pairs * h * w * 32;
    GRID_STRIDE(i, total) {
        const int q = (int)(i & 31);
        const int64_t m = i >> 5;
        const int xx = (int)(m % w), y = (int)((m / w) % h), p = (int)(m / ((int64_t)w * h));
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            float val = 0.f;
            if (k < 98) {
                const int tap = k >> 1, c = k & 1;
                const int ky = tap / 7, kx = tap - 7 * ky;
                const int yy = y - 3 + ky, x2 = xx - 3 + kx;
                if (yy >= 0 && yy < h && x2 >= 0 && x2 < w) val = flow[(((int64_t)p * h + yy) * w + x2) * 2 + c];
            }
            v[j] = val;
        }
        *reinterpret_cast<f32x4*>(out + m * 128 + 4 * q) = v;
    }
}

// ---------------------------------------------------------------------------------------
// EW_AVGPOOL2: F.avg_pool2d(corr, 2, stride=2) (corr.py:25-27) on [rows][hs][ws] -> [rows][hs/2][ws/2]
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_avgpool2(const float* __restrict__ src, int64_t rows, int hs, int ws, float* __restrict__ dst)
{
    const int hd = hs / 2, wd = ws / 2;
    const int64_t total = rows * hd * wd;
    GRID_STRIDE(i, total) {
        const int xx = (int)(i % wd), y = (int)((i / wd) % hd);
        const int64_t r = i / ((int64_t)wd * hd);
        const float* s = src + (r * hs + 2 * y) * ws + 2 * xx;
        dst[i] = (s[0] + s[1] + s[ws] + s[ws + 1]) * 0.25f;
    }
}

// ---------------------------------------------------------------------------------------
// EW_CORR_TRANSPOSE: dst[p] = src[p]^T for n square planes of hw x hw floats (raft_plan.cpp: the backward pair-directions' correlation
// volumes).  64 x 64 tiles through LDS (row pitch 65: conflict-free both ways); both the reads and the writes are 256-byte runs per wave.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_corr_transpose(const float* __restrict__ src, float* __restrict__ dst, int hw, int tiles)
{
    __shared__ float tile[64][65];
    const int64_t plane = (int64_t)hw * hw;
    const int p = blockIdx.x / (tiles * tiles), tt = blockIdx.x - p * tiles * tiles;
    const int ty = tt / tiles, tx = tt - ty * tiles;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const float* s = src + p * plane;
    float* d = dst + p * plane;
    const int x = tx * 64 + lx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int y = ty * 64 + ly + 4 * r;
        if (x < hw && y < hw) tile[ly + 4 * r][lx] = s[(int64_t)y * hw + x];
    }
    __syncthreads();
    const int ox = ty * 64 + lx;                 // output column = input row
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oy = tx * 64 + ly + 4 * r;     // output row = input column
        if (ox < hw && oy < hw) d[(int64_t)oy * hw + ox] = tile[lx][ly + 4 * r];
    }
}

// ---------------------------------------------------------------------------------------
// EW_CORR_LOOKUP: CorrBlock.__call__ (corr.py:29-50) + bilinear_sampler (utils/utils.py:57-70; grid_sample,
// align_corners=True, zero padding).  Row m, column lvl*81 + 9*i + j samples level lvl at
// (x, y) = (coords.x / 2^lvl + (i-4), coords.y / 2^lvl + (j-4)) -- the reference's meshgrid(dy, dx) is added to (x, y),
// so the first window index walks x.  Columns 324..ld-1 are zero (K padding of convc1).
// ---------------------------------------------------------------------------------------
struct RaftLevels {
    const float* base[4];
    int h[4], w[4];
};
__global__ void __launch_bounds__(256)
k_raft_corr_lookup(RaftLevels L, const float* __restrict__ coords, int64_t M, int ld, float* __restrict__ out)
{
    const int64_t total = M * ld;
    GRID_STRIDE(i, total) {
        const int col = (int)(i % ld);
        const int64_t m = i / ld;
        float val = 0.f;
        if (col < 324) {
            const int lvl = col / 81, rem = col - 81 * lvl;
            const int wi = rem / 9, wj = rem - 9 * wi;
            const int hh = L.h[lvl], ww = L.w[lvl];
            const float scale = (float)(1 << lvl);
            const float x = coords[2 * m] / scale + (float)(wi - 4);
            const float y = coords[2 * m + 1] / scale + (float)(wj - 4);
            // bilinear_sampler: xgrid = 2*x/(W-1) - 1 ; grid_sample un-normalises ((g + 1) / 2) * (W - 1)
            const float gx = 2.0f * x / (float)(ww - 1) - 1.0f, gy = 2.0f * y / (float)(hh - 1) - 1.0f;
            const float ix = ((gx + 1.0f) / 2.0f) * (float)(ww - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(hh - 1);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float ax = ix - fx0, ay = iy - fy0;
            const float* lv = L.base[lvl] + m * hh * ww;
            const bool xin0 = x0 >= 0 && x0 < ww, xin1 = x0 + 1 >= 0 && x0 + 1 < ww;
            const bool yin0 = y0 >= 0 && y0 < hh, yin1 = y0 + 1 >= 0 && y0 + 1 < hh;
            const float nw = (yin0 && xin0) ? lv[(int64_t)y0 * ww + x0] : 0.f;
            const float ne = (yin0 && xin1) ? lv[(int64_t)y0 * ww + x0 + 1] : 0.f;
            const float sw = (yin1 && xin0) ? lv[(int64_t)(y0 + 1) * ww + x0] : 0.f;
            const float se = (yin1 && xin1) ? lv[(int64_t)(y0 + 1) * ww + x0 + 1] : 0.f;
            val = nw * ((1.0f - ax) * (1.0f - ay)) + ne * (ax * (1.0f - ay)) + sw * ((1.0f - ax) * ay) + se * (ax * ay);
        }
        out[i] = val;
    }
}

// ---------------------------------------------------------------------------------------
// SepConvGRU gate arithmetic (update.py:47-58).  zr [M][256] holds the raw convz | convr outputs, q [M][128] the raw convq.
// EW_GRU_RH: state[chRH..] = sigmoid(r) * h        EW_GRU_UPDATE: h = (1 - z) * h + z * tanh(q), z = sigmoid(z)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_gru_rh(const float* __restrict__ zr, float* __restrict__ hxr, int pairs, int h, int w, int halo, int Chx, int chH, int chRH)
{
    const int64_t total = (int64_t)pairs * h * w * 32;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i & 31);
        const int64_t m = i >> 5;
        const int xx = (int)(m % w), y = (int)((m / w) % h), p = (int)(m / ((int64_t)w * h));
        float* s = hxr + (((int64_t)p * Hp + y + halo) * Wp + xx + halo) * Chx;
        const f32x4 r = *reinterpret_cast<const f32x4*>(zr + m * 256 + 128 + 4 * c4);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(s + chH + 4 * c4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = sigmoidf_(r[j]) * hv[j];
        *reinterpret_cast<f32x4*>(s + chRH + 4 * c4) = o;
    }
}
__global__ void __launch_bounds__(256)
k_raft_gru_update(const float* __restrict__ zr, const float* __restrict__ q, float* __restrict__ hxr, int pairs, int h, int w, int halo,
                  int Chx, int chH)
{
    const int64_t total = (int64_t)pairs * h * w * 32;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i & 31);
        const int64_t m = i >> 5;
        const int xx = (int)(m % w), y = (int)((m / w) % h), p = (int)(m / ((int64_t)w * h));
        float* s = hxr + (((int64_t)p * Hp + y + halo) * Wp + xx + halo) * Chx + chH + 4 * c4;
        const f32x4 z = *reinterpret_cast<const f32x4*>(zr + m * 256 + 4 * c4);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + m * 128 + 4 * c4);
        f32x4 hv = *reinterpret_cast<f32x4*>(s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float zz = sigmoidf_(z[j]);
            hv[j] = (1.0f - zz) * hv[j] + zz * tanhf(qv[j]);
        }
        *reinterpret_cast<f32x4*>(s) = hv;
    }
}

// ---------------------------------------------------------------------------------------
// EW_CONVEX_UP: RAFT.upsample_flow (raft.py:72-84): mask [M][576] viewed as (9, 8, 8), softmax over the 9 neighbours,
// convex combination of the 3x3 neighbourhood of 8*flow -> out [pairs][2][8h][8w]
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_raft_convex_up(const float* __restrict__ flow, const float* __restrict__ mask, int pairs, int h, int w, float* __restrict__ out)
{
    const int64_t total = (int64_t)pairs * h * w * 64;
    const int H = 8 * h, W = 8 * w;
    GRID_STRIDE(i, total) {
        const int sub = (int)(i & 63);
        const int64_t m = i >> 6;
        const int xx = (int)(m % w), y = (int)((m / w) % h), p = (int)(m / ((int64_t)w * h));
        const float* mk = mask + m * 576 + sub;
        float e[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = mk[64 * k]; mx = fmaxf(mx, e[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = expf(e[k] - mx); sum += e[k]; }
        float ux = 0.f, uy = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, x2 = xx + k % 3 - 1;
            float fx = 0.f, fy = 0.f;
            if (yy >= 0 && yy < h && x2 >= 0 && x2 < w) {
                const float* fl = flow + (((int64_t)p * h + yy) * w + x2) * 2;
                fx = 8.0f * fl[0];
                fy = 8.0f * fl[1];
            }
            const float wgt = e[k] / sum;
            ux += wgt * fx;
            uy += wgt * fy;
        }
        const int oy = 8 * y + (sub >> 3), ox = 8 * xx + (sub & 7);
        out[(((int64_t)p * 2 + 0) * H + oy) * W + ox] = ux;
        out[(((int64_t)p * 2 + 1) * H + oy) * W + ox] = uy;
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
#define LAUNCH(kernel, total, ...)                                                                                  \
    do {                                                                                                            \
        if ((total) <= 0) return 0;                                                                                 \
        hipLaunchKernelGGL(kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);          \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                            \
    } while (0)

extern "C" int vsr_raft_launch_im2col7_u8(const uint8_t* img, int n, int H, int W, int bgr, float* out, void* stream)
{
    LAUNCH(k_raft_im2col7_u8, (int64_t)n * (H / 2) * (W / 2) * 40, img, n, H, W, bgr, out);
}
extern "C" int vsr_raft_launch_inorm_stats(const float* x, int n, int H, int W, int C, int halo, double* acc, float* stats, void* stream)
{
    if (n <= 0) return 0;
    if (C % 32) return -1;
    if (hipMemsetAsync(acc, 0, (size_t)n * C * 2 * sizeof(double), (hipStream_t)stream) != hipSuccess) return -1;
    int slices = (int)(((int64_t)H * W + 4095) / 4096);            // >= 4096 pixels per workgroup, at most 64 slices
    if (slices > 64) slices = 64;
    if (slices < 1) slices = 1;
    if (C <= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(k_raft_inorm_partial4, dim3(n * slices), dim3(256), 0, (hipStream_t)stream, x, H, W, C, halo, slices, acc);
    else
    hipLaunchKernelGGL(k_raft_inorm_partial, dim3(n * (C / 32) * slices), dim3(256), 0, (hipStream_t)stream, x, H, W, C, halo, slices, acc);
    hipLaunchKernelGGL(k_raft_inorm_finish, dim3(grid_for((int64_t)n * C)), dim3(256), 0, (hipStream_t)stream, acc, n, C,
                       (int64_t)H * W, stats);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int vsr_raft_launch_inorm_apply(float* x, int n, int H, int W, int C, int halo, const float* stats, int relu, const float* res,
                                           int resHalo, void* stream)
{
    LAUNCH(k_raft_inorm_apply, (int64_t)n * H * W * (C / 4), x, n, H, W, C, halo, stats, relu, res, resHalo);
}
extern "C" int vsr_raft_launch_ctx_split(const float* cmap, const int32_t* frameOf, int pairs, int h, int w, int halo, int Chx, float* hxr,
                                         void* stream)
{
    LAUNCH(k_raft_ctx_split, (int64_t)pairs * h * w * 64, cmap, frameOf, pairs, h, w, halo, Chx, hxr);
}
extern "C" int vsr_raft_launch_flow_update(const float* delta, int ldDelta, float* coords, float* flow, float* hxr, int pairs, int h, int w,
                                           int init, int halo, int Chx, int chFlow, void* stream)
{
    LAUNCH(k_raft_flow_update, (int64_t)pairs * h * w, delta, ldDelta, coords, flow, hxr, pairs, h, w, init, halo, Chx, chFlow);
}
extern "C" int vsr_raft_launch_im2col7_flow(const float* flow, int pairs, int h, int w, float* out, void* stream)
{
    LAUNCH(k_raft_im2col7_flow, (int64_t)pairs * h * w * 32, flow, pairs, h, w, out);
}
extern "C" int vsr_raft_launch_avgpool2(const float* src, int64_t rows, int hs, int ws, float* dst, void* stream)
{
    LAUNCH(k_raft_avgpool2, rows * (hs / 2) * (ws / 2), src, rows, hs, ws, dst);
}
extern "C" int vsr_raft_launch_corr_transpose(const float* src, float* dst, int n, int hw, void* stream)
{
    if (n <= 0 || hw <= 0) return 0;
    const int tiles = (hw + 63) / 64;
    if ((int64_t)n * tiles * tiles > 2147483647LL) return -1;
    hipLaunchKernelGGL(k_raft_corr_transpose, dim3((unsigned)(n * tiles * tiles)), dim3(256), 0, (hipStream_t)stream, src, dst, hw, tiles);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int vsr_raft_launch_corr_lookup(const float* const* levels, const int* lvlH, const int* lvlW, const float* coords, int64_t M, int ld,
                                           float* out, void* stream)
{
    RaftLevels L;
    for (int l = 0; l < 4; ++l) { L.base[l] = levels[l]; L.h[l] = lvlH[l]; L.w[l] = lvlW[l]; }
    LAUNCH(k_raft_corr_lookup, M * ld, L, coords, M, ld, out);
}
extern "C" int vsr_raft_launch_gru_rh(const float* zr, float* hxr, int pairs, int h, int w, int halo, int Chx, int chH, int chRH, void* stream)
{
    LAUNCH(k_raft_gru_rh, (int64_t)pairs * h * w * 32, zr, hxr, pairs, h, w, halo, Chx, chH, chRH);
}
extern "C" int vsr_raft_launch_gru_update(const float* zr, const float* q, float* hxr, int pairs, int h, int w, int halo, int Chx, int chH,
                                          void* stream)
{
    LAUNCH(k_raft_gru_update, (int64_t)pairs * h * w * 32, zr, q, hxr, pairs, h, w, halo, Chx, chH);
}
extern "C" int vsr_raft_launch_convex_up(const float* flow, const float* mask, int pairs, int h, int w, float* out, void* stream)
{
    LAUNCH(k_raft_convex_up, (int64_t)pairs * h * w * 64, flow, mask, pairs, h, w, out);
}

// =======================================================================================
// Recurrent flow completion (rfc_plan.h; reference backend/inpaint/video/model/recurrent_flow_completion.py)
// =======================================================================================

// ---------------------------------------------------------------------------------------
// EW_RFC_IM2COL5: forward_bidirect_flow's masking (:322-333) + cat((masked_flows, masks)) (:281) fused with the im2col of
// the stem Conv3d (1,5,5), stride (1,2,2), replicate padding (:209-211).  Two sequences of T = t-1 steps: s = 0 the
// forward flows with masks[:-1], s = 1 the backward flows with masks[1:], both flipped in time.  Row ((i*2+s), oy, ox),
// 96 columns, k = (ky*5+kx)*3 + c with c = (flow_x*(1-m), flow_y*(1-m), m); 75.. zero.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rfc_im2col5(const float* __restrict__ ff, const float* __restrict__ fb, const uint8_t* __restrict__ mask, int t, int H, int W,
              float* __restrict__ out)
{
    const int T = t - 1, oh = H / 2, ow = W / 2;
    const int64_t total = (int64_t)2 * T * oh * ow * 24;
    GRID_STRIDE(i, total) {
        const int q = (int)(i % 24);
        const int64_t m = i / 24;
        const int ox = (int)(m % ow), oy = (int)((m / ow) % oh);
        const int fs = (int)(m / ((int64_t)ow * oh));
        const int s = fs & 1, step = fs >> 1;
        const float* fl = s == 0 ? ff + (int64_t)step * 2 * H * W : fb + (int64_t)(T - 1 - step) * 2 * H * W;
        const uint8_t* mk = mask + (int64_t)(s == 0 ? step : T - step) * H * W;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            float val = 0.f;
            if (k < 75) {
                const int tap = k / 3, c = k - 3 * tap;
                const int ky = tap / 5, kx = tap - 5 * ky;
                int y = 2 * oy - 2 + ky, x = 2 * ox - 2 + kx;
                y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);            // padding_mode='replicate'
                x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
                const float mv = mk[(int64_t)y * W + x] ? 1.0f : 0.0f;
                val = c == 2 ? mv : fl[((int64_t)c * H + y) * W + x] * (1.0f - mv);
            }
            v[j] = val;
        }
        *reinterpret_cast<f32x4*>(out + m * 96 + 4 * q) = v;
    }
}

// ---------------------------------------------------------------------------------------
// EW_DEFORM_COLS: SecondOrderDeformableAlignment.forward (:31-46) up to the contraction: offset = 5*tanh(o[0:288]),
// mask = sigmoid(o[288:432]); torchvision.ops.deform_conv2d's column matrix for x = cat[srcA, srcB] (2 x 128 channels,
// 16 offset groups of 16 channels, 3x3, stride 1, pad 1): cols[m][((ci/32)*9 + k)*32 + ci%32] =
// mask[g,k] * bilinear(x[ci], y - 1 + ky + off_y[g,k], x - 1 + kx + off_x[g,k]), zero outside the image
// (offset channel g*18 + 2k (+1), mask channel g*9 + k).  One thread per (pixel, group, tap): 16 channels.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_deform_cols(const float* __restrict__ srcA, const float* __restrict__ srcB, const float* __restrict__ off, int ldOff, float maxMag, int n,
              int h, int w, int halo, int C, float* __restrict__ cols)
{
    const int64_t total = (int64_t)n * h * w * 144;
    const int Wp = w + 2 * halo, Hp = h + 2 * halo;
    GRID_STRIDE(i, total) {
        const int gk = (int)(i % 144);
        const int64_t m = i / 144;
        const int g = gk / 9, k = gk - 9 * g;
        const int xx = (int)(m % w), y = (int)((m / w) % h), f = (int)(m / ((int64_t)w * h));
        const float* o = off + m * ldOff;
        const float dy = maxMag * tanhf(o[g * 18 + 2 * k]), dx = maxMag * tanhf(o[g * 18 + 2 * k + 1]);
        const float mk = sigmoidf_(o[288 + g * 9 + k]);
        const float py = (float)(y - 1 + k / 3) + dy, px = (float)(xx - 1 + k % 3) + dx;
        const float fy0 = floorf(py), fx0 = floorf(px);
        const int y0 = (int)fy0, x0 = (int)fx0;
        const float ly = py - fy0, lx = px - fx0;
        const float* src = (g < 8 ? srcA : srcB) + (int64_t)f * Hp * Wp * C + (g & 7) * 16;      // channels 16g.. of cat[srcA, srcB]
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
            const int yy = y0 + (corner >> 1), x2 = x0 + (corner & 1);
            const float wgt = ((corner >> 1) ? ly : 1.0f - ly) * ((corner & 1) ? lx : 1.0f - lx);
            if (yy >= 0 && yy < h && x2 >= 0 && x2 < w) {
                const float* p = src + ((int64_t)(yy + halo) * Wp + x2 + halo) * C;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = acc[j][e] + v[e] * wgt;
                }
            }
        }
        const int ci0 = g * 16;                                     // channel of cat[srcA, srcB]
        float* dst = cols + m * (9 * 2 * C) + ((ci0 / 32) * 9 + k) * 32 + (ci0 % 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = acc[j][e] * mk;
            *reinterpret_cast<f32x4*>(dst + 4 * j) = acc[j];
        }
    }
}

// ---------------------------------------------------------------------------------------
// EW_RFC_COMBINE: un-flip the backward sequence (:334-336) and combine_flow (:341-348):
// out = pred * mask + flow * (1 - mask), planar [T][2][H][W]; pred [(i*2+s)][H][W][ld] (columns 0, 1)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rfc_combine(const float* __restrict__ pred, int ld, const float* __restrict__ ff, const float* __restrict__ fb,
              const uint8_t* __restrict__ mask, int t, int H, int W, float* __restrict__ outF, float* __restrict__ outB)
{
    const int T = t - 1;
    const int64_t total = (int64_t)2 * T * H * W;
    GRID_STRIDE(i, total) {
        const int xx = (int)(i % W), y = (int)((i / W) % H);
        const int fs = (int)(i / ((int64_t)W * H));
        const int s = fs & 1, step = fs >> 1;
        const int fi = s == 0 ? step : T - 1 - step;               // flow index inside its direction
        const float mv = mask[((int64_t)(s == 0 ? fi : fi + 1) * H + y) * W + xx] ? 1.0f : 0.0f;
        const float* fl = (s == 0 ? ff : fb) + (int64_t)fi * 2 * H * W;
        float* o = (s == 0 ? outF : outB) + (int64_t)fi * 2 * H * W;
        const float* p = pred + i * ld;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t at = ((int64_t)c * H + y) * W + xx;
            o[at] = p[c] * mv + (fl[at] * (1.0f - mv)) * (1.0f - mv);
        }
    }
}

extern "C" int vsr_rfc_launch_im2col5(const float* ff, const float* fb, const uint8_t* mask, int t, int H, int W, float* out, void* stream)
{
    LAUNCH(k_rfc_im2col5, (int64_t)2 * (t - 1) * (H / 2) * (W / 2) * 24, ff, fb, mask, t, H, W, out);
}
extern "C" int vsr_rfc_launch_deform_cols(const float* srcA, const float* srcB, const float* off, int ldOff, float maxMag, int n, int h, int w,
                                          int halo, int C, float* cols, void* stream)
{
    LAUNCH(k_deform_cols, (int64_t)n * h * w * 144, srcA, srcB, off, ldOff, maxMag, n, h, w, halo, C, cols);
}
extern "C" int vsr_rfc_launch_combine(const float* pred, int ld, const float* ff, const float* fb, const uint8_t* mask, int t, int H, int W,
                                      float* outF, float* outB, void* stream)
{
    LAUNCH(k_rfc_combine, (int64_t)2 * (t - 1) * H * W, pred, ld, ff, fb, mask, t, H, W, outF, outB);
}
