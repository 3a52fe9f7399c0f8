// HBM-bound kernels of the ProPainter generator path (plan ops of kind OP_EW, pp_plan.h): SURVEY.md 8(a) row a16.
// Each kernel cites the reference lines it stands for (backend/inpaint/video/model/propainter.py and
// model/modules/{flow_loss_utils,sparse_transformer}.py).  Plain expressions under "fp contract(off)"; sampling
// coordinates follow torch's op order (normalise to [-1,1], un-normalise) so the CPU oracle and this file round alike.
#include <hip/hip_runtime.h>
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

// flow_warp's sampling position (flow_loss_utils.py:33-38 + grid_sample's un-normalisation, align_corners=True)
__device__ __forceinline__ float warp_coord(float pos, int size)
{
    const float d = (float)(size - 1 > 1 ? size - 1 : 1);
    const float g = 2.0f * pos / d - 1.0f;
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}

// bilinear sample of a planar image (zeros outside), grid_sample semantics
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

// fbConsistencyCheck (propainter.py:24-33) at one pixel: 1 where the forward-backward flow error is small
__device__ __forceinline__ float fb_valid(const float* __restrict__ fprop, const float* __restrict__ fcheck, int h, int w, int y, int x)
{
    const int64_t hw = (int64_t)h * w, at = (int64_t)y * w + x;
    const float fx = fprop[at], fy = fprop[hw + at];
    const float ix = warp_coord((float)x + fx, w), iy = warp_coord((float)y + fy, h);
    const float bx = sample_bilinear(fcheck, h, w, iy, ix), by = sample_bilinear(fcheck + hw, h, w, iy, ix);
    const float dx = fx + bx, dy = fy + by;
    const float diff = dx * dx + dy * dy;
    const float mag = (fx * fx + fy * fy) + (bx * bx + by * by);
    return diff < 0.01f * mag + 0.5f ? 1.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------
// EW_PP_IMGPROP: one step of the non-learnable BidirectionalPropagation (img_propagation, propainter.py:136-165,316-319):
// flow-consistency check, nearest-neighbour warp of the propagated frame, bilinear warp of the propagated mask,
// union / update rules.  first = 1: prop = current (step 0 of a direction).  Planar fp32 [C][h][w] images, masks [h][w].
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pp_imgprop(const float* __restrict__ prevProp, const float* __restrict__ prevMask, const float* __restrict__ cur, const float* __restrict__ mcur,
             const float* __restrict__ fprop, const float* __restrict__ fcheck, int C, int h, int w, int first, float* __restrict__ prop,
             float* __restrict__ mprop)
{
    const int64_t hw = (int64_t)h * w;
    GRID_STRIDE(i, hw) {
        const int x = (int)(i % w), y = (int)(i / w);
        if (first) {
            for (int c = 0; c < C; ++c) prop[c * hw + i] = cur[c * hw + i];
            mprop[i] = mcur[i];
            continue;
        }
        const float valid = fb_valid(fprop, fcheck, h, w, y, x);
        const float ix = warp_coord((float)x + fprop[i], w), iy = warp_coord((float)y + fprop[hw + i], h);
        // mask_prop_valid = binary(flow_warp(mask_prop, flow)) -- bilinear (flow_warp's default), threshold 0.1
        const float mvalid = sample_bilinear(prevMask, h, w, iy, ix) > 0.1f ? 1.0f : 0.0f;
        const float mc = mcur[i];
        const float uni = mc * valid * (1.0f - mvalid) > 0.1f ? 1.0f : 0.0f;
        // nearest-neighbour warp of the propagated frame: grid_sample(mode='nearest') rounds half to even
        const int nx = (int)nearbyintf(ix), ny = (int)nearbyintf(iy);
        const bool inb = nx >= 0 && nx < w && ny >= 0 && ny < h;
        for (int c = 0; c < C; ++c) {
            const float wv = inb ? prevProp[c * hw + (int64_t)ny * w + nx] : 0.f;
            prop[c * hw + i] = uni * wv + (1.0f - uni) * cur[c * hw + i];
        }
        mprop[i] = mc * (1.0f - (valid * (1.0f - mvalid))) > 0.1f ? 1.0f : 0.0f;
    }
}

// EW_PP_MASK_F32 / output conversion: u8 masks (non-zero = hole) <-> fp32 {0,1}
__global__ void __launch_bounds__(256) k_pp_mask_f32(const uint8_t* __restrict__ src, int64_t n, float* __restrict__ dst)
{
    GRID_STRIDE(i, n) dst[i] = src[i] ? 1.0f : 0.0f;
}
__global__ void __launch_bounds__(256) k_pp_mask_u8(const float* __restrict__ src, int64_t n, uint8_t* __restrict__ dst)
{
    GRID_STRIDE(i, n) dst[i] = src[i] > 0.5f ? 1 : 0;
}
extern "C" int vsr_pp_launch_mask_f32(const uint8_t* src, int64_t n, float* dst, void* stream) { LAUNCH(k_pp_mask_f32, n, src, n, dst); }
extern "C" int vsr_pp_launch_mask_u8(const float* src, int64_t n, uint8_t* dst, void* stream) { LAUNCH(k_pp_mask_u8, n, src, n, dst); }

extern "C" int vsr_pp_launch_imgprop(const float* prevProp, const float* prevMask, const float* cur, const float* mcur, const float* fprop,
                                     const float* fcheck, int C, int h, int w, int first, float* prop, float* mprop, void* stream)
{
    LAUNCH(k_pp_imgprop, (int64_t)h * w, prevProp, prevMask, cur, mcur, fprop, fcheck, C, h, w, first, prop, mprop);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Plugin glue of PropainterInpaint.inpaint (propainter_inpaint.py:190-361) on the device: the frame batch stays in HBM as the
// uint8 BGR crops the caller uploaded; these three kernels replace the reference's cvtColor / to_tensors / numpy blends.
// ---------------------------------------------------------------------------------------------------------------------------
// frames = to_tensors()(RGB frames) * 2 - 1 (:193-213: x.float().div(255) * 2 - 1); masked = frames * (1 - masks_dilated) (:298)
__global__ __launch_bounds__(256) void k_pp_prepare(const uint8_t* __restrict__ bgr, const uint8_t* __restrict__ mask, int n, int h, int w,
                                                    float* __restrict__ masked)
{
    const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
    GRID_STRIDE(i, total) {
        const int64_t p = i % hw, f = i / hw;
        const float keep = 1.0f - (mask[p] ? 1.0f : 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = ((float)bgr[i * 3 + (2 - c)] / 255.0f) * 2.0f - 1.0f;
            masked[(f * 3 + c) * hw + p] = x * keep;
        }
    }
}

// updated_frames = frames * (1 - masks_dilated) + prop_imgs * masks_dilated (:314)
__global__ __launch_bounds__(256) void k_pp_compose(const uint8_t* __restrict__ bgr, const uint8_t* __restrict__ mask, const float* __restrict__ prop,
                                                    int n, int h, int w, float* __restrict__ out)
{
    const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
    GRID_STRIDE(i, total) {
        const int64_t p = i % hw, f = i / hw;
        const float m = mask[p] ? 1.0f : 0.0f;
        const float keep = 1.0f - m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = ((float)bgr[i * 3 + (2 - c)] / 255.0f) * 2.0f - 1.0f;
            const int64_t at = (f * 3 + c) * hw + p;
            out[at] = x * keep + prop[at] * m;
        }
    }
}

// one window of :345-357: img = ((pred + 1) / 2 * 255).astype(u8) * binary + frame * (1 - binary);
// comp = first visit ? img : (comp.astype(f32) * 0.5 + img.astype(f32) * 0.5).astype(u8)   (truncation after every average).
// comp is kept in the caller's BGR order (the reference swaps back at the end, :360).
__global__ __launch_bounds__(256) void k_pp_blend(const float* __restrict__ pred, const uint8_t* __restrict__ bgr, const uint8_t* __restrict__ mask,
                                                  const int32_t* __restrict__ frameIdx, const int32_t* __restrict__ first, int lt, int h, int w,
                                                  uint8_t* __restrict__ comp)
{
    const int64_t hw = (int64_t)h * w, total = (int64_t)lt * hw;
    GRID_STRIDE(i, total) {
        const int64_t p = i % hw;
        const int k = (int)(i / hw);
        const int idx = frameIdx[k];
        const bool fst = first[k] != 0, hole = mask[p] != 0;
        const int64_t at = ((int64_t)idx * hw + p) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {                        // c: RGB channel of the network output
            float v = (pred[((int64_t)k * 3 + c) * hw + p] + 1.0f) / 2.0f;
            v = v * 255.0f;
            const uint8_t img = hole ? (uint8_t)v : bgr[at + (2 - c)];       // astype(np.uint8): wraps mod 256 only outside [0, 256)
            uint8_t* o = comp + at + (2 - c);
            *o = fst ? img : (uint8_t)((float)*o * 0.5f + (float)img * 0.5f);
        }
    }
}

extern "C" {

int vsr_pp_prepare_frames(const uint8_t* bgr_dev, const uint8_t* mask_dev, int n, int h, int w, float* masked_dev, void* stream)
{
    LAUNCH(k_pp_prepare, (int64_t)n * h * w, bgr_dev, mask_dev, n, h, w, masked_dev);
}

int vsr_pp_compose_frames(const uint8_t* bgr_dev, const uint8_t* mask_dev, const float* prop_dev, int n, int h, int w, float* out_dev, void* stream)
{
    LAUNCH(k_pp_compose, (int64_t)n * h * w, bgr_dev, mask_dev, prop_dev, n, h, w, out_dev);
}

int vsr_pp_blend_window(const float* pred_dev, const uint8_t* bgr_dev, const uint8_t* mask_dev, const int32_t* frame_idx_dev,
                        const int32_t* first_dev, int lt, int h, int w, uint8_t* comp_dev, void* stream)
{
    LAUNCH(k_pp_blend, (int64_t)lt * h * w, pred_dev, bgr_dev, mask_dev, frame_idx_dev, first_dev, lt, h, w, comp_dev);
}

} // extern "C"
