// Kernels of the scene-cut detector that --inpaint-mode propainter runs before inpainting (SURVEY.md 8(f) rank 4;
// reference backend/tools/subtitle_detect.py:158-170 -> backend/scenedetect/detectors/content_detector.py:138-172):
// per frame cv2.cvtColor(BGR2HSV) on uint8, then per consecutive frame pair the sum of |difference| of every plane
// (_mean_pixel_distance, :28-35).  Integer work, HBM-bound, bit-exact; the host (backend/tools/scene_detect.py) turns
// the sums into scores and cuts.  The down-scaling in front of it (scene_manager.py:499-504) is vsr_launch_resize_u8.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vsr_hip.h"

#define DONE() return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP

// cvRound(a / (double)i) for 0 < i <= 255: round half to even of an exact rational (a tie is exactly representable and
// every other quotient is further than 1/510 from a tie, so the double division of OpenCV's tables rounds alike)
__device__ __forceinline__ int div_round_half_even(int a, int i)
{
    const int q = a / i, r2 = 2 * (a - q * i);
    return q + ((r2 > i) || (r2 == i && (q & 1)));
}

// OpenCV RGB2HSV_b (hrange 180, hsv_shift 12): sdiv_table[v] = cvRound((255 << 12) / v), hdiv_table180[d] = cvRound((180 << 12) / (6 d));
// s = (d * sdiv[v] + 2048) >> 12; h = sector offset + channel difference, (h * hdiv[d] + 2048) >> 12 (arithmetic shift), +180 if negative
__global__ void __launch_bounds__(256) k_bgr2hsv_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t npix)
{
    for (int64_t p = blockIdx.x * (int64_t)256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const int b = src[3 * p], g = src[3 * p + 1], r = src[3 * p + 2];
        const int v = max(b, max(g, r)), vmin = min(b, min(g, r)), d = v - vmin;
        const int sdiv = v ? div_round_half_even(255 << 12, v) : 0;
        const int hdiv = d ? div_round_half_even((180 << 12) / 6, d) : 0;
        const int s = (d * sdiv + (1 << 11)) >> 12;
        int h = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * d) : (r - g + 4 * d);
        h = (h * hdiv + (1 << 11)) >> 12;
        if (h < 0) h += 180;
        dst[3 * p] = (uint8_t)min(max(h, 0), 255);
        dst[3 * p + 1] = (uint8_t)s;
        dst[3 * p + 2] = (uint8_t)v;
    }
}

// sums[pair][c] += sum over pixels |hsv[pair + 1][p][c] - hsv[pair][p][c]|; blockIdx.y = pair
__global__ void __launch_bounds__(256) k_absdiff_sums(const uint8_t* __restrict__ hsv, int64_t pix, unsigned long long* __restrict__ sums)
{
    const uint8_t* a = hsv + (int64_t)blockIdx.y * pix * 3;
    const uint8_t* b = a + pix * 3;
    unsigned s0 = 0, s1 = 0, s2 = 0;
    for (int64_t p = blockIdx.x * (int64_t)256 + threadIdx.x; p < pix; p += (int64_t)gridDim.x * 256) {
        s0 += (unsigned)abs((int)a[3 * p] - (int)b[3 * p]);
        s1 += (unsigned)abs((int)a[3 * p + 1] - (int)b[3 * p + 1]);
        s2 += (unsigned)abs((int)a[3 * p + 2] - (int)b[3 * p + 2]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[blockIdx.y * 3 + 0], (unsigned long long)s0);
        atomicAdd(&sums[blockIdx.y * 3 + 1], (unsigned long long)s1);
        atomicAdd(&sums[blockIdx.y * 3 + 2], (unsigned long long)s2);
    }
}

extern "C" {

int vsr_launch_bgr2hsv_u8(const uint8_t* bgr_dev, uint8_t* hsv_dev, int64_t npix, void* stream)
{
    if (npix <= 0) return 0;
    if (!bgr_dev || !hsv_dev) return VSR_ERR_ARG;
    int64_t g = (npix + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(k_bgr2hsv_u8, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, bgr_dev, hsv_dev, npix);
    DONE();
}

int vsr_launch_absdiff_sums_u8x3(const uint8_t* hsv_dev, int npairs, int64_t pix_per_frame, uint64_t* sums_dev, void* stream)
{
    if (npairs <= 0) return 0;
    if (!hsv_dev || !sums_dev || pix_per_frame <= 0 || npairs > 65535) return VSR_ERR_ARG;
    if (hipMemsetAsync(sums_dev, 0, (size_t)npairs * 3 * sizeof(uint64_t), (hipStream_t)stream) != hipSuccess) return VSR_ERR_HIP;
    int64_t g = (pix_per_frame + 256 * 8 - 1) / (256 * 8);          // ~8 pixels per thread
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_absdiff_sums, dim3((unsigned)g, (unsigned)npairs), dim3(256), 0, (hipStream_t)stream, hsv_dev, pix_per_frame,
                       (unsigned long long*)sums_dev);
    DONE();
}

} // extern "C"
