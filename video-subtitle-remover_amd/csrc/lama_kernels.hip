// HBM-bound kernels of the LaMa path (plan ops of kind OP_EW, lama_plan.h) -- SURVEY.md 8(a) row a12.
// Reference arithmetic: backend/inpaint/utils/lama_util.py (get_image :12-29, pad_img_to_modulo :52-60, prepare_img_and_mask
// :63-80), backend/inpaint/lama_inpaint.py (_inpaint_batch :45-58: clip(x * 255, 0, 255).astype(uint8)), and the exported
// module's forward (masked = image * (1 - mask); out = mask * pred + (1 - mask) * image).  fp32 expressions are written in the
// reference's op order under "fp contract(off)" so that truncation to uint8 falls on the same side.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lama_kernels.h"

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

// np.pad(..., mode='symmetric') at the bottom / right (lama_util.py:56-60): index n + i -> n - 1 - i
__device__ __forceinline__ int sym_index(int i, int n) { return i < n ? i : 2 * n - 1 - i; }
// ReflectionPad2d / padding_mode='reflect': -i -> i, (n - 1) + i -> (n - 1) - i
__device__ __forceinline__ int reflect_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// One thread per (padded pixel, tap): 4 floats (masked B, G, R as given; mask) at cols[pixel][tap * 4 .. +3]; K 196 -> 224, the
// 28 pad columns are written as zeros by the tap-0 thread's neighbours (taps 49..55).
// get_image: x.astype(float32) / 255; mask: (m / 255 > 0) * 1; masked = image * (1 - mask)  (int mask promoted to float)
__global__ __launch_bounds__(256) void k_lama_im2col7(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, int B, int H, int W,
                                                      int Hp, int Wp, float* __restrict__ cols)
{
    const int64_t total = (int64_t)B * Hp * Wp * 56;
    GRID_STRIDE(i, total) {
        const int tap = (int)(i % 56);
        const int64_t pixel = i / 56;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tap < 49) {
            const int X = (int)(pixel % Wp), Y = (int)((pixel / Wp) % Hp), b = (int)(pixel / ((int64_t)Wp * Hp));
            const int py = reflect_index(Y + tap / 7 - 3, Hp), px = reflect_index(X + tap % 7 - 3, Wp);
            const int sy = sym_index(py, H), sx = sym_index(px, W);
            const int64_t at = ((int64_t)b * H + sy) * W + sx;
            const float m = mask[at] > 0 ? 1.f : 0.f;
            const float keep = 1.f - m;
            v[0] = ((float)img[at * 3 + 0] / 255.f) * keep;
            v[1] = ((float)img[at * 3 + 1] / 255.f) * keep;
            v[2] = ((float)img[at * 3 + 2] / 255.f) * keep;
            v[3] = m;
        }
        *(f32x4*)(cols + pixel * 224 + tap * 4) = v;
    }
}

// reflect halo of an NHWC activation, in place: one thread per (border pixel, 4 channels)
__global__ __launch_bounds__(256) void k_lama_halo(float* __restrict__ x, int n, int H, int W, int C, int halo)
{
    const int Hp = H + 2 * halo, Wp = W + 2 * halo, C4 = C / 4;
    const int border = Hp * Wp - H * W;
    const int64_t total = (int64_t)n * border * C4;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        int e = (int)(r % border);
        const int f = (int)(r / border);
        // enumerate the border: `halo` full rows on top, `halo` at the bottom, then the side columns of the interior rows
        int Y, X;
        if (e < 2 * halo * Wp) {
            Y = e / Wp;
            X = e % Wp;
            if (Y >= halo) Y += H;
        } else {
            e -= 2 * halo * Wp;
            Y = halo + e / (2 * halo);
            X = e % (2 * halo);
            if (X >= halo) X += W;
        }
        const int sy = reflect_index(Y - halo, H) + halo, sx = reflect_index(X - halo, W) + halo;
        float* base = x + (int64_t)f * Hp * Wp * C;
        *(f32x4*)(base + ((int64_t)Y * Wp + X) * C + c4 * 4) = *(const f32x4*)(base + ((int64_t)sy * Wp + sx) * C + c4 * 4);
    }
}

// dst = a + b (FFCResnetBlock.forward: id + x, both channel groups) over the padded frame: halo pixels take the reflected
// interior sum when `reflect`, and are left alone (zero halo for the transposed convs) otherwise
__global__ __launch_bounds__(256) void k_lama_add_halo(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, int n,
                                                       int H, int W, int C, int halo, int reflect)
{
    const int Hp = H + 2 * halo, Wp = W + 2 * halo, C4 = C / 4;
    const int64_t total = (int64_t)n * Hp * Wp * C4;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int X = (int)(r % Wp);
        r /= Wp;
        const int Y = (int)(r % Hp);
        const int f = (int)(r / Hp);
        const bool inside = Y >= halo && Y < halo + H && X >= halo && X < halo + W;
        if (!inside && !reflect) continue;
        const int sy = reflect_index(Y - halo, H) + halo, sx = reflect_index(X - halo, W) + halo;
        const int64_t src = (((int64_t)f * Hp + sy) * Wp + sx) * C + c4 * 4, at = (((int64_t)f * Hp + Y) * Wp + X) * C + c4 * 4;
        const f32x4 va = *(const f32x4*)(a + src), vb = *(const f32x4*)(b + src);
        *(f32x4*)(dst + at) = va + vb;
    }
}

// torch.sigmoid, out = mask * pred + (1 - mask) * image, np.clip(out * 255, 0, 255).astype('uint8'), crop to H x W
__global__ __launch_bounds__(256) void k_lama_out(const float* __restrict__ logits, const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                  int B, int H, int W, int Hp, int Wp, uint8_t* __restrict__ out)
{
    const int64_t total = (int64_t)B * H * W;
    GRID_STRIDE(i, total) {
        const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
        const float m = mask[i] > 0 ? 1.f : 0.f;
        const float keep = 1.f - m;
        // logits of the 7x7 conv: [image][block row][block col][(dy, dx, c) of the 4 x 4 block, padded to 64]
        const float* lg = logits + ((((int64_t)b * (Hp / 4) + (Y >> 2)) * (Wp / 4) + (X >> 2)) * 64) + (((Y & 3) * 4 + (X & 3)) * 3);
        for (int c = 0; c < 3; ++c) {
            const float p = 1.f / (1.f + expf(-lg[c]));
            const float im = (float)img[i * 3 + c] / 255.f;
            float v = (m * p + keep * im) * 255.f;
            v = fminf(fmaxf(v, 0.f), 255.f);
            out[i * 3 + c] = (uint8_t)v;
        }
    }
}

extern "C" {

int vsr_lama_launch_im2col7(const uint8_t* img, const uint8_t* mask, int B, int H, int W, int Hp, int Wp, float* cols, void* stream)
{
    LAUNCH(k_lama_im2col7, (int64_t)B * Hp * Wp * 56, img, mask, B, H, W, Hp, Wp, cols);
}

int vsr_lama_launch_halo(float* x, int n, int H, int W, int C, int halo, void* stream)
{
    if (halo <= 0) return 0;
    const int64_t border = (int64_t)(H + 2 * halo) * (W + 2 * halo) - (int64_t)H * W;
    LAUNCH(k_lama_halo, (int64_t)n * border * (C / 4), x, n, H, W, C, halo);
}

int vsr_lama_launch_add_halo(const float* a, const float* b, float* dst, int n, int H, int W, int C, int halo, int reflect, void* stream)
{
    LAUNCH(k_lama_add_halo, (int64_t)n * (H + 2 * halo) * (W + 2 * halo) * (C / 4), a, b, dst, n, H, W, C, halo, reflect);
}

int vsr_lama_launch_out(const float* logits, const uint8_t* img, const uint8_t* mask, int B, int H, int W, int Hp, int Wp, uint8_t* out,
                        void* stream)
{
    LAUNCH(k_lama_out, (int64_t)B * H * W, logits, img, mask, B, H, W, Hp, Wp, out);
}

} // extern "C"
