// Launchers of the ProPainter generator's elementwise kernels (pp_kernels.hip); device pointers; 0 or -1.
#pragma once
#include <stdint.h>

extern "C" {
int vsr_pp_launch_mask_f32(const uint8_t* src, int64_t n, float* dst, void* stream);
int vsr_pp_launch_imgprop(const float* prevProp, const float* prevMask, const float* cur, const float* mcur, const float* fprop,
                          const float* fcheck, int C, int h, int w, int first, float* prop, float* mprop, void* stream);
int vsr_pp_launch_mask_u8(const float* src, int64_t n, uint8_t* dst, void* stream);
}
