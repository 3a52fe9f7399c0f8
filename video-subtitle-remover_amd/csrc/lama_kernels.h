// Launchers of the HBM-bound kernels of the LaMa path (lama_kernels.hip); one per OP_EW sub-kind of lama_plan.h.
// All pointers are device pointers; return 0 or -1 (launch error).
#pragma once
#include <stdint.h>

extern "C" {
int vsr_lama_launch_im2col7(const uint8_t* img, const uint8_t* mask, int B, int H, int W, int Hp, int Wp, float* cols, void* stream);
int vsr_lama_launch_halo(float* x, int n, int H, int W, int C, int halo, void* stream);
int vsr_lama_launch_add_halo(const float* a, const float* b, float* dst, int n, int H, int W, int C, int halo, int reflect, void* stream);
int vsr_lama_launch_out(const float* logits, const uint8_t* img, const uint8_t* mask, int B, int H, int W, int Hp, int Wp, uint8_t* out,
                        void* stream);
}
