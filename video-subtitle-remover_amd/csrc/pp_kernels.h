// Launchers of the ProPainter generator's elementwise kernels (pp_kernels.hip); device pointers; 0 or -1.
#pragma once
#include <stdint.h>

extern "C" {
int vsr_pp_launch_mask_f32(const uint8_t* src, int64_t n, float* dst, void* stream);
int vsr_pp_launch_imgprop(const float* prevProp, const float* prevMask, const float* cur, const float* mcur, const float* fprop,
                          const float* fcheck, int C, int h, int w, int first, float* prop, float* mprop, void* stream);
int vsr_pp_launch_mask_u8(const float* src, int64_t n, uint8_t* dst, void* stream);
}

// InpaintGenerator.forward (pp_gen_kernels.hip)
extern "C" {
int vsr_pp_launch_im2col3(const float* frames, const uint8_t* m1, const uint8_t* m2, int n, int H, int W, float* out, void* stream);
int vsr_pp_launch_ds_flow(const float* src, int n2, int H, int W, float* dst, void* stream);
int vsr_pp_launch_ds_mask(const uint8_t* m1, const uint8_t* m2, int n, int H, int W, float* slots, int halo, int C, void* stream);
int vsr_pp_launch_featprop_prep(const float* prop, const float* fprop, const float* fcheck, const float* maskSlot, int h, int w, int halo, int C,
                                float* warped, float* misc, void* stream);
int vsr_pp_launch_deform_cols(const float* src, const float* off, int ldOff, const float* flow, float maxMag, int h, int w, int halo, int C,
                              float* cols, void* stream);
int vsr_pp_launch_layernorm(const float* x, const float* gamma, const float* beta, int t, int fh, int fw, int C, int gh, int gw, float* y,
                            void* stream);
int vsr_pp_launch_pool(const float* y, const float* wgt, const float* bias, int t, int gh, int gw, int C, int ph, int pw, float* out, void* stream);
int vsr_pp_launch_fold(const float* vec, int ld, int t, int fh, int fw, int h, int w, int C, int halo, int normalize, float* out, void* stream);
int vsr_pp_launch_unfold_gelu(const float* map, int t, int fh, int fw, int h, int w, int C, int ld, float* out, void* stream);
int vsr_pp_launch_tanh_out(const float* y, int ld, int n, int H, int W, float* out, void* stream);
// the pair fold -> unfold + GELU with the GELU applied once per map element inside the fold (bit-identical token rows)
int vsr_pp_launch_fold_gelu(const float* vec, int ld, int t, int fh, int fw, int h, int w, int C, int halo, int normalize, float* out, void* stream);
int vsr_pp_launch_unfold_plain(const float* map, int t, int fh, int fw, int h, int w, int C, int ld, float* out, void* stream);
}
