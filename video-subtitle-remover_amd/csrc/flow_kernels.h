// Launchers of the RAFT elementwise / gather kernels (flow_kernels.hip); one per OP_EW sub-kind of raft_plan.h.
// All pointers are device pointers; return 0 or -1 (launch error).
#pragma once
#include <stdint.h>

extern "C" {
int vsr_raft_launch_im2col7_u8(const uint8_t* img, int n, int H, int W, int bgr, float* out, void* stream);
// acc: n*C*2 doubles of scratch (zeroed here)
int vsr_raft_launch_inorm_stats(const float* x, int n, int H, int W, int C, int halo, double* acc, float* stats, void* stream);
int vsr_raft_launch_inorm_apply(float* x, int n, int H, int W, int C, int halo, const float* stats, int relu, const float* res,
                                int resHalo, void* stream);
int vsr_raft_launch_ctx_split(const float* cmap, const int32_t* frameOf, int pairs, int h, int w, int halo, int Chx, float* hxr,
                              void* stream);
int vsr_raft_launch_flow_update(const float* delta, int ldDelta, float* coords, float* flow, float* hxr, int pairs, int h, int w,
                                int init, int halo, int Chx, int chFlow, void* stream);
int vsr_raft_launch_im2col7_flow(const float* flow, int pairs, int h, int w, float* out, void* stream);
int vsr_raft_launch_avgpool2(const float* src, int64_t rows, int hs, int ws, float* dst, void* stream);
// dst[p] = src[p]^T, n square planes of hw x hw floats
int vsr_raft_launch_corr_transpose(const float* src, float* dst, int n, int hw, void* stream);
int vsr_raft_launch_corr_lookup(const float* const* levels, const int* lvlH, const int* lvlW, const float* coords, int64_t M, int ld,
                                float* out, void* stream);
int vsr_raft_launch_gru_rh(const float* zr, float* hxr, int pairs, int h, int w, int halo, int Chx, int chH, int chRH, void* stream);
int vsr_raft_launch_gru_update(const float* zr, const float* q, float* hxr, int pairs, int h, int w, int halo, int Chx, int chH,
                               void* stream);
int vsr_raft_launch_convex_up(const float* flow, const float* mask, int pairs, int h, int w, float* out, void* stream);
}

// recurrent flow completion (rfc_plan.h)
extern "C" {
int vsr_rfc_launch_im2col5(const float* ff, const float* fb, const uint8_t* mask, int t, int H, int W, float* out, void* stream);
int vsr_rfc_launch_deform_cols(const float* srcA, const float* srcB, const float* off, int ldOff, float maxMag, int n, int h, int w,
                               int halo, int C, float* cols, void* stream);
int vsr_rfc_launch_combine(const float* pred, int ld, const float* ff, const float* fb, const uint8_t* mask, int t, int H, int W,
                           float* outF, float* outB, void* stream);
}
