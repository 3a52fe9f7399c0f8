// Internal view of the HBM-bound kernels (elementwise.hip); SMProblem lives in the public header.
#pragma once
#include <stdint.h>
#include "../../include/vsr_hip.h"

// d_probs: DEVICE array (rowStart filled, multiples of 4); totalRows = padded row count
extern "C" int vsr_launch_softmax_dev(const SMProblem* d_probs, int nprobs, int totalRows, void* stream);

// precision mode 2: the same kernels writing (and, for the upsampler, reading) split-format tensors
// (gather_gemm_v5.h); *Split = 0 is the plain fp32 layout of the public entry points
extern "C" int vsr_launch_norm_im2col_fmt(const uint8_t* img, int ih, int iw, int nframes, float* out, int premask,
                                          const uint8_t* mask, int outSplit, void* stream);
extern "C" int vsr_launch_reduce_scatter_fmt(const float* part, int nsplit, int64_t splitStride, int M, int N,
                                             const int32_t* rowC, const int32_t* colC, float* out, int outSplit,
                                             const float* lsum, int ldL, void* stream);   // lsum [nsplit][ldL]: divide row m by their sum (fused attention), or NULL
// y rows are 2x4 pixel blocks of a blkW-wide image with columns (dy, dx, channel) -- the blocked 64 -> 3 conv (sttn_plan.cpp); 0 = vsr_launch_decode_out
extern "C" int vsr_launch_decode_out_blk(const float* y, int ldy, int pix, int nframes, const int32_t* frameIdx,
                                         const int32_t* first, float* comp, const uint8_t* inBGR, const uint8_t* mask,
                                         int blkW, void* stream);
extern "C" int vsr_launch_upsample2x_fmt(const float* src, int H, int W, int C, int haloS, float* dst, int haloD,
                                         int nframes, int split, void* stream);
// KN operand (B(k, n) = B[rowB[k] + colB[n / 32] + n % 32], split format) -> NK operand dst[n * ld + k] in split format; K, N multiples of 32
extern "C" int vsr_launch_kn_to_nk_split(const float* B, const int32_t* rowB, const int32_t* colB, int K, int N, int64_t ld, float* dst, void* stream);
// the same with a range: output rows [oyLo, oyHi) of the upsampled image / pixels [pLo, pLo + pCnt) of every decoded frame (whole image
// rows, whole 2-row blocks in the blocked form) -- the decoder of a plan that was given the rows its caller will read (Plan::decLo)
extern "C" int vsr_launch_upsample2x_rows(const float* src, int H, int W, int C, int haloS, float* dst, int haloD,
                                          int nframes, int split, int oyLo, int oyHi, void* stream);
extern "C" int vsr_launch_decode_out_rows(const float* y, int ldy, int pix, int nframes, const int32_t* frameIdx,
                                          const int32_t* first, float* comp, const uint8_t* inBGR, const uint8_t* mask,
                                          int blkW, int pLo, int pCnt, void* stream);
