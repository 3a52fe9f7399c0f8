// Internal view of the grouped gather-GEMM: the descriptor (GGProblem) and its enums live in
// the public C-ABI header; this adds the device-descriptor launcher used by the engine.
//
// Every dense contraction on the STTN hot path (reference backend/inpaint/sttn/auto_sttn.py:
// 75-95 encoder/decoder convs, 172-174 QKV 1x1, 140-145 QK^T and PV, 162-164 output_linear,
// 214-218 FFN convs) is one GGProblem: row / 32-element-chunk offset tables turn im2col,
// zero padding (physical halos), patch (un)folding, window frame gathers and channel slicing
// into pure addressing -- no tensor on this path is permuted or copied to be multiplied.
#pragma once
#include <stdint.h>
#include "../../include/vsr_hip.h"

// d_probs: DEVICE array (tileStart filled); totalBlocks = sum tilesM*tilesN*splitK.
// variant 1: one workgroup per tile; 2 / 3: persistent kernels pulling tile ids from queue[0..7] (must be 0);
// nQueues = 8: per-XCD tile ranges with stealing, 1: one global queue (v3 only).
// variant 4 = split-half f16-MFMA kernels; rangeFlag (device, nullable) is OR-ed with 1 on a non-finite result.
extern "C" int vsr_launch_gather_gemm_dev(const GGProblem* d_probs, int nprobs, int totalBlocks, int tileCfg,
                                          int bmode, unsigned int* queue, int variant, int nQueues,
                                          unsigned int* rangeFlag, void* stream);

// the 256 x 256 fp16-operand kernel (tile config VSR_TILE_256x256, variant 6, NK; gather_gemm_v7.h): cuts one problem into a body of
// whole rounds of 256-row tiles and a remainder of short tiles (out: room for 2 problems; returns how many); CUs of the device
extern "C" int vsr_v7_split(const GGProblem* p, int cus, GGProblem* out);
extern "C" int vsr_gg_cus(void);

// at most four output columns: the dot-product kernel (gather_gemm_narrow.h).  Problems laid out for the 256 x 32 tile (tilesN = 1),
// splitK = 1, NK, no residual, N * K <= vsr_gg_narrow_cap(); maxN / maxK = the largest N / K among them
extern "C" int vsr_gg_narrow_cap(void);
extern "C" int vsr_launch_gather_gemm_narrow_dev(const GGProblem* d_probs, int nprobs, int totalBlocks, int maxN, int maxK, void* stream);
