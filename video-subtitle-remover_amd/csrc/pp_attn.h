// Fused window attention of the ProPainter generator (pp_attn_kernels.hip): one problem per (window, head) of a transformer block.
#pragma once
#include <stdint.h>

struct PpAttnProblem {
    const float* Q;          // query rows: Q + qrow[m] + d        (the head's 128 columns start at Q)
    const float* K;          // key rows:   K + krow[k] + d
    const float* V;          // value rows: V + krow[k] + d        (q, k, v of a token lie in one row of the fused QKV tensor)
    float* O;                // output:     O + orow[m] + d
    const int32_t* qrow;     // M entries (element offsets)
    const int32_t* krow;     // nk entries
    const int32_t* orow;     // M entries
    int M, nk;               // queries, keys
    int tileStart;           // first workgroup (128-query tile) of this problem in the launch
    float scale;             // log2(e) / sqrt(D)
};

// f16 = 0: exact fp32 (v_mfma_f32_32x32x2_f32); 1: fp16 operands, fp32 accumulation and softmax (rangeFlag OR-ed with 1 on a non-finite output)
extern "C" int vsr_pp_launch_flash_attn(const PpAttnProblem* d_probs, int nprobs, int totalTiles, int f16, unsigned int* rangeFlag, void* stream);
