// Fused window attention of the ProPainter generator: softmax(Q K^T / sqrt(D)) V of SparseWindowAttention.forward (reference
// backend/inpaint/video/model/modules/sparse_transformer.py:238-262) for every (window, head) problem of a transformer block in ONE
// launch, with no score or probability matrix in memory.
//
// Why it can be fused here and not in the STTN attention: the head dimension is D = 128 (512 channels / 4 heads), so a
// tile of 32 queries keeps its whole output row block (32 x 128) in registers; in STTN a "token" is a patch of 960 .. 76 800 values.
// The three-op form this replaces (QK^T into PG_S, k_softmax_rows into PG_P, P.V) moved 2 x 11 MB per (masked window, head) through
// HBM three times and spent a third of the attention's time in the softmax pass (profiles/r05_propainter_ops_by_time.log).
//
// Arithmetic: exact fp32 (v_mfma_f32_32x32x2_f32 = an fmaf chain per output), fp32 online softmax (running row maximum and row
// sum, rescaled accumulators: the flash-attention recurrence), exp2 with log2(e) / sqrt(D) folded into the score scale.
//
// Structure -- everything is laid out so that the probabilities never leave the registers they were computed in:
//   * a wave owns 32 queries; a workgroup is 4 waves = 128 queries of one problem and walks that problem's keys 32 at a time;
//   * scores are computed TRANSPOSED, S^T[key][query] = K Q^T: MFMA A operand = K rows from LDS (lane l31 = key, ds_read_b128 of
//     four d values), B operand = the wave's Q rows, held in 64 registers for the whole key loop (lane l31 = query).  The
//     accumulator then gives lane (query = l31, hi) the 16 keys (r & 3) + 8 (r >> 2) + 4 hi of the tile: all softmax statistics of a
//     query live in ONE lane pair (l31, l31 + 32) -- a per-lane max / sum over 16 registers and one cross-half exchange;
//   * O^T[d][query] += V^T P^T: the B operand of MFMA step r is exactly accumulator register r of S^T after the exponential (lane
//     (query, hi) must supply P[query][key(r, hi)] -- what it holds); the A operand V[key(r, hi)][d] is a ds_read_b32 from the V tile.
//     The output accumulators (4 d-blocks x 16 registers) have the query in the lane again, so the rescale by exp2(m_old - m_new)
//     is a per-lane scalar multiply, skipped (wave-uniform test) when no row maximum of the wave moved;
//   * K / V tiles (32 rows x 128 floats each) are gathered by row tables -- keys of a masked window are its own tokens, the rolled
//     windows' tokens and the pooled tokens of every second frame (pp_plan.cpp attention()) -- through a register prefetch of the
//     next tile while the current one is contracted; 34 KB of LDS and <= 256 VGPRs: two workgroups per CU, so one workgroup's
//     softmax VALU work runs under the other's MFMAs.
// Per 32-key tile a wave issues 64 + 64 MFMAs (8 192 matrix-pipe cycles) against ~150 VALU and 80 LDS instructions: MFMA-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pp_attn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 128;          // head dimension
constexpr int LD = D + 4;       // LDS row stride in floats: row r starts at bank 4 r -- conflict-free b128 reads across 16 rows, b32 reads across d
constexpr int BQ = 128, BK = 32;
constexpr float NEG = -1.0e30f; // "minus infinity" that survives subtraction

__global__ void __launch_bounds__(256, 2)
k_pp_flash_attn_f32(const PpAttnProblem* __restrict__ probs, int nprobs)
{
    __shared__ __attribute__((aligned(16))) float Ks[BK * LD];
    __shared__ __attribute__((aligned(16))) float Vs[BK * LD];

    const int bid = blockIdx.x;
    int pi = 0;
    for (int lo = 0, hi_ = nprobs - 1; lo < hi_;) {       // last problem whose first tile id is <= bid
        const int mid = (lo + hi_ + 1) >> 1;
        if (bid >= probs[mid].tileStart) lo = mid; else hi_ = mid - 1;
        pi = lo;
    }
    const PpAttnProblem* __restrict__ P = probs + pi;
    const int M = P->M, nk = P->nk;
    const float scale = P->scale;
    const int tq = bid - P->tileStart;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int q0 = tq * BQ + wave * 32;
    const bool active = q0 < M;                            // wave-uniform: a wave beyond the problem's queries only helps staging

    // ---- the wave's Q rows as the B operand of every score MFMA: lane (query l31, hi) holds d = 8 g + 4 hi + j
    f32x4 qf[D / 8];
    {
        int qi = q0 + l31;
        if (qi > M - 1) qi = M - 1;
        const float* qp = P->Q + P->qrow[qi] + 4 * hi;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qp + 8 * g);
    }
    f32x16 o[D / 32];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float mrun = NEG, lrun = 0.f;

    // ---- staging: thread -> rows tid / 32 + 8 i (i = 0..3), float4 column tid % 32 of the K tile and of the V tile
    const int s_r = tid >> 5, s_c = (tid & 31) * 4;
    f32x4 kr[4], vr[4];
    const float* Kb = P->K;
    const float* Vb = P->V;
    const int32_t* krow = P->krow;
    auto prefetch = [&](int it) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = it * BK + s_r + 8 * i;
            if (key > nk - 1) key = nk - 1;                 // rows beyond the key set are fetched (valid memory) and masked out of the softmax
            const int ro = krow[key];
            kr[i] = *reinterpret_cast<const f32x4*>(Kb + ro + s_c);
            vr[i] = *reinterpret_cast<const f32x4*>(Vb + ro + s_c);
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&Ks[(s_r + 8 * i) * LD + s_c]) = kr[i];
            *reinterpret_cast<f32x4*>(&Vs[(s_r + 8 * i) * LD + s_c]) = vr[i];
        }
    };

    const int ntiles = (nk + BK - 1) / BK;
    prefetch(0);
    stash();
    __syncthreads();
    for (int it = 0; it < ntiles; ++it) {
        const bool more = it + 1 < ntiles;
        if (more) prefetch(it + 1);
        if (active) {
            // ---- S^T = K Q^T for the 32 keys of this tile
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int g = 0; g < D / 8; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[l31 * LD + 8 * g + 4 * hi]);
#pragma unroll
                for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[g][j], s, 0, 0, 0);
            }
            // ---- online softmax: this lane's 16 keys, its partner half's 16
            const int kbase = it * BK + 4 * hi;
            float mloc = NEG;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                s[r] = key < nk ? s[r] * scale : NEG;
                mloc = fmaxf(mloc, s[r]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float mnew = fmaxf(mrun, mloc);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
                lsum += s[r];
            }
            lrun = lrun * alpha + lsum;
            mrun = mnew;
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int b = 0; b < D / 32; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            }
            // ---- O^T += V^T P^T: step r contracts the key pair {key(r, 0), key(r, 1)}
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vrow = &Vs[((r & 3) + 8 * (r >> 2) + 4 * hi) * LD + l31];
#pragma unroll
                for (int b = 0; b < D / 32; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * b], s[r], o[b], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) stash();
        __syncthreads();
    }

    // ---- normalise and store: lane (query, hi) holds d = 32 b + (r & 3) + 8 (r >> 2) + 4 hi -> float4 runs of its output row
    if (active) {
        const float ltot = lrun + __shfl_xor(lrun, 32, 64);
        if (q0 + l31 < M) {
            const float inv = 1.0f / ltot;
            float* op = P->O + P->orow[q0 + l31] + 4 * hi;
#pragma unroll
            for (int b = 0; b < D / 32; ++b)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {o[b][4 * q4] * inv, o[b][4 * q4 + 1] * inv, o[b][4 * q4 + 2] * inv, o[b][4 * q4 + 3] * inv};
                    *reinterpret_cast<f32x4*>(op + 32 * b + 8 * q4) = v;
                }
        }
    }
}

extern "C" int vsr_pp_launch_flash_attn(const PpAttnProblem* d_probs, int nprobs, int totalTiles, void* stream)
{
    if (nprobs <= 0 || totalTiles <= 0) return 0;
    hipLaunchKernelGGL(k_pp_flash_attn_f32, dim3(totalTiles), dim3(256), 0, (hipStream_t)stream, d_probs, nprobs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
