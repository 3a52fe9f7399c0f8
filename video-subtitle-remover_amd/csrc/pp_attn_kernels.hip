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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same attention on fp16 operands with fp32 accumulation (v_mfma_f32_32x32x16_f16) -- the generator's arithmetic in the
// reference's GPU mode (propainter_inpaint.py:140-146: the module runs under .half()); softmax statistics, the running output and the
// normalisation stay fp32 (more than `.half()` keeps).  Same skeleton as the fp32 kernel; what changes is the operand geometry:
//   * S^T = K Q^T: 8 MFMAs of 16 d-values.  A = K tile in LDS as fp16 [32 keys][128 d] (256-byte rows, 16-byte slots XOR-swizzled by
//     key & 15: the 16 lanes of a ds_read_b128 group hit 16 different slots), B = the wave's Q rows as 8 x f16x8 registers;
//   * O^T += V^T P^T: 2 MFMAs of 16 keys per 32-d block.  The B operand of MFMA m must hold, in lane (query, hi), the probabilities of
//     the keys in its k-slots 8 hi .. 8 hi + 7 -- and the lane OWNS accumulator registers r = 8 m .. 8 m + 7 = keys 16 m + (i & 3) +
//     8 (i >> 2) + 4 hi, i = 0..7.  The key <-> k-slot assignment of a contraction is free as long as both operands agree, so P is
//     packed as it lies (eight v_cvt_pk) and the V tile is staged TRANSPOSED with its keys permuted to match: Vt[d][16 m + 8 hi + i]
//     = V[16 m + (i & 3) + 8 (i >> 2) + 4 hi][d].  A thread stages 4 consecutive keys x 4 d: the four keys are consecutive k-slots, so
//     it writes four 8-byte groups (one per d row; 8-byte slots XOR-swizzled by (d >> 2) & 7), and a fragment is two ds_read_b64.
// Range: probabilities are in [0, 1]; q / k / v beyond the fp16 range turn into inf and surface as a non-finite output row, which sets
// *rangeFlag -- the engine then redoes the call in exact fp32 (flow_engine.hip range guard).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2)
k_pp_flash_attn_f16(const PpAttnProblem* __restrict__ probs, int nprobs, unsigned int* __restrict__ rangeFlag)
{
    __shared__ __attribute__((aligned(16))) char Kh[BK * 256];            // fp16 [32 keys][128 d], swizzled 16-byte slots
    __shared__ __attribute__((aligned(16))) char Vt[D * 64];              // fp16 [128 d][32 permuted keys], swizzled 8-byte slots

    const int bid = blockIdx.x;
    int pi = 0;
    for (int lo = 0, hi_ = nprobs - 1; lo < hi_;) {
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
    const bool active = q0 < M;

    // ---- Q rows: lane (query, hi) holds d = 16 s + 8 hi .. + 7 of MFMA step s
    f16x8 qf[D / 16];
    {
        int qi = q0 + l31;
        if (qi > M - 1) qi = M - 1;
        const float* qp = P->Q + P->qrow[qi] + 8 * hi;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * s), b = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
            qf[s] = f16x8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
        }
    }
    f32x16 o[D / 32];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float mrun = NEG, lrun = 0.f;

    // ---- staging.  K: thread -> key tid / 16 + 16 i (i = 0, 1), d = 8 (tid % 16) .. + 7 (two float4 -> one 16-byte slot).
    //                V: thread -> keys 4 (tid / 32) .. + 3, d = 4 (tid % 32) .. + 3 (four float4 -> four 8-byte groups, one per d)
    const int k_r = tid >> 4, k_s = tid & 15;
    const int v_g = tid >> 5, v_c = (tid & 31) * 4;
    f32x4 kr[4], vr[4];
    const float* Kb = P->K;
    const float* Vb = P->V;
    const int32_t* krow = P->krow;
    auto prefetch = [&](int it) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = it * BK + k_r + 16 * i;
            if (key > nk - 1) key = nk - 1;
            const float* p = Kb + krow[key] + 8 * k_s;
            kr[2 * i] = *reinterpret_cast<const f32x4*>(p);
            kr[2 * i + 1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = it * BK + 4 * v_g + i;
            if (key > nk - 1) key = nk - 1;
            vr[i] = *reinterpret_cast<const f32x4*>(Vb + krow[key] + v_c);
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k_r + 16 * i;
            const f32x4 a = kr[2 * i], b = kr[2 * i + 1];
            const f16x8 h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
            *reinterpret_cast<f16x8*>(Kh + key * 256 + ((k_s ^ (key & 15)) << 4)) = h;
        }
        // keys 4 g .. 4 g + 3 of the tile: m = g >> 2, hi' = g & 1, i = 4 ((g >> 1) & 1) .. + 3 -> k-slots 16 m + 8 hi' + i: 8-byte group 4 m + 2 hi' + ((g >> 1) & 1)
        const int grp = 4 * (v_g >> 2) + 2 * (v_g & 1) + ((v_g >> 1) & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = v_c + j;
            const f16x4 h = {(_Float16)vr[0][j], (_Float16)vr[1][j], (_Float16)vr[2][j], (_Float16)vr[3][j]};
            *reinterpret_cast<f16x4*>(Vt + d * 64 + ((grp ^ ((d >> 2) & 7)) << 3)) = h;
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
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 16; ++st) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(Kh + l31 * 256 + (((2 * st + hi) ^ (l31 & 15)) << 4));
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], s, 0, 0, 0);
            }
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
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f16x8 pf = {(_Float16)s[8 * m], (_Float16)s[8 * m + 1], (_Float16)s[8 * m + 2], (_Float16)s[8 * m + 3],
                                  (_Float16)s[8 * m + 4], (_Float16)s[8 * m + 5], (_Float16)s[8 * m + 6], (_Float16)s[8 * m + 7]};
#pragma unroll
                for (int b = 0; b < D / 32; ++b) {
                    const int d = 32 * b + l31;
                    const char* row = Vt + d * 64;
                    const int sw = (d >> 2) & 7, g0 = 4 * m + 2 * hi;
                    const f16x4 lo4 = *reinterpret_cast<const f16x4*>(row + ((g0 ^ sw) << 3));
                    const f16x4 hi4 = *reinterpret_cast<const f16x4*>(row + (((g0 + 1) ^ sw) << 3));
                    const f16x8 vf = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) stash();
        __syncthreads();
    }

    if (active) {
        const float ltot = lrun + __shfl_xor(lrun, 32, 64);
        bool bad = false;
        if (q0 + l31 < M) {
            const float inv = 1.0f / ltot;
            float* op = P->O + P->orow[q0 + l31] + 4 * hi;
#pragma unroll
            for (int b = 0; b < D / 32; ++b)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = {o[b][4 * q4] * inv, o[b][4 * q4 + 1] * inv, o[b][4 * q4 + 2] * inv, o[b][4 * q4 + 3] * inv};
                    bad |= !(__builtin_fabsf(v[0]) <= 3.0e38f) || !(__builtin_fabsf(v[1]) <= 3.0e38f) || !(__builtin_fabsf(v[2]) <= 3.0e38f) ||
                           !(__builtin_fabsf(v[3]) <= 3.0e38f);
                    *reinterpret_cast<f32x4*>(op + 32 * b + 8 * q4) = v;
                }
        }
        if (rangeFlag != nullptr && __any(bad) && lane == 0) atomicOr(rangeFlag, 1u);
    }
}

// f16 != 0: fp16 operands / fp32 accumulation (rangeFlag: device word OR-ed with 1 on a non-finite output, nullable)
extern "C" int vsr_pp_launch_flash_attn(const PpAttnProblem* d_probs, int nprobs, int totalTiles, int f16, unsigned int* rangeFlag, void* stream)
{
    if (nprobs <= 0 || totalTiles <= 0) return 0;
    if (f16) hipLaunchKernelGGL(k_pp_flash_attn_f16, dim3(totalTiles), dim3(256), 0, (hipStream_t)stream, d_probs, nprobs, rangeFlag);
    else hipLaunchKernelGGL(k_pp_flash_attn_f32, dim3(totalTiles), dim3(256), 0, (hipStream_t)stream, d_probs, nprobs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
