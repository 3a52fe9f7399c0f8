// gather_gemm_f32_narrow: gather-GEMM problems with at most four output columns -- the convs that end a head (RAFT's flow head
// conv2, 256 -> 2, update.py:6-12; the flow-completion net's upsample.2, 32 -> 2, recurrent_flow_completion.py:271-276; the
// generator's last decoder conv, 64 -> 3, propainter.py:268-276).
//
// Why (profiles/r06_propainter_f32_*_kernel_stats.csv): on the 256 x 32 tile of gather_gemm_f32_v3 such a problem multiplies
// 32 columns to keep 2 or 3 -- 72 chunks x 32 MFMAs per wave and tile for RAFT's flow head, 184 us of matrix pipe per CU
// against 5 us of useful arithmetic -- and its operand rows go through the LDS for nothing: 105 launches and 98 ms of a 68-frame
// ProPainter batch at 6-8 TFLOP/s.  These problems are reads of the A operand and nothing else, so this kernel is a dot product:
//   * a workgroup owns 256 rows (the tile bookkeeping of the 256 x 32 tile: same tables, same tile ids); the N x K weights sit in
//     LDS (<= 40 KB, sized per launch: a small K leaves room for more workgroups per CU), staged once per workgroup;
//   * 8 lanes share a row -- a lane holds one float4 of the row's 128-byte chunk -- and own 8 rows each, 32 apart, so a wave's load
//     instruction covers 8 NEIGHBOURING rows (8 x 128 contiguous bytes for a 32-channel map) and a lane has 8 independent 16-byte
//     loads in flight per chunk;
//   * fp32 FMAs in the K order of the tables, one partial sum per lane and column, a 3-step butterfly over the row's 8 lanes at
//     the end; bias / alpha / activation as in the MFMA kernels' epilogues.  Exact fp32 in every arithmetic mode of the engines.
// The 3 x 3 taps of a row overlap its neighbours': with the channel-major K order consecutive chunks are the nine taps of one
// channel block, so the x-taps hit in L1 and the y-taps in the XCD's L2 (tile ids are dealt to XCDs in contiguous runs as in
// gather_gemm_f32).
#pragma once

#define GG_NARROW_WCAP 10240      // floats of LDS for the weights: N x K <= this (flow_engine.hip checks before it picks the kernel)

template <int NN>
__global__ void __launch_bounds__(256)
gather_gemm_f32_narrow(const GGProblem* __restrict__ probs, int nprobs)
{
    constexpr int BM = 256;
    extern __shared__ __attribute__((aligned(16))) float wsm[];      // NN x (the launch's largest K) floats

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 3, q = lane & 7;          // row group of the wave, float4 slot of the chunk

    const int bid = blockIdx.x;
    int pi = 0;
    for (int lo_ = 0, hi_ = nprobs - 1; lo_ < hi_;) {
        const int mid_ = (lo_ + hi_ + 1) >> 1;
        if (bid >= probs[mid_].tileStart) lo_ = mid_; else hi_ = mid_ - 1;
        pi = lo_;
    }
    const GGProblem* __restrict__ P = probs + pi;
    const int M = P->M, N = P->N, K = P->K;
    const int nblk = P->tilesM;                      // tilesN = 1, splitK = 1 (checked by the host)
    int tm = bid - P->tileStart;
    {   // workgroups with equal (id & 7) share an XCD: each XCD gets a contiguous run of tiles
        const int xcd = tm & 7, qq = nblk >> 3, r = nblk & 7;
        tm = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (tm >> 3);
    }
    const int nchunks = K / VSR_GG_KC;

    const gcf32 A = (gcf32)P->A;
    const gcf32 B = (gcf32)P->B;
    const gci32 rowA = (gci32)P->rowA;
    const cci32 colA = (cci32)P->colA;
    const gci32 rowB = (gci32)P->rowB;
    const gci32 colB = (gci32)P->colB;

    // weights -> LDS, [NN][K] (rows beyond N: zeros)
    for (int i = tid; i < NN * (K >> 2); i += 256) {
        const int n = i / (K >> 2), k = (i - n * (K >> 2)) << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < N) v = *reinterpret_cast<gcf32x4>(B + (rowB[n] + colB[k >> 5] + (k & 31)));
        *reinterpret_cast<f32x4*>(wsm + n * K + k) = v;
    }

    int rows[8], aoff[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        int m = tm * BM + p * 32 + wave * 8 + g;
        rows[p] = m;
        if (m > M - 1) m = M - 1;
        aoff[p] = rowA[m] + 4 * q;
    }
    float acc[8][NN];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int n = 0; n < NN; ++n) acc[p][n] = 0.f;
    __syncthreads();

    int ca = colA[0];
    for (int c = 0; c < nchunks; ++c) {
        const int caNext = colA[c + 1 < nchunks ? c + 1 : c];      // (scalar load, a chunk ahead)
        f32x4 a[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) a[p] = *reinterpret_cast<gcf32x4>(A + (aoff[p] + ca));
        f32x4 w[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) w[n] = *reinterpret_cast<const f32x4*>(wsm + n * K + c * VSR_GG_KC + 4 * q);
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                float s = acc[p][n];
                s = __builtin_fmaf(a[p][0], w[n][0], s);
                s = __builtin_fmaf(a[p][1], w[n][1], s);
                s = __builtin_fmaf(a[p][2], w[n][2], s);
                s = __builtin_fmaf(a[p][3], w[n][3], s);
                acc[p][n] = s;
            }
        ca = caNext;
    }
    // the row's 8 lanes: butterfly, every lane ends with the whole sum
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            float s = acc[p][n];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            acc[p][n] = s;
        }
    // lane q of a row stores column q
    if (q < N) {
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const gcf32 bias = (gcf32)P->bias;
        const float bq = bias != nullptr ? bias[q] : 0.f;
        const gci32 rowC = (gci32)P->rowC;
        const int cc = ((cci32)P->colC)[0] + q;
        const gf32 C = (gf32)P->C;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float s = acc[p][0];
#pragma unroll
            for (int n = 1; n < NN; ++n) s = q == n ? acc[p][n] : s;
            float v = s * alpha + bq;
            if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
            else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
            if (rows[p] < M) C[rowC[rows[p]] + cc] = v;
        }
    }
}
