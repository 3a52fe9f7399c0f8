// gather_gemm_f32_v2: persistent, dynamically scheduled variant of the grouped gather-GEMM
// (same GGProblem semantics and MFMA fragment mapping as gather_gemm_f32 in gather_gemm.hip).
//
// Why (measured on MI355X with scripts/gg_ablate.hip, conv-shaped 72000 x 256 x 2304):
//   * the hardware dispatcher packs the LAST partial round of workgroups greedily onto few CUs,
//     so a launch with 1.47 rounds of workgroups costs two full rounds: the MFMA-only ablation
//     of v1 already stops at 103-127 TF of the 157 TF peak.  v2 launches exactly one resident
//     round (CUs x occupancy workgroups) and every workgroup pulls tiles from an atomic queue
//     (largest problems first), so all CUs drain together and the tail is < 1 tile.
//   * v1's single LDS buffer needs two barriers per 32-deep chunk and leaves ~1.5k cycles per
//     chunk between a wave's MFMA runs; with one wave per workgroup per SIMD the four SIMDs of
//     a CU serve the co-resident workgroups in different orders and the barriers convoy them
//     (84 % MFMA-pipe utilisation in steady state).  v2 double-buffers LDS (one barrier per
//     chunk), writes chunk k+1 and issues the loads of chunk k+2 behind the first MFMAs of
//     chunk k, and drops the row padding for an XOR swizzle so two buffers still fit 2-3
//     workgroups per CU.
//
// LDS image of a [rows][32] tile: 128-byte rows, 16-byte group q of row r stored at group
// q ^ ((r >> 1) & 7): ds_write_b128 (8 lanes = one row) and the fragment ds_read_b128
// (16-lane groups over rows {0-3,12-15,20-27}+32k at one q) are both conflict-free.
#pragma once

template <int BM, int BN, int WM, int WN, int BMODE GG_ABL_PARAM>
__global__ void __launch_bounds__(256)
gather_gemm_f32_v2(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int A_IT = BM / 32;
    constexpr int LDB_KN = BN + 4;
    constexpr int TPR = BN / 4;
    constexpr int RPP = 256 / TPR;
    constexpr int B_IT = (BMODE == VSR_BMODE_NK) ? (BN / 32) : (32 / RPP);
    constexpr int AS_FLOATS = BM * 32;
    constexpr int BS_FLOATS = (BMODE == VSR_BMODE_NK) ? BN * 32 : 32 * LDB_KN;
    constexpr int BUF_FLOATS = AS_FLOATS + BS_FLOATS;
    static_assert(WM * WN == 4, "4 waves");

    __shared__ __attribute__((aligned(16))) float smem[2 * BUF_FLOATS + 4];
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + 2 * BUF_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int s_r = tid >> 3, s_q = tid & 7;
    const int k_r = tid / TPR, k_q = tid % TPR;
    // swizzled float offsets inside a 32-float row
    const int stOff = ((s_q ^ ((s_r >> 1) & 7)) << 2);                 // staging store (row = s_r + 32*it)
    int rdOff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rdOff[g] = (((2 * g + hi) ^ ((l31 >> 1) & 7)) << 2); // fragment read, row = .. + l31

    if (tid == 0) *nextTile = (int)atomicAdd(queue, 1u);
    __syncthreads();

    for (;;) {
        const int bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();                       // everyone has read the slot before it is refilled
        if (bid >= totalTiles) break;
        if (tid == 0) *nextTile = (int)atomicAdd(queue, 1u);   // prefetch the next tile id

        // last problem whose first tile id is <= bid (tileStart is non-decreasing; binary search: a grouped launch
        // can carry thousands of problems, e.g. ProPainter's per-window / per-frame attention)
        int pi = 0;
        for (int lo_ = 0, hi_ = nprobs - 1; lo_ < hi_;) {
            const int mid_ = (lo_ + hi_ + 1) >> 1;
            if (bid >= probs[mid_].tileStart) lo_ = mid_; else hi_ = mid_ - 1;
            pi = lo_;
        }
        const GGProblem* __restrict__ P = probs + pi;
        const int M = P->M, N = P->N;
        const int tilesN = P->tilesN, splitK = P->splitK;
        const int tilesMN = P->tilesM * tilesN;
        const int t = bid - P->tileStart;
        // tiles are handed out in queue order: consecutive ids share A rows (tn fastest)
        const int split = t / tilesMN;
        const int rem = t - split * tilesMN;
        const int tm = rem / tilesN;
        const int tn = rem - tm * tilesN;
        const int nchunksTotal = P->K / VSR_GG_KC;
        const int kcBeg = split * P->chunksPerSplit;
        int kcEnd = kcBeg + P->chunksPerSplit;
        if (kcEnd > nchunksTotal) kcEnd = nchunksTotal;

        const gcf32 A = (gcf32)P->A;
        const gcf32 B = (gcf32)P->B;
        const gci32 rowA = (gci32)P->rowA;
        const cci32 colA = (cci32)P->colA;
        const gci32 rowB = (gci32)P->rowB;
        const cci32 colB = (cci32)P->colB;

        int aoff[A_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aoff[it] = rowA[tm * BM + s_r + 32 * it] + 4 * s_q;
        int boff[B_IT], boffNext[B_IT];
        int bcolKN = 0;
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) boff[it] = rowB[tn * BN + s_r + 32 * it] + 4 * s_q;
        } else {
            bcolKN = colB[(tn * BN) / VSR_GG_KC + (k_q >> 3)] + 4 * (k_q & 7);
        }

        f32x4 ra[A_IT], rb[B_IT];
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        auto load_rowB_KN = [&](int kc, int (&dst)[B_IT]) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) dst[it] = rowB[kc * VSR_GG_KC + k_r + RPP * it];
        };
        auto load_tile = [&](int kc) {
            const int ca = colA[kc];
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *(gcf32x4)(A + (aoff[it] + ca));
            if constexpr (BMODE == VSR_BMODE_NK) {
                const int cb = colB[kc];
#pragma unroll
                for (int it = 0; it < B_IT; ++it) rb[it] = *(gcf32x4)(B + (boff[it] + cb));
            } else {
#pragma unroll
                for (int it = 0; it < B_IT; ++it) rb[it] = *(gcf32x4)(B + (boff[it] + bcolKN));
            }
        };
        auto store_tile = [&](int buf) {
            float* As = smem + buf * BUF_FLOATS;
            float* Bs = As + AS_FLOATS;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                *reinterpret_cast<f32x4*>(&As[(s_r + 32 * it) * 32 + stOff]) = ra[it];
            if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    *reinterpret_cast<f32x4*>(&Bs[(s_r + 32 * it) * 32 + stOff]) = rb[it];
            } else {
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    *reinterpret_cast<f32x4*>(&Bs[(k_r + RPP * it) * LDB_KN + 4 * k_q]) = rb[it];
            }
        };
        auto compute_group = [&](int buf, int g) {
            const float* As = smem + buf * BUF_FLOATS;
            const float* Bs = As + AS_FLOATS;
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const f32x4*>(&As[(wm * WTM + mi * 32 + l31) * 32 + rdOff[g]]);
            if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bf[ni] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WTN + ni * 32 + l31) * 32 + rdOff[g]]);
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bf[ni][j] = Bs[(8 * g + 4 * hi + j) * LDB_KN + wn * WTN + ni * 32 + l31];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
        };

        if (kcBeg < kcEnd) {
            if constexpr (BMODE == VSR_BMODE_KN) load_rowB_KN(kcBeg, boff);
            load_tile(kcBeg);
            if constexpr (BMODE == VSR_BMODE_KN) {
                if (kcBeg + 1 < kcEnd) load_rowB_KN(kcBeg + 1, boffNext);
            }
            store_tile(0);
            if (kcBeg + 1 < kcEnd) {
                if constexpr (BMODE == VSR_BMODE_KN) {
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                    if (kcBeg + 2 < kcEnd) load_rowB_KN(kcBeg + 2, boffNext);
                }
                load_tile(kcBeg + 1);
            }
            __syncthreads();
            int cur = 0;
            for (int kc = kcBeg; kc < kcEnd; ++kc) {
                // registers hold chunk kc+1; LDS[cur] holds chunk kc
                compute_group(cur, 0);
                if (kc + 1 < kcEnd) {
                    store_tile(cur ^ 1);       // last read in iteration kc-1, fenced by its barrier
                    if (kc + 2 < kcEnd) {
                        if constexpr (BMODE == VSR_BMODE_KN) {
#pragma unroll
                            for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                            if (kc + 3 < kcEnd) load_rowB_KN(kc + 3, boffNext);
                        }
                        load_tile(kc + 2);
                    }
                }
                compute_group(cur, 1);
                compute_group(cur, 2);
                compute_group(cur, 3);
                __syncthreads();
                cur ^= 1;
            }
        }

        // ---- epilogue (identical to v1) ----
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;   // relu(act(..) + R): residual blocks of RAFT
        const bool partial = (splitK > 1);
        const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
        const gcf32 R = partial ? (gcf32) nullptr : (gcf32)P->R;
        const gci32 rowC = (gci32)P->rowC;
        const cci32 colC = (cci32)P->colC;
        const gci32 rowR = (gci32)P->rowR;
        const gf32 C = (gf32)(P->C + (partial ? (int64_t)split * P->splitStride : (int64_t)0));
        int ccol[NI];
        float bv[NI];
        bool nok[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n0 = tn * BN + wn * WTN + ni * 32;
            ccol[ni] = colC[n0 / VSR_GG_KC] + l31;
            nok[ni] = (n0 + l31) < N;
            bv[ni] = (bias != nullptr && nok[ni]) ? bias[n0 + l31] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int m = tm * BM + row;
                const int rc = rowC[m];
                const int rr = (R != nullptr) ? rowR[m] : 0;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    float v = acc[mi][ni][r] * alpha + bv[ni];
                    if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
                        else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
                    if (m < M && nok[ni]) {
                        if (R != nullptr) { v += R[rr + ccol[ni]]; if (postRelu) v = fmaxf(v, 0.f); }
                        C[rc + ccol[ni]] = v;
                    }
                }
            }
        }
        __syncthreads();   // nextTile written by tid 0 above is visible; LDS buffers free for the next tile
    }
}
