// gather_gemm_f32_aexp: the KN kernel of gather_gemm.hip (one workgroup per tile, register-staged single LDS buffer) for the
// second half of a fused attention, softmax(S) . V (reference backend/inpaint/sttn/auto_sttn.py:141-145) with no probability
// matrix in memory.  A problem that carries VSR_ACT_A_EXP hands over the SCORES as its A operand and the row maxima the score
// GEMM left behind (VSR_ACT_ROW_MAX, gather_gemm_v3.h); while a 128x32 score tile sits in registers between its global load and
// its LDS store every value becomes 2^(s - max_row) (the scores carry the factor log2(e) / sqrt(D)) -- sixteen v_exp_f32 per lane and chunk, issued in the shadow of the other
// workgroups' MFMAs -- and joins its row's running sum.  The thread -> row assignment of the staging is the same for every chunk,
// so the sums live in four registers per lane and meet once, after the loop.  splitK 1: the epilogue divides by the row sum;
// splitK > 1 (the 4800-token scale is cut in three): partial planes stay unnormalised and the N-tile 0 of every split writes its
// partial sums for the reduce-scatter pass (elementwise.hip) to divide by their total -- fixed summation order, so the result is
// deterministic.  Problems without the flag (the coarse scales of the same grouped launch, whose scores were split along K and
// therefore normalised by k_softmax_rows) run exactly as in gather_gemm_f32<BM, BN, WM, WN, KN>.
#pragma once

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256)
gather_gemm_f32_aexp(const GGProblem* __restrict__ probs, int nprobs)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int A_IT = BM / 32;
    constexpr int LDB_KN = BN + 4;
    constexpr int TPR = BN / 4;    // threads per k-row of the B tile
    constexpr int RPP = 256 / TPR; // k-rows per pass
    constexpr int B_IT = 32 / RPP;
    static_assert(WM * WN == 4, "4 waves");

    __shared__ __attribute__((aligned(16))) float smem[BM * GG_LD + 32 * LDB_KN];
    float* As = smem;
    float* Bs = smem + BM * GG_LD;

    const int bid = blockIdx.x;
    int pi = 0;
    for (int lo_ = 0, hi_ = nprobs - 1; lo_ < hi_;) {
        const int mid_ = (lo_ + hi_ + 1) >> 1;
        if (bid >= probs[mid_].tileStart) lo_ = mid_; else hi_ = mid_ - 1;
        pi = lo_;
    }
    const GGProblem* __restrict__ P = probs + pi;

    const int M = P->M, N = P->N;
    const int tilesM = P->tilesM, tilesN = P->tilesN, splitK = P->splitK;
    const int tilesMN = tilesM * tilesN;
    const int nblk = tilesMN * splitK;
    int t = bid - P->tileStart;
    {   // XCD-aware remap (as gather_gemm_f32)
        const int xcd = t & 7, q = nblk >> 3, r = nblk & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
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
    const bool aexp = (P->act & VSR_ACT_A_EXP) != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int s_r = tid >> 3, s_q = tid & 7;
    int aoff[A_IT];
    float mrow[A_IT], lsum[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        aoff[it] = rowA[tm * BM + s_r + 32 * it] + 4 * s_q;
        lsum[it] = 0.f;
        mrow[it] = 0.f;
    }
    if (aexp) {
        const unsigned int __attribute__((address_space(1)))* rmax = (const unsigned int __attribute__((address_space(1)))*)P->bias;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) mrow[it] = f32_from_ordered(rmax[tm * BM + s_r + 32 * it]);   // array padded to tilesM * BM
    }

    int boff[B_IT], boffNext[B_IT];
    const int k_r = tid / TPR, k_q = tid % TPR;
    const int bcolKN = colB[(tn * BN) / VSR_GG_KC + (k_q >> 3)] + 4 * (k_q & 7);

    f32x4 ra[A_IT], rb[B_IT];
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto load_rowB = [&](int kc, int (&dst)[B_IT]) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it) dst[it] = rowB[kc * VSR_GG_KC + k_r + RPP * it];
    };
    auto load_tile = [&](int kc) {
        const int ca = colA[kc];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) ra[it] = *(gcf32x4)(A + (aoff[it] + ca));
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rb[it] = *(gcf32x4)(B + (boff[it] + bcolKN));
    };
    // the row sums are only read where they are used: by every tile of an unsplit product (its epilogue divides), by the N-tile 0 of
    // a split one (it writes them for the reduce pass)
    const bool needSum = aexp && (splitK == 1 || tn == 0);
    auto store_tile = [&]() {
        if (aexp) {            // scores arrive scaled by log2(e) / sqrt(D) (the score GEMM's alpha): one subtract and one v_exp_f32 each
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[it][j] = __builtin_amdgcn_exp2f(ra[it][j] - mrow[it]);
            if (needSum) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) lsum[it] += (ra[it][0] + ra[it][1]) + (ra[it][2] + ra[it][3]);
            }
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            *reinterpret_cast<f32x4*>(&As[(s_r + 32 * it) * GG_LD + 4 * s_q]) = ra[it];
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            *reinterpret_cast<f32x4*>(&Bs[(k_r + RPP * it) * LDB_KN + 4 * k_q]) = rb[it];
    };
    auto compute_tile = [&]() {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const f32x4*>(&As[(wm * WTM + mi * 32 + l31) * GG_LD + 8 * g + 4 * hi]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bf[ni][j] = Bs[(8 * g + 4 * hi + j) * LDB_KN + wn * WTN + ni * 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
        }
    };

    if (kcBeg < kcEnd) {
        load_rowB(kcBeg, boff);
        if (kcBeg + 1 < kcEnd) load_rowB(kcBeg + 1, boffNext);
        load_tile(kcBeg);
        store_tile();
        __syncthreads();
        for (int kc = kcBeg; kc < kcEnd; ++kc) {
            const bool hasNext = (kc + 1 < kcEnd);
            if (hasNext) {
#pragma unroll
                for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                load_tile(kc + 1);
                if (kc + 2 < kcEnd) load_rowB(kc + 2, boffNext);
            }
            compute_tile();
            __syncthreads();
            if (hasNext) store_tile();
            __syncthreads();
        }
    }

    // ---- row sums of the exponentials: the 8 lanes that staged one row meet by shuffle, the tile's BM sums go to LDS
    float* lrow = smem;                               // [BM]; the loop ended on a barrier
    const bool partial = (splitK > 1);
    if (aexp) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            float s = lsum[it];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (s_q == 0) lrow[s_r + 32 * it] = s;
        }
        __syncthreads();
        if (partial && tn == 0 && tid < BM)           // one writer per (split, row): the reduce pass adds the splits in order
            const_cast<float*>(P->R)[(int64_t)split * tilesM * BM + tm * BM + tid] = lrow[tid];
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float alpha = P->alpha;
    const int act = P->act & 0xff;
    const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;
    const gcf32 bias = (partial || aexp) ? (gcf32) nullptr : (gcf32)P->bias;
    const gcf32 R = (partial || aexp) ? (gcf32) nullptr : (gcf32)P->R;
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
            const float inv = (aexp && !partial) ? 1.f / lrow[row] : 1.f;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float v = acc[mi][ni][r] * alpha * inv + bv[ni];
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
}
