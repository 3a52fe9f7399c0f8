// gather_gemm_f32_v3: persistent + LDS-DMA variant of the grouped gather-GEMM.
//
// Same GGProblem semantics, tile shapes and MFMA fragment mapping as v1/v2; what changes is how
// operand tiles reach LDS:
//   * global_load_lds_dwordx4 (LDS-DMA): each lane fetches 16 bytes of its gathered row and the
//     hardware writes them to LDS at (wave-uniform base + lane*16) -- no VGPR round trip, no
//     ds_write pass, no staging registers.  The LDS image must therefore be lane-linear: a
//     [rows][32] tile is stored with 128-byte rows (8 lanes per row, 8 rows = 1 KiB per wave
//     instruction) and the bank-conflict swizzle is applied on the SOURCE side: the lane that
//     owns slot q of row r fetches 16-byte group q ^ ((r >> 1) & 7) of that row, and fragment
//     reads use the same XOR (linear destination + permuted source + permuted read).
//   * two LDS buffers, one barrier per 32-deep chunk: the DMA of chunk k+1 is issued right after
//     the barrier that retires chunk k-1 and is drained (vmcnt(0), which hipcc attaches to
//     __syncthreads() while an LDS-DMA is in flight) at the barrier that ends chunk k.
//   * chunk offsets (colA / colB tables) are held in a VGPR (one table entry per lane, refreshed
//     every 64 chunks, prefetched one refresh ahead) and picked with v_readlane: no scalar load
//     and no lgkmcnt(0) stall in front of every chunk.
//   * persistent workgroups pulling tiles from per-XCD atomic queues with stealing (see v2 for the
//     measurement that motivates persistence: the dispatcher packs a partial last round onto few
//     CUs; one global queue scattered neighbouring tiles over all XCDs and tripled the fabric reads).
#pragma once
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_vptr;

// 16-byte LDS-DMA: global (per lane) -> LDS (wave-uniform base + lane*16).  The builtin needs the
// gfx950 target features, so the host pass of hipcc (which still has to emit the kernel's launch
// stub) sees an empty body.
__device__ __forceinline__ void glds16(gcf32 src, lds_vptr dst)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

#ifdef GG_ABLATE
// 64: wave 0 of the first 1024 workgroups logs s_memtime at tile start, after the prologue barrier,
// after every chunk barrier and after the epilogue (256 stamps per workgroup)
__device__ unsigned long long gg_trace[1024 * 256];
#define GG_STAMP()                                                                     \
    if constexpr (GG_ABL(64)) {                                                        \
        if (tid == 0 && blockIdx.x < 1024 && tr_ < 256)                                \
            gg_trace[blockIdx.x * 256 + tr_] = __builtin_readcyclecounter();           \
        ++tr_;                                                                         \
    }
#else
#define GG_STAMP()
#endif

template <int BM, int BN, int WM, int WN, int BMODE GG_ABL_PARAM>
__global__ void __launch_bounds__(256)
gather_gemm_f32_v3(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue, int nQueues)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int A_IT = BM / 32;
    constexpr int TPR = BN / 4;
    constexpr int RPP = 256 / TPR;
    constexpr int B_IT = (BMODE == VSR_BMODE_NK) ? (BN / 32) : (32 / RPP);
    constexpr int AS_FLOATS = BM * 32;
    constexpr int BS_FLOATS = BN * 32;          // NK: [BN][32] swizzled ; KN: [32][BN] linear
    constexpr int BUF_FLOATS = AS_FLOATS + BS_FLOATS;
    static_assert(WM * WN == 4, "4 waves");

    // [2 operand buffers][rowC | rowR offsets of the tile's BM rows][next tile id]
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF_FLOATS + 2 * BM + 4];
    int* rowTab = reinterpret_cast<int*>(smem + 2 * BUF_FLOATS);
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + 2 * BUF_FLOATS + 2 * BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int s_r = tid >> 3, s_q = tid & 7;            // [rows][32] images: row-in-pass, slot
    const int k_r = tid / TPR, k_q = tid % TPR;         // KN B image: k-row-in-pass, float4-in-row
    const int srcSwz = ((s_q ^ ((s_r >> 1) & 7)) << 2); // float offset of the 16-byte group this lane fetches
    int rdOff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rdOff[g] = (((2 * g + hi) ^ ((l31 >> 1) & 7)) << 2);

    // Tile queue: the flat tile-id space is cut into 8 contiguous ranges, one per XCD, each with its
    // own counter (queue[0..7], zeroed by the host).  A workgroup drains the range of the XCD it runs
    // on (blockIdx % 8 -- observed placement, used for L2 locality only: tiles that share A rows or
    // weight columns are neighbours in id space and so meet in one L2) and then steals from the
    // following ranges, so every id is handed out exactly once whatever the placement is.
    int qFirst = 0;                                    // ranges before (home + qFirst) are known to be empty
    auto fetchTile = [&]() -> int {
        const int home = blockIdx.x % nQueues;             // nQueues = 8 (one per XCD) or 1 (single global queue)
        for (; qFirst < nQueues; ++qFirst) {
            const int x = (home + qFirst) % nQueues;
            const int lo = (int)(((long long)totalTiles * x) / nQueues), hi = (int)(((long long)totalTiles * (x + 1)) / nQueues);
            if (lo < hi) {
                // (a plain-load peek in front of the atomic was tried here: it removes the same-address atomics on exhausted
                // counters at the tail of a launch, but adds a round trip to every steal -- the fp32 bench lost 2 %,
                // profiles/r02_peek_ab.log; only the short-tile fp16 kernel v6 keeps it)
                const int i = lo + (int)atomicAdd(queue + x, 1u);
                if (i < hi) return i;
            }
        }
        return totalTiles;
    };
    if (tid == 0) *nextTile = fetchTile();
    __syncthreads();
#ifdef GG_ABLATE
    int tr_ = 0;
#endif

    for (;;) {
        const int bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();
        if (bid >= totalTiles) break;
        if (tid == 0) *nextTile = fetchTile();
        GG_STAMP()   // tile start

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
        const gci32 colA = (gci32)P->colA;
        const gci32 rowB = (gci32)P->rowB;
        const gci32 colB = (gci32)P->colB;

        {   // output / residual row offsets of this tile -> LDS (read back in the epilogue; the
            // main loop's barriers order the two), so the epilogue starts without a dependent
            // global table read per row
            const gci32 rowCt = (gci32)P->rowC;
            const gci32 rowRt = (gci32)P->rowR;
            const bool hasR = (P->R != nullptr) && (splitK == 1) && !(P->act & VSR_ACT_ROW_MAX);
#pragma unroll
            for (int i = tid; i < 2 * BM; i += 256)
                rowTab[i] = i < BM ? rowCt[tm * BM + i] : (hasR ? rowRt[tm * BM + i - BM] : 0);
        }
        int aoff[A_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aoff[it] = rowA[tm * BM + s_r + 32 * it] + srcSwz;
        int boff[B_IT], boffNext[B_IT];
        int bcolKN = 0;
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) boff[it] = rowB[tn * BN + s_r + 32 * it] + srcSwz;
        } else {
            bcolKN = colB[(tn * BN) / VSR_GG_KC + (k_q >> 3)] + 4 * (k_q & 7);
        }

        // chunk-offset tables: lane i holds entry (base + i); refreshed every 64 chunks
        int colBase = kcBeg;                                   // chunk index held by lane 0 of vcolA/vcolB
        auto fetchCols = [&](int base, int& va, int& vb) {
            const int idx = base + lane < nchunksTotal ? base + lane : nchunksTotal - 1;
            va = colA[idx];
            if constexpr (BMODE == VSR_BMODE_NK) vb = colB[idx]; else vb = 0;
        };
        int vcolA = 0, vcolB = 0, vcolAn = 0, vcolBn = 0;
        fetchCols(colBase, vcolA, vcolB);
        fetchCols(colBase + 64, vcolAn, vcolBn);

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
        // LDS-DMA of chunk kc into buffer buf (destination = wave-uniform base + lane*16)
        auto dma_tile = [&](int kc, int buf) {
            float* As = smem + buf * BUF_FLOATS;
            float* Bs = As + AS_FLOATS;
            const int ca = __builtin_amdgcn_readlane(vcolA, kc - colBase);
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                glds16(A + (aoff[it] + ca), (lds_vptr)(As + (wave * 8 + 32 * it) * 32));
            if constexpr (BMODE == VSR_BMODE_NK) {
                const int cb = __builtin_amdgcn_readlane(vcolB, kc - colBase);
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    glds16(B + (boff[it] + cb), (lds_vptr)(Bs + (wave * 8 + 32 * it) * 32));
            } else {
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    glds16(B + (boff[it] + bcolKN), (lds_vptr)(Bs + (wave * (64 / TPR) + RPP * it) * BN));
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
                        bf[ni][j] = Bs[(8 * g + 4 * hi + j) * BN + wn * WTN + ni * 32 + l31];
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
            if constexpr (BMODE == VSR_BMODE_KN) {
                load_rowB_KN(kcBeg, boff);
                if (kcBeg + 1 < kcEnd) load_rowB_KN(kcBeg + 1, boffNext);
            }
            dma_tile(kcBeg, 0);
            __syncthreads();                       // drains the DMA (vmcnt(0)) and publishes buffer 0
            GG_STAMP()   // prologue done
            int cur = 0;
            for (int kc = kcBeg; kc < kcEnd; ++kc) {
                if (kc + 1 < kcEnd) {
                    if (kc + 1 - colBase >= 64) {  // next 64 table entries become current
                        colBase += 64;
                        vcolA = vcolAn; vcolB = vcolBn;
                        fetchCols(colBase + 64, vcolAn, vcolBn);
                    }
                    if constexpr (BMODE == VSR_BMODE_KN) {
#pragma unroll
                        for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                        if (kc + 2 < kcEnd) load_rowB_KN(kc + 2, boffNext);
                    }
                    dma_tile(kc + 1, cur ^ 1);     // buffer last read in iteration kc-1, fenced by its barrier
                }
                compute_group(cur, 0);
                compute_group(cur, 1);
                compute_group(cur, 2);
                compute_group(cur, 3);
                __syncthreads();                   // vmcnt(0) + barrier: chunk kc+1 landed, chunk kc retired
                GG_STAMP()   // chunk done
                cur ^= 1;
            }
        }

        // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
        // Row offsets come from LDS, the residual reads of 16 rows are issued back to back.
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;   // relu(act(..) + R): residual blocks of RAFT
        const bool partial = (splitK > 1);
        const bool rowMax = (P->act & VSR_ACT_ROW_MAX) != 0;       // R is the row-maximum array of the scores, not a residual
        const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
        const gcf32 R = (partial || rowMax) ? (gcf32) nullptr : (gcf32)P->R;
        const cci32 colC = (cci32)P->colC;
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
        // interior tiles (the vast majority) take a branch-free path: per-element predicates put every
        // store into its own basic block, and hipcc then drains vmcnt(0) in front of each one -- 32-64
        // serialised store round trips (~45k cycles per tile, measured) instead of a pipelined burst
        const bool fullTile = (tm * BM + BM <= M) && (tn * BN + BN <= N);
        auto epilogue = [&](auto fullTag, auto resTag) {
            constexpr bool FULL = decltype(fullTag)::value, HASR = decltype(resTag)::value;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int rc[16], rr[16];
                float rv[16][NI];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    rc[r] = rowTab[row];
                    if constexpr (HASR) rr[r] = rowTab[BM + row];
                }
                if constexpr (HASR) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool mok = FULL || (tm * BM + row) < M;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            rv[r][ni] = (mok && (FULL || nok[ni])) ? R[rr[r] + ccol[ni]] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool mok = FULL || (tm * BM + row) < M;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        float v = acc[mi][ni][r] * alpha + bv[ni];
                        if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
                        else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
                        if constexpr (HASR) { v += rv[r][ni]; if (postRelu) v = fmaxf(v, 0.f); }
                        if (mok && (FULL || nok[ni])) C[rc[r] + ccol[ni]] = v;
                    }
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (fullTile) { if (R != nullptr) epilogue(T_{}, T_{}); else epilogue(T_{}, F_{}); }
        else          { if (R != nullptr) epilogue(F_{}, T_{}); else epilogue(F_{}, F_{}); }
        if (rowMax) {
            // VSR_ACT_ROW_MAX (the QK^T of a fused attention): the largest score of every row of this tile joins the row's running
            // maximum -- 32 lanes hold a row's columns, the waves side by side meet in LDS (the operand buffers are idle: the main
            // loop ended on a barrier), then one atomic per row in a coalesced burst.  The P.V kernel subtracts it (VSR_ACT_A_EXP).
            float* scr = smem;                                      // [BM][WN]
            // A lane holds one column of 16 * MI rows; a row's 32 columns sit in the 32 lanes of its half wave.  Halving butterfly:
            // at every step a lane keeps one half of its rows and hands the other half to its partner, so after log2(32) steps
            // lane l31 holds the maximum of row-slot l31 -- 31 exchanges per lane instead of 5 for each of the 32 rows.
            static_assert(MI * 16 == 32 || MI * 16 == 16, "row slots per lane");
            constexpr int NV = MI * 16;
            float v[NV];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        if (nok[ni]) mx = fmaxf(mx, acc[mi][ni][r] * alpha + bv[ni]);
                    v[mi * 16 + r] = mx;
                }
            if constexpr (NV == 16) {                               // 16 row slots: the first exchange is a plain maximum
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], __shfl_xor(v[i], 16, 64));
            }
#pragma unroll
            for (int half = (NV == 32 ? 16 : 8), bit = (NV == 32 ? 16 : 8); half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (l31 & bit) != 0;                   // this lane keeps slots [half, 2 half), its partner [0, half)
#pragma unroll
                for (int i = 0; i < half; ++i) {
                    const float lo = v[i], hi_ = v[i + half];
                    const float got = __shfl_xor(up ? lo : hi_, bit, 64);
                    v[i] = fmaxf(up ? hi_ : lo, got);
                }
            }
            {   // slot -> row of the wave's sub-tile: slot = mi * 16 + r, row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi
                const int slot = (NV == 32) ? l31 : (l31 & 15);
                const int r = slot & 15, mi = slot >> 4;
                if (NV == 32 || l31 < 16) scr[(wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * WN + wn] = v[0];
            }
            __syncthreads();
            if (tid < BM && tm * BM + tid < M) {
                float mx = scr[tid * WN];
#pragma unroll
                for (int w = 1; w < WN; ++w) mx = fmaxf(mx, scr[tid * WN + w]);
                atomicMax(reinterpret_cast<unsigned int*>(const_cast<float*>(P->R)) + tm * BM + tid, f32_ordered(mx));
            }
        }
        __syncthreads();
        GG_STAMP()   // epilogue done
    }
}
