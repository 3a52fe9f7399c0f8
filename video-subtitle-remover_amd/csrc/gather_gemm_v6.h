// gather_gemm_f16_v6: the fp16-operand mode (BASELINE.json config 5, kernel variant 6) for NK problems on SPLIT-FORMAT tensors.
//
// v5 in its HI_ONLY mode still streamed whole 128-byte chunks ([32 hi | 32 lo] halves) through L2 -> LDS and used half of
// them, and paid one barrier per 32-deep chunk although a chunk is only MI*NI*2 fp16 MFMAs per wave: it ran at 0.10 of the
// dense-f16 roof, bound by the operand stream and by the per-chunk fixed costs (round-1 VERDICT, weak #3).  v6 keeps v5's
// tensors, offset tables, LDS geometry (128-byte rows of 8 sixteen-byte pieces, bank-conflict XOR applied on the source side of
// the LDS-DMA and mirrored on the fragment read) and epilogue, but a row of an LDS stage now holds the hi halves of TWO
// consecutive K chunks -- pieces 0-3 = chunk 2j, pieces 4-7 = chunk 2j+1; the lane that fills piece slot q of row r fetches
// logical piece q ^ ((r >> 1) & 7) and picks its chunk by that piece's upper bit.  Per barrier a workgroup therefore moves the
// bytes v5 moved and contracts 64 k-values instead of 32, none of the bytes wasted: half the L2 -> LDS traffic and half the
// barriers per FLOP.  An odd chunk count ends with a half stage (its second half re-fetches the first chunk and is not
// contracted).  Operands: fp16 hi halves, fp32 accumulation (v_mfma_f32_32x32x16_f16), range guard as in v4 / v5.
#pragma once
#include <type_traits>

template <int BM, int BN, int WM, int WN, int STAGES>
__global__ void __launch_bounds__(256, (STAGES * (BM + BN) * 128 + 2 * BM * 4 <= 78 * 1024) ? 2 : 1)
gather_gemm_f16_v6(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue, int nQueues,
                   unsigned int* __restrict__ rangeFlag)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int AS_FLOATS = BM * 32, BS_FLOATS = BN * 32;     // [rows][128 bytes]
    constexpr int BUF_FLOATS = AS_FLOATS + BS_FLOATS;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(STAGES >= 2 && STAGES <= 4, "2..4 operand stages");

    // ONE __shared__ object (a second one makes hipcc drain the LDS-DMA queue in front of every fragment read)
    __shared__ __attribute__((aligned(16))) float smem[STAGES * BUF_FLOATS + 2 * BM + 4];
    int* rowTab = reinterpret_cast<int*>(smem + STAGES * BUF_FLOATS);
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + STAGES * BUF_FLOATS + 2 * BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int s_r = tid >> 3, s_q = tid & 7;                    // LDS-DMA: row-in-pass, piece slot
    const int lp = s_q ^ ((s_r >> 1) & 7);                      // logical piece this lane fetches
    const bool second = (lp & 4) != 0;                          // ... of the second chunk of the pair
    const int srcSwz = (lp & 3) << 2;                           // float offset of the 16-byte group inside the chunk's hi half
    // MFMA step st (k = 16 st .. 16 st + 15 of the 64-deep pair): lane (l31, hi) reads logical piece 2 st + hi
    int rd[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) rd[st] = (((2 * st + hi) ^ ((l31 >> 1) & 7)) << 4);

    int qFirst = 0;
    auto fetchTile = [&]() -> int {
        const int home = blockIdx.x % nQueues;
        for (; qFirst < nQueues; ++qFirst) {
            const int x = (home + qFirst) % nQueues;
            const int lo = (int)(((long long)totalTiles * x) / nQueues), hi_ = (int)(((long long)totalTiles * (x + 1)) / nQueues);
            if (lo < hi_) {
                const int i = lo + (int)atomicAdd(queue + x, 1u);
                if (i < hi_) return i;
            }
        }
        return totalTiles;
    };
    if (tid == 0) *nextTile = fetchTile();
    __syncthreads();

    for (;;) {
        const int bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();
        if (bid >= totalTiles) break;
        if (tid == 0) *nextTile = fetchTile();

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

        {
            const gci32 rowCt = (gci32)P->rowC;
            const gci32 rowRt = (gci32)P->rowR;
            const bool hasR = (P->R != nullptr) && (splitK == 1);
#pragma unroll
            for (int i = tid; i < 2 * BM; i += 256)
                rowTab[i] = i < BM ? rowCt[tm * BM + i] : (hasR ? rowRt[tm * BM + i - BM] : 0);
        }
        int aoff[A_IT], boff[B_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aoff[it] = rowA[tm * BM + s_r + 32 * it] + srcSwz;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) boff[it] = rowB[tn * BN + s_r + 32 * it] + srcSwz;

        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // LDS-DMA of one pair of chunks: ca0 / ca1 (cb0 / cb1) = wave-uniform chunk offsets of the pair in A (B)
        auto dma_pair = [&](int buf, int ca0, int ca1, int cb0, int cb1) {
            float* As = smem + buf * BUF_FLOATS;
            float* Bs = As + AS_FLOATS;
            const int ca = second ? ca1 : ca0, cb = second ? cb1 : cb0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                glds16(A + (aoff[it] + ca), (lds_vptr)(As + (wave * 8 + 32 * it) * 32));
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                glds16(B + (boff[it] + cb), (lds_vptr)(Bs + (wave * 8 + 32 * it) * 32));
        };
        auto compute_step = [&](int buf, int st) {
            const char* As = reinterpret_cast<const char*>(smem + buf * BUF_FLOATS);
            const char* Bs = As + AS_FLOATS * 4;
            f16x8 ah[MI], bh[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                ah[mi] = *reinterpret_cast<const f16x8*>(As + (wm * WTM + mi * 32 + l31) * 128 + rd[st]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bh[ni] = *reinterpret_cast<const f16x8*>(Bs + (wn * WTN + ni * 32 + l31) * 128 + rd[st]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        };

        // Operand pipeline, D = STAGES - 1 pairs deep, counted vmcnt (each wave waits only for ITS pieces of the pair it is about
        // to read; raw s_barrier: __syncthreads() would drain the DMA queue).  No ordinary (VGPR-destination) load inside the
        // loop: hipcc would wait vmcnt(0) for it.  The chunk-offset tables are fetched per super-block of 128 chunks (2 VGPRs
        // per table); 128 is even, so a pair never straddles two super-blocks.
        constexpr int D = STAGES - 1;
        constexpr int PIECES = A_IT + B_IT;
        static_assert(PIECES * (D - 1) <= 63, "vmcnt is 6 bits");
        for (int sb = kcBeg; sb < kcEnd; sb += 128) {
            const int sbEnd = sb + 128 < kcEnd ? sb + 128 : kcEnd;
            const int i0 = sb + lane < nchunksTotal ? sb + lane : nchunksTotal - 1;
            const int i1 = sb + 64 + lane < nchunksTotal ? sb + 64 + lane : nchunksTotal - 1;
            const int ca0v = colA[i0], ca1v = colA[i1], cb0v = colB[i0], cb1v = colB[i1];
            asm volatile("" ::"v"(ca0v), "v"(ca1v), "v"(cb0v), "v"(cb1v));
            auto pick = [&](int v0, int v1, int i) { return i < 64 ? __builtin_amdgcn_readlane(v0, i) : __builtin_amdgcn_readlane(v1, i - 64); };
            auto issue = [&](int kc, int buf) {                 // pair (kc, kc + 1); a lone last chunk is fetched twice
                const int i = kc - sb, j = kc + 1 < sbEnd ? i + 1 : i;
                dma_pair(buf, pick(ca0v, ca1v, i), pick(ca0v, ca1v, j), pick(cb0v, cb1v, i), pick(cb0v, cb1v, j));
            };
            if (sb != kcBeg) {                                  // stages of the previous super-block are still being read
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (sb + 2 * d < sbEnd) issue(sb + 2 * d, d);
            int cur = 0, nxt = D % STAGES;
            for (int kc = sb; kc < sbEnd; kc += 2) {
                const int ahead = (sbEnd - 1 - kc) >> 1;        // younger pairs already issued: min(ahead, D - 1)
                if (D >= 3 && ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (D >= 3 ? 2 : 0)) : "memory");
                else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (D >= 2 ? 1 : 0)) : "memory");
                else                           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kc + 2 * D < sbEnd) issue(kc + 2 * D, nxt);
                compute_step(cur, 0);
                compute_step(cur, 1);
                if (kc + 1 < sbEnd) {
                    compute_step(cur, 2);
                    compute_step(cur, 3);
                }
                cur = cur + 1 == STAGES ? 0 : cur + 1;
                nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
            }
        }
        __syncthreads();                           // rowTab visible even when the k range is empty; LDS-DMA queue empty

        // ---- epilogue (as v5): C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;
        const bool cSplit = (P->act & VSR_ACT_OUT_SPLIT) != 0;
        const float vmax = cSplit ? 65504.f : 3.0e38f;
        bool nonFinite = false;
        const bool partial = (splitK > 1);
        const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
        const gcf32 R = partial ? (gcf32) nullptr : (gcf32)P->R;
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
                if constexpr (HASR) {        // residual tensors are GEMM operands too: split format
                    typedef const _Float16 __attribute__((address_space(1)))* gch;
                    const gch R16 = (gch)R;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool mok = FULL || (tm * BM + row) < M;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            float x = 0.f;
                            if (mok && (FULL || nok[ni])) {
                                const int e = 2 * (rr[r] + ccol[ni] - l31) + l31;
                                x = (float)R16[e] + (float)R16[e + 32];
                            }
                            rv[r][ni] = x;
                        }
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
                        nonFinite |= !(__builtin_fabsf(v) <= vmax);
                        if (mok && (FULL || nok[ni])) {
                            if (cSplit) {
                                typedef _Float16 __attribute__((address_space(1)))* gh;
                                const gh C16 = (gh)C;
                                const int e = 2 * (rc[r] + ccol[ni] - l31) + l31;
                                const _Float16 h = (_Float16)v;
                                C16[e] = h;
                                C16[e + 32] = (_Float16)(v - (float)h);
                            } else {
                                C[rc[r] + ccol[ni]] = v;
                            }
                        }
                    }
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (fullTile) { if (R != nullptr) epilogue(T_{}, T_{}); else epilogue(T_{}, F_{}); }
        else          { if (R != nullptr) epilogue(F_{}, T_{}); else epilogue(F_{}, F_{}); }
        if (rangeFlag != nullptr && __any(nonFinite) && lane == 0) atomicOr(rangeFlag, 1u);
        __syncthreads();
    }
}
