// gather_gemm_f32_v5: split-half operands on the f16 matrix cores with SPLIT-FORMAT tensors in HBM.
//
// v4 showed that the split-half arithmetic (a = hi + lo in fp16, a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi,
// fp32 accumulation) is as accurate as the fp32 kernels on this network, but that splitting inside the
// GEMM costs more than the MFMAs it feeds: 235 TF with the global->VGPR->cvt->ds_write staging vs 400 TF
// without it (scripts/attic/gg_ablate.hip).  v5 therefore moves the split to the PRODUCER: every tensor that is
// a GEMM operand lives in HBM in "split format" -- each 32-float chunk (128 B) holds [32 hi halves | 32 lo
// halves] of the same 32 values -- so all offset tables (which address 32-element chunks) are unchanged,
// an operand tile reaches LDS by LDS-DMA exactly like v3 (no VGPR staging, no conversion, no ds_write), and
// its 128-byte rows already are the [pieces 0-3 = hi | pieces 4-7 = lo] image the f16 MFMA fragments read.
// The epilogue splits its fp32 results once and writes split format (or plain fp32 for tensors that are
// not GEMM operands: attention scores, P.V partial planes, the 3-channel decoder output); residuals are
// read in split format.  KN problems (P.V) gather their B operand (V, split format) with 2-byte loads and
// transpose in registers.  Range guard as in v4.
#pragma once
#include <type_traits>

template <int BM, int BN, int WM, int WN, int BMODE, int STAGES, bool HI_ONLY GG_ABL_PARAM>
__global__ void __launch_bounds__(256, (STAGES == 2 && BM * BN <= 128 * 64) ? 3 : 1)   // 50 KB of LDS: 3 workgroups per CU,
                                                                                        // so hold hipcc to 3 waves' worth of registers
gather_gemm_f32_v5(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue, int nQueues,
                   unsigned int* __restrict__ rangeFlag)
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

    // [STAGES operand buffers][rowC | rowR offsets of the tile's BM rows][next tile id] -- ONE __shared__ object:
    // a second one makes hipcc drain the LDS-DMA queue (vmcnt(0)) in front of every fragment read
    static_assert(STAGES >= 2 && STAGES <= 4, "2..4 operand buffers");
    static_assert(BMODE == VSR_BMODE_NK || STAGES == 2, "the register-transposed KN operand is double-buffered");
    __shared__ __attribute__((aligned(16))) float smem[STAGES * BUF_FLOATS + 2 * BM + 4];
    int* rowTab = reinterpret_cast<int*>(smem + STAGES * BUF_FLOATS);
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + STAGES * BUF_FLOATS + 2 * BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int s_r = tid >> 3, s_q = tid & 7;            // [rows][32] images: row-in-pass, slot
    const int srcSwz = ((s_q ^ ((s_r >> 1) & 7)) << 2); // float offset of the 16-byte group this lane fetches
    // fragment of MFMA step st (k = 16st .. 16st+15): lane (l31, hi) reads piece 2st+hi (hi halves) and 4+2st+hi (lo)
    int rdHi[2], rdLo[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        rdHi[st] = (((2 * st + hi) ^ ((l31 >> 1) & 7)) << 4);
        rdLo[st] = (((4 + 2 * st + hi) ^ ((l31 >> 1) & 7)) << 4);
    }

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

    for (;;) {
        const int bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();
        if (bid >= totalTiles) break;
        if (tid == 0) *nextTile = fetchTile();

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
            const bool hasR = (P->R != nullptr) && (splitK == 1);
#pragma unroll
            for (int i = tid; i < 2 * BM; i += 256)
                rowTab[i] = i < BM ? rowCt[tm * BM + i] : (hasR ? rowRt[tm * BM + i - BM] : 0);
        }
        int aoff[A_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aoff[it] = rowA[tm * BM + s_r + 32 * it] + srcSwz;
        int boff[B_IT];
        // KN (P.V): a thread owns ONE column n of the B tile and gathers 8 k-values of it per group as
        // (hi, lo) halves (2-byte loads, lanes of a wave = 64 consecutive n of one k), then writes one 16-byte
        // hi piece + one lo piece of LDS row n: the transpose happens in registers (cf. v4).
        constexpr int KN_PAIRS = (BN * 4) / 256 > 0 ? (BN * 4) / 256 : 1;
        constexpr int KN_KGSTEP = 256 / BN;
        const int kn_n = tid % BN, kn_kg0 = __builtin_amdgcn_readfirstlane(tid / BN);
        int bcolKN = 0;                                  // ushort index of this column's hi half inside a row chunk
        int vrowB = 0, vrowBn = 0, rowBaseChunk = kcBeg;
        auto fetchRows = [&](int baseChunk, int& v) {
            const int idx = baseChunk * VSR_GG_KC + lane < P->K ? baseChunk * VSR_GG_KC + lane : P->K - 1;
            v = rowB[idx];
        };
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) boff[it] = rowB[tn * BN + s_r + 32 * it] + srcSwz;
        } else {
            bcolKN = colB[(tn * BN) / VSR_GG_KC + (kn_n >> 5)] * 2 + (kn_n & 31);
            fetchRows(rowBaseChunk, vrowB);
            fetchRows(rowBaseChunk + 2, vrowBn);
        }

        // chunk-offset tables: lane i holds entry (base + i); refreshed every 64 chunks
        int colBase = kcBeg;                                   // chunk index held by lane 0 of vcolA/vcolB
        auto fetchCols = [&](int base, int& va, int& vb) {
            const int idx = base + lane < nchunksTotal ? base + lane : nchunksTotal - 1;
            va = colA[idx];
            if constexpr (BMODE == VSR_BMODE_NK) vb = colB[idx]; else vb = 0;
        };
        int vcolA = 0, vcolB = 0, vcolAn = 0, vcolBn = 0;
        if constexpr (BMODE == VSR_BMODE_KN) {                 // NK fetches its tables per super-block (below)
            fetchCols(colBase, vcolA, vcolB);
            fetchCols(colBase + 64, vcolAn, vcolBn);
        }

        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        typedef const unsigned short __attribute__((address_space(1)))* gcu16;
        const gcu16 B16 = (gcu16)P->B;
        unsigned short kh[KN_PAIRS][8], kl[KN_PAIRS][8];  // KN: halves of 8 k-values of this thread's column
        // LDS-DMA of chunk kc: A always, B when it is k-contiguous (NK)
        auto dma_tile = [&](int buf, int ca, int cb) {      // ca / cb: chunk offsets (wave-uniform) of A / B
            float* As = smem + buf * BUF_FLOATS;
            float* Bs = As + AS_FLOATS;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                glds16(A + (aoff[it] + ca), (lds_vptr)(As + (wave * 8 + 32 * it) * 32));
            if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
                for (int it = 0; it < B_IT; ++it)
                    glds16(B + (boff[it] + cb), (lds_vptr)(Bs + (wave * 8 + 32 * it) * 32));
            }
        };
        auto load_B_KN = [&](int kc) {
            if (kc - rowBaseChunk >= 2) {                // the VGPR holds the row offsets of two chunks
                rowBaseChunk += 2;
                vrowB = vrowBn;
                fetchRows(rowBaseChunk + 2, vrowBn);
            }
            const int kb = (kc - rowBaseChunk) * VSR_GG_KC;
#pragma unroll
            for (int p = 0; p < KN_PAIRS; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ro = __builtin_amdgcn_readlane(vrowB, kb + 8 * (kn_kg0 + KN_KGSTEP * p) + j);
                    kh[p][j] = B16[2 * ro + bcolKN];
                    kl[p][j] = B16[2 * ro + bcolKN + 32];
                }
        };
        auto store_B_KN = [&](int buf) {
            char* Bs = reinterpret_cast<char*>(smem + buf * BUF_FLOATS + AS_FLOATS);
            const int sw = (kn_n >> 1) & 7;
#pragma unroll
            for (int p = 0; p < KN_PAIRS; ++p) {
                const int kg = kn_kg0 + KN_KGSTEP * p;
                f16x8 h, l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    h[j] = __builtin_bit_cast(_Float16, kh[p][j]);
                    l[j] = __builtin_bit_cast(_Float16, kl[p][j]);
                }
                *reinterpret_cast<f16x8*>(Bs + kn_n * 128 + ((kg ^ sw) << 4)) = h;
                *reinterpret_cast<f16x8*>(Bs + kn_n * 128 + (((4 + kg) ^ sw) << 4)) = l;
            }
        };
        auto compute_step = [&](int buf, int st) {
            const char* As = reinterpret_cast<const char*>(smem + buf * BUF_FLOATS);
            const char* Bs = As + AS_FLOATS * 4;
            f16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const char* row = As + (wm * WTM + mi * 32 + l31) * 128;
                ah[mi] = *reinterpret_cast<const f16x8*>(row + rdHi[st]);
                al[mi] = *reinterpret_cast<const f16x8*>(row + rdLo[st]);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const char* row = Bs + (wn * WTN + ni * 32 + l31) * 128;
                bh[ni] = *reinterpret_cast<const f16x8*>(row + rdHi[st]);
                bl[ni] = *reinterpret_cast<const f16x8*>(row + rdLo[st]);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        };

        // fp16 mode (HI_ONLY, kernel variant 6): operands are the hi halves alone -- one MFMA per product instead of
        // three, half the fragment reads; the tensors keep the split format (producers still write lo, the
        // residual add and the elementwise kernels still use it)
        auto compute_step_hi = [&](int buf, int st) {
            const char* As = reinterpret_cast<const char*>(smem + buf * BUF_FLOATS);
            const char* Bs = As + AS_FLOATS * 4;
            f16x8 ah[MI], bh[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                ah[mi] = *reinterpret_cast<const f16x8*>(As + (wm * WTM + mi * 32 + l31) * 128 + rdHi[st]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bh[ni] = *reinterpret_cast<const f16x8*>(Bs + (wn * WTN + ni * 32 + l31) * 128 + rdHi[st]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        };

        auto compute_chunk = [&](int buf) {
            if constexpr (HI_ONLY) {
                compute_step_hi(buf, 0);
                compute_step_hi(buf, 1);
            } else {
                compute_step(buf, 0);
                compute_step(buf, 1);
            }
        };
        auto refreshCols = [&](int kc) {           // chunk kc is about to be fetched: lane (kc - colBase) must hold it
            if (kc - colBase >= 64) {
                colBase += 64;
                vcolA = vcolAn; vcolB = vcolBn;
                fetchCols(colBase + 64, vcolAn, vcolBn);
            }
        };

        if constexpr (BMODE == VSR_BMODE_NK) {
            // Operand pipeline, D = STAGES-1 chunks deep.  At 3 f16 MFMAs per product a chunk is ~400 MFMA cycles per
            // wave, far less than an L2 round trip, so ONE chunk in flight per workgroup (the fp32 kernels' double
            // buffer) leaves the matrix cores waiting on memory latency.  Each wave waits only for ITS pieces of
            // chunk kc (counted vmcnt: the D-1 younger chunks stay in flight), the barrier then publishes the chunk
            // and retires the buffer of chunk kc-1, which the DMA of chunk kc+D overwrites.  Raw s_barrier, not
            // __syncthreads(): its fence would drain the DMA queue (vmcnt(0)).
            // The loop body must not contain an ordinary (VGPR-destination) load: hipcc waits vmcnt(0) for those,
            // which drains the pipeline every iteration.  The chunk-offset tables are therefore fetched per
            // super-block of 128 chunks (2 VGPRs per table, lane i = chunk base+i / base+64+i); every conv / QKV /
            // QK^T range of this network fits one super-block.
            constexpr int D = STAGES - 1;
            constexpr int PIECES = A_IT + B_IT;    // LDS-DMA instructions per wave and chunk
            static_assert(PIECES * (D - 1) <= 63, "vmcnt is 6 bits");
            for (int sb = kcBeg; sb < kcEnd; sb += 128) {
                const int sbEnd = sb + 128 < kcEnd ? sb + 128 : kcEnd;
                const int i0 = sb + lane < nchunksTotal ? sb + lane : nchunksTotal - 1;
                const int i1 = sb + 64 + lane < nchunksTotal ? sb + 64 + lane : nchunksTotal - 1;
                const int ca0 = colA[i0], ca1 = colA[i1], cb0 = colB[i0], cb1 = colB[i1];
                // a use in front of the first DMA: the compiler's wait for these four loads lands here, not
                // (as vmcnt(0)) behind the prologue's DMAs
                asm volatile("" ::"v"(ca0), "v"(ca1), "v"(cb0), "v"(cb1));
                auto issue = [&](int kc, int buf) {
                    const int i = kc - sb;
                    const int ca = i < 64 ? __builtin_amdgcn_readlane(ca0, i) : __builtin_amdgcn_readlane(ca1, i - 64);
                    const int cb = i < 64 ? __builtin_amdgcn_readlane(cb0, i) : __builtin_amdgcn_readlane(cb1, i - 64);
                    dma_tile(buf, ca, cb);
                };
                if (sb != kcBeg) {                 // buffers of the previous super-block are still being read
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (sb + d < sbEnd) issue(sb + d, d);
                int cur = 0, nxt = D % STAGES;
                for (int kc = sb; kc < sbEnd; ++kc) {
                    const int ahead = sbEnd - 1 - kc;  // younger chunks already issued: min(ahead, D-1)
                    if (D >= 3 && ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (D >= 3 ? 2 : 0)) : "memory");
                    else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (D >= 2 ? 1 : 0)) : "memory");
                    else                           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my fragment reads of chunk kc-1 have left the LDS
                    __builtin_amdgcn_s_barrier();
                    if (kc + D < sbEnd) issue(kc + D, nxt);
                    compute_chunk(cur);
                    cur = cur + 1 == STAGES ? 0 : cur + 1;
                    nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
                }
            }
            // the tile-end barrier below keeps the next tile's DMA off the buffers still being read
        } else if (kcBeg < kcEnd) {
            dma_tile(0, __builtin_amdgcn_readlane(vcolA, kcBeg - colBase), 0);
            load_B_KN(kcBeg);
            store_B_KN(0);
            __syncthreads();                       // drains the DMA (vmcnt(0)) and publishes buffer 0
            int cur = 0;
            for (int kc = kcBeg; kc < kcEnd; ++kc) {
                const bool more = kc + 1 < kcEnd;
                if (more) {
                    refreshCols(kc + 1);
                    dma_tile(cur ^ 1, __builtin_amdgcn_readlane(vcolA, kc + 1 - colBase), 0);   // buffer last read in iteration kc-1
                    load_B_KN(kc + 1);
                }
                compute_chunk(cur);
                if (more) store_B_KN(cur ^ 1);
                __syncthreads();                   // vmcnt(0) + barrier: chunk kc+1 landed, chunk kc retired
                cur ^= 1;
            }
        }
        __syncthreads();                           // rowTab visible even when the k range is empty; LDS-DMA queue empty

        // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
        // Row offsets come from LDS, the residual reads of 16 rows are issued back to back.
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;   // relu(act(..) + R): residual blocks of RAFT
        const bool cSplit = (P->act & VSR_ACT_OUT_SPLIT) != 0;   // output in split format (a later GEMM operand)
        const float vmax = cSplit ? 65504.f : 3.0e38f;           // a split-format value must fit its fp16 hi half
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
                                const int e = 2 * (rr[r] + ccol[ni] - l31) + l31;     // ushort index of the hi half
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
                        nonFinite |= !(__builtin_fabsf(v) <= vmax);      // also catches NaN
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
