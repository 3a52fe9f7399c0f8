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
// 32: per-workgroup cycle sums of wave 0's chunk phases (operand DMA issue | fragment reads + MFMAs | vmcnt(0) wait | barrier wait)
#define GG_PH(k)                                                                       \
    if constexpr (GG_ABL(32)) {                                                        \
        const unsigned long long now_ = __builtin_readcyclecounter();                  \
        ph_[k] += now_ - tl_;                                                          \
        tl_ = now_;                                                                    \
    }
#else
#define GG_STAMP()
#define GG_PH(k)
#endif

// (second launch bound = waves per SIMD: three workgroups of the 128x64 tile share a CU, which needs <= 168 registers)
template <int BM, int BN, int WM, int WN, int BMODE GG_ABL_PARAM>
__global__ void __launch_bounds__(256, (BM + BN) * 32 * 4 * 2 <= 50 * 1024 ? 3 : 2)
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
    unsigned long long ph_[5] = {0, 0, 0, 0, 0}, tl_ = 0;
    const unsigned long long kc0_ = __builtin_readcyclecounter(), kr0_ = __builtin_amdgcn_s_memrealtime();   // shader clock vs 100 MHz
#endif

    for (;;) {
        const int bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();
        if (bid >= totalTiles) break;
        if (tid == 0) *nextTile = fetchTile();
        GG_STAMP()   // tile start
        if constexpr (!GG_ABL(1024)) __builtin_amdgcn_s_setprio(3);

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
        // LDS-DMA of chunk kc into buffer buf (destination = wave-uniform base + lane*16).  Round 3: when every row offset of
        // this wave is in [0, 2^30) floats (always, except for tensors beyond 4 GB) a piece is addressed as
        // (scalar base = operand + chunk offset) + (unsigned 32-bit byte offset of the lane's row, constant over the tile):
        // the saddr form of global_load_lds, no vector ALU work per piece.  (Per piece the 64-bit vector address cost a
        // v_add, a v_ashrrev and a v_lshl_add_u64, each of which waits behind the co-resident waves' MFMAs.)
        bool narrow = true;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) narrow = narrow && ((unsigned)aoff[it] < (1u << 30));
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) narrow = narrow && ((unsigned)boff[it] < (1u << 30));
        } else {
            narrow = false;                     // KN: the B row offsets change with every chunk; keep the vector form
        }
        narrow = __all(narrow) != 0;
        unsigned aoffB[A_IT], boffB[B_IT];          // byte offsets of the lane's rows
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aoffB[it] = (unsigned)aoff[it] << 2;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) boffB[it] = (unsigned)boff[it] << 2;
        auto dma_tile = [&](int kc, auto bufTag) {
            constexpr int buf = decltype(bufTag)::value;
            float* As = smem + buf * BUF_FLOATS;
            float* Bs = As + AS_FLOATS;
            const int ca = __builtin_amdgcn_readlane(vcolA, kc - colBase);
            if (narrow) {
                const char __attribute__((address_space(1)))* baseA = (const char __attribute__((address_space(1)))*)A + (long long)ca * 4;
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    asm volatile("" : "+v"(aoffB[it]));       // keeps the zero-extension next to the load, where hipcc folds it into the saddr form
                    glds16((gcf32)(baseA + aoffB[it]), (lds_vptr)(As + (wave * 8 + 32 * it) * 32));
                }
                if constexpr (BMODE == VSR_BMODE_NK) {
                    const int cb = __builtin_amdgcn_readlane(vcolB, kc - colBase);
                    const char __attribute__((address_space(1)))* baseB = (const char __attribute__((address_space(1)))*)B + (long long)cb * 4;
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) {
                        asm volatile("" : "+v"(boffB[it]));
                        glds16((gcf32)(baseB + boffB[it]), (lds_vptr)(Bs + (wave * 8 + 32 * it) * 32));
                    }
                }
                return;
            }
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
        // Fragment reads of group g (four k-steps); the MFMAs take the WEIGHT fragment as their first operand and the
        // ACTIVATION fragment as their second, i.e. they accumulate the transposed tile: lane l31 of an accumulator owns output
        // ROW l31 and its 16 registers are columns (r&3) + 8*(r>>2) + 4*hi -- four runs of four consecutive columns, so the
        // epilogue moves 16 bytes per lane and instruction (round 3; before, a lane owned a column and stored 16 single floats).
        auto read_group = [&](auto bufTag, int g, f32x4 (&af)[MI], f32x4 (&bf)[NI]) {
            constexpr int buf = decltype(bufTag)::value;             // compile-time buffer: the reads fold it into their offset field
            const float* As = smem + buf * BUF_FLOATS;
            const float* Bs = As + AS_FLOATS;
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
        };
        auto mfma_group = [&](const f32x4 (&af)[MI], const f32x4 (&bf)[NI]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[ni][j], af[mi][j], acc[mi][ni], 0, 0, 0);
        };
        // one chunk: the reads of group g+1 are in flight while the MFMAs of group g issue (two fragment sets), so a wave
        // stalls on LDS latency once per chunk instead of four times -- with three waves per SIMD the matrix pipe idles
        // whenever all three wait at once (timeline in profiles/r03_v3_probe.log: 6975 cycles per chunk against 6144 pipe-bound)
        auto compute_chunk = [&](auto buf) {
            f32x4 af0[MI], bf0[NI], af1[MI], bf1[NI];
            // (sched_barrier: hipcc otherwise sinks every read group back in front of its own MFMAs)
            read_group(buf, 0, af0, bf0);
            read_group(buf, 1, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            read_group(buf, 2, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            read_group(buf, 3, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(af0, bf0);
            mfma_group(af1, bf1);
        };

        if (kcBeg < kcEnd) {
            if constexpr (BMODE == VSR_BMODE_KN) {
                load_rowB_KN(kcBeg, boff);
                if (kcBeg + 1 < kcEnd) load_rowB_KN(kcBeg + 1, boffNext);
            }
            using B0_ = std::integral_constant<int, 0>;
            using B1_ = std::integral_constant<int, 1>;
            dma_tile(kcBeg, B0_{});
            __syncthreads();                       // drains the DMA (vmcnt(0)) and publishes buffer 0
            if constexpr (!GG_ABL(1024) || !GG_ABL(512)) __builtin_amdgcn_s_setprio(0);
            GG_STAMP()   // prologue done
            const int kcEndU = __builtin_amdgcn_readfirstlane(kcEnd);      // scalar loop control
            // one chunk out of buffer `cur` while the DMA of the next one fills the other buffer
            auto chunk = [&](int kc, auto cur, auto nxt) {
#ifdef GG_ABLATE
                if constexpr (GG_ABL(32)) { tl_ = __builtin_readcyclecounter(); ph_[4] += 1; }
#endif
                if constexpr (!GG_ABL(256)) __builtin_amdgcn_s_setprio(3);    // the few non-MFMA instructions of a chunk go first: they issue in the shadow of the other waves' MFMAs
                if (kc + 1 < kcEndU) {
                    if (kc + 1 - colBase >= 64) {  // next 64 table entries become current
                        colBase += 64;
                        vcolA = vcolAn; vcolB = vcolBn;
                        fetchCols(colBase + 64, vcolAn, vcolBn);
                    }
                    if constexpr (BMODE == VSR_BMODE_KN) {
#pragma unroll
                        for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                        if (kc + 2 < kcEndU) load_rowB_KN(kc + 2, boffNext);
                    }
                    if constexpr (!GG_ABL(2)) dma_tile(kc + 1, nxt);         // buffer last read in iteration kc-1, fenced by its barrier
                }
                if constexpr (!GG_ABL(256)) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); }
                GG_PH(0)
                compute_chunk(cur);
                GG_PH(1)
#ifdef GG_ABLATE
                if constexpr (GG_ABL(32)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
                GG_PH(2)
                if constexpr (!GG_ABL(1)) __syncthreads();                   // vmcnt(0) + barrier: chunk kc+1 landed, chunk kc retired
                GG_PH(3)
                GG_STAMP()   // chunk done
            };
            int kc = __builtin_amdgcn_readfirstlane(kcBeg);
            for (; kc + 1 < kcEndU; kc += 2) {
                chunk(kc, B0_{}, B1_{});
                chunk(kc + 1, B1_{}, B0_{});
            }
            if (kc < kcEndU) chunk(kc, B0_{}, B1_{});
        }

        if constexpr (!GG_ABL(512)) __builtin_amdgcn_s_setprio(3);     // epilogue, queue and tables of the next tile at priority: +2 % (profiles/r03_v3_probe.log)
        // ---- epilogue.  Transposed accumulators (see read_group): lane l31 owns output row l31 of its 32x32 block, register
        // r is column (r&3) + 8*(r>>2) + 4*hi.  One row offset per lane, and interior tiles whose output (and residual) rows are
        // 16-byte aligned move float4s: 4 stores per accumulator instead of 16 (the store tail is issue-bound, not bandwidth-bound).
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;   // relu(act(..) + R): residual blocks of RAFT
        const bool partial = (splitK > 1);
        const bool rowMax = (P->act & VSR_ACT_ROW_MAX) != 0;       // R is the row-maximum array of the scores, not a residual
        const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
        const gcf32 R = (partial || rowMax) ? (gcf32) nullptr : (gcf32)P->R;
        const cci32 colC = (cci32)P->colC;
        const gf32 C = (gf32)(P->C + (partial ? (int64_t)split * P->splitStride : (int64_t)0));
        int ccol[NI];            // offset of column (32-block start + 4*hi) of this lane
        int ncol[NI];            // its global column index
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n0 = tn * BN + wn * WTN + ni * 32;
            ccol[ni] = colC[n0 / VSR_GG_KC] + 4 * hi;
            ncol[ni] = n0 + 4 * hi;
        }
        const bool fullTile = (tm * BM + BM <= M) && (tn * BN + BN <= N);
        auto activate = [&](float v) {
            if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
            else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
            return v;
        };
        // float4 epilogue of an interior tile.  ACTK: the activation as a compile-time constant (0 none, 1 LeakyReLU 0.2, 2 ReLU,
        // 3 LeakyReLU 0.1), -1 = decided per element at run time (POST_RELU and anything else).  Every bias and residual load of
        // the wave is issued before the first value is touched: one exposed memory latency per tile (the first version loaded the
        // bias inside the store loop and paid eight serialised round trips, 50 k cycles per tile in profiles/r03_v3_probe.log).
        auto epilogue_vec = [&](auto resTag, auto actTag) {
            constexpr bool HASR = decltype(resTag)::value;
            constexpr int ACTK = decltype(actTag)::value;
            typedef const f32x4 __attribute__((address_space(1)))* gv4;
            f32x4 bq[NI][4];
            f32x4 rv[HASR ? MI : 1][NI][4];
            int rc[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (bias != nullptr) bq[ni][q] = *reinterpret_cast<gv4>(bias + (ncol[ni] + 8 * q));
                    else bq[ni][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * WTM + mi * 32 + l31;
                rc[mi] = rowTab[row];
                if constexpr (HASR) {
                    const int rr = rowTab[BM + row];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[mi][ni][q] = *reinterpret_cast<gv4>(R + (rr + ccol[ni] + 8 * q));
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[mi][ni][4 * q + e] * alpha + bq[ni][q][e];
                            if constexpr (ACTK == 1) v = fmaxf(v, 0.2f * v);          // == v > 0 ? v : 0.2 v
                            else if constexpr (ACTK == 2) v = fmaxf(v, 0.f);
                            else if constexpr (ACTK == 3) v = fmaxf(v, 0.1f * v);
                            else if constexpr (ACTK < 0) v = activate(v);
                            if constexpr (HASR) {
                                v += rv[mi][ni][q][e];
                                if constexpr (ACTK < 0) { if (postRelu) v = fmaxf(v, 0.f); }
                            }
                            o[e] = v;
                        }
                        *reinterpret_cast<f32x4 __attribute__((address_space(1)))*>(C + (rc[mi] + ccol[ni] + 8 * q)) = o;
                    }
        };
        // border tiles and unaligned outputs: one float at a time, predicated
        auto epilogue_scalar = [&](auto resTag) {
            constexpr bool HASR = decltype(resTag)::value;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * WTM + mi * 32 + l31;
                const int rc = rowTab[row];
                const int rr = HASR ? rowTab[BM + row] : 0;
                const bool mok = (tm * BM + row) < M;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cofs = (r & 3) + 8 * (r >> 2);
                        const bool ok = mok && (ncol[ni] + cofs) < N;
                        float v = acc[mi][ni][r] * alpha + ((bias != nullptr && ok) ? bias[ncol[ni] + cofs] : 0.f);
                        v = activate(v);
                        if constexpr (HASR) { if (ok) v += R[rr + ccol[ni] + cofs]; if (postRelu) v = fmaxf(v, 0.f); }
                        if (ok) C[rc + ccol[ni] + cofs] = v;
                    }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        // float4 path: an interior tile whose row / column offsets keep 16-byte alignment for every lane of this wave
        bool vec = fullTile && ((reinterpret_cast<uintptr_t>(P->C) | (uintptr_t)(partial ? P->splitStride * 4 : 0)) & 15) == 0 &&
                   (bias == nullptr || (reinterpret_cast<uintptr_t>(P->bias) & 15) == 0) && (R == nullptr || (reinterpret_cast<uintptr_t>(P->R) & 15) == 0);
        {
            int low = 0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) low |= ccol[ni];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                low |= rowTab[wm * WTM + mi * 32 + l31];
                if (R != nullptr) low |= rowTab[BM + wm * WTM + mi * 32 + l31];
            }
            vec = vec && __all((low & 3) == 0);
        }
        if constexpr (GG_ABL(128)) {           // ablation: no epilogue (one store keeps the accumulators alive)
            if (acc[0][0][0] == 12345.678f) C[0] = acc[MI - 1][NI - 1][15];
        } else
        if (vec) {
            using IC = std::integral_constant<int, -1>;
            const int ak = postRelu ? -1 : act;
            if (R != nullptr) {
                if (ak == VSR_ACT_NONE) epilogue_vec(T_{}, std::integral_constant<int, 0>{});
                else if (ak == VSR_ACT_LRELU02) epilogue_vec(T_{}, std::integral_constant<int, 1>{});
                else epilogue_vec(T_{}, IC{});
            } else {
                if (ak == VSR_ACT_NONE) epilogue_vec(F_{}, std::integral_constant<int, 0>{});
                else if (ak == VSR_ACT_LRELU02) epilogue_vec(F_{}, std::integral_constant<int, 1>{});
                else if (ak == VSR_ACT_RELU) epilogue_vec(F_{}, std::integral_constant<int, 2>{});
                else epilogue_vec(F_{}, IC{});
            }
        } else {
            if (R != nullptr) epilogue_scalar(T_{}); else epilogue_scalar(F_{});
        }
        if (rowMax) {
            // VSR_ACT_ROW_MAX (the QK^T of a fused attention): the largest score of every row of this tile joins the row's running
            // maximum.  A lane holds 16 * NI columns of one row per mi, its partner lane ^ 32 the other half of the row's 32-block;
            // the waves side by side meet in LDS (the operand buffers are idle: the main loop ended on a barrier), then one atomic
            // per row in a coalesced burst.  The P.V kernel subtracts it (VSR_ACT_A_EXP).
            float* scr = smem;                                      // [BM][WN]
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float mx = -INFINITY;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = ncol[ni] + (r & 3) + 8 * (r >> 2);
                        const float bvv = (bias != nullptr && n < N) ? bias[n] : 0.f;
                        if (n < N) mx = fmaxf(mx, acc[mi][ni][r] * alpha + bvv);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (hi == 0) scr[(wm * WTM + mi * 32 + l31) * WN + wn] = mx;
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
#ifdef GG_ABLATE
    if constexpr (GG_ABL(32)) {
        if (tid == 0 && blockIdx.x < 4096)
            for (int k = 0; k < 5; ++k) gg_dbg[blockIdx.x * 8 + k] = ph_[k];
        if (tid == 0 && blockIdx.x < 4096) {
            gg_dbg[blockIdx.x * 8 + 5] = __builtin_readcyclecounter() - kc0_;
            gg_dbg[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime() - kr0_;
        }
    }
#endif
}
