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

// Round 3: WM x WN = 8 waves (512 threads) runs the 256x128 tile, each wave a 64x64 block: twice the FLOPs per operand byte of the
// 128x64 tile at the same bytes in flight per CU (one workgroup of three 48 KB stages instead of two of three 24 KB stages) -- the
// kernel is bound by the latency of its operand pieces, not by bandwidth or by the matrix cores (profiles/r02_f16_ablation_*.log).
template <int BM, int BN, int WM, int WN, int STAGES GG_ABL_PARAM>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN == 8) ? 2 : ((STAGES * (BM + BN) * 128 + 2 * BM * 4 <= 78 * 1024) ? 2 : 1))
gather_gemm_f16_v6(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue, int nQueues,
                   unsigned int* __restrict__ rangeFlag)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int NT = WM * WN * 64;                            // threads of the workgroup
    constexpr int RP = NT / 8;                                  // operand rows one pass of LDS-DMA pieces covers (8 lanes per 128-byte row)
    constexpr int A_IT = BM / RP, B_IT = BN / RP;
    constexpr int AS_FLOATS = BM * 32, BS_FLOATS = BN * 32;     // [rows][128 bytes]
    constexpr int BUF_FLOATS = AS_FLOATS + BS_FLOATS;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    static_assert(BM % RP == 0 && BN % RP == 0, "tile rows in whole DMA passes");
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

    // Tile queue (see gather_gemm_v5.h).  The home range's counter is bumped at the START of a tile and its answer is only looked
    // at after the main loop (the round trip of the atomic hides behind the tile); once the home range is empty the workgroup
    // steals, but PEEKS at a foreign counter with a plain load first -- at the tail of a launch every workgroup scans the other
    // ranges, and hundreds of same-address atomics on exhausted counters serialise in L2 (tens of microseconds).
    int qFirst = 0;
    const int home = blockIdx.x % nQueues;
    auto rangeOf = [&](int x, int& lo, int& hi_) {
        lo = (int)(((long long)totalTiles * x) / nQueues);
        hi_ = (int)(((long long)totalTiles * (x + 1)) / nQueues);
    };
    auto stealTile = [&]() -> int {
        for (; qFirst < nQueues; ++qFirst) {
            const int x = (home + qFirst) % nQueues;
            int lo, hi_;
            rangeOf(x, lo, hi_);
            if (lo >= hi_) continue;
            if (qFirst > 0 && (int)__hip_atomic_load(queue + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= hi_ - lo) continue;
            const int i = lo + (int)atomicAdd(queue + x, 1u);
            if (i < hi_) return i;
        }
        return totalTiles;
    };
    if (tid == 0) *nextTile = stealTile();
    __syncthreads();

#ifdef GG_ABLATE
    int tr_ = 0;                                     // 256: wall-clock stamps of wave 0 (100 MHz), 4 per tile
#define V6_STAMP(drain)                                                                                        \
    if constexpr (GG_ABL(256)) {                                                                               \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                 \
        if (tid == 0 && blockIdx.x < 1024 && tr_ < 256) gg_trace[blockIdx.x * 256 + tr_] = wall_clock64();     \
        ++tr_;                                                                                                 \
    }
#else
#define V6_STAMP(drain)
#endif
    for (;;) {
        const int bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();
        if (bid >= totalTiles) break;
        V6_STAMP(0)
        unsigned int pend = 0;
        if (tid == 0 && qFirst == 0) pend = atomicAdd(queue + home, 1u);

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
            for (int i = tid; i < 2 * BM; i += NT)
                rowTab[i] = i < BM ? rowCt[tm * BM + i] : (hasR ? rowRt[tm * BM + i - BM] : 0);
        }
        int aoff[A_IT], boff[B_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aoff[it] = rowA[tm * BM + s_r + RP * it] + srcSwz;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) boff[it] = rowB[tn * BN + s_r + RP * it] + srcSwz;

        V6_STAMP(1)
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // LDS-DMA of one pair of chunks: ca0 / ca1 (cb0 / cb1) = wave-uniform chunk offsets of the pair in A (B)
        // byte offsets of the lane's operand rows; narrow: all of them in [0, 2^30) floats (always, except for tensors beyond 4 GB)
        unsigned aoffB[A_IT], boffB[B_IT];
        bool narrow = true;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) { narrow = narrow && ((unsigned)aoff[it] < (1u << 30)); aoffB[it] = (unsigned)aoff[it] << 2; }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) { narrow = narrow && ((unsigned)boff[it] < (1u << 30)); boffB[it] = (unsigned)boff[it] << 2; }
        narrow = __all(narrow) != 0;
        auto dma_pair = [&](auto bufTag, int ca0, int ca1, int cb0, int cb1) __attribute__((always_inline)) {
            constexpr int buf = decltype(bufTag)::value;
            float* As = smem + buf * BUF_FLOATS;
            float* Bs = As + AS_FLOATS;
            if constexpr (GG_ABL(2)) { if (buf >= 0) return; }     // ablation: no operand fetch at all
            if constexpr (GG_ABL(8)) { ca0 = ca1 = cb0 = cb1 = 0; }                                          // 8: one hot chunk
            if (narrow) {
                // round 3: the saddr form of global_load_lds -- scalar base = operand + the SMALLER chunk offset of the pair, per-lane
                // unsigned byte offset = row offset + what its chunk lies above that (one select per operand and pair, one add per piece)
                typedef const char __attribute__((address_space(1)))* gcc8;
                const int ma = ca0 < ca1 ? ca0 : ca1, mb = cb0 < cb1 ? cb0 : cb1;
                const gcc8 baseA = (gcc8)A + (long long)ma * 4, baseB = (gcc8)B + (long long)mb * 4;
                const unsigned da = (unsigned)((second ? ca1 : ca0) - ma) << 2, db = (unsigned)((second ? cb1 : cb0) - mb) << 2;
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    unsigned vo = aoffB[it] + da;
                    asm volatile("" : "+v"(vo));
                    glds16((gcf32)(baseA + vo), (lds_vptr)(As + (wave * 8 + RP * it) * 32));
                }
#pragma unroll
                for (int it = 0; it < B_IT; ++it) {
                    unsigned vo = boffB[it] + db;
                    asm volatile("" : "+v"(vo));
                    glds16((gcf32)(baseB + vo), (lds_vptr)(Bs + (wave * 8 + RP * it) * 32));
                }
                return;
            }
            const int ca = second ? ca1 : ca0, cb = second ? cb1 : cb0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                glds16(A + (aoff[it] + ca), (lds_vptr)(As + (wave * 8 + RP * it) * 32));
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                glds16(B + (boff[it] + cb), (lds_vptr)(Bs + (wave * 8 + RP * it) * 32));
        };
        struct Frag { f16x8 a[MI], b[NI]; };
        auto load_frag = [&](auto bufTag, int st, Frag& f) __attribute__((always_inline)) {
            constexpr int buf = decltype(bufTag)::value;        // compile-time stage: the reads fold it into their offset field
            const char* As = reinterpret_cast<const char*>(smem + buf * BUF_FLOATS);
            const char* Bs = As + AS_FLOATS * 4;
            if constexpr (GG_ABL(16)) {                             // ablation: MFMA on register operands (no fragment reads)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < 8; ++j) f.a[mi][j] = (_Float16)(float)(lane + st + mi + j);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 8; ++j) f.b[ni][j] = (_Float16)(float)(lane - st - ni - j);
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    f.a[mi] = *reinterpret_cast<const f16x8*>(As + (wm * WTM + mi * 32 + l31) * 128 + rd[st]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    f.b[ni] = *reinterpret_cast<const f16x8*>(Bs + (wn * WTN + ni * 32 + l31) * 128 + rd[st]);
            }
        };
        auto mfma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[ni], f.a[mi], acc[mi][ni], 0, 0, 0);   // transposed tile: a lane owns an output ROW (epilogue below)
        };
        auto compute_step = [&](auto buf, int st) __attribute__((always_inline)) {
            if constexpr (GG_ABL(4)) return;                        // ablation: no fragment reads, no MFMA
            Frag f;
            load_frag(buf, st, f);
            mfma_frag(f);
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
            auto pick = [&](int v0, int v1, int i) {           // both halves read, scalar select: no branch in the loop
                const int a = __builtin_amdgcn_readlane(v0, i & 63), b = __builtin_amdgcn_readlane(v1, i & 63);
                return i < 64 ? a : b;
            };
            auto issue = [&](int kc, auto bufTag) __attribute__((always_inline)) {     // pair (kc, kc + 1); a lone last chunk is fetched twice
                const int i = kc - sb, j = kc + 1 < sbEnd ? i + 1 : i;
                dma_pair(bufTag, pick(ca0v, ca1v, i), pick(ca0v, ca1v, j), pick(cb0v, cb1v, i), pick(cb0v, cb1v, j));
            };
            if (sb != kcBeg) {                                  // stages of the previous super-block are still being read
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            using S0_ = std::integral_constant<int, 0>;
            using S1_ = std::integral_constant<int, 1>;
            using S2_ = std::integral_constant<int, 2 % STAGES>;
            using S3_ = std::integral_constant<int, 3 % STAGES>;
            if (sb < sbEnd) issue(sb, S0_{});
            if (D >= 2 && sb + 2 < sbEnd) issue(sb + 2, S1_{});
            if (D >= 3 && sb + 4 < sbEnd) issue(sb + 4, S2_{});
            // one pair out of stage `cur` while the DMA of pair kc + 2 D goes into stage `nxt` (the one retired by this step's barrier)
            auto step = [&](int kc, auto cur, auto nxt) __attribute__((always_inline)) {
                const int ahead = (sbEnd - 1 - kc) >> 1;        // younger pairs already issued: min(ahead, D - 1)
                if (D >= 3 && ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (D >= 3 ? 2 : 0)) : "memory");
                else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (D >= 2 ? 1 : 0)) : "memory");
                else                           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                if (kc + 2 * D < sbEnd) issue(kc + 2 * D, nxt);
                compute_step(cur, 0);
                compute_step(cur, 1);
                if (kc + 1 < sbEnd) {
                    compute_step(cur, 2);
                    compute_step(cur, 3);
                }
            };
            // the stage a pair lives in is a compile-time constant of a loop unrolled by STAGES (no address arithmetic per read)
            int kc = sb;
            if constexpr (STAGES == 2) {
                for (; kc < sbEnd; kc += 4) {
                    step(kc, S0_{}, S1_{});
                    if (kc + 2 < sbEnd) step(kc + 2, S1_{}, S0_{});
                }
            } else if constexpr (STAGES == 3) {
                for (; kc < sbEnd; kc += 6) {
                    step(kc, S0_{}, S2_{});
                    if (kc + 2 < sbEnd) step(kc + 2, S1_{}, S0_{});
                    if (kc + 4 < sbEnd) step(kc + 4, S2_{}, S1_{});
                }
            } else {
                for (; kc < sbEnd; kc += 8) {
                    step(kc, S0_{}, S3_{});
                    if (kc + 2 < sbEnd) step(kc + 2, S1_{}, S0_{});
                    if (kc + 4 < sbEnd) step(kc + 4, S2_{}, S1_{});
                    if (kc + 6 < sbEnd) step(kc + 6, S3_{}, S2_{});
                }
            }
        }
        if (tid == 0) {                            // the next tile: the home counter's answer, else steal
            int nt = totalTiles;
            if (qFirst == 0) {
                int lo, hi_;
                rangeOf(home, lo, hi_);
                if (lo + (int)pend < hi_) nt = lo + (int)pend; else qFirst = 1;
            }
            if (nt == totalTiles) nt = stealTile();
            *nextTile = nt;
        }
        __syncthreads();                           // rowTab visible even when the k range is empty; LDS-DMA queue empty
        V6_STAMP(0)

        // ---- epilogue.  Transposed accumulators (round 3, as gather_gemm_v3.h): lane l31 owns output ROW l31 of its 32x32 block,
        // register r is column (r & 3) + 8 (r >> 2) + 4 hi -- four runs of four consecutive columns.  In split format a run is 8 bytes
        // of hi halves and, 64 bytes on, 8 bytes of lo halves: two 8-byte stores where the column-owning layout issued eight 2-byte
        // ones (the epilogue was a fifth of a tile's time, profiles/r02_f16_ablation_after.log).
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;
        const bool cSplit = (P->act & VSR_ACT_OUT_SPLIT) != 0;
        const float vmax = cSplit ? 65504.f : 3.0e38f;
        bool nonFinite = false;
        const bool partial = (splitK > 1);
        const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
        const gcf32 R = (partial || GG_ABL(64)) ? (gcf32) nullptr : (gcf32)P->R;
        const cci32 colC = (cci32)P->colC;
        const gf32 C = (gf32)(P->C + (partial ? (int64_t)split * P->splitStride : (int64_t)0));
        int cbase[NI];           // float offset of the 32-column block this lane's columns belong to
        int ncol[NI];            // global index of the lane's first column (block start + 4 hi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n0 = tn * BN + wn * WTN + ni * 32;
            cbase[ni] = colC[n0 / VSR_GG_KC];
            ncol[ni] = n0 + 4 * hi;
        }
        const bool fullTile = (tm * BM + BM <= M) && (tn * BN + BN <= N);
        auto activate = [&](float v) __attribute__((always_inline)) {
            if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
            else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
            return v;
        };
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        // interior tile, 16-byte aligned rows: vector loads / stores.  ACTK: 0 none, 1 LeakyReLU 0.2, -1 run-time
        auto epilogue_vec = [&](auto resTag, auto actTag) __attribute__((always_inline)) {
            constexpr bool HASR = decltype(resTag)::value;
            constexpr int ACTK = decltype(actTag)::value;
            typedef const f32x4 __attribute__((address_space(1)))* gv4;
            typedef const f16x4 __attribute__((address_space(1)))* gh4;
            f32x4 bq[NI][4];
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
                const int rc = rowTab[row];
                f16x4 rh[NI][4], rl[NI][4];
                if constexpr (HASR) {        // residual tensors are GEMM operands too: split format
                    const int rr = rowTab[BM + row];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const gh4 p = reinterpret_cast<gh4>(reinterpret_cast<const char __attribute__((address_space(1)))*>(R) + 4 * (long long)(rr + cbase[ni]) + 8 * hi + 16 * q);
                            rh[ni][q] = p[0];
                            rl[ni][q] = p[8];            // + 64 bytes
                        }
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[mi][ni][4 * q + e] * alpha + bq[ni][q][e];
                            if constexpr (ACTK == 1) v = v > 0.f ? v : 0.2f * v;
                            else if constexpr (ACTK < 0) v = activate(v);
                            if constexpr (HASR) {
                                v += (float)rh[ni][q][e] + (float)rl[ni][q][e];
                                if constexpr (ACTK < 0) { if (postRelu) v = fmaxf(v, 0.f); }
                            }
                            nonFinite |= !(__builtin_fabsf(v) <= vmax);
                            o[e] = v;
                        }
                        if constexpr (GG_ABL(32)) { if (o[0] == 12345.678f) C[0] = o[1]; }   // ablation: no output stores
                        else if (cSplit) {
                            f16x4 h, l;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { h[e] = (_Float16)o[e]; l[e] = (_Float16)(o[e] - (float)h[e]); }
                            typedef f16x4 __attribute__((address_space(1)))* gwh4;
                            const gwh4 p = reinterpret_cast<gwh4>(reinterpret_cast<char __attribute__((address_space(1)))*>(C) + 4 * (long long)(rc + cbase[ni]) + 8 * hi + 16 * q);
                            p[0] = h;
                            p[8] = l;
                        } else {
                            *reinterpret_cast<f32x4 __attribute__((address_space(1)))*>(C + (rc + cbase[ni] + 4 * hi + 8 * q)) = o;
                        }
                    }
            }
        };
        // border tiles and unaligned outputs: one value at a time, predicated
        auto epilogue_scalar = [&](auto resTag) __attribute__((always_inline)) {
            constexpr bool HASR = decltype(resTag)::value;
            typedef const _Float16 __attribute__((address_space(1)))* gch;
            typedef _Float16 __attribute__((address_space(1)))* gh;
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
                        const int cofs = 4 * hi + (r & 3) + 8 * (r >> 2);          // column inside the 32-block
                        const bool ok = mok && (ncol[ni] - 4 * hi + cofs) < N;
                        float v = acc[mi][ni][r] * alpha + ((bias != nullptr && ok) ? bias[ncol[ni] - 4 * hi + cofs] : 0.f);
                        v = activate(v);
                        if constexpr (HASR) {
                            if (ok) { const long long e = 2 * (long long)(rr + cbase[ni]) + cofs; v += (float)((gch)R)[e] + (float)((gch)R)[e + 32]; }
                            if (postRelu) v = fmaxf(v, 0.f);
                        }
                        nonFinite |= !(__builtin_fabsf(v) <= vmax);
                        if constexpr (GG_ABL(32)) { if (v == 12345.678f) C[0] = v; }
                        else if (ok) {
                            if (cSplit) {
                                const long long e = 2 * (long long)(rc + cbase[ni]) + cofs;
                                const _Float16 h = (_Float16)v;
                                ((gh)C)[e] = h;
                                ((gh)C)[e + 32] = (_Float16)(v - (float)h);
                            } else {
                                C[rc + cbase[ni] + cofs] = v;
                            }
                        }
                    }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        bool vec = fullTile && ((reinterpret_cast<uintptr_t>(P->C) | (uintptr_t)(partial ? P->splitStride * 4 : 0)) & 15) == 0 &&
                   (bias == nullptr || (reinterpret_cast<uintptr_t>(P->bias) & 15) == 0) && (R == nullptr || (reinterpret_cast<uintptr_t>(P->R) & 15) == 0);
        {
            int low = 0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) low |= cbase[ni];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                low |= rowTab[wm * WTM + mi * 32 + l31];
                if (R != nullptr) low |= rowTab[BM + wm * WTM + mi * 32 + l31];
            }
            vec = vec && __all((low & 3) == 0);
        }
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
                else epilogue_vec(F_{}, IC{});
            }
        } else {
            if (R != nullptr) epilogue_scalar(T_{}); else epilogue_scalar(F_{});
        }
        if (rangeFlag != nullptr && __any(nonFinite) && lane == 0) atomicOr(rangeFlag, 1u);
        V6_STAMP(1)
        __syncthreads();
    }
}
