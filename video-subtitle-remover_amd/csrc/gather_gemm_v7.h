// gather_gemm_f16_v7: the fp16-operand mode (BASELINE.json config 5) for LARGE NK problems on split-format tensors -- a 256 x 256
// output tile per 8-wave workgroup, one workgroup per CU.
//
// Why (profiles/r02_f16_ablation_*.log, r03_f16_tile_256x128_ab.log): v6's 128x64 tile contracts 64 k-values per barrier with
// 8 MFMAs per wave -- 256 matrix-pipe cycles against ~1 400 cycles of stage (barrier, counted wait, six LDS-DMA pieces, fragment
// read latency); it moves 24.5 KB of operands through L2 -> LDS per 2 x 128 x 64 x 64 FLOP.  The kernel is bound by those fixed
// costs and by bytes in flight, not by the matrix cores (0.17 of their roof).  A 256 x 256 tile changes both ratios at once: per
// barrier a wave issues 32 MFMAs (1 024 pipe cycles, two waves per SIMD = 2 048) and the workgroup moves 64 KB per 2 x 256 x 256 x 64
// FLOP -- 5.3 x fewer operand bytes per FLOP, 4 x more matrix work per barrier, and with N = 256 (every conv of the model) an
// activation row is fetched once per tap instead of once per tap and N tile.
//
// Structure
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns four 32-row blocks {wm, wm+2, wm+4, wm+6} x 64 columns = 4 x 2
//     accumulators of v_mfma_f32_32x32x16_f16 (128 registers), transposed as in v3 / v6 (a lane owns an output row).
//   * LDS: two stages of [256 A rows | 256 B rows] x 128 bytes.  A 128-byte row holds the fp16 hi halves of TWO consecutive K
//     chunks exactly as in v6 (pieces 0-3 chunk 2j, 4-7 chunk 2j+1; XOR swizzle on the source side of the LDS-DMA, mirrored on
//     the fragment read); a stage is filled by 8 LDS-DMA instructions per wave.
//   * one barrier per 64-deep stage: wait for the stage's pieces, barrier, issue the next stage's pieces into the other buffer,
//     then 4 k-steps x (6 ds_read_b128 + 8 MFMAs).
//   * DYNAMIC TILE HEIGHT.  One workgroup per CU means a launch is a whole number of rounds, and M = T x 4800 never is one:
//     a tile covers R = roundup32(ceil(M / tilesM)) <= 256 rows, R derived in the kernel from the problem's own M and tilesM, and
//     32-row blocks beyond R are skipped by the waves that own them (wave-uniform).  The host (vsr_v7_split) cuts a problem into a
//     body of whole rounds of 256-row tiles and a remainder problem of one short tile per CU, so the tail of a launch costs a few
//     short tiles instead of a second round.
//   * tiles: the first round is static (tile id = workgroup id), later ones come from one atomic counter whose answer is only
//     read after the main loop.
// Operands: fp16 hi halves of split-format tensors, fp32 accumulation, fp32 epilogue (bias, activation, residual), output in
// split format (VSR_ACT_OUT_SPLIT) or plain fp32, range guard as in v4 - v6.  Table indices are clamped to the problem's M / N,
// so the offset tables need no padding beyond M and N.
#pragma once
#include <type_traits>

template <int SPLIT GG_ABL_PARAM>
__global__ void __launch_bounds__(512, 2)
gather_gemm_f16_v7(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue,
                   unsigned int* __restrict__ rangeFlag, int order)
{
    constexpr int BM = 256, BN = 256, NT = 512;
    constexpr int MI = 4, NI = 2;
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    constexpr int A_IT = 4, B_IT = 4;                            // LDS-DMA passes of 64 rows
    static_assert(SPLIT == 0 || SPLIT == 1, "0: fp16 hi halves as operands (variant 6); 1: split-half operands, three MFMAs per product (variant 5)");

    // ONE __shared__ object (a second one makes hipcc drain the LDS-DMA queue in front of every fragment read)
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_BYTES / 4 + 2 * BM + 4];
    int* rowTab = reinterpret_cast<int*>(smem + 2 * STAGE_BYTES / 4);
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + 2 * STAGE_BYTES / 4 + 2 * BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int s_r = tid >> 3, s_q = tid & 7;                    // LDS-DMA: row-in-pass (0..63), piece slot
    const int lp = s_q ^ ((s_r >> 1) & 7);                      // logical piece this lane fetches
    // SPLIT 0: a row holds the hi halves of a PAIR of chunks (pieces 0-3 chunk 2j, 4-7 chunk 2j+1); SPLIT 1: ONE chunk, its 128
    // bytes as they lie in memory (pieces 0-3 hi halves, 4-7 lo halves)
    const bool second = SPLIT == 0 && (lp & 4) != 0;            // the piece belongs to the second chunk of the pair
    const int srcSwz = (SPLIT ? lp : (lp & 3)) << 2;            // float offset of the 16-byte group inside its chunk
    int rd[4];                                                  // logical piece p of a row lies at rd-style offset (p ^ swizzle) << 4
#pragma unroll
    for (int st = 0; st < 4; ++st) rd[st] = (((2 * st + hi) ^ ((l31 >> 1) & 7)) << 4);   // SPLIT 0: k-step st reads piece 2 st + hi; SPLIT 1: k-step st < 2 reads hi piece rd[st] and lo piece rd[2 + st]

#ifdef GG_ABLATE
    int tr_ = 0;                                     // 256: wall-clock stamps of wave 0 (100 MHz), 4 per tile
#define V7_STAMP(drain)                                                                                        \
    if constexpr (GG_ABL(256)) {                                                                               \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                 \
        if (tid == 0 && blockIdx.x < 1024 && tr_ < 256) gg_trace[blockIdx.x * 256 + tr_] = wall_clock64();     \
        ++tr_;                                                                                                 \
    }
#else
#define V7_STAMP(drain)
#endif

    // Tile order.  order 0: first round static (tile id = workgroup id), later rounds from one atomic counter.  order 1 (XCD-aware,
    // static): workgroup b runs on XCD b % 8 (observed placement, used for locality only) and takes slot b of every round of
    // gridDim.x tile ids; within a round XCD x owns the x-th eighth of the ids, so the ~32 tiles an XCD works on at a time are
    // NEIGHBOURS in id space -- they share A row blocks (the N tiles of one M tile) or sit in adjacent M tiles -- and meet in that
    // XCD's L2.  In plain order a tile's 31 neighbours on its XCD are ids 8 apart: different M tile and different N tile, nothing
    // shared, every operand block crosses the fabric once per tile (the split-format modes are bound by exactly that stream).
    const int G = (int)gridDim.x;
    auto xcd_slot = [&](int id) -> int {
        const int r = id / G, s = id - r * G;
        const int cnt = totalTiles - r * G < G ? totalTiles - r * G : G;      // ids of this round
        const int x = s & 7, k = s >> 3, q = cnt >> 3, rm = cnt & 7;
        if (s >= cnt) return totalTiles;                                      // a slot beyond a partial last round
        // XCD x owns q + (x < rm) ids starting at x q + min(x, rm); slot (x, k) exists for k < ceil((cnt - x) / 8) = that count
        return r * G + x * q + (x < rm ? x : rm) + k;
    };
    int slotId = blockIdx.x;                         // order 1: the slot sequence b, b + G, b + 2 G, ...
    int bid = order ? xcd_slot(slotId) : (int)blockIdx.x;      // first round: static
    for (;;) {
        if (bid >= totalTiles) break;
        V7_STAMP(0)
        unsigned int pend = 0;
        if (tid == 0 && !order) pend = atomicAdd(queue, 1u);   // the tile after this one; the answer is read after the main loop

        int pi = 0;
        for (int lo_ = 0, hi_ = nprobs - 1; lo_ < hi_;) {
            const int mid_ = (lo_ + hi_ + 1) >> 1;
            if (bid >= probs[mid_].tileStart) lo_ = mid_; else hi_ = mid_ - 1;
            pi = lo_;
        }
        const GGProblem* __restrict__ P = probs + __builtin_amdgcn_readfirstlane(pi);
        // (everything below is wave-uniform; say so, or the loop control and the table indices live in vector registers)
        const int M = __builtin_amdgcn_readfirstlane(P->M), N = __builtin_amdgcn_readfirstlane(P->N);
        const int tilesM = __builtin_amdgcn_readfirstlane(P->tilesM), tilesN = __builtin_amdgcn_readfirstlane(P->tilesN);
        const int splitK = __builtin_amdgcn_readfirstlane(P->splitK);
        const int tilesMN = tilesM * tilesN;
        const int t = bid - __builtin_amdgcn_readfirstlane(P->tileStart);
        const int split = __builtin_amdgcn_readfirstlane(t / tilesMN);
        const int rem = t - split * tilesMN;
        const int tm = __builtin_amdgcn_readfirstlane(rem / tilesN);
        const int tn = rem - tm * tilesN;
        const int nchunksTotal = __builtin_amdgcn_readfirstlane(P->K / VSR_GG_KC);
        const int kcBeg = __builtin_amdgcn_readfirstlane(split * P->chunksPerSplit);
        int kcEnd = kcBeg + __builtin_amdgcn_readfirstlane(P->chunksPerSplit);
        if (kcEnd > nchunksTotal) kcEnd = nchunksTotal;
        kcEnd = __builtin_amdgcn_readfirstlane(kcEnd);
        // tile height: the rows of the problem spread evenly over its M tiles, in whole 32-row blocks
        int R = (((M + tilesM - 1) / tilesM) + 31) & ~31;
        if (R > BM) R = BM;
        R = __builtin_amdgcn_readfirstlane(R);
        const int m0 = tm * R, n0 = tn * BN;
        const int nblk = R >> 5;                                // 32-row blocks of this tile
        const int MIact = (nblk - wm + 1) >> 1;                  // blocks wm, wm + 2, ... < nblk owned by this wave

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
            int m = m0 + (tid & (BM - 1));
            if (m > M - 1) m = M - 1;
            rowTab[tid] = tid < BM ? rowCt[m] : (hasR ? rowRt[m] : 0);
        }
        // byte offsets of the lane's operand rows (the saddr form adds them to a scalar base as unsigned 32-bit values: every row
        // offset must lie in [0, 2^30) floats -- true for any tensor below 4 GB; otherwise the launch reports it through rangeFlag
        // bit 1 and its results are not to be used)
        unsigned aoffB[A_IT], boffB[B_IT];
        bool narrow = true;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int m = m0 + s_r + 64 * it;
            if (m > M - 1) m = M - 1;
            const int o = rowA[m] + srcSwz;
            narrow = narrow && ((unsigned)o < (1u << 30));
            aoffB[it] = (unsigned)o << 2;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int n = n0 + s_r + 64 * it;
            if (n > N - 1) n = N - 1;
            const int o = rowB[n] + srcSwz;
            narrow = narrow && ((unsigned)o < (1u << 30));
            boffB[it] = (unsigned)o << 2;
        }
        if (rangeFlag != nullptr && !narrow) atomicOr(rangeFlag, 2u);
        const int aPasses = (R + 63) >> 6;                      // 64-row passes that hold rows of this tile

        V7_STAMP(1)
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // LDS-DMA of one pair of chunks into a stage, in two parts (0: the A rows, 1: the B rows) so that the pieces can be spread
        // between the k-steps of the stage that is being computed; ca0 / ca1 (cb0 / cb1) = wave-uniform chunk offsets of the pair
        auto dma_part = [&](auto bufTag, auto partTag, int c0, int c1) __attribute__((always_inline)) {
            constexpr int buf = decltype(bufTag)::value;
            constexpr int part = decltype(partTag)::value;
            char* Ls = reinterpret_cast<char*>(smem) + buf * STAGE_BYTES + part * A_BYTES;
            if constexpr (GG_ABL(2)) { if (buf >= 0) return; }     // ablation: no operand fetch at all
            if constexpr (GG_ABL(8)) { c0 = c1 = 0; }              // 8: one hot chunk
            const gcf32 X = part == 0 ? A : B;
            const int passes = part == 0 ? aPasses : B_IT;
            // saddr form of global_load_lds: scalar base = operand + the SMALLER chunk offset of the pair, per-lane unsigned byte
            // offset = row offset + what its chunk lies above that
            typedef const char __attribute__((address_space(1)))* gcc8;
            const int mn = c0 < c1 ? c0 : c1;
            const gcc8 base = (gcc8)X + (long long)mn * 4;
            const unsigned dl = (unsigned)((second ? c1 : c0) - mn) << 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (it < passes) {
                    unsigned vo = (part == 0 ? aoffB[it] : boffB[it]) + dl;
                    asm volatile("" : "+v"(vo));
                    glds16((gcf32)(base + vo), (lds_vptr)(Ls + (wave * 8 + 64 * it) * 128));
                }
            }
        };

        // The main loop, specialised on the number of 32-row blocks this wave owns in this tile (MIA = 1..4: no branch around an
        // MFMA).  Operand pipeline: two stages, one barrier per pair of chunks.  No ordinary (VGPR-destination) load inside the loop:
        // hipcc would wait vmcnt(0) for it in the middle of a stage.  The chunk-offset tables are fetched per super-block of 128
        // chunks (2 VGPRs per table); 128 is even, so a pair never straddles two super-blocks.
        auto main_loop = [&](auto miaTag) __attribute__((always_inline)) {
            constexpr int MIA = decltype(miaTag)::value;
            struct Frag { f16x8 a[MIA > 0 ? MIA : 1], b[NI]; };
            // fragment reads of k-step st (16 k-values) of a stage / its MFMAs (transposed: the weight or key fragment is the
            // first operand, a lane owns an output ROW)
            auto read_frag = [&](auto bufTag, int st, Frag& f) __attribute__((always_inline)) {
                constexpr int buf = decltype(bufTag)::value;
                const char* As = reinterpret_cast<const char*>(smem) + buf * STAGE_BYTES;
                const char* Bs = As + A_BYTES;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    f.b[ni] = *reinterpret_cast<const f16x8*>(Bs + (wn * 64 + ni * 32 + l31) * 128 + rd[st]);
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi)
                    f.a[mi] = *reinterpret_cast<const f16x8*>(As + ((wm + 2 * mi) * 32 + l31) * 128 + rd[st]);
            };
            auto mfma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[ni], f.a[mi], acc[mi][ni], 0, 0, 0);
            };
            using S0_ = std::integral_constant<int, 0>;
            using S1_ = std::integral_constant<int, 1>;
            using PA_ = std::integral_constant<int, 0>;
            using PB_ = std::integral_constant<int, 1>;
            for (int sb = kcBeg; sb < kcEnd; sb += 128) {
                const int sbEnd = sb + 128 < kcEnd ? sb + 128 : kcEnd;
                const int i0 = sb + lane < nchunksTotal ? sb + lane : nchunksTotal - 1;
                const int i1 = sb + 64 + lane < nchunksTotal ? sb + 64 + lane : nchunksTotal - 1;
                const int ca0v = colA[i0], ca1v = colA[i1], cb0v = colB[i0], cb1v = colB[i1];
                asm volatile("" ::"v"(ca0v), "v"(ca1v), "v"(cb0v), "v"(cb1v));
                auto pick = [&](int v0, int v1, int i) {           // both halves read, scalar select: no branch in the loop
                    const int a_ = __builtin_amdgcn_readlane(v0, i & 63), b_ = __builtin_amdgcn_readlane(v1, i & 63);
                    return i < 64 ? a_ : b_;
                };
                if (sb != kcBeg) {                                  // stages of the previous super-block are still being read
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                {   // pair (sb, sb + 1) into stage 0; a lone last chunk is fetched twice
                    const int j = sb + 1 < sbEnd ? 1 : 0;
                    dma_part(S0_{}, PA_{}, pick(ca0v, ca1v, 0), pick(ca0v, ca1v, j));
                    dma_part(S0_{}, PB_{}, pick(cb0v, cb1v, 0), pick(cb0v, cb1v, j));
                }
                // one pair out of stage `cur`; the DMA of the next pair goes into stage `nxt`, which this step's barrier retires
                // (spreading its pieces between the k-steps instead was measured: no difference, profiles/r04_v7_probe_a.log)
                auto step = [&](int kc, auto cur, auto nxt) __attribute__((always_inline)) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                    const bool more = kc + 2 < sbEnd;
                    const int i = kc + 2 - sb, j = kc + 3 < sbEnd ? i + 1 : i;
                    const int na0 = pick(ca0v, ca1v, i), na1 = pick(ca0v, ca1v, j), nb0 = pick(cb0v, cb1v, i), nb1 = pick(cb0v, cb1v, j);
                    if (more) { dma_part(nxt, PA_{}, na0, na1); dma_part(nxt, PB_{}, nb0, nb1); }
                    if constexpr (GG_ABL(4)) return;
                    Frag f0;
                    read_frag(cur, 0, f0);
                    mfma_frag(f0);
                    read_frag(cur, 1, f0);
                    mfma_frag(f0);
                    if (kc + 1 < sbEnd) {
                        read_frag(cur, 2, f0);
                        mfma_frag(f0);
                        read_frag(cur, 3, f0);
                        mfma_frag(f0);
                    }
                };
                for (int kc = sb; kc < sbEnd; kc += 4) {
                    step(kc, S0_{}, S1_{});
                    if (kc + 2 < sbEnd) step(kc + 2, S1_{}, S0_{});
                }
            }
        };

        // ---- SPLIT 1 (kernel variant 5): split-half operands, a.b = a_lo.b_hi + a_hi.b_lo + a_hi.b_hi in that order per k-step (the
        // order of gather_gemm_f32_v5, so the two kernels agree bit for bit).  A stage is ONE 32-deep chunk -- its 128-byte rows are
        // the tensor's own [32 hi | 32 lo] lines, no byte fetched in vain -- and carries 48 MFMAs per wave: three times the matrix
        // work of the fp16-operand form per LDS byte and per barrier.
        auto main_loop_split = [&](auto miaTag) __attribute__((always_inline)) {
            constexpr int MIA = decltype(miaTag)::value;
            struct Frag { f16x8 ah[MIA > 0 ? MIA : 1], al[MIA > 0 ? MIA : 1], bh[NI], bl[NI]; };
            auto read_frag = [&](auto bufTag, int st, Frag& f) __attribute__((always_inline)) {
                constexpr int buf = decltype(bufTag)::value;
                const char* As = reinterpret_cast<const char*>(smem) + buf * STAGE_BYTES;
                const char* Bs = As + A_BYTES;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const char* row = Bs + (wn * 64 + ni * 32 + l31) * 128;
                    f.bh[ni] = *reinterpret_cast<const f16x8*>(row + rd[st]);
                    f.bl[ni] = *reinterpret_cast<const f16x8*>(row + rd[2 + st]);
                }
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi) {
                    const char* row = As + ((wm + 2 * mi) * 32 + l31) * 128;
                    f.ah[mi] = *reinterpret_cast<const f16x8*>(row + rd[st]);
                    f.al[mi] = *reinterpret_cast<const f16x8*>(row + rd[2 + st]);
                }
            };
            auto mfma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ni], f.al[mi], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ni], f.ah[mi], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ni], f.ah[mi], acc[mi][ni], 0, 0, 0);
                    }
            };
            using S0_ = std::integral_constant<int, 0>;
            using S1_ = std::integral_constant<int, 1>;
            using PA_ = std::integral_constant<int, 0>;
            using PB_ = std::integral_constant<int, 1>;
            for (int sb = kcBeg; sb < kcEnd; sb += 128) {
                const int sbEnd = sb + 128 < kcEnd ? sb + 128 : kcEnd;
                const int i0 = sb + lane < nchunksTotal ? sb + lane : nchunksTotal - 1;
                const int i1 = sb + 64 + lane < nchunksTotal ? sb + 64 + lane : nchunksTotal - 1;
                const int ca0v = colA[i0], ca1v = colA[i1], cb0v = colB[i0], cb1v = colB[i1];
                asm volatile("" ::"v"(ca0v), "v"(ca1v), "v"(cb0v), "v"(cb1v));
                auto pick = [&](int v0, int v1, int i) {
                    const int a_ = __builtin_amdgcn_readlane(v0, i & 63), b_ = __builtin_amdgcn_readlane(v1, i & 63);
                    return i < 64 ? a_ : b_;
                };
                if (sb != kcBeg) {                                  // stages of the previous super-block are still being read
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                {
                    const int ca = pick(ca0v, ca1v, 0), cb = pick(cb0v, cb1v, 0);
                    dma_part(S0_{}, PA_{}, ca, ca);
                    dma_part(S0_{}, PB_{}, cb, cb);
                }
                auto step = [&](int kc, auto cur, auto nxt) __attribute__((always_inline)) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                    const int i = kc + 1 - sb;
                    const int na = pick(ca0v, ca1v, i), nb = pick(cb0v, cb1v, i);
                    if (kc + 1 < sbEnd) { dma_part(nxt, PA_{}, na, na); dma_part(nxt, PB_{}, nb, nb); }
                    if constexpr (GG_ABL(4)) return;
                    Frag f;
                    read_frag(cur, 0, f);
                    mfma_frag(f);
                    read_frag(cur, 1, f);
                    mfma_frag(f);
                };
                for (int kc = sb; kc < sbEnd; kc += 2) {
                    step(kc, S0_{}, S1_{});
                    if (kc + 1 < sbEnd) step(kc + 1, S1_{}, S0_{});
                }
            }
        };
        if constexpr (SPLIT == 1) {
            if (MIact >= 4) main_loop_split(std::integral_constant<int, 4>{});
            else if (MIact == 3) main_loop_split(std::integral_constant<int, 3>{});
            else if (MIact == 2) main_loop_split(std::integral_constant<int, 2>{});
            else if (MIact == 1) main_loop_split(std::integral_constant<int, 1>{});
            else main_loop_split(std::integral_constant<int, 0>{});
        } else
        if (MIact >= 4) main_loop(std::integral_constant<int, 4>{});
        else if (MIact == 3) main_loop(std::integral_constant<int, 3>{});
        else if (MIact == 2) main_loop(std::integral_constant<int, 2>{});
        else if (MIact == 1) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 0>{});   // a wave without a block in this tile (R = 32, wm = 1) still fetches its share of the operands and meets every barrier
        if (tid == 0) {
            if (order) {
                int nx = totalTiles;
                for (slotId += G; slotId < (totalTiles + G - 1) / G * G; slotId += G) {      // (a slot beyond a partial round has no tile)
                    nx = xcd_slot(slotId);
                    if (nx < totalTiles) break;
                }
                *nextTile = nx < totalTiles ? nx : totalTiles;
            } else {
                *nextTile = (int)gridDim.x + (int)pend;
            }
        }
        __syncthreads();                           // rowTab visible even when the k range is empty; LDS-DMA queue empty; nextTile published
        V7_STAMP(0)

        // ---- epilogue.  Transposed accumulators: lane l31 owns output ROW l31 of its 32x32 block, register r is column
        // (r & 3) + 8 (r >> 2) + 4 hi.  Stored from there, a lane moves 8-byte runs of 32 different rows per instruction and the
        // epilogue of a 256 x 256 tile took 33 us of its 113 (profiles/r04_v7_probe_a.log: 128 half-used memory instructions per
        // lane, store-issue bound).  So a wave first turns its 32 x 64 block (one mi, both ni) through a PRIVATE 8.5 KB patch of the
        // idle operand stages: written as it lies in the accumulators (row pitch 272 bytes: conflict-free float4 writes), read
        // back with lane -> (row = lane / 8 + 8 pass, 8 consecutive columns = lane % 8).  In split format 8 columns are 16 bytes of
        // hi halves and, 64 bytes on, 16 bytes of lo halves: four lanes fill a half line, every load / store is 16 bytes per lane
        // and the lane's columns -- hence its bias values -- are the same in every pass.  No barrier: the patch is the wave's own.
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;
        const bool cSplit = (P->act & VSR_ACT_OUT_SPLIT) != 0;
        const float vmax = cSplit ? 65504.f : 3.0e38f;
        bool nonFinite = false;
        const bool partial = (splitK > 1);
        const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
        const gcf32 Rr = (partial || GG_ABL(64)) ? (gcf32) nullptr : (gcf32)P->R;
        const cci32 colC = (cci32)P->colC;
        const gf32 C = (gf32)(P->C + (partial ? (int64_t)split * P->splitStride : (int64_t)0));
        auto activate = [&](float v) __attribute__((always_inline)) {
            if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
            else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
            return v;
        };
        constexpr int PITCH = 272;                               // bytes per patch row (64 floats + 16 bytes)
        char* patch = reinterpret_cast<char*>(smem) + wave * (32 * PITCH);
        const int e_r = lane >> 3, e_c = lane & 7;               // read-back: row-in-pass, group of 8 columns
        const int e_ni = e_c >> 2;                               // ... which lies in this 32-column block of the wave
        const int nbE = n0 + wn * 64 + e_ni * 32;                // first column of that block
        const bool colOk = nbE < N;
        const int cbaseE = colC[(colOk ? nbE : 0) / VSR_GG_KC];  // float offset of the block in an output row
        const int ncolE = nbE + 8 * (e_c & 3);                   // the lane's first column
        // whole 32-column blocks, 16-byte aligned rows: the vector path
        bool vec = (N % 32 == 0) && ((reinterpret_cast<uintptr_t>(P->C) | (uintptr_t)(partial ? P->splitStride * 4 : 0)) & 15) == 0 &&
                   (bias == nullptr || (reinterpret_cast<uintptr_t>(P->bias) & 15) == 0) && (Rr == nullptr || (reinterpret_cast<uintptr_t>(P->R) & 15) == 0);
        {   // (a wave decides for itself: the patch is its own and nothing below meets another wave)
            int low = cbaseE;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (mi >= MIact) continue;
                low |= rowTab[(wm + 2 * mi) * 32 + l31];
                if (Rr != nullptr) low |= rowTab[BM + (wm + 2 * mi) * 32 + l31];
            }
            vec = vec && __all((low & 3) == 0);
        }
        if (vec) {
            typedef const f32x4 __attribute__((address_space(1)))* gv4;
            typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
            typedef const f16x8v __attribute__((address_space(1)))* gh8;
            typedef f16x8v __attribute__((address_space(1)))* gwh8;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && colOk) { b0 = *reinterpret_cast<gv4>(bias + ncolE); b1 = *reinterpret_cast<gv4>(bias + ncolE + 4); }
            const int ak = postRelu ? -1 : act;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (mi >= MIact) continue;
                // the block as it lies in the accumulators: lane (l31, hi) writes row l31, columns ni 32 + 8 q + 4 hi ..+3
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v4 = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(patch + l31 * PITCH + (ni * 32 + 8 * q + 4 * hi) * 4) = v4;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int blockRow = (wm + 2 * mi) * 32;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int row = blockRow + 8 * ps + e_r;
                    const bool ok = colOk && (m0 + row) < M;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(patch + (8 * ps + e_r) * PITCH + e_c * 32);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(patch + (8 * ps + e_r) * PITCH + e_c * 32 + 16);
                    const int rc = rowTab[row];
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = x0[e] * alpha + b0[e]; v[4 + e] = x1[e] * alpha + b1[e]; }
                    if (ak == VSR_ACT_LRELU02) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                    } else if (ak != VSR_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = activate(v[e]);
                    }
                    if (Rr != nullptr) {             // residual tensors are GEMM operands too: split format
                        const int rr = rowTab[BM + row];
                        const gh8 pr = reinterpret_cast<gh8>(reinterpret_cast<const char __attribute__((address_space(1)))*>(Rr) + 4 * (long long)(rr + cbaseE) + 16 * (e_c & 3));
                        f16x8v rh = {0, 0, 0, 0, 0, 0, 0, 0}, rl = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (ok) { rh = pr[0]; rl = pr[4]; }          // + 64 bytes
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] += (float)rh[e] + (float)rl[e];
                            if (postRelu) v[e] = fmaxf(v[e], 0.f);
                        }
                    }
                    if (ok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) nonFinite |= !(__builtin_fabsf(v[e]) <= vmax);
                    }
                    if constexpr (GG_ABL(32)) { if (v[0] == 12345.678f) C[0] = v[1]; }   // ablation: no output stores
                    else if (ok) {
                        if (cSplit) {
                            f16x8v h, l;
#pragma unroll
                            for (int e = 0; e < 8; ++e) { h[e] = (_Float16)v[e]; l[e] = (_Float16)(v[e] - (float)h[e]); }
                            const gwh8 pw = reinterpret_cast<gwh8>(reinterpret_cast<char __attribute__((address_space(1)))*>(C) + 4 * (long long)(rc + cbaseE) + 16 * (e_c & 3));
                            pw[0] = h;
                            pw[4] = l;
                        } else {
                            typedef f32x4 __attribute__((address_space(1)))* gw4;
                            const gw4 pw = reinterpret_cast<gw4>(C + (rc + cbaseE + 8 * (e_c & 3)));
                            pw[0] = f32x4{v[0], v[1], v[2], v[3]};
                            pw[1] = f32x4{v[4], v[5], v[6], v[7]};
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next block
            }
        } else {
            // unaligned outputs or N not a multiple of 32: one value at a time, predicated, straight from the accumulators
            typedef const _Float16 __attribute__((address_space(1)))* gch;
            typedef _Float16 __attribute__((address_space(1)))* gh;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (mi >= MIact) continue;
                const int row = (wm + 2 * mi) * 32 + l31;
                const int rc = rowTab[row];
                const int rr = rowTab[BM + row];
                const bool mok = (m0 + row) < M;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int nb = n0 + wn * 64 + ni * 32;
                    const int cb = colC[(nb < N ? nb : 0) / VSR_GG_KC];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cofs = 4 * hi + (r & 3) + 8 * (r >> 2);          // column inside the 32-block
                        const bool ok = mok && (nb + cofs) < N;
                        float v = acc[mi][ni][r] * alpha + ((bias != nullptr && ok) ? bias[nb + cofs] : 0.f);
                        v = activate(v);
                        if (Rr != nullptr) {
                            if (ok) { const long long e = 2 * (long long)(rr + cb) + cofs; v += (float)((gch)Rr)[e] + (float)((gch)Rr)[e + 32]; }
                            if (postRelu) v = fmaxf(v, 0.f);
                        }
                        if (ok) nonFinite |= !(__builtin_fabsf(v) <= vmax);
                        if constexpr (GG_ABL(32)) { if (v == 12345.678f) C[0] = v; }
                        else if (ok) {
                            if (cSplit) {
                                const long long e = 2 * (long long)(rc + cb) + cofs;
                                const _Float16 h = (_Float16)v;
                                ((gh)C)[e] = h;
                                ((gh)C)[e + 32] = (_Float16)(v - (float)h);
                            } else {
                                C[rc + cb + cofs] = v;
                            }
                        }
                    }
                }
            }
        }
        if (rangeFlag != nullptr && __any(nonFinite) && lane == 0) atomicOr(rangeFlag, 1u);
        V7_STAMP(1)
        bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();                           // every wave has read nextTile and is out of rowTab
    }
}
