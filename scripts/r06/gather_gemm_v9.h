// EXPERIMENT, NOT IN THE LIBRARY (round 6; result: profiles/r06_v9_probe.log -- bit-identical to v7, +4 % / -12 % / +9 % on the three conv
// shapes, -4 % on the score shapes: the premise below was wrong, see DESIGN 8 item 1 for what the probe showed instead).
// gather_gemm_f16_v9: gather_gemm_f16_v7's tile (256 x 256 per 8-wave workgroup, one workgroup per CU, fp16 hi halves of split-format
// tensors as operands, dynamic tile height, LDS-turned epilogue -- see gather_gemm_v7.h) with a different OPERAND PIPELINE.
//
// What v7's probe numbers said (profiles/r04_v7_probe_c_ring.log): on the conv shape a tile's main loop takes 58 us where
// its MFMAs alone take 27 and its operand fetch alone 38 -- additive, at a sixth of the L2 bandwidth.  After each barrier BOTH waves of a
// SIMD issue their eight LDS-DMA instructions (~150 cycles each inside a phase that carries fragment reads: in-order issue, the wave
// multiplies nothing meanwhile) and then both multiply: the DMA-issue phases coincide and the MFMA phases share one pipe.
// Here the two waves of a SIMD (w and w + 4) take COMPLEMENTARY roles in every stage:
//     waves 0-3:  barrier | issue their share of stage i + 3 | multiply stage i
//     waves 4-7:  barrier | multiply stage i                 | issue their share of stage i + 3
// so one wave's DMA issue runs beside the other's MFMAs.  The late issuers' pieces need time to land: a RING of four 32-deep stages
// (one chunk's hi halves per stage, 64-byte rows; 4 x 32 KB = v7's 128 KB) -- a piece is issued three stages before it is read, i.e.
// at least two full stages (~1 us) earlier.  (Round 4 measured the ring with SYNCHRONOUS roles: nothing -- it removes no coincidence.)
// One barrier per stage; each wave waits for ITS OWN pieces of the stage (vmcnt counts 4 per stage in flight behind it), the barrier
// makes everybody's visible.  Same MFMAs on the same operands in the same k order per accumulator as v7: bit-identical output.
#pragma once
#include <type_traits>

template <int ROLES GG_ABL_PARAM>
__global__ void __launch_bounds__(512, 2)
gather_gemm_f16_v9(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue,
                   unsigned int* __restrict__ rangeFlag, int order)
{
    constexpr int SPLIT = 0;                                     // fp16 hi halves only (kernel variant 6)
    constexpr int BM = 256, BN = 256;
    constexpr int MI = 4, NI = 2;
    constexpr int A_BYTES = BM * 64, STAGE_BYTES = (BM + BN) * 64, NSTAGE = 4;
    constexpr int A_IT = 2, B_IT = 2;                            // LDS-DMA passes of 128 rows (64-byte rows: 4 lanes per row)
    static_assert(ROLES == 0 || ROLES == 1, "1: complementary roles of the two waves of a SIMD; 0: every wave issues first (A/B runs)");

    // ONE __shared__ object (a second one makes hipcc drain the LDS-DMA queue in front of every fragment read)
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE_BYTES / 4 + 2 * BM + 4];
    int* rowTab = reinterpret_cast<int*>(smem + NSTAGE * STAGE_BYTES / 4);
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + NSTAGE * STAGE_BYTES / 4 + 2 * BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int s_r = tid >> 2, s_q = tid & 3;                    // LDS-DMA: row-in-pass (0..127), piece slot (a 64-byte row: the hi halves of ONE chunk)
    const int lp = s_q ^ ((s_r >> 2) & 3);                      // logical piece this lane fetches (16 rows x 4 pieces fill the 64 banks once)
    const int srcSwz = lp << 2;                                 // float offset of the 16-byte group inside its chunk's hi halves
    int rd[2];                                                  // k-step st reads logical piece 2 st + hi of a row: at ((2 st + hi) ^ swizzle) << 4
#pragma unroll
    for (int st = 0; st < 2; ++st) rd[st] = (((2 * st + hi) ^ ((l31 >> 2) & 3)) << 4);
    const bool lateIssuer = ROLES == 1 && wave >= 4;            // (waves w and w + 4 share a SIMD)

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
            int m = m0 + s_r + 128 * it;
            if (m > M - 1) m = M - 1;
            const int o = rowA[m] + srcSwz;
            narrow = narrow && ((unsigned)o < (1u << 30));
            aoffB[it] = (unsigned)o << 2;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int n = n0 + s_r + 128 * it;
            if (n > N - 1) n = N - 1;
            const int o = rowB[n] + srcSwz;
            narrow = narrow && ((unsigned)o < (1u << 30));
            boffB[it] = (unsigned)o << 2;
        }
        if (rangeFlag != nullptr && !narrow) atomicOr(rangeFlag, 2u);
        const int aPasses = (R + 127) >> 7;                     // 128-row passes that hold rows of this tile

        V7_STAMP(1)
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // LDS-DMA of ONE chunk (its hi halves) into ring stage `buf`: this wave's share -- 2 A passes + 2 B passes of 128 rows, 16 rows
        // (1 KB) per wave and pass.  ca / cb = wave-uniform chunk offsets
        auto dma_stage = [&](int buf, int ca, int cb) __attribute__((always_inline)) {
            char* Ls = reinterpret_cast<char*>(smem) + buf * STAGE_BYTES;
            typedef const char __attribute__((address_space(1)))* gcc8;
            const gcc8 baseA = (gcc8)A + (long long)ca * 4, baseB = (gcc8)B + (long long)cb * 4;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                if (it < aPasses) {
                    unsigned vo = aoffB[it];
                    asm volatile("" : "+v"(vo));
                    glds16((gcf32)(baseA + vo), (lds_vptr)(Ls + (wave * 16 + 128 * it) * 64));
                }
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                unsigned vo = boffB[it];
                asm volatile("" : "+v"(vo));
                glds16((gcf32)(baseB + vo), (lds_vptr)(Ls + A_BYTES + (wave * 16 + 128 * it) * 64));
            }
        };
        // (a wave whose A pass holds no row of a short tile issues 2 or 3 instructions per stage instead of 4: the counted waits below
        // are in units of that)
        const int perStage = (aPasses < A_IT ? aPasses : A_IT) + B_IT;

        auto main_loop = [&](auto miaTag) __attribute__((always_inline)) {
            constexpr int MIA = decltype(miaTag)::value;
            struct Frag { f16x8 a[MIA > 0 ? MIA : 1], b[NI]; };
            auto read_frag = [&](int buf, int st, Frag& f) __attribute__((always_inline)) {
                const char* As = reinterpret_cast<const char*>(smem) + buf * STAGE_BYTES;
                const char* Bs = As + A_BYTES;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    f.b[ni] = *reinterpret_cast<const f16x8*>(Bs + (wn * 64 + ni * 32 + l31) * 64 + rd[st]);
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi)
                    f.a[mi] = *reinterpret_cast<const f16x8*>(As + ((wm + 2 * mi) * 32 + l31) * 64 + rd[st]);
            };
            auto mfma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[ni], f.a[mi], acc[mi][ni], 0, 0, 0);
            };
            // counted wait: at most `behind` stages of this wave's pieces may still be in flight (each perStage instructions)
            auto wait_own = [&](int behind) __attribute__((always_inline)) {
                if (perStage == 4) {
                    if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (perStage == 3) {
                    if (behind >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (behind == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    if (behind >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (behind == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            };
            for (int sb = kcBeg; sb < kcEnd; sb += 128) {
                const int sbEnd = sb + 128 < kcEnd ? sb + 128 : kcEnd;
                const int nst = sbEnd - sb;                                    // stages (chunks) of this super-block
                const int i0 = sb + lane < nchunksTotal ? sb + lane : nchunksTotal - 1;
                const int i1 = sb + 64 + lane < nchunksTotal ? sb + 64 + lane : nchunksTotal - 1;
                const int ca0v = colA[i0], ca1v = colA[i1], cb0v = colB[i0], cb1v = colB[i1];
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(ca0v), "v"(ca1v), "v"(cb0v), "v"(cb1v) : "memory");     // the tables are ordinary loads: out of the counted queue
                auto pick = [&](int v0, int v1, int i) {
                    const int a_ = __builtin_amdgcn_readlane(v0, i & 63), b_ = __builtin_amdgcn_readlane(v1, i & 63);
                    return i < 64 ? a_ : b_;
                };
                if (sb != kcBeg) {                                  // the ring of the previous super-block is still being read
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                // prime the ring: stages 0, 1, 2 by every wave
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < nst) dma_stage(j, pick(ca0v, ca1v, j), pick(cb0v, cb1v, j));
                for (int i = 0; i < nst; ++i) {
                    const int rem = nst - 1 - i;                   // stages behind this one
                    wait_own(rem < 2 ? rem : 2);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                    const bool more = i + 3 < nst;
                    const int na = pick(ca0v, ca1v, more ? i + 3 : i), nb = pick(cb0v, cb1v, more ? i + 3 : i);
                    if (more && !lateIssuer) dma_stage((i + 3) & 3, na, nb);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!GG_ABL(4)) {
                        Frag f0;
                        read_frag(i & 3, 0, f0);
                        mfma_frag(f0);
                        read_frag(i & 3, 1, f0);
                        mfma_frag(f0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && lateIssuer) dma_stage((i + 3) & 3, na, nb);
                }
            }
        };
        if (MIact >= 4) main_loop(std::integral_constant<int, 4>{});
        else if (MIact == 3) main_loop(std::integral_constant<int, 3>{});
        else if (MIact == 2) main_loop(std::integral_constant<int, 2>{});
        else if (MIact == 1) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 0>{});   // a wave without a block in this tile still fetches its share of the operands and meets every barrier
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
