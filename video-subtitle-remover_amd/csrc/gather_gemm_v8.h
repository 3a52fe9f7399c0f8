// gather_gemm_f32_v8: the exact-fp32 gather-GEMM (v_mfma_f32_32x32x2_f32) for LARGE, LONG-K NK problems -- the 3x3 256 -> 256
// convolutions that are 59 % of the STTN step -- as ONE tile of up to 288 x 256 outputs per 8-wave workgroup, one workgroup per CU.
//
// Why (profiles/r03_v3_probe_*.log, r03_mfma_ceiling.log, r04_v8_probe_*.log).  v3's 128 x 64 tile keeps the matrix pipe busy 97 %
// of the time INSIDE its chunk loop, and still a T = 15 conv launch takes 679 us against 555 us of MFMA issue: 2 252 tiles of
// 190 us each (three per CU) leave a ragged tail in which CUs run one or two workgroups -- a single 4-wave workgroup cannot hide
// its own barriers and fragment latency -- and every tile pays a prologue / epilogue that its neighbours only partly cover; it
// also moves 24.5 KB of operands through L2 -> LDS per 0.5 MFLOP and reads every LDS byte twice.  With N = 256 in ONE tile an
// activation row is fetched once per tap (not once per tap and N tile), a wave contracts a 160 x 64 block per fragment set
// (0.19 fragment reads per MFMA instead of 0.375, 0.06 LDS-DMA pieces instead of 0.19), and a launch is ONE round of equal tiles
// that end together: no tail.
//
// Structure (the frame of gather_gemm_f16_v7, the arithmetic and the k order of gather_gemm_f32_v3 -- results are bit-identical
// to v3's):
//   * 512 threads = 8 waves as 2 (M) x 4 (N).  A wave owns the 32-row blocks {wm, wm + 2, ...} of the tile x 64 columns: up to
//     5 x 2 accumulators (160 registers), transposed as in v3 (the weight fragment is the MFMA's first operand: a lane owns an
//     output ROW).  Waves w and w + 4 share a SIMD, so every SIMD carries the same 9 x 2 blocks whatever the split 5 / 4.
//   * LDS: two stages of [288 A rows | 256 B rows] x 128 bytes (one 32-deep K chunk; 136 KB in all), filled by LDS-DMA
//     (global_load_lds_dwordx4, saddr form, XOR swizzle on the source side mirrored on the fragment read -- v3's image).
//   * one barrier per chunk, INSIDE the chunk (between its third and fourth fragment group, see main_loop): the fourth group's
//     MFMAs cover the barrier, the LDS-DMA issue of chunk k + 2 and the first fragment reads of chunk k + 1; a chunk is
//     4 groups x ((MI + 2) ds_read_b128 + 8 MI MFMAs) with the next group's fragments always in flight.
//   * DYNAMIC TILE HEIGHT (as v7): a tile covers R = roundup32(ceil(M / tilesM)) <= 288 rows, derived in the kernel from the
//     problem's own M and tilesM; blocks beyond R are skipped by the waves that own them and their rows are not fetched.  The host
//     (vsr_v8_split) picks tilesM so that a launch is whole rounds of equal tiles: M = 72 000 (T = 15) -> 250 tiles of 288 rows.
//   * tiles: first round static (tile id = workgroup id), later ones from one atomic counter read after the main loop.
//   * epilogue: a wave turns each 32 x 64 block through a private LDS patch (the idle operand stages) so that a lane stores
//     32 contiguous bytes of one output row and 8 lanes fill two whole 128-byte lines; bias / activation / residual as in v3.
// Not here: split-K partial planes are supported (C + split * splitStride, no bias / residual); VSR_ACT_ROW_MAX and the KN form
// are not (the host never routes them: short-K score / QKV products are better off on v3's three overlapping workgroups).
#pragma once
#include <type_traits>

template <int MAXB GG_ABL_PARAM>
__global__ void __launch_bounds__(512, 2)
gather_gemm_f32_v8(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue)
{
    constexpr int BM = MAXB * 32, BN = 256;
    constexpr int NI = 2, MI = (MAXB + 1) / 2;
    constexpr int A_BYTES = BM * 128, STAGE_BYTES = (BM + BN) * 128;
    constexpr int A_IT = (BM + 63) / 64, B_IT = 4;               // LDS-DMA passes of 64 rows
    static_assert(MAXB >= 2 && MAXB <= 9, "tile height in 32-row blocks");
    static_assert(2 * STAGE_BYTES + 8 * BM + 16 <= 160 * 1024, "LDS");
    static_assert(8 * 32 * 272 <= 2 * STAGE_BYTES, "epilogue patches fit the operand stages");

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
    const int srcSwz = (s_q ^ ((s_r >> 1) & 7)) << 2;           // float offset of the 16-byte group this lane fetches
    int rd[4];                                                  // byte offset of fragment group g in a row of this lane
#pragma unroll
    for (int g = 0; g < 4; ++g) rd[g] = (((2 * g + hi) ^ ((l31 >> 1) & 7)) << 4);

#ifdef GG_ABLATE
    int tr_ = 0;                                     // 256: wall-clock stamps of wave 0 (100 MHz), 4 per tile
#define V8_STAMP(drain)                                                                                        \
    if constexpr (GG_ABL(256)) {                                                                               \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                 \
        if (tid == 0 && blockIdx.x < 1024 && tr_ < 256) gg_trace[blockIdx.x * 256 + tr_] = wall_clock64();     \
        ++tr_;                                                                                                 \
    }
#else
#define V8_STAMP(drain)
#endif

#ifdef GG_ABLATE
    const unsigned long long kc0_ = __builtin_readcyclecounter(), kr0_ = __builtin_amdgcn_s_memrealtime();   // shader clock vs 100 MHz
#endif
    int bid = blockIdx.x;                            // first round: static
    for (;;) {
        if (bid >= totalTiles) break;
        V8_STAMP(0)
        unsigned int pend = 0;
        if (tid == 0) pend = atomicAdd(queue, 1u);   // the tile after this one; the answer is read after the main loop

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
            for (int i = tid; i < 2 * BM; i += 512) {
                int m = m0 + (i < BM ? i : i - BM);
                if (m > M - 1) m = M - 1;
                rowTab[i] = i < BM ? rowCt[m] : (hasR ? rowRt[m] : 0);
            }
        }
        // byte offsets of the lane's operand rows (the saddr form adds them to a scalar base as unsigned 32-bit values: every row
        // offset must lie in [0, 2^30) floats -- true for any tensor below 4 GB, which the host checks before it routes a problem here)
        unsigned aoffB[A_IT], boffB[B_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int m = m0 + s_r + 64 * it;
            if (m > M - 1) m = M - 1;
            aoffB[it] = (unsigned)(rowA[m] + srcSwz) << 2;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int n = n0 + s_r + 64 * it;
            if (n > N - 1) n = N - 1;
            boffB[it] = (unsigned)(rowB[n] + srcSwz) << 2;
        }

        V8_STAMP(1)
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        // LDS-DMA of one chunk into a stage, in two parts (0: the A rows of this tile, 1: the B rows); c = the chunk's offset
        auto dma_part = [&](auto bufTag, auto partTag, int c) __attribute__((always_inline)) {
            constexpr int buf = decltype(bufTag)::value;
            constexpr int part = decltype(partTag)::value;
            char* Ls = reinterpret_cast<char*>(smem) + buf * STAGE_BYTES + part * A_BYTES;
            if constexpr (GG_ABL(2)) { if (buf >= 0) return; }     // ablation: no operand fetch at all
            if constexpr (GG_ABL(8)) { c = 0; }                    // 8: one hot chunk
            typedef const char __attribute__((address_space(1)))* gcc8;
            const gcc8 base = (gcc8)(part == 0 ? A : B) + (long long)c * 4;
#pragma unroll
            for (int it = 0; it < (part == 0 ? A_IT : B_IT); ++it) {
                if (part == 1 || 64 * it + wave * 8 < R) {         // (wave-uniform: R is a multiple of 32, a wave fills 8 rows)
                    unsigned vo = part == 0 ? aoffB[it] : boffB[it];
                    asm volatile("" : "+v"(vo));                   // keeps the zero-extension next to the load, where hipcc folds it into the saddr form
                    glds16((gcf32)(base + vo), (lds_vptr)(Ls + (wave * 8 + 64 * it) * 128));
                }
            }
        };

        // The main loop, specialised on the number of 32-row blocks this wave owns in this tile (MIA = 0..MI: no branch around an
        // MFMA).  No ordinary (VGPR-destination) load inside the loop: hipcc would wait vmcnt(0) for it in the middle of a chunk.
        // The chunk-offset tables are fetched per super-block of 128 chunks (2 VGPRs per table) and picked with v_readlane.
        auto main_loop = [&](auto miaTag) __attribute__((always_inline)) {
            constexpr int MIA = decltype(miaTag)::value;
            struct Frag { f32x4 a[MIA > 0 ? MIA : 1], b[NI]; };
            // fragment reads of group g (four k-steps: lane (l31, hi) owns k = 8 g + 4 hi + j) / its MFMAs (transposed: the weight
            // fragment is the first operand).  Per accumulator the products arrive in (chunk, g, j) order: v3's order.
            auto read_frag = [&](auto bufTag, int g, Frag& f) __attribute__((always_inline)) {
                constexpr int buf = decltype(bufTag)::value;
                const char* As = reinterpret_cast<const char*>(smem) + buf * STAGE_BYTES;
                const char* Bs = As + A_BYTES;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    f.b[ni] = *reinterpret_cast<const f32x4*>(Bs + (wn * 64 + ni * 32 + l31) * 128 + rd[g]);
#pragma unroll
                for (int mi = 0; mi < MIA; ++mi)
                    f.a[mi] = *reinterpret_cast<const f32x4*>(As + ((wm + 2 * mi) * 32 + l31) * 128 + rd[g]);
            };
            auto mfma_frag = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[ni][j], f.a[mi][j], acc[mi][ni], 0, 0, 0);
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
                if constexpr (GG_ABL(512)) {
                // ---- the plain sequence (kept for A/B runs of the probe): barrier, next chunk's pieces, four groups
                dma_part(S0_{}, PA_{}, pick(ca0v, ca1v, 0));
                dma_part(S0_{}, PB_{}, pick(cb0v, cb1v, 0));
                auto step = [&](int kc, auto cur, auto nxt) __attribute__((always_inline)) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                    const int i = kc + 1 - sb;
                    const int na = pick(ca0v, ca1v, i), nb = pick(cb0v, cb1v, i);
                    Frag f0, f1;
                    if constexpr (MIA > 0 && !GG_ABL(4)) {
                        read_frag(cur, 0, f0);
                        read_frag(cur, 1, f1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (kc + 1 < sbEnd) { dma_part(nxt, PA_{}, na); dma_part(nxt, PB_{}, nb); }
                    if constexpr (GG_ABL(4)) return;
                    if constexpr (MIA > 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_frag(f0);
                        __builtin_amdgcn_sched_barrier(0);
                        read_frag(cur, 2, f0);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_frag(f1);
                        __builtin_amdgcn_sched_barrier(0);
                        read_frag(cur, 3, f1);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_frag(f0);
                        mfma_frag(f1);
                    }
                };
                for (int kc = sb; kc < sbEnd; kc += 2) {
                    step(kc, S0_{}, S1_{});
                    if (kc + 1 < sbEnd) step(kc + 1, S1_{}, S0_{});
                }
                } else {
                // ---- the pipelined sequence.  With one workgroup per CU nothing covers a bubble that all eight waves share, and a
                // barrier at a chunk boundary is one: every wave waits for the slowest, then issues its LDS-DMA pieces, then waits for
                // its first fragments (profiles/r04_v8_probe_a.log: 47 us of a 600 us main loop).  So the barrier sits INSIDE a chunk,
                // between the third and the fourth fragment group -- all LDS reads of the chunk are done by then, its stage is free and
                // the next chunk's pieces (issued a whole chunk earlier) have landed -- and the fourth group's MFMAs cover what follows:
                //     A: MFMAs of group 0          | read group 2
                //     B: MFMAs of group 1          | read group 3
                //     C: MFMAs of group 2          | wait, BARRIER, read group 0 of the NEXT chunk
                //     D: MFMAs of group 3, j by j  | LDS-DMA pieces of chunk k + 2 (into the stage just retired) between the runs,
                //                                  | read group 1 of the next chunk behind the last
                // The next chunk starts with its fragments in registers; no barrier at the boundary: 600 -> 590 us (_b.log).
                // (Measured and dropped, _c.log: ONE A set refilled block row by block row -- 8 MFMAs on a[mi], then the ds_read of the
                // next group's a[mi] -- to spread the fragment reads between the MFMAs: the two accumulators of a block row alternate
                // at distance 2 and the loop went 590 -> 603 us.)
                const int n = sbEnd - sb;
                dma_part(S0_{}, PA_{}, pick(ca0v, ca1v, 0));
                dma_part(S0_{}, PB_{}, pick(cb0v, cb1v, 0));
                if (n > 1) { dma_part(S1_{}, PA_{}, pick(ca0v, ca1v, 1)); dma_part(S1_{}, PB_{}, pick(cb0v, cb1v, 1)); }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                constexpr bool COMPUTE = MIA > 0 && !GG_ABL(4);
                Frag f0, f1;
                if constexpr (COMPUTE) {
                    read_frag(S0_{}, 0, f0);
                    read_frag(S0_{}, 1, f1);
                }
                auto mfma_j = [&](const Frag& f, int j) __attribute__((always_inline)) {
#pragma unroll
                    for (int mi = 0; mi < MIA; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[ni][j], f.a[mi][j], acc[mi][ni], 0, 0, 0);
                };
                auto step = [&](int c, auto cur, auto nxt) __attribute__((always_inline)) {
                    const bool more = c + 1 < n, more2 = c + 2 < n;        // wave-uniform
                    const int i2 = more2 ? c + 2 : c;
                    const int na = pick(ca0v, ca1v, i2), nb = pick(cb0v, cb1v, i2);
                    if constexpr (COMPUTE) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_frag(f0);                                      // A
                        __builtin_amdgcn_sched_barrier(0);
                        read_frag(cur, 2, f0);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_frag(f1);                                      // B
                        __builtin_amdgcn_sched_barrier(0);
                        read_frag(cur, 3, f1);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_frag(f0);                                      // C
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (more) {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        if constexpr (!GG_ABL(1)) __builtin_amdgcn_s_barrier();
                    }
                    // (behind the last chunk of a tile the refills read stale LDS that nothing uses: no branch around them)
                    if constexpr (COMPUTE) {
                        read_frag(nxt, 0, f0);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_j(f1, 0);                                      // D
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (more2) dma_part(cur, PA_{}, na);
                    if constexpr (COMPUTE) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_j(f1, 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (more2) dma_part(cur, PB_{}, nb);
                    if constexpr (COMPUTE) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_j(f1, 2);
                        mfma_j(f1, 3);
                        __builtin_amdgcn_sched_barrier(0);
                        read_frag(nxt, 1, f1);
                    }
                };
                for (int c = 0; c < n; c += 2) {
                    step(c, S0_{}, S1_{});
                    if (c + 1 < n) step(c + 1, S1_{}, S0_{});
                }
                }
            }
        };
        if (MIact >= 5) { if constexpr (MI >= 5) main_loop(std::integral_constant<int, 5>{}); }
        else if (MIact == 4) { if constexpr (MI >= 4) main_loop(std::integral_constant<int, 4>{}); }
        else if (MIact == 3) { if constexpr (MI >= 3) main_loop(std::integral_constant<int, 3>{}); }
        else if (MIact == 2) { if constexpr (MI >= 2) main_loop(std::integral_constant<int, 2>{}); }
        else if (MIact == 1) main_loop(std::integral_constant<int, 1>{});
        else main_loop(std::integral_constant<int, 0>{});   // a wave without a block in this tile (R = 32, wm = 1) still fetches its share of the operands and meets every barrier
        if (tid == 0) *nextTile = (int)gridDim.x + (int)pend;
        __syncthreads();                           // rowTab visible even when the k range is empty; LDS-DMA queue empty; nextTile published
        V8_STAMP(0)

        // ---- epilogue.  Transposed accumulators: lane l31 owns output ROW l31 of its 32x32 block, register r is column
        // (r & 3) + 8 (r >> 2) + 4 hi.  A wave turns its 32 x 64 block (one mi, both ni) through a PRIVATE 8.5 KB patch of the idle
        // operand stages: written as it lies in the accumulators (row pitch 272 bytes: conflict-free float4 writes), read back with
        // lane -> (row = lane / 8 + 8 pass, 8 consecutive columns = lane % 8): every load / store is 2 x 16 bytes per lane, 8 lanes
        // fill two whole lines of an output row, and the lane's columns -- hence its bias values -- are the same in every pass.  No
        // barrier: the patch is the wave's own.
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;
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
        const int cbaseE = colC[(colOk ? nbE : 0) / VSR_GG_KC] + 8 * (e_c & 3);   // float offset of the lane's first column in an output row
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
        if constexpr (GG_ABL(128)) {               // ablation: no epilogue (one store keeps the accumulators alive)
            float s_ = 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) s_ += acc[mi][ni][0] + acc[mi][ni][15];
            if (s_ == 12345.678f) C[0] = s_;
        } else
        if (vec) {
            typedef const f32x4 __attribute__((address_space(1)))* gv4;
            typedef f32x4 __attribute__((address_space(1)))* gw4;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && colOk) { b0 = *reinterpret_cast<gv4>(bias + ncolE); b1 = *reinterpret_cast<gv4>(bias + ncolE + 4); }
            const int ak = postRelu ? -1 : act;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (mi >= MIact) continue;
                const int blockRow = (wm + 2 * mi) * 32;
                // the block as it lies in the accumulators: lane (l31, hi) writes row l31, columns ni 32 + 8 q + 4 hi ..+3
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v4 = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(patch + l31 * PITCH + (ni * 32 + 8 * q + 4 * hi) * 4) = v4;
                    }
                // (the block's accumulators are dead from here on: the residual of its four passes takes their registers, every
                // load in flight before the first value is touched)
                f32x4 r0[4], r1[4];
                bool okp[4];
                int rcp[4];
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int row = blockRow + 8 * ps + e_r;
                    okp[ps] = colOk && (m0 + row) < M;
                    rcp[ps] = rowTab[row];
                    r0[ps] = f32x4{0.f, 0.f, 0.f, 0.f}; r1[ps] = r0[ps];
                    if (Rr != nullptr && okp[ps]) {
                        const gv4 pr = reinterpret_cast<gv4>(Rr + (rowTab[BM + row] + cbaseE));
                        r0[ps] = pr[0]; r1[ps] = pr[1];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is written
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(patch + (8 * ps + e_r) * PITCH + e_c * 32);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(patch + (8 * ps + e_r) * PITCH + e_c * 32 + 16);
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
                    if (Rr != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] += r0[ps][e]; v[4 + e] += r1[ps][e];
                            if (postRelu) { v[e] = fmaxf(v[e], 0.f); v[4 + e] = fmaxf(v[4 + e], 0.f); }
                        }
                    }
                    if constexpr (GG_ABL(32)) { if (v[0] == 12345.678f) C[0] = v[1]; }   // ablation: no output stores
                    else if (okp[ps]) {
                        const gw4 pw = reinterpret_cast<gw4>(C + (rcp[ps] + cbaseE));
                        pw[0] = f32x4{v[0], v[1], v[2], v[3]};
                        pw[1] = f32x4{v[4], v[5], v[6], v[7]};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next block
            }
        } else {
            // unaligned outputs or N not a multiple of 32: one value at a time, predicated, straight from the accumulators
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
                            if (ok) v += Rr[rr + cb + cofs];
                            if (postRelu) v = fmaxf(v, 0.f);
                        }
                        if constexpr (GG_ABL(32)) { if (v == 12345.678f) C[0] = v; }
                        else if (ok) C[rc + cb + cofs] = v;
                    }
                }
            }
        }
        V8_STAMP(1)
        bid = __builtin_amdgcn_readfirstlane(*nextTile);
        __syncthreads();                           // every wave has read nextTile and is out of rowTab and of its patch
    }
#ifdef GG_ABLATE
    if constexpr (GG_ABL(256)) {
        if (tid == 0 && blockIdx.x < 4096) {
            gg_dbg[blockIdx.x * 8 + 5] = __builtin_readcyclecounter() - kc0_;
            gg_dbg[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime() - kr0_;
        }
    }
#endif
}
