// gather_gemm_f32_v3: persistent + LDS-DMA variant of the grouped gather-GEMM (exact fp32, v_mfma_f32_32x32x2_f32).
//
// Same GGProblem semantics and tile shapes as v1; how operand tiles reach LDS and how tiles follow each other:
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
//   * persistent workgroups pulling tiles from per-XCD atomic queues with stealing (the dispatcher packs
//     a partial last round onto few CUs; one global queue scattered neighbouring tiles over all XCDs and
//     tripled the fabric reads).
// Round 3 (profiles/r03_mfma_ceiling.log, r03_v3_probe_*.log: the board holds 2.39 GHz under this kernel and a bare MFMA loop
// reaches 153 TF, so the 29 % the pipe idled were the kernel's own):
//   * a LEAN chunk loop: an operand piece is addressed as (scalar base = operand + chunk offset) + (unsigned 32-bit byte offset of
//     the lane's row), the saddr form of global_load_lds -- no vector ALU per piece; the two LDS buffers are compile-time
//     constants of a loop unrolled by two, so fragment reads carry their buffer in the offset field; loop control is scalar.
//     Every vector instruction of a wave that is not an MFMA queues behind the co-resident waves' MFMAs (about one per 64 cycles),
//     and a chunk used to carry ~45 of them;
//   * TRANSPOSED accumulators (the weight fragment is the MFMA's first operand): a lane owns an output ROW and four runs of four
//     consecutive columns, so the epilogue moves float4s with every bias / residual load issued before the first value is touched,
//     the activation is a compile-time constant of the path, and the row maxima of a score tile need one shuffle;
//   * tiles are PIPELINED through the workgroup: the next tile id is claimed (one atomic, not waited for) when a tile starts, its
//     row tables are loaded during the third chunk and its first operand chunk is fetched under the last chunk's MFMAs, so a tile
//     ends with its epilogue and the next one starts computing at once -- no queue round trip, no table round trip, no
//     first-chunk round trip between two tiles (8-chunk tiles such as the QKV GEMM spent more time there than in their MFMAs).
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

// a wave-uniform pointer / int the compiler can keep in scalar registers (values that went through LDS or a loop of loads
// are otherwise treated as divergent: every later field load of the problem descriptor becomes a vector load + v_readfirstlane)
template <typename T>
__device__ __forceinline__ T* gg_uniform_ptr(T* p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

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
    // A tile's CONTROL BLOCK in LDS (two slots, so that the next tile's can be written under the current tile): everything the
    // epilogue needs, fetched with the tile's tables (i.e. a tile ahead) instead of by a chain of dependent scalar / global loads
    // after the last MFMA:  [rowC offsets BM][rowR offsets BM][bias slice BN][colC chunk offsets BN/32][16 control words]
    constexpr int CB_BIAS = 2 * BM, CB_COLC = 2 * BM + BN, CB_CTL = CB_COLC + BN / 32;
    constexpr int CB_WORDS = (CB_CTL + 16 + 3) / 4 * 4;
    constexpr int RT_IT = (CB_WORDS + 255) / 256; // control-block words per thread
    // control words: 0 M, 1 N, 2 alpha, 3 act, 4 splitK, 5/6 C (split plane), 7/8 R as given, 9 has bias
    enum { CW_M = 0, CW_N, CW_ALPHA, CW_ACT, CW_SPLITK, CW_CLO, CW_CHI, CW_RLO, CW_RHI, CW_HASBIAS };
    // tiles follow each other without a round trip (see the header) for NK problems; the KN form keeps the plain sequence
    constexpr bool PIPE = (BMODE == VSR_BMODE_NK) && !GG_ABL(2048);
    static_assert(WM * WN == 4, "4 waves");

    // [2 operand buffers][2 control blocks][row maxima scratch BM x WN][next tile id]
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF_FLOATS + 2 * CB_WORDS + BM * WN + 4];
    int* rowTabs = reinterpret_cast<int*>(smem + 2 * BUF_FLOATS);
    float* scr = smem + 2 * BUF_FLOATS + 2 * CB_WORDS;
    volatile int* nextTile = reinterpret_cast<volatile int*>(smem + 2 * BUF_FLOATS + 2 * CB_WORDS + BM * WN);

    // nQueues: 8 or 1 tile ranges; bit 8 (VSR_GG_SWIZZLE): problems with many N tiles walk their tiles in groups of 8 M-rows, N
    // outermost inside a group (see locate())
    const bool swizzle = (nQueues & 0x100) != 0;
    nQueues &= 0xff;
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
    for (int g_ = 0; g_ < 4; ++g_) rdOff[g_] = (((2 * g_ + hi) ^ ((l31 >> 1) & 7)) << 2);

    // Tile queue: the flat tile-id space is cut into 8 contiguous ranges, one per XCD, each with its
    // own counter (queue[0..7], zeroed by the host).  A workgroup drains the range of the XCD it runs
    // on (blockIdx % 8 -- observed placement, used for L2 locality only: tiles that share A rows or
    // weight columns are neighbours in id space and so meet in one L2) and then steals from the
    // following ranges, so every id is handed out exactly once whatever the placement is.
    int qFirst = 0;                                    // ranges before (home + qFirst) are known to be empty (thread 0)
    const int qShift = nQueues == 8 ? 3 : 0;
    auto fetchTile = [&]() __attribute__((always_inline)) -> int {
        // nQueues = 8 (one per XCD) or 1 (single global queue): powers of two, no division in here
        for (; qFirst < nQueues; ++qFirst) {
            const int x = (blockIdx.x + qFirst) & (nQueues - 1);
            const int lo = (int)(((long long)totalTiles * x) >> qShift), hi_ = (int)(((long long)totalTiles * (x + 1)) >> qShift);
            if (lo < hi_) {
                // (a plain-load peek in front of the atomic was tried here: it removes the same-address atomics on exhausted
                // counters at the tail of a launch, but adds a round trip to every steal -- the fp32 bench lost 2 %,
                // profiles/r02_peek_ab.log; only the short-tile fp16 kernel v6 keeps it)
                const int i = lo + (int)atomicAdd(queue + x, 1u);
                if (i < hi_) return i;
            }
        }
        return totalTiles;
    };
    // the same claim in two halves: the atomic on the first range that may hold work goes out when a tile starts and nothing
    // depends on its answer until claim_end(), a chunk later, when it has long returned
    int clVal = 0, clHi = -1;
    auto claim_begin = [&]() __attribute__((always_inline)) {
        clHi = -1;
        if (qFirst < nQueues) {
            const int x = (blockIdx.x + qFirst) & (nQueues - 1);
            const int lo = (int)(((long long)totalTiles * x) >> qShift);
            clHi = (int)(((long long)totalTiles * (x + 1)) >> qShift);
            clVal = lo < clHi ? lo + (int)atomicAdd(queue + x, 1u) : clHi;
        }
    };
    auto claim_end = [&]() __attribute__((always_inline)) -> int {
        if (clHi >= 0) {
            if (clVal < clHi) return clVal;
            ++qFirst;                                  // that range is exhausted
        }
        return fetchTile();
    };

    // ---- where a tile id points: problem, tile coordinates, K range (all wave-uniform)
    struct Geo {
        const GGProblem* P;
        int tm, tn, split, kcBeg, kcEnd, nchunksTotal;
    };
    auto locate = [&](int bid_) __attribute__((always_inline)) -> Geo {
        // last problem whose first tile id is <= bid (tileStart is non-decreasing; binary search: a grouped launch
        // can carry thousands of problems, e.g. ProPainter's per-window / per-frame attention)
        int pi = 0;
        for (int lo_ = 0, hi_ = nprobs - 1; lo_ < hi_;) {
            const int mid_ = (lo_ + hi_ + 1) >> 1;
            if (bid_ >= probs[mid_].tileStart) lo_ = mid_; else hi_ = mid_ - 1;
            pi = lo_;
        }
        Geo q;
        q.P = probs + __builtin_amdgcn_readfirstlane(pi);
        const int tilesN = q.P->tilesN;
        const int tilesMN = q.P->tilesM * tilesN;
        const int t_ = bid_ - q.P->tileStart;
        q.split = t_ / tilesMN;
        const int rem = t_ - q.split * tilesMN;
        q.tm = rem / tilesN;
        q.tn = rem - q.tm * tilesN;
        if (swizzle && tilesN > 4) {
            // Many N tiles per row block (QK^T: 75, QKV: 12): in plain row-major order the ~96 tiles an XCD works on at a time
            // are one or two row blocks times ALL column blocks -- every B block is fetched once per row block and the 4 MB L2
            // holds a fraction of them (rocprofv3: 64 % L2 hits, 788 MB fetched per 4800-token QK^T launch against 37 MB of
            // operands, profiles/r03_pmc_l2.log).  In groups of 8 row blocks with the column block outermost, the same 96
            // tiles are an 8 x 12 patch: every A block is shared by 12 and every B block by 8 concurrent tiles.
            constexpr int G = 8;
            const int gs = G * tilesN;
            const int gid = rem / gs;
            const int r2 = rem - gid * gs;
            const int first = gid * G;
            const int gsz = q.P->tilesM - first < G ? q.P->tilesM - first : G;
            q.tn = r2 / gsz;
            q.tm = first + (r2 - q.tn * gsz);
        }
        q.nchunksTotal = q.P->K / VSR_GG_KC;
        q.kcBeg = q.split * q.P->chunksPerSplit;
        q.kcEnd = q.kcBeg + q.P->chunksPerSplit;
        if (q.kcEnd > q.nchunksTotal) q.kcEnd = q.nchunksTotal;
        q.tm = __builtin_amdgcn_readfirstlane(q.tm); q.tn = __builtin_amdgcn_readfirstlane(q.tn);
        q.split = __builtin_amdgcn_readfirstlane(q.split); q.nchunksTotal = __builtin_amdgcn_readfirstlane(q.nchunksTotal);
        q.kcBeg = __builtin_amdgcn_readfirstlane(q.kcBeg); q.kcEnd = __builtin_amdgcn_readfirstlane(q.kcEnd);
        return q;
    };
    // ---- the per-lane tables of a tile: gathered row offsets of the operand pieces this lane fetches, the first 128 chunk
    // offsets (lane i holds entry base + i; refreshed every 64 chunks), this thread's share of the output / residual row table
    // (load_tab only LOADS: any arithmetic on a loaded value would make hipcc wait for the load -- and, the counter being in
    // order, for the operand DMA issued before it -- in the middle of a chunk)
    struct Tab {
        int aoff[A_IT];
        int boff[B_IT];
        int vcolA, vcolB, vcolAn, vcolBn;
        int rt[RT_IT];
        int bcolKN;
    };
    // byte offsets of the lane's operand rows (pre-shifted: no vector ALU for them in the chunk loop), valid when `narrow`: every
    // row offset of the wave is in [0, 2^30) floats -- always, except for tensors beyond 4 GB
    struct Rows {
        unsigned aB[A_IT];
        unsigned bB[B_IT];
        bool narrow;
    };
    auto fetchCols = [&](const Geo& q, int base, int& va, int& vb) __attribute__((always_inline)) {
        const GGProblem* Pq = gg_uniform_ptr(q.P);
        const gci32 colA = (gci32)gg_uniform_ptr(Pq->colA);
        const gci32 colB = (gci32)gg_uniform_ptr(Pq->colB);
        const int idx = base + lane < q.nchunksTotal ? base + lane : q.nchunksTotal - 1;
        va = colA[idx];
        if constexpr (BMODE == VSR_BMODE_NK) vb = colB[idx]; else vb = 0;
    };
    auto load_tab = [&](const Geo& q_, Tab& tb) __attribute__((always_inline)) {
        Geo q = q_;
        q.P = gg_uniform_ptr(q_.P);
        const gci32 rowA = (gci32)gg_uniform_ptr(q.P->rowA);
        const gci32 rowB = (gci32)gg_uniform_ptr(q.P->rowB);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) tb.aoff[it] = rowA[q.tm * BM + s_r + 32 * it];
        tb.bcolKN = 0;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) tb.boff[it] = 0;
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) tb.boff[it] = rowB[q.tn * BN + s_r + 32 * it];
        } else {
            const gci32 colB = (gci32)q.P->colB;
            tb.bcolKN = colB[(q.tn * BN) / VSR_GG_KC + (k_q >> 3)];
        }
        fetchCols(q, q.kcBeg, tb.vcolA, tb.vcolB);
        fetchCols(q, q.kcBeg + 64, tb.vcolAn, tb.vcolBn);
        // this thread's words of the control block
        const gci32 rowCt = (gci32)gg_uniform_ptr(q.P->rowC);
        const gci32 rowRt = (gci32)gg_uniform_ptr(q.P->rowR);
        const int pact = q.P->act, psplitK = q.P->splitK, pN = q.P->N;
        const bool partial = psplitK > 1;
        const bool hasR = (q.P->R != nullptr) && !partial && !(pact & VSR_ACT_ROW_MAX);
        const gcf32 biasp = partial ? (gcf32) nullptr : (gcf32)gg_uniform_ptr(q.P->bias);
        const cci32 colCp = (cci32)gg_uniform_ptr(q.P->colC);
        const unsigned long long cptr = reinterpret_cast<unsigned long long>(q.P->C + (partial ? (int64_t)q.split * q.P->splitStride : (int64_t)0));
        const unsigned long long rptr = reinterpret_cast<unsigned long long>(q.P->R);
#pragma unroll
        for (int k = 0; k < RT_IT; ++k) {
            const int i = tid + 256 * k;
            int v = 0;
            if (i < BM) v = rowCt[q.tm * BM + i];
            else if (i < 2 * BM) { if (hasR) v = rowRt[q.tm * BM + i - BM]; }
            else if (i < CB_COLC) { const int n = q.tn * BN + i - CB_BIAS; if (biasp != nullptr && n < pN) v = __float_as_int(biasp[n]); }
            else if (i < CB_CTL) v = colCp[(q.tn * BN) / VSR_GG_KC + i - CB_COLC];
            else {
                const int w = i - CB_CTL;
                v = w == CW_M ? q.P->M : w == CW_N ? pN : w == CW_ALPHA ? __float_as_int(q.P->alpha) : w == CW_ACT ? pact : w == CW_SPLITK ? psplitK
                  : w == CW_CLO ? (int)(unsigned)cptr : w == CW_CHI ? (int)(unsigned)(cptr >> 32) : w == CW_RLO ? (int)(unsigned)rptr
                  : w == CW_RHI ? (int)(unsigned)(rptr >> 32) : w == CW_HASBIAS ? (biasp != nullptr ? 1 : 0) : 0;
            }
            tb.rt[k] = v;
        }
    };
    auto make_rows = [&](const Tab& tb, Rows& rw) __attribute__((always_inline)) {
        bool nar = true;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int o = tb.aoff[it] + srcSwz;
            nar = nar && ((unsigned)o < (1u << 30));
            rw.aB[it] = (unsigned)o << 2;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rw.bB[it] = 0;
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int o = tb.boff[it] + srcSwz;
                nar = nar && ((unsigned)o < (1u << 30));
                rw.bB[it] = (unsigned)o << 2;
            }
        } else {
            nar = false;                        // KN: the B row offsets change with every chunk; keep the vector form
        }
        rw.narrow = __all(nar) != 0;
    };
    auto store_rowtab = [&](const Tab& tb, int slot_) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < RT_IT; ++k) {
            const int i = tid + 256 * k;
            if (i < CB_WORDS) rowTabs[slot_ * CB_WORDS + i] = tb.rt[k];
        }
    };
    // LDS-DMA of one chunk of tile q into buffer buf (destination = wave-uniform base + lane*16).  A piece is addressed as (scalar
    // base = operand + chunk offset) + (unsigned 32-bit byte offset of the lane's row, constant over the tile) -- the saddr form of
    // global_load_lds, no vector ALU per piece.  Tensors beyond 4 GB (tb.narrow false) and the KN form, whose B rows change with
    // every chunk, take 64-bit vector addresses; the former read their row offsets from the tables again (never seen on this path).
    auto dma_pieces = [&](const Geo& q, gcf32 opA, gcf32 opB, Rows& rw, const int (&boKN)[B_IT], int bcolKN, int ca, int cb, auto bufTag) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufTag)::value;
        float* As = smem + buf * BUF_FLOATS;
        float* Bs = As + AS_FLOATS;
        typedef const char __attribute__((address_space(1)))* gcc8;
        if (rw.narrow) {
            const gcc8 baseA = (gcc8)opA + (long long)ca * 4;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                asm volatile("" : "+v"(rw.aB[it]));      // keeps the zero-extension next to the load, where hipcc folds it into the saddr form
                glds16((gcf32)(baseA + rw.aB[it]), (lds_vptr)(As + (wave * 8 + 32 * it) * 32));
            }
            if constexpr (BMODE == VSR_BMODE_NK) {
                const gcc8 baseB = (gcc8)opB + (long long)cb * 4;
#pragma unroll
                for (int it = 0; it < B_IT; ++it) {
                    asm volatile("" : "+v"(rw.bB[it]));
                    glds16((gcf32)(baseB + rw.bB[it]), (lds_vptr)(Bs + (wave * 8 + 32 * it) * 32));
                }
            }
            return;
        }
        const gci32 rowA = (gci32)q.P->rowA;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            glds16(opA + ((long long)rowA[q.tm * BM + s_r + 32 * it] + srcSwz + ca), (lds_vptr)(As + (wave * 8 + 32 * it) * 32));
        if constexpr (BMODE == VSR_BMODE_NK) {
            const gci32 rowB_ = (gci32)q.P->rowB;
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                glds16(opB + ((long long)rowB_[q.tn * BN + s_r + 32 * it] + srcSwz + cb), (lds_vptr)(Bs + (wave * 8 + 32 * it) * 32));
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                glds16(opB + (boKN[it] + bcolKN), (lds_vptr)(Bs + (wave * (64 / TPR) + RPP * it) * BN));
        }
    };

    using B0_ = std::integral_constant<int, 0>;
    using B1_ = std::integral_constant<int, 1>;
    using T_ = std::true_type;
    using F_ = std::false_type;

#ifdef GG_ABLATE
    int tr_ = 0;
    unsigned long long ph_[5] = {0, 0, 0, 0, 0}, tl_ = 0;
    const unsigned long long kc0_ = __builtin_readcyclecounter(), kr0_ = __builtin_amdgcn_s_memrealtime();   // shader clock vs 100 MHz
#endif

    // ---- first tile: the plain sequence (claim, locate, tables, first chunk)
    if (tid == 0) *nextTile = fetchTile();
    __syncthreads();
    int bid = __builtin_amdgcn_readfirstlane(*nextTile);
    Geo g{};
    Tab t{};
    int slot = 0, startBuf = 0;
    bool ready = false;            // the current tile's tables are in registers / LDS and its first chunk is in buffer startBuf

    for (;;) {
        if (bid >= totalTiles) break;
        GG_STAMP()   // tile start
        __builtin_amdgcn_s_setprio(3);     // everything that is not the MFMA stream goes first: it issues in the shadow of the other waves' MFMAs
        if (!ready) {
            __syncthreads();               // every wave has read *nextTile and left the previous epilogue (row table, scratch)
            g = locate(bid);
            load_tab(g, t);
            slot = 0;
            startBuf = 0;
            store_rowtab(t, 0);
        }
        if (tid == 0) claim_begin();       // the tile after this one: one atomic, answer picked up in chunk 1 (or after the tile)

        const GGProblem* __restrict__ P = gg_uniform_ptr(g.P);
        const int tm = g.tm, tn = g.tn, split = g.split;
        const gcf32 A = (gcf32)gg_uniform_ptr(P->A);
        const gcf32 B = (gcf32)gg_uniform_ptr(P->B);
        const gci32 rowB = (gci32)P->rowB;
        int* rowTab = rowTabs + slot * CB_WORDS;

        int boff[B_IT], boffNext[B_IT];             // KN: row offsets of this / the next chunk's B rows
#pragma unroll
        for (int it = 0; it < B_IT; ++it) { boff[it] = 0; boffNext[it] = 0; }
        Rows rows;
        make_rows(t, rows);
        const int bcolKN = t.bcolKN + 4 * (k_q & 7);
        int colBase = g.kcBeg;                                 // chunk index held by lane 0 of vcolA/vcolB
        int vcolA = t.vcolA, vcolB = t.vcolB, vcolAn = t.vcolAn, vcolBn = t.vcolBn;

        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        auto load_rowB_KN = [&](int kc, int (&dst)[B_IT]) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) dst[it] = rowB[kc * VSR_GG_KC + k_r + RPP * it];
        };
        auto dma_tile = [&](int kc, auto bufTag) __attribute__((always_inline)) {
            const int ca = __builtin_amdgcn_readlane(vcolA, kc - colBase);
            int cb = 0;
            if constexpr (BMODE == VSR_BMODE_NK) cb = __builtin_amdgcn_readlane(vcolB, kc - colBase);
            dma_pieces(g, A, B, rows, boff, bcolKN, ca, cb, bufTag);
        };
        // Fragment reads of group g (four k-steps); the MFMAs take the WEIGHT fragment as their first operand and the
        // ACTIVATION fragment as their second, i.e. they accumulate the transposed tile: lane l31 of an accumulator owns output
        // ROW l31 and its 16 registers are columns (r&3) + 8*(r>>2) + 4*hi -- four runs of four consecutive columns.
        auto read_group = [&](auto bufTag, int g_, f32x4 (&af)[MI], f32x4 (&bf)[NI]) __attribute__((always_inline)) {
            constexpr int buf = decltype(bufTag)::value;             // compile-time buffer: the reads fold it into their offset field
            const float* As = smem + buf * BUF_FLOATS;
            const float* Bs = As + AS_FLOATS;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const f32x4*>(&As[(wm * WTM + mi * 32 + l31) * 32 + rdOff[g_]]);
            if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bf[ni] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WTN + ni * 32 + l31) * 32 + rdOff[g_]]);
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bf[ni][j] = Bs[(8 * g_ + 4 * hi + j) * BN + wn * WTN + ni * 32 + l31];
            }
        };
        auto mfma_group = [&](const f32x4 (&af)[MI], const f32x4 (&bf)[NI]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[ni][j], af[mi][j], acc[mi][ni], 0, 0, 0);
        };
        // one chunk: the reads of group g+1 are in flight while the MFMAs of group g issue (two fragment sets)
        auto compute_chunk = [&](auto buf) __attribute__((always_inline)) {
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

        // ---- the tile after this one (PIPE): claimed when this tile starts, published in chunk 1, located and its tables loaded
        // in chunk 2, its first chunk fetched under the last chunk of this tile
        Geo gN{};
        Tab tN{};
        int bidN = totalTiles;
        bool haveN = false;        // gN / tN are valid: the next tile's first chunk goes out with this tile's last chunk
        bool published = false;    // *nextTile holds the claimed id

        const int kcEndU = __builtin_amdgcn_readfirstlane(g.kcEnd);    // scalar loop control
        const int kcBegU = __builtin_amdgcn_readfirstlane(g.kcBeg);
        if (!ready) {
            if (kcBegU < kcEndU) {
                if constexpr (BMODE == VSR_BMODE_KN) {
                    load_rowB_KN(kcBegU, boff);
                    if (kcBegU + 1 < kcEndU) load_rowB_KN(kcBegU + 1, boffNext);
                }
                dma_tile(kcBegU, B0_{});
            }
            __syncthreads();                       // drains the DMA (vmcnt(0)), publishes buffer 0 and the row table
        }
        __builtin_amdgcn_s_setprio(0);
        GG_STAMP()   // prologue done

        // One chunk out of buffer `cur` while the DMA of the next one fills the other buffer.  MODE (compile time, so that the steady
        // state carries none of it): 0 plain; 1 also publishes the claimed tile id; 2 also reads it, locates the next tile and issues
        // its table loads; 3 the last chunk of a pipelined tile: the next TILE's first chunk goes out instead of this tile's next.
        auto chunk = [&](int kc, auto cur, auto nxt, auto modeTag) __attribute__((always_inline)) {
            constexpr int MODE = decltype(modeTag)::value;
#ifdef GG_ABLATE
            if constexpr (GG_ABL(32)) { tl_ = __builtin_readcyclecounter(); ph_[4] += 1; }
#endif
            __builtin_amdgcn_s_setprio(3);
            if constexpr (MODE == 1) {             // (the barrier that ended chunk 0 drained the claim's atomic: nothing to wait for)
                if (tid == 0) *nextTile = claim_end();
                published = true;
            }
            if (MODE != 3 && (MODE == 2 || kc + 1 < kcEndU)) {
                if (kc + 1 - colBase >= 64) {  // next 64 table entries become current
                    colBase += 64;
                    vcolA = vcolAn; vcolB = vcolBn;
                    fetchCols(g, colBase + 64, vcolAn, vcolBn);
                }
                if constexpr (BMODE == VSR_BMODE_KN) {
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                    if (kc + 2 < kcEndU) load_rowB_KN(kc + 2, boffNext);
                }
                if constexpr (!GG_ABL(2)) dma_tile(kc + 1, nxt);         // buffer last read in iteration kc-1, fenced by its barrier
            }
            if constexpr (MODE == 2) {             // the barrier that ended chunk 1 published the id; at least one more chunk follows
                bidN = __builtin_amdgcn_readfirstlane(*nextTile);
                if (bidN < totalTiles) {
                    gN = locate(bidN);
                    if (gN.kcBeg < gN.kcEnd) {     // (an empty K range takes the plain sequence)
                        load_tab(gN, tN);
                        haveN = true;
                    }
                }
            }
            if constexpr (MODE == 3) {
                if (haveN) {
                    // the first chunk of the next tile goes into the free buffer, its row table into the free slot
                    Rows rN;
                    make_rows(tN, rN);
                    const int ca = __builtin_amdgcn_readlane(tN.vcolA, 0);
                    const int cb = __builtin_amdgcn_readlane(tN.vcolB, 0);
                    const GGProblem* PN = gg_uniform_ptr(gN.P);
                    dma_pieces(gN, (gcf32)gg_uniform_ptr(PN->A), (gcf32)gg_uniform_ptr(PN->B), rN, boff, 0, ca, cb, nxt);
                    store_rowtab(tN, slot ^ 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
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
        using M0_ = std::integral_constant<int, 0>;
        using M1_ = std::integral_constant<int, 1>;
        using M2_ = std::integral_constant<int, 2>;
        using M3_ = std::integral_constant<int, 3>;
        auto run = [&](auto s0, auto s1) __attribute__((always_inline)) {
            int kc = kcBegU;
            if (PIPE && kcEndU - kcBegU >= 4) {
                chunk(kc, s0, s1, M0_{});
                chunk(kc + 1, s1, s0, M1_{});
                chunk(kc + 2, s0, s1, M2_{});
                kc += 3;
                for (; kc + 2 < kcEndU; kc += 2) {         // the steady state
                    chunk(kc, s1, s0, M0_{});
                    chunk(kc + 1, s0, s1, M0_{});
                }
                if (kc + 1 < kcEndU) {
                    chunk(kc, s1, s0, M0_{});
                    chunk(kc + 1, s0, s1, M3_{});
                } else {
                    chunk(kc, s1, s0, M3_{});
                }
                return;
            }
            for (; kc + 1 < kcEndU; kc += 2) {
                chunk(kc, s0, s1, M0_{});
                chunk(kc + 1, s1, s0, M0_{});
            }
            if (kc < kcEndU) chunk(kc, s0, s1, M0_{});
        };
        if (startBuf == 0) run(B0_{}, B1_{}); else run(B1_{}, B0_{});

        __builtin_amdgcn_s_setprio(3);
        // ---- epilogue.  Transposed accumulators (see read_group): lane l31 owns output row l31 of its 32x32 block, register
        // r is column (r&3) + 8*(r>>2) + 4*hi.  One row offset per lane, and interior tiles whose output (and residual) rows are
        // 16-byte aligned move float4s: 4 stores per accumulator instead of 16 (the store tail is issue-bound, not bandwidth-bound).
        // (nothing below reads the problem descriptor: the control block was fetched a tile ahead)
        const float* biasL = reinterpret_cast<const float*>(rowTab + CB_BIAS);
        const int* ctl = rowTab + CB_CTL;
        const int M = __builtin_amdgcn_readfirstlane(ctl[CW_M]), N = __builtin_amdgcn_readfirstlane(ctl[CW_N]);
        const int splitK = __builtin_amdgcn_readfirstlane(ctl[CW_SPLITK]);
        const float alpha = __int_as_float(__builtin_amdgcn_readfirstlane(ctl[CW_ALPHA]));
        const int pact = __builtin_amdgcn_readfirstlane(ctl[CW_ACT]);
        const int act = pact & 0xff;
        const bool postRelu = (pact & VSR_ACT_POST_RELU) != 0;     // relu(act(..) + R): residual blocks of RAFT
        const bool partial = (splitK > 1);
        const bool rowMax = (pact & VSR_ACT_ROW_MAX) != 0;         // R is the row-maximum array of the scores, not a residual
        const bool hasBias = __builtin_amdgcn_readfirstlane(ctl[CW_HASBIAS]) != 0;
        const unsigned long long cptr = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ctl[CW_CHI]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(ctl[CW_CLO]);
        const unsigned long long rptr = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ctl[CW_RHI]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(ctl[CW_RLO]);
        const gcf32 Rgiven = (gcf32) reinterpret_cast<const float*>(rptr);
        const gcf32 R = (partial || rowMax) ? (gcf32) nullptr : Rgiven;
        const gf32 C = (gf32) reinterpret_cast<float*>(cptr);
        int ccol[NI];            // offset of column (32-block start + 4*hi) of this lane
        int ncol[NI];            // its global column index
        int lcol[NI];            // its column inside the tile
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            lcol[ni] = wn * WTN + ni * 32 + 4 * hi;
            ccol[ni] = rowTab[CB_COLC + (wn * WTN + ni * 32) / VSR_GG_KC] + 4 * hi;
            ncol[ni] = tn * BN + lcol[ni];
        }
        const bool fullTile = (tm * BM + BM <= M) && (tn * BN + BN <= N);
        auto activate = [&](float v) __attribute__((always_inline)) {
            if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
            else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
            return v;
        };
        // float4 epilogue of an interior tile.  ACTK: the activation as a compile-time constant (0 none, 1 LeakyReLU 0.2, 2 ReLU),
        // -1 = decided per element at run time (POST_RELU and anything else).  Every bias and residual load of the wave is issued
        // before the first value is touched: one exposed memory latency per tile (the first version loaded the bias inside the
        // store loop and paid eight serialised round trips, 50 k cycles per tile).
        auto epilogue_vec = [&](auto resTag, auto actTag) __attribute__((always_inline)) {
            constexpr bool HASR = decltype(resTag)::value;
            constexpr int ACTK = decltype(actTag)::value;
            typedef const f32x4 __attribute__((address_space(1)))* gv4;
            f32x4 bq[NI][4];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bq[ni][q] = *reinterpret_cast<const f32x4*>(biasL + (lcol[ni] + 8 * q));       // zeros where there is no bias
                }
            // (the residual of one 32-row block at a time: holding both blocks' took the kernel over the 168 registers that three
            // workgroups per CU allow)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * WTM + mi * 32 + l31;
                const int rc = rowTab[row];
                f32x4 rv[NI][4];
                if constexpr (HASR) {
                    const int rr = rowTab[BM + row];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[ni][q] = *reinterpret_cast<gv4>(R + (rr + ccol[ni] + 8 * q));
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
                            else if constexpr (ACTK == 2) v = fmaxf(v, 0.f);
                            else if constexpr (ACTK < 0) v = activate(v);
                            if constexpr (HASR) {
                                v += rv[ni][q][e];
                                if constexpr (ACTK < 0) { if (postRelu) v = fmaxf(v, 0.f); }
                            }
                            o[e] = v;
                        }
                        *reinterpret_cast<f32x4 __attribute__((address_space(1)))*>(C + (rc + ccol[ni] + 8 * q)) = o;
                    }
            }
        };
        // border tiles and unaligned outputs: one float at a time, predicated
        auto epilogue_scalar = [&](auto resTag) __attribute__((always_inline)) {
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
                        float v = acc[mi][ni][r] * alpha + biasL[lcol[ni] + cofs];
                        v = activate(v);
                        if constexpr (HASR) { if (ok) v += R[rr + ccol[ni] + cofs]; if (postRelu) v = fmaxf(v, 0.f); }
                        if (ok) C[rc + ccol[ni] + cofs] = v;
                    }
            }
        };
        // float4 path: an interior tile whose row / column offsets keep 16-byte alignment for every lane of this wave
        bool vec = fullTile && (cptr & 15) == 0 && (R == nullptr || (rptr & 15) == 0);
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
            // the waves side by side meet in LDS (a scratch of its own: with pipelined tiles the operand buffers already hold the
            // next tile's first chunk), then one atomic per row in a coalesced burst.  The P.V kernel subtracts it (VSR_ACT_A_EXP).
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float mx = -INFINITY;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cofs = (r & 3) + 8 * (r >> 2);
                        if (ncol[ni] + cofs < N) mx = fmaxf(mx, acc[mi][ni][r] * alpha + biasL[lcol[ni] + cofs]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (hi == 0) scr[(wm * WTM + mi * 32 + l31) * WN + wn] = mx;
            }
            __syncthreads();
            if (tid < BM && tm * BM + tid < M) {
                float mx = scr[tid * WN];
#pragma unroll
                for (int w = 1; w < WN; ++w) mx = fmaxf(mx, scr[tid * WN + w]);
                atomicMax(reinterpret_cast<unsigned int*>(reinterpret_cast<float*>(rptr)) + tm * BM + tid, f32_ordered(mx));
            }
        }
        GG_STAMP()   // epilogue done

        // ---- next tile
        if (PIPE && haveN) {
            // its tables are in registers, its row table in the other slot and its first chunk in the buffer the last chunk left
            // free (the barrier that ended the last chunk published all three): no round trip, no barrier
            startBuf ^= ((kcEndU - kcBegU) & 1);
            slot ^= 1;
            g = gN;
            t = tN;
            bid = bidN;
            ready = true;
        } else {
            // the plain sequence: publish the claimed id (if the chunk loop did not get to it), meet, read it
            __syncthreads();                       // every wave is out of the chunk loop and its epilogue
            if (tid == 0 && !published) *nextTile = claim_end();
            __syncthreads();
            bid = __builtin_amdgcn_readfirstlane(*nextTile);
            ready = false;
        }
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
