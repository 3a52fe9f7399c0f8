// gather_gemm_f32_v4: the persistent gather-GEMM with SPLIT-HALF operands on the f16 matrix cores.
//
// fp32-in MFMA runs at the f32 vector rate (1/16 of the f16/bf16 MFMA rate) and, measured on this
// board, at ~1.9 GHz under load: ~120 TF is the practical ceiling of v1-v3.  v4 keeps fp32 data in
// HBM, fp32 accumulation and the GGProblem semantics, but feeds the matrix cores with each fp32
// operand split into two halves  x = hi + lo,  hi = fp16(x), lo = fp16(x - hi)  (22 significand
// bits kept, i.e. within 2 bits of fp32) and contracts
//        a*b  ~=  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (the dropped a_lo*b_lo is ~2^-22 relative)
// with three v_mfma_f32_32x32x16_f16 per 32x32x16 step: fp16 products are exact in fp32, sums are
// accumulated in fp32 by the MFMA.  3 MFMAs at 16x the rate = 5.3x the fp32-MFMA arithmetic rate.
// Range: operands must stay below 65504 in magnitude (fp16); values below 2^-14*2^-11 lose their lo part.
//
// Structure: v3's persistent tile queue and epilogue; operands are staged through registers
// (global fp32 -> split -> two ds_write_b64 per float4) into a double-buffered LDS image whose
// 128-byte row is [32 hi halves | 32 lo halves] of one 32-deep chunk -- the same bytes per row as the
// fp32 image, the same XOR swizzle of the 16-byte pieces, conflict-free ds_read_b128 fragments
// (one read = the 8 k-values a lane feeds to one MFMA).  NK problems only (convs, QKV, QK^T).
//
// HI_ONLY (kernel variant 7): the fp16-operand / fp32-accumulate arithmetic of the reference's own GPU mode for flow completion and
// the ProPainter generator (propainter_inpaint.py:140-146,249-251: `.half()` modules under autocast-free fp16).  Operands are rounded
// to fp16 on the way into LDS (the lo halves are neither computed, stored nor read), ONE v_mfma_f32_32x32x16_f16 per product; the
// tensors in HBM, bias / activation / residual and the accumulation stay fp32, so the result is at least as accurate as an fp16
// module's (which also rounds every activation it stores).  Same range guard as the split mode.
#pragma once
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WM, int WN, int BMODE, bool HI_ONLY GG_ABL_PARAM>
__global__ void __launch_bounds__(256)
gather_gemm_f32_v4(const GGProblem* __restrict__ probs, int nprobs, int totalTiles, unsigned int* __restrict__ queue, int nQueues,
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
    // LDS row (128 B) = 8 pieces of 16 B: pieces 0-3 = hi halves of k 0..31, pieces 4-7 = lo halves.
    // A thread's float4 (k = 4*s_q..+3) becomes 8 B of piece s_q>>1 (hi) and 8 B of piece 4+(s_q>>1) (lo);
    // pieces are XOR-swizzled with (row >> 1) & 7 exactly like the fp32 image.
    const int stSw = (s_r >> 1) & 7;
    const int stHi = ((((s_q >> 1)) ^ stSw) << 4) + ((s_q & 1) << 3);        // byte offset inside the row
    const int stLo = ((((s_q >> 1) + 4) ^ stSw) << 4) + ((s_q & 1) << 3);
    // fragment of MFMA step s (k = 16s .. 16s+15): lane (l31, hi) reads piece 2s+hi (hi halves) and 4+2s+hi (lo)
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
        for (int it = 0; it < A_IT; ++it) aoff[it] = rowA[tm * BM + s_r + 32 * it] + 4 * s_q;
        int boff[B_IT];
        // KN (P.V): B(k, n) has n contiguous in memory but the f16 MFMA wants 8 consecutive k per lane, so a
        // thread owns ONE column n and gathers 8 k-values of it per group (lanes of a wave read 64 consecutive
        // n of one k: coalesced 4-byte loads), splits them and writes one 16-byte hi piece + one lo piece of
        // LDS row n -- the transpose happens in registers.  Row offsets rowB[k] of 64 consecutive k (two
        // chunks) live in a VGPR (lane = k) and are picked with v_readlane, like the chunk offsets.
        constexpr int KN_PAIRS = (BN * 4) / 256 > 0 ? (BN * 4) / 256 : 1;   // (column, k-group) pairs per thread (KN tiles have BN >= 64)
        constexpr int KN_KGSTEP = 256 / BN;             // k-groups covered per pass
        const int kn_n = tid % BN, kn_kg0 = __builtin_amdgcn_readfirstlane(tid / BN);
        int bcolKN = 0;
        int vrowB = 0, vrowBn = 0, rowBaseChunk = kcBeg;
        auto fetchRows = [&](int baseChunk, int& v) {
            const int idx = baseChunk * VSR_GG_KC + lane < P->K ? baseChunk * VSR_GG_KC + lane : P->K - 1;
            v = rowB[idx];
        };
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) boff[it] = rowB[tn * BN + s_r + 32 * it] + 4 * s_q;
        } else {
            bcolKN = colB[(tn * BN) / VSR_GG_KC + (kn_n >> 5)] + (kn_n & 31);
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
        fetchCols(colBase, vcolA, vcolB);
        fetchCols(colBase + 64, vcolAn, vcolBn);

        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        f32x4 ra[A_IT], rb[B_IT];
        float rk[KN_PAIRS][8];                           // KN: 8 k-values of this thread's column per pair
        auto load_tile = [&](int kc) {
            const int ca = __builtin_amdgcn_readlane(vcolA, kc - colBase);
#pragma unroll
            for (int it = 0; it < A_IT; ++it) ra[it] = *(gcf32x4)(A + (aoff[it] + ca));
            if constexpr (BMODE == VSR_BMODE_NK) {
                const int cb = __builtin_amdgcn_readlane(vcolB, kc - colBase);
#pragma unroll
                for (int it = 0; it < B_IT; ++it) rb[it] = *(gcf32x4)(B + (boff[it] + cb));
            } else {
                if (kc - rowBaseChunk >= 2) {            // the VGPR holds the row offsets of two chunks
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
                        rk[p][j] = B[ro + bcolKN];
                    }
            }
        };
        auto split_store = [&](char* rowBase, const f32x4& v) {
            f16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (GG_ABL(128)) {             // ablation: no split arithmetic (wrong values)
                    h[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, v[j]) >> 16));
                    l[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, v[j]) & 0x3fff));
                } else {
                    h[j] = (_Float16)v[j];
                    if constexpr (!HI_ONLY) l[j] = (_Float16)(v[j] - (float)h[j]);
                }
            }
            *reinterpret_cast<f16x4*>(rowBase + stHi) = h;
            if constexpr (!HI_ONLY) *reinterpret_cast<f16x4*>(rowBase + stLo) = l;
        };
        auto store_tile = [&](int buf) {
            char* As = reinterpret_cast<char*>(smem + buf * BUF_FLOATS);
            char* Bs = As + AS_FLOATS * 4;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) split_store(As + (s_r + 32 * it) * 128, ra[it]);
            if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
                for (int it = 0; it < B_IT; ++it) split_store(Bs + (s_r + 32 * it) * 128, rb[it]);
            } else {
                const int sw = (kn_n >> 1) & 7;
#pragma unroll
                for (int p = 0; p < KN_PAIRS; ++p) {
                    const int kg = kn_kg0 + KN_KGSTEP * p;
                    f16x8 h, l;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        h[j] = (_Float16)rk[p][j];
                        if constexpr (!HI_ONLY) l[j] = (_Float16)(rk[p][j] - (float)h[j]);
                    }
                    *reinterpret_cast<f16x8*>(Bs + kn_n * 128 + ((kg ^ sw) << 4)) = h;
                    if constexpr (!HI_ONLY) *reinterpret_cast<f16x8*>(Bs + kn_n * 128 + (((4 + kg) ^ sw) << 4)) = l;
                }
            }
        };
        auto compute_step = [&](int buf, int st) {
            const char* As = reinterpret_cast<const char*>(smem + buf * BUF_FLOATS);
            const char* Bs = As + AS_FLOATS * 4;
            f16x8 ah[MI], al[MI], bh[NI], bl[NI];
            if constexpr (GG_ABL(8)) {                   // ablation: MFMA only
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) { ah[mi] = __builtin_bit_cast(f16x8, ra[mi % A_IT]); al[mi] = ah[mi]; }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) { bh[ni] = __builtin_bit_cast(f16x8, ra[(ni + 1) % A_IT]); bl[ni] = bh[ni]; }
            } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const char* row = As + (wm * WTM + mi * 32 + l31) * 128;
                ah[mi] = *reinterpret_cast<const f16x8*>(row + rdHi[st]);
                if constexpr (!HI_ONLY) al[mi] = *reinterpret_cast<const f16x8*>(row + rdLo[st]);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const char* row = Bs + (wn * WTN + ni * 32 + l31) * 128;
                bh[ni] = *reinterpret_cast<const f16x8*>(row + rdHi[st]);
                if constexpr (!HI_ONLY) bl[ni] = *reinterpret_cast<const f16x8*>(row + rdLo[st]);
            }
            }
            // small cross terms first, the hi*hi term last
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (!HI_ONLY) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    }
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        };

        if (kcBeg < kcEnd) {
            load_tile(kcBeg);
            store_tile(0);
            if (kcBeg + 1 < kcEnd) load_tile(kcBeg + 1);
            __syncthreads();
            int cur = 0;
            for (int kc = kcBeg; kc < kcEnd; ++kc) {
                // registers hold chunk kc+1 (fp32); LDS[cur] holds chunk kc (split halves)
                compute_step(cur, 0);
                if (kc + 1 < kcEnd) {
                    if constexpr (!GG_ABL(4)) store_tile(cur ^ 1);   // buffer last read in iteration kc-1, fenced by its barrier
                    if (kc + 2 < kcEnd && !GG_ABL(2)) {
                        if (kc + 2 - colBase >= 64) {
                            colBase += 64;
                            vcolA = vcolAn; vcolB = vcolBn;
                            fetchCols(colBase + 64, vcolAn, vcolBn);
                        }
                        load_tile(kc + 2);
                    }
                }
                compute_step(cur, 1);
                if constexpr (!GG_ABL(1)) __syncthreads();
                cur ^= 1;
            }
        }

        // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
        // Row offsets come from LDS, the residual reads of 16 rows are issued back to back.
        const float alpha = P->alpha;
        const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;   // relu(act(..) + R): residual blocks of RAFT
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
        // range guard: an operand beyond the fp16 range became +-inf in its hi half, which makes the
        // accumulator non-finite; the host then recomputes the chunk with the exact fp32 kernels
        bool nonFinite = false;
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
                        nonFinite |= !(__builtin_fabsf(v) <= 3.0e38f);
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
        if (rangeFlag != nullptr && __any(nonFinite) && lane == 0) atomicOr(rangeFlag, 1u);
        __syncthreads();
    }
}
