// Grouped gather-GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One launch runs a list of GGProblem descriptors (gather_gemm.h); a flat 1-D grid is
// mapped problem -> (split-K slice, M tile, N tile) with an XCD-aware bijective remap so
// that tiles sharing A rows / B columns sit on one XCD's L2.
//
// Structure (4 waves, one per SIMD; 3-4 workgroups per CU give the MFMA pipe its cover):
//   global --(16 B/lane, 128 B per gathered row chunk)--> VGPR prefetch of chunk k+1
//   LDS image A[BM][32+4], B[BN][32+4] (NK) or B[32][BN+4] (KN), single-buffered
//   fragments by ds_read_b128 (A, B-NK: conflict-free at row stride 36) / ds_read_b32 (B-KN)
//   each wave: (BM/WM)x(BN/WN) outputs as MIxNI 32x32 accumulators, 16 k-steps per chunk
//
// fp32-in/fp32-accumulate MFMA is exact f32 (an fmaf chain) and runs at the f32 vector
// rate (64 cycles per 32x32x2), i.e. one chunk is 64 MFMAs = 4096 issue cycles per wave:
// staging (8 x 16-B loads + 8 ds_write_b128 per lane) hides completely under it.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "gather_gemm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GG_LD 36  // LDS row stride (floats) of a [rows][32] image

// Pointers read out of a descriptor in memory are generic to the compiler (flat_load, which
// also ticks lgkmcnt and so fences the LDS pipeline); they are all global, say so.
typedef const float __attribute__((address_space(1)))* gcf32;
typedef float __attribute__((address_space(1)))* gf32;
typedef const int32_t __attribute__((address_space(1)))* gci32;
typedef const f32x4 __attribute__((address_space(1)))* gcf32x4;
// wave-uniform, read-only table entries (chunk offsets): constant address space -> s_load
typedef const int32_t __attribute__((address_space(4)))* cci32;

// monotone unsigned image of a float (0 sorts below every number): row maxima of the attention scores are kept with atomicMax
__device__ __forceinline__ unsigned int f32_ordered(float f)
{
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(unsigned int e)
{
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// Ablation hooks (scripts/attic/gg_ablate.hip defines GG_ABLATE and adds an ABL template argument that
// switches single mechanisms off to price them); compiled out of the product.
#ifdef GG_ABLATE
#define GG_ABL_PARAM , int ABL
#define GG_ABL(x) ((ABL & (x)) != 0)
__device__ unsigned long long gg_dbg[4096 * 8];   // 32: per-workgroup phase cycle sums (wave 0)
#define GG_T(slot)                                                                              \
    if constexpr (GG_ABL(32)) {                                                                  \
        const unsigned long long now_ = __builtin_readcyclecounter();                            \
        if (tid == 0 && bid < 4096) gg_dbg[bid * 8 + (slot)] += now_ - tlast_;                   \
        tlast_ = now_;                                                                           \
    }
#else
#define GG_T(slot)
#define GG_ABL_PARAM
#define GG_ABL(x) false
#endif
// 1: no barriers  2: no global loads in the loop  4: no LDS stores in the loop  8: no LDS fragment reads
// 16: chunk offsets by arithmetic instead of table s_loads

template <int BM, int BN, int WM, int WN, int BMODE GG_ABL_PARAM>
__global__ void __launch_bounds__(256)
gather_gemm_f32(const GGProblem* __restrict__ probs, int nprobs)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int A_IT = BM / 32;
    constexpr int LDB_KN = BN + 4;
    constexpr int TPR = BN / 4;    // KN: threads per k-row of the B tile
    constexpr int RPP = 256 / TPR; // KN: k-rows per pass
    constexpr int B_IT = (BMODE == VSR_BMODE_NK) ? (BN / 32) : (32 / RPP);
    constexpr int BS_FLOATS = (BMODE == VSR_BMODE_NK) ? BN * GG_LD : 32 * LDB_KN;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(MI >= 1 && NI >= 1, "wave tile");

    __shared__ __attribute__((aligned(16))) float smem[BM * GG_LD + BS_FLOATS];
    float* As = smem;
    float* Bs = smem + BM * GG_LD;

    // ---- which problem does this workgroup belong to (uniform) ----
    const int bid = blockIdx.x;
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
    const int tilesM = P->tilesM, tilesN = P->tilesN, splitK = P->splitK;
    const int tilesMN = tilesM * tilesN;
    const int nblk = tilesMN * splitK;
    int t = bid - P->tileStart;
    {   // XCD-aware remap: workgroups with equal (t & 7) share an XCD; give each XCD a
        // contiguous run of logical tiles (bijective for any nblk)
        const int xcd = t & 7, q = nblk >> 3, r = nblk & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
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
    const cci32 colA = (cci32)P->colA;
    const gci32 rowB = (gci32)P->rowB;
    const cci32 colB = (cci32)P->colB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- per-thread staging coordinates ----
    const int s_r = tid >> 3, s_q = tid & 7; // row-in-pass, float4-in-chunk ([rows][32] images)
    int aoff[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
        aoff[it] = rowA[tm * BM + s_r + 32 * it] + 4 * s_q;

    int boff[B_IT];      // NK: row offsets (fixed) ; KN: row offsets of the chunk being loaded
    int boffNext[B_IT];  // KN: row offsets one chunk ahead (table read is a dependent load)
    const int k_r = tid / TPR, k_q = tid % TPR; // KN: k-row-in-pass, float4-in-row
    int bcolKN = 0;
    if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            boff[it] = rowB[tn * BN + s_r + 32 * it] + 4 * s_q;
    } else {
        bcolKN = colB[(tn * BN) / VSR_GG_KC + (k_q >> 3)] + 4 * (k_q & 7);
    }

    f32x4 ra[A_IT], rb[B_IT];
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto load_rowB_KN = [&](int kc, int (&dst)[B_IT]) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            dst[it] = rowB[kc * VSR_GG_KC + k_r + RPP * it];
    };
    auto load_tile = [&](int kc) {
        const int ca = GG_ABL(16) ? kc * VSR_GG_KC : colA[kc];
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            ra[it] = *(gcf32x4)(A + (aoff[it] + ca));
        if constexpr (BMODE == VSR_BMODE_NK) {
            const int cb = GG_ABL(16) ? kc * VSR_GG_KC : colB[kc];
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                rb[it] = *(gcf32x4)(B + (boff[it] + cb));
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                rb[it] = *(gcf32x4)(B + (boff[it] + bcolKN));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            *reinterpret_cast<f32x4*>(&As[(s_r + 32 * it) * GG_LD + 4 * s_q]) = ra[it];
        if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                *reinterpret_cast<f32x4*>(&Bs[(s_r + 32 * it) * GG_LD + 4 * s_q]) = rb[it];
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                *reinterpret_cast<f32x4*>(&Bs[(k_r + RPP * it) * LDB_KN + 4 * k_q]) = rb[it];
        }
    };
    auto compute_tile = [&]() {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // lane (l31, hi) owns k = 8g + 4hi + j, j = 0..3: MFMA step j contracts the
            // pair {8g + j, 8g + 4 + j}; A and B use the same assignment.
            f32x4 af[MI], bf[NI];
            if constexpr (GG_ABL(8)) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = ra[mi % A_IT];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bf[ni] = rb[ni % B_IT];
            } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const f32x4*>(
                    &As[(wm * WTM + mi * 32 + l31) * GG_LD + 8 * g + 4 * hi]);
            if constexpr (BMODE == VSR_BMODE_NK) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bf[ni] = *reinterpret_cast<const f32x4*>(
                        &Bs[(wn * WTN + ni * 32 + l31) * GG_LD + 8 * g + 4 * hi]);
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bf[ni][j] = Bs[(8 * g + 4 * hi + j) * LDB_KN + wn * WTN + ni * 32 + l31];
            }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
        }
    };

    // ---- main loop: prefetch chunk k+1 into VGPRs while chunk k is contracted from LDS ----
    if (kcBeg < kcEnd) {
        if constexpr (BMODE == VSR_BMODE_KN) {
            load_rowB_KN(kcBeg, boff);
            if (kcBeg + 1 < kcEnd) load_rowB_KN(kcBeg + 1, boffNext);
        }
        load_tile(kcBeg);
        store_tile();
        __syncthreads();
#ifdef GG_ABLATE
        unsigned long long tlast_ = __builtin_readcyclecounter();
#endif
        for (int kc = kcBeg; kc < kcEnd; ++kc) {
            const bool hasNext = (kc + 1 < kcEnd);
            if (hasNext && !GG_ABL(2)) {
                if constexpr (BMODE == VSR_BMODE_KN) {
#pragma unroll
                    for (int it = 0; it < B_IT; ++it) boff[it] = boffNext[it];
                }
                load_tile(kc + 1);
                if constexpr (BMODE == VSR_BMODE_KN) {
                    if (kc + 2 < kcEnd) load_rowB_KN(kc + 2, boffNext);
                }
            }
            GG_T(0) // issue of the next chunk's loads
            compute_tile();
            GG_T(1) // ds_read + MFMA
            if constexpr (!GG_ABL(1)) __syncthreads();
            GG_T(2) // barrier 1 (everyone done reading LDS)
            if (hasNext && !GG_ABL(4)) store_tile();
            GG_T(3) // vmcnt wait + ds_write
            if constexpr (!GG_ABL(1)) __syncthreads();
            GG_T(4) // barrier 2
        }
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float alpha = P->alpha;
    const int act = P->act & 0xff;
        const bool postRelu = (P->act & VSR_ACT_POST_RELU) != 0;   // relu(act(..) + R): residual blocks of RAFT
    const bool partial = (splitK > 1);
    const gcf32 bias = partial ? (gcf32) nullptr : (gcf32)P->bias;
    const gcf32 R = partial ? (gcf32) nullptr : (gcf32)P->R;
    const gci32 rowC = (gci32)P->rowC;
    const cci32 colC = (cci32)P->colC;
    const gci32 rowR = (gci32)P->rowR;
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
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int m = tm * BM + row;
            const int rc = rowC[m];
            const int rr = (R != nullptr) ? rowR[m] : 0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float v = acc[mi][ni][r] * alpha + bv[ni];
                if (act == VSR_ACT_LRELU02) v = v > 0.f ? v : 0.2f * v;
                        else if (act == VSR_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (act == VSR_ACT_LRELU01) v = v > 0.f ? v : 0.1f * v;
                if (m < M && nok[ni]) {
                    if (R != nullptr) { v += R[rr + ccol[ni]]; if (postRelu) v = fmaxf(v, 0.f); }
                    C[rc + ccol[ni]] = v;
                }
            }
        }
    }
}

#include "gather_gemm_v3.h"
#ifndef GG_ABLATE
#include "gather_gemm_pvx.h"
#endif
#include "gather_gemm_v4.h"
#include "gather_gemm_v5.h"
#include "gather_gemm_v6.h"
#include "gather_gemm_v7.h"
#include "gather_gemm_narrow.h"
#ifdef GG_WITH_V9_PROBE          // scripts/r06/v9_probe.hip only: an experiment that did not make it into the library
#include "../../scripts/r06/gather_gemm_v9.h"
#endif

#ifndef GG_ABLATE
// resident workgroups for the persistent kernel: CUs x occupancy of that instantiation (cached)
template <typename K>
static int resident_blocks(K kernel, int threads = 256)
{
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, 0) != hipSuccess || occ <= 0) occ = 1;
    return cus * occ;
}

// operand buffers of the split-format kernel.  Measured on the bench workload (128x64 tiles): 2 buffers with 3
// workgroups per CU beat 3 buffers with 2 workgroups and 4 buffers with 1 -- the operand stream is bound by the
// L2 -> LDS path, which more resident workgroups keep busier than a deeper per-workgroup queue does.
static constexpr int v5_stages(int bm, int bn, int bmode)
{
    return bmode == VSR_BMODE_KN ? 2 : ((bm == 128 && bn == 64) ? 2 : 3);
}

template <int BM, int BN, int WM, int WN, int MODE, int ST>
static void launch_v5_st(const GGProblem* d_probs, int nprobs, int totalBlocks, unsigned int* queue, int nQueues,
                         unsigned int* rangeFlag, bool hiOnly, hipStream_t stream)
{
    if (hiOnly) {
        static const int resident = resident_blocks(gather_gemm_f32_v5<BM, BN, WM, WN, MODE, ST, true>);
        const int g = totalBlocks < resident ? totalBlocks : resident;
        hipLaunchKernelGGL((gather_gemm_f32_v5<BM, BN, WM, WN, MODE, ST, true>), dim3(g), dim3(256), 0, stream, d_probs,
                           nprobs, totalBlocks, queue, nQueues, rangeFlag);
    } else {
        static const int resident = resident_blocks(gather_gemm_f32_v5<BM, BN, WM, WN, MODE, ST, false>);
        const int g = totalBlocks < resident ? totalBlocks : resident;
        hipLaunchKernelGGL((gather_gemm_f32_v5<BM, BN, WM, WN, MODE, ST, false>), dim3(g), dim3(256), 0, stream, d_probs,
                           nprobs, totalBlocks, queue, nQueues, rangeFlag);
    }
}
// fp16-operand mode of NK problems: v6 (two K chunks of hi halves per LDS stage).  VSR_F16_KERNEL=5 keeps v5's HI_ONLY path (A/B
// runs), VSR_V6_STAGES = 2..4 overrides the stage count.
template <int BM, int BN, int WM, int WN, int ST>
static void launch_v6_st(const GGProblem* d_probs, int nprobs, int totalBlocks, unsigned int* queue, int nQueues, unsigned int* rangeFlag,
                         hipStream_t stream)
{
    static const int resident = resident_blocks(gather_gemm_f16_v6<BM, BN, WM, WN, ST>, WM * WN * 64);
    const int g = totalBlocks < resident ? totalBlocks : resident;
    hipLaunchKernelGGL((gather_gemm_f16_v6<BM, BN, WM, WN, ST>), dim3(g), dim3(WM * WN * 64), 0, stream, d_probs, nprobs, totalBlocks, queue, nQueues,
                       rangeFlag);
}

// How a problem is cut for the 256 x 256 kernel (one workgroup per CU: a launch runs in whole rounds).  out[0] = the body, whole
// rounds of 256-row tiles; out[1] = the remaining rows as one short tile per workgroup of a round (tile height =
// roundup32(ceil(M / tilesM)), which the kernel derives again from M and tilesM).  Returns the number of problems written (1 or 2);
// tileStart is left to the caller.
extern "C" int vsr_v7_split(const GGProblem* p, int cus, GGProblem* out)
{
    if (cus <= 0) cus = 256;
    const int tilesN = (p->N + 255) / 256;
    const int splitK = p->splitK > 0 ? p->splitK : 1;
    const int perM = tilesN * splitK;                                  // workgroups one M tile needs (N tiles x K slices)
    const int perRound = cus / perM > 0 ? cus / perM : 1;              // M tiles of one round
    // split-K problems write partial planes indexed by the split: they stay one problem (tilesM is free, see the kernel)
    const int body = splitK > 1 ? 0 : (p->M / 256) / perRound * perRound;   // whole 256-row tiles in whole rounds
    int n = 0;
    if (body > 0) {
        out[n] = *p;
        out[n].tilesN = tilesN; out[n].M = body * 256; out[n].tilesM = body;
        ++n;
    }
    const int rem = p->M - body * 256;
    if (rem > 0) {
        out[n] = *p;
        out[n].tilesN = tilesN; out[n].M = rem;
        out[n].rowA = p->rowA + (size_t)body * 256; out[n].rowC = p->rowC + (size_t)body * 256;
        out[n].rowR = p->rowR ? p->rowR + (size_t)body * 256 : nullptr;
        const int rounds = (rem + 256 * perRound - 1) / (256 * perRound);                 // rounds the remainder needs at full height
        int rows = ((rem + rounds * perRound - 1) / (rounds * perRound) + 31) & ~31;      // rows per tile when those rounds are full
        if (rows > 256) rows = 256;
        out[n].tilesM = (rem + rows - 1) / rows;
        ++n;
    }
    return n;
}
extern "C" int vsr_gg_cus(void)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
}

// fp16-operand mode, large NK problems: the 256 x 256 tile of 8 waves, one workgroup per CU (gather_gemm_v7.h)
static void launch_v7(const GGProblem* d_probs, int nprobs, int totalBlocks, unsigned int* queue, unsigned int* rangeFlag, bool hiOnly, hipStream_t stream)
{
    // tile order: 1 = XCD-aware static slots (default), 0 = first round static, then one atomic counter (VSR_V7_ORDER=0, A/B runs)
    static const int order = [] { const char* e = getenv("VSR_V7_ORDER"); return (e && atoi(e) == 0) ? 0 : 1; }();
    if (hiOnly) {
        static const int resident = resident_blocks(gather_gemm_f16_v7<0>, 512);
        const int g = totalBlocks < resident ? totalBlocks : resident;
        hipLaunchKernelGGL((gather_gemm_f16_v7<0>), dim3(g), dim3(512), 0, stream, d_probs, nprobs, totalBlocks, queue, rangeFlag, order);
    } else {
        static const int resident = resident_blocks(gather_gemm_f16_v7<1>, 512);
        const int g = totalBlocks < resident ? totalBlocks : resident;
        hipLaunchKernelGGL((gather_gemm_f16_v7<1>), dim3(g), dim3(512), 0, stream, d_probs, nprobs, totalBlocks, queue, rangeFlag, order);
    }
}

template <int BM, int BN, int WM, int WN, int MODE>
static void launch_v5(const GGProblem* d_probs, int nprobs, int totalBlocks, unsigned int* queue, int nQueues,
                      unsigned int* rangeFlag, bool hiOnly, hipStream_t stream)
{
    if constexpr (MODE == VSR_BMODE_NK && BN >= 64) {
        static const bool useV6 = [] { const char* e = getenv("VSR_F16_KERNEL"); return !(e && atoi(e) == 5); }();
        static const int st6 = [] { const char* e = getenv("VSR_V6_STAGES"); int x = e ? atoi(e) : 0; return (x < 2 || x > 4) ? 3 : x; }();
        if (hiOnly && useV6) {
            if (st6 == 2) { launch_v6_st<BM, BN, WM, WN, 2>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, stream); return; }
            if constexpr ((BM + BN) * 128 * 4 <= 150 * 1024) {
                if (st6 == 4) { launch_v6_st<BM, BN, WM, WN, 4>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, stream); return; }
            }
            launch_v6_st<BM, BN, WM, WN, 3>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, stream);
            return;
        }
    }
    // VSR_V5_STAGES = 2..4 overrides the buffer count of the 128-row NK tiles (A/B runs)
    static const int over = [] { const char* e = getenv("VSR_V5_STAGES"); int x = e ? atoi(e) : 0; return (x < 2 || x > 4) ? 0 : x; }();
    if constexpr (MODE == VSR_BMODE_NK && BM == 128) {
        // fp16 operands (a third of the MFMA work per byte): the deeper queue pays, 376 vs 360 fps on the bench
        const int st = over ? over : ((hiOnly && BN == 64) ? 3 : v5_stages(BM, BN, MODE));
        if (st == 2) launch_v5_st<BM, BN, WM, WN, MODE, 2>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, hiOnly, stream);
        else if (st == 3) launch_v5_st<BM, BN, WM, WN, MODE, 3>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, hiOnly, stream);
        else launch_v5_st<BM, BN, WM, WN, MODE, 4>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, hiOnly, stream);
    } else {
        launch_v5_st<BM, BN, WM, WN, MODE, v5_stages(BM, BN, MODE)>(d_probs, nprobs, totalBlocks, queue, nQueues, rangeFlag, hiOnly, stream);
    }
}

#define GG_LAUNCH(BM, BN, WM, WN, MODE)                                                                         \
    do {                                                                                                        \
        if (queue && (variant == 5 || variant == 6)) {                                                          \
            launch_v5<BM, BN, WM, WN, MODE>(d_probs, nprobs, totalBlocks, queue, nQueues == 8 ? 8 : 1, rangeFlag,       \
                                            variant == 6, stream);                                              \
        } else if (queue && variant == 7) {                                                                     \
            static const int resident = resident_blocks(gather_gemm_f32_v4<BM, BN, WM, WN, MODE, true>);       \
            const int g = totalBlocks < resident ? totalBlocks : resident;                                     \
            hipLaunchKernelGGL((gather_gemm_f32_v4<BM, BN, WM, WN, MODE, true>), dim3(g), block, 0, stream,    \
                               d_probs, nprobs, totalBlocks, queue, nQueues == 8 ? 8 : 1, rangeFlag);          \
        } else if (queue && variant == 4) {                                                                     \
            static const int resident = resident_blocks(gather_gemm_f32_v4<BM, BN, WM, WN, MODE, false>);      \
            const int g = totalBlocks < resident ? totalBlocks : resident;                                     \
            hipLaunchKernelGGL((gather_gemm_f32_v4<BM, BN, WM, WN, MODE, false>), dim3(g), block, 0, stream,   \
                               d_probs, nprobs, totalBlocks, queue, nQueues == 8 ? 8 : 1, rangeFlag);          \
        } else if (queue) {                                                                                     \
            static const int resident = resident_blocks(gather_gemm_f32_v3<BM, BN, WM, WN, MODE>);             \
            const int g = totalBlocks < resident ? totalBlocks : resident;                                     \
            hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, MODE>), dim3(g), block, 0, stream, d_probs, \
                               nprobs, totalBlocks, queue, ((nQueues & 0xff) == 8 ? 8 : 1) | (nQueues & 0x100));    \
        } else {                                                                                                \
            hipLaunchKernelGGL((gather_gemm_f32<BM, BN, WM, WN, MODE>), dim3(totalBlocks), block, 0, stream,   \
                               d_probs, nprobs);                                                               \
        }                                                                                                       \
    } while (0)

// variant 1 (or queue == nullptr): one workgroup per tile.  variant 3 / 4 / 5 / 6: persistent kernels pulling tile
// ids from queue[0..7] (must be 0): 3 = LDS-DMA double buffer (2 is accepted as an alias of 3),
// 4 = split-half operands on the f16 matrix cores (fp32 tensors, split in the kernel), 5 = the same arithmetic
// on SPLIT-FORMAT tensors (A, B, R and -- with VSR_ACT_OUT_SPLIT in act -- C; see gather_gemm_v5.h), 6 = variant 5
// with the fp16 hi halves alone as operands (one MFMA per product), 7 = variant 4's fp32 tensors with the operands rounded to
// fp16 in the kernel (one MFMA per product; the flow engines' fp16 mode).
// nQueues (v3): 8 = one tile range per XCD with stealing (few N tiles per A row block: neighbours share A
// through one L2), 1 = a single global queue (many N tiles per row block: spreading them over the XCDs
// avoids hammering one L2 with the same lines -- measured 101 vs 86 TF on the QKV GEMM).
// rangeFlag (variants 4, 5, 6): device word that is OR-ed with 1 when an accumulator comes out non-finite.
extern "C" int vsr_launch_gather_gemm_dev(const GGProblem* d_probs, int nprobs, int totalBlocks, int tileCfg,
                                          int bmode, unsigned int* queue, int variant, int nQueues,
                                          unsigned int* rangeFlag, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (totalBlocks <= 0) return 0;
    dim3 block(256);
    if (variant & VSR_VARIANT_A_EXP) {   // P.V of a fused attention: the KN kernel that exponentiates its A operand (gather_gemm_pvx.h)
        if ((variant & 0xff) != 1 || bmode != VSR_BMODE_KN) return -1;
        if (tileCfg == VSR_TILE_128x64) hipLaunchKernelGGL((gather_gemm_f32_aexp<128, 64, 2, 2>), dim3(totalBlocks), block, 0, stream, d_probs, nprobs);
        else if (tileCfg == VSR_TILE_128x128) hipLaunchKernelGGL((gather_gemm_f32_aexp<128, 128, 2, 2>), dim3(totalBlocks), block, 0, stream, d_probs, nprobs);
        else return -1;
        return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
    }
    if (variant <= 1) queue = nullptr;
    if (tileCfg == VSR_TILE_256x128) {         // the 8-wave tile of the fp16-operand mode
        if (bmode != VSR_BMODE_NK || variant != 6 || !queue) return -1;
        launch_v6_st<256, 128, 4, 2, 3>(d_probs, nprobs, totalBlocks, queue, nQueues == 8 ? 8 : 1, rangeFlag, stream);
        return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
    }
    if (tileCfg == VSR_TILE_256x256) {         // the 8-wave 256 x 256 tile of the split-format modes (variant 5: split-half, 6: fp16 operands; dynamic tile height, see gather_gemm_v7.h)
        if (bmode != VSR_BMODE_NK || (variant != 6 && variant != 5) || !queue) return -1;
        launch_v7(d_probs, nprobs, totalBlocks, queue, rangeFlag, variant == 6, stream);
        return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
    }
    if (tileCfg == VSR_TILE_128x128 && bmode == VSR_BMODE_NK) GG_LAUNCH(128, 128, 2, 2, VSR_BMODE_NK);
    else if (tileCfg == VSR_TILE_128x128 && bmode == VSR_BMODE_KN) GG_LAUNCH(128, 128, 2, 2, VSR_BMODE_KN);
    else if (tileCfg == VSR_TILE_256x32 && bmode == VSR_BMODE_NK) GG_LAUNCH(256, 32, 4, 1, VSR_BMODE_NK);
    else if (tileCfg == VSR_TILE_256x64 && bmode == VSR_BMODE_NK) GG_LAUNCH(256, 64, 4, 1, VSR_BMODE_NK);
    else if (tileCfg == VSR_TILE_128x64 && bmode == VSR_BMODE_NK) GG_LAUNCH(128, 64, 2, 2, VSR_BMODE_NK);
    else if (tileCfg == VSR_TILE_128x64 && bmode == VSR_BMODE_KN) GG_LAUNCH(128, 64, 2, 2, VSR_BMODE_KN);
    else
        return -1;
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}

// problems of at most four output columns (gather_gemm_narrow.h): NK, tiles of 256 rows (tilesN = 1, splitK = 1), no residual,
// N x K <= vsr_gg_narrow_cap() floats -- the caller checks (flow_engine.hip); maxN / maxK = the largest N / K of the launch
extern "C" int vsr_gg_narrow_cap(void) { return GG_NARROW_WCAP; }
extern "C" int vsr_launch_gather_gemm_narrow_dev(const GGProblem* d_probs, int nprobs, int totalBlocks, int maxN, int maxK, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (nprobs <= 0 || totalBlocks <= 0) return 0;
    const int nn = maxN <= 2 ? 2 : 4;
    if (maxN < 1 || maxN > 4 || maxK < VSR_GG_KC || (int64_t)nn * maxK > GG_NARROW_WCAP) return -1;
    const size_t lds = (size_t)nn * maxK * sizeof(float);
    if (nn == 2) hipLaunchKernelGGL((gather_gemm_f32_narrow<2>), dim3(totalBlocks), dim3(256), lds, stream, d_probs, nprobs);
    else hipLaunchKernelGGL((gather_gemm_f32_narrow<4>), dim3(totalBlocks), dim3(256), lds, stream, d_probs, nprobs);
    return hipGetLastError() == hipSuccess ? 0 : VSR_ERR_HIP;
}
#endif // GG_ABLATE
