// Where does gather_gemm_f32_v3 lose its 29 % against the MFMA pipe?  (profiles/r03_mfma_ceiling.log: the board holds 2.39 GHz,
// bare MFMA 153 TF, the kernel's loop structure on L2-resident operands 135 TF, the kernel itself 112-119 TF.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/r03/v3_probe.hip -o video-subtitle-remover_amd/build/v3_probe
// The FFN conv of the STTN step (M = 72000 rows of a [15,34,164,CS] halo'd NHWC tensor, N = 256, K = 2304, channel-major K
// order) through the shipped kernel with
//   * the pixel stride CS and the weight row stride KS varied (1 KB / 9 KB strides put the 128 gathered lines of a chunk on few L2 channels?),
//   * every row pointing at ONE hot row (perfect caching),
//   * single mechanisms removed (results wrong on purpose): operand DMA in the loop, the chunk barrier, the epilogue,
//   * a per-tile timeline (s_memtime of wave 0) summarised per XCD.
#define GG_ABLATE 1
#include "../../video-subtitle-remover_amd/csrc/gather_gemm.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <string.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Prob {
    GGProblem* d = nullptr;
    int blocks = 0;
    double gflop = 0;
    std::vector<void*> owned;
    void free_all() { for (void* p : owned) hipFree(p); hipFree(d); }
};

template <int BM, int BN>
static Prob make(int T, int CS, int KS, bool hot, bool residual)
{
    const int H = 30, W = 160, C = 256, halo = 2, Hp = H + 2 * halo, Wp = W + 2 * halo;
    const int M = T * H * W, N = 256, K = 9 * C;
    const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
    std::vector<int32_t> rowA(tilesM * BM), colA(K / 32), rowB(tilesN * BN), colB(K / 32), rowC(tilesM * BM), colC(tilesN * BN / 32);
    for (int m = 0; m < tilesM * BM; ++m) {
        const int mm = m < M ? m : 0;
        const int t = mm / (H * W), y = (mm / W) % H, x = mm % W;
        rowC[m] = ((t * Hp + y + halo) * Wp + x + halo) * CS;
        rowA[m] = hot ? ((halo)*Wp + halo) * CS : rowC[m];
    }
    int i = 0;
    for (int c0 = 0; c0 < C; c0 += 32)
        for (int ky = -1; ky <= 1; ++ky)
            for (int kx = -1; kx <= 1; ++kx) colA[i++] = hot ? 0 : (ky * Wp + kx) * CS + c0;
    for (int n = 0; n < tilesN * BN; ++n) rowB[n] = hot ? 0 : (n < N ? n : 0) * KS;
    for (int k = 0; k < K / 32; ++k) colB[k] = hot ? 0 : 32 * k;
    for (int n = 0; n < tilesN * BN / 32; ++n) colC[n] = 32 * n;
    const size_t actElems = (size_t)T * Hp * Wp * CS;
    float *A, *B, *Cc, *bias;
    Prob pr;
    CK(hipMalloc(&A, actElems * 4)); CK(hipMalloc(&Cc, actElems * 4)); CK(hipMalloc(&B, (size_t)N * KS * 4)); CK(hipMalloc(&bias, N * 4));
    std::vector<float> hA(actElems), hB((size_t)N * KS);
    unsigned s = 12345;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.f - 1.f) * 0.02f; }
    CK(hipMemcpy(A, hA.data(), actElems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), (size_t)N * KS * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Cc, 0, actElems * 4)); CK(hipMemset(bias, 0, N * 4));
    int32_t *dRowA, *dColA, *dRowB, *dColB, *dRowC, *dColC;
#define UP(d, h) CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); pr.owned.push_back(d)
    UP(dRowA, rowA); UP(dColA, colA); UP(dRowB, rowB); UP(dColB, colB); UP(dRowC, rowC); UP(dColC, colC);
    pr.owned.push_back(A); pr.owned.push_back(B); pr.owned.push_back(Cc); pr.owned.push_back(bias);
    GGProblem p{};
    p.A = A; p.B = B; p.C = Cc; p.bias = bias; p.R = residual ? A : nullptr;
    p.rowA = dRowA; p.colA = dColA; p.rowB = dRowB; p.colB = dColB; p.rowC = dRowC; p.colC = dColC; p.rowR = dRowC;
    p.M = M; p.N = N; p.K = K; p.tilesM = tilesM; p.tilesN = tilesN; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 1; p.alpha = 1.f; p.splitStride = 0;
    CK(hipMalloc(&pr.d, sizeof(p))); CK(hipMemcpy(pr.d, &p, sizeof(p), hipMemcpyHostToDevice));
    pr.blocks = tilesM * tilesN;
    pr.gflop = 2.0 * M * N * (double)K / 1e9;
    return pr;
}

// plain dense problem: A [M][K] row-major, B [N][K], C [M][N]
template <int BM, int BN>
static Prob make_dense(int M, int N, int K)
{
    const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
    std::vector<int32_t> rowA(tilesM * BM), colA(K / 32), rowB(tilesN * BN), colB(K / 32), rowC(tilesM * BM), colC(tilesN * BN / 32);
    for (int m = 0; m < tilesM * BM; ++m) { rowA[m] = (m < M ? m : 0) * K; rowC[m] = (m < M ? m : 0) * N; }
    for (int k = 0; k < K / 32; ++k) { colA[k] = 32 * k; colB[k] = 32 * k; }
    for (int n = 0; n < tilesN * BN; ++n) rowB[n] = (n < N ? n : 0) * K;
    for (int n = 0; n < tilesN * BN / 32; ++n) colC[n] = 32 * n;
    float *A, *B, *Cc, *bias;
    Prob pr;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&Cc, (size_t)M * N * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&bias, N * 4));
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 999;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.f - 1.f) * 0.05f; }
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Cc, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, N * 4));
    int32_t *dRowA, *dColA, *dRowB, *dColB, *dRowC, *dColC;
    UP(dRowA, rowA); UP(dColA, colA); UP(dRowB, rowB); UP(dColB, colB); UP(dRowC, rowC); UP(dColC, colC);
    pr.owned.push_back(A); pr.owned.push_back(B); pr.owned.push_back(Cc); pr.owned.push_back(bias);
    GGProblem p{};
    p.A = A; p.B = B; p.C = Cc; p.bias = bias; p.R = nullptr;
    p.rowA = dRowA; p.colA = dColA; p.rowB = dRowB; p.colB = dColB; p.rowC = dRowC; p.colC = dColC; p.rowR = dRowC;
    p.M = M; p.N = N; p.K = K; p.tilesM = tilesM; p.tilesN = tilesN; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 0; p.alpha = 1.f; p.splitStride = 0;
    CK(hipMalloc(&pr.d, sizeof(p))); CK(hipMemcpy(pr.d, &p, sizeof(p), hipMemcpyHostToDevice));
    pr.blocks = tilesM * tilesN;
    pr.gflop = 2.0 * M * N * (double)K / 1e9;
    return pr;
}

template <int BM, int BN, int WM, int WN, int ABL>
static float time_v3(const Prob& pr, int residentPerCU, int iters = 12, int nq = 8)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, ABL>, 256, 0);
    if (residentPerCU > 0 && residentPerCU < occ) occ = residentPerCU;
    const int grid = pr.blocks < 256 * occ ? pr.blocks : 256 * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, ABL>), dim3(grid), dim3(256), 0, 0, pr.d, 1, pr.blocks, q + 8 * i, nq);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    return best;
}

template <int BM, int BN, int WM, int WN>
static void trace(const Prob& pr, int K, int nq = 8)
{
    std::vector<unsigned long long> z(1024 * 256, 0), h(1024 * 256);
    hipMemcpyToSymbol(HIP_SYMBOL(gg_trace), z.data(), z.size() * 8);
    unsigned int* q; hipMalloc(&q, 32); hipMemset(q, 0, 32);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 64>, 256, 0);
    const int grid = 256 * occ;
    hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 64>), dim3(grid), dim3(256), 0, 0, pr.d, 1, pr.blocks, q, nq);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_trace), h.size() * 8);
    hipFree(q);
    const int nch = K / 32, per = nch + 3;            // stamps per tile: start, prologue, nch chunks, epilogue
    // per XCD (blockIdx % 8): mean cycles per chunk, prologue, epilogue, tiles done, over the workgroups' FIRST TWO tiles (all resident)
    printf("    timeline (wave 0 of %d workgroups, occupancy %d/CU), cycles; pipe-bound chunk = %d\n", grid, occ, occ * (BM / WM / 32) * (BN / WN / 32) * 16 * 64);
    for (int x = 0; x < 8; ++x) {
        double cs = 0, ps = 0, es = 0, cmax = 0; long n = 0, nt = 0;
        std::vector<double> chunkAvg;
        for (int w = x; w < grid && w < 1024; w += 8) {
            const unsigned long long* s = &h[w * 256];
            for (int tile = 0; tile < (per <= 40 ? 3 : 2); ++tile) {
                if (per <= 40 && tile == 0) continue;      // short tiles: skip the first (plain-sequence) tile
                const unsigned long long* p = s + tile * per;
                if ((tile + 1) * per > 256 || !p[0] || !p[per - 1]) break;
                double sum = 0;
                for (int c = 0; c < nch; ++c) { const double dd = (double)(p[2 + c] - p[1 + c]); sum += dd; cmax = std::max(cmax, dd); }
                cs += sum; n += nch; ps += (double)(p[1] - p[0]); es += (double)(p[per - 1] - p[per - 2]); ++nt;
                chunkAvg.push_back(sum / nch);
            }
        }
        std::sort(chunkAvg.begin(), chunkAvg.end());
        if (nt) printf("      xcd %d: chunk mean %6.0f  (per-tile means min %6.0f med %6.0f max %6.0f, single max %6.0f)  prologue %6.0f  epilogue %6.0f  [%ld tiles]\n", x, cs / n,
                       chunkAvg.front(), chunkAvg[chunkAvg.size() / 2], chunkAvg.back(), cmax, ps / nt, es / nt, nt);
    }
}

template <int BM, int BN, int WM, int WN>
static void phases(const Prob& pr, int residentPerCU)
{
    std::vector<unsigned long long> z(4096 * 8, 0), h(4096 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(gg_dbg), z.data(), z.size() * 8);
    unsigned int* q; hipMalloc(&q, 32); hipMemset(q, 0, 32);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 32>, 256, 0);
    if (residentPerCU > 0 && residentPerCU < occ) occ = residentPerCU;
    const int grid = 256 * occ;
    {   // the DVFS loop settles over tens of milliseconds: the same load first, then the stamped launch
        unsigned int* q2; hipMalloc(&q2, 64 * 32); hipMemset(q2, 0, 64 * 32);
        for (int i = 0; i < 60; ++i)
            hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 0>), dim3(grid), dim3(256), 0, 0, pr.d, 1, pr.blocks, q2 + 8 * i, 8);
        hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 32>), dim3(grid), dim3(256), 0, 0, pr.d, 1, pr.blocks, q, 8);
        hipDeviceSynchronize();
        hipFree(q2);
    }
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_dbg), h.size() * 8);
    hipFree(q);
    double t[4] = {0, 0, 0, 0}, n = 0;
    for (int w = 0; w < grid; ++w) { for (int k = 0; k < 4; ++k) t[k] += (double)h[w * 8 + k]; n += (double)h[w * 8 + 4]; }
    const double all = t[0] + t[1] + t[2] + t[3];
    {
        std::vector<double> mhz;
        for (int w = 0; w < grid; ++w) if (h[w * 8 + 6]) mhz.push_back((double)h[w * 8 + 5] / (double)h[w * 8 + 6] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        printf("    shader clock over the workgroups' lifetime (s_memtime / s_memrealtime): median %.0f MHz [%.0f .. %.0f]\n", mhz[mhz.size() / 2], mhz.front(), mhz.back());
    }
    printf("    phases of wave 0 (%d workgroups/CU; cycles per chunk, all tiles): DMA issue %.0f | reads + 32 MFMA %.0f | vmcnt(0) wait %.0f | barrier wait %.0f | total %.0f (pipe-bound %d)\n",
           occ, t[0] / n, t[1] / n, t[2] / n, t[3] / n, all / n, occ * 2048);
}

// the same launch over and over: any difference between two results of identical inputs is a race
template <int BM, int BN, int WM, int WN>
static void stress(const Prob& pr, size_t cElems, float* Cdev, int launches, int nq)
{
    std::vector<float> ref(cElems), got(cElems);
    unsigned int* q; hipMalloc(&q, 64 * 8 * 4);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 0>, 256, 0);
    const int grid = pr.blocks < 256 * occ ? pr.blocks : 256 * occ;
    int bad = 0;
    for (int it = 0; it < launches; ++it) {
        if (it % 25 == 0) { hipMemset(q, 0, 64 * 8 * 4); hipMemset(Cdev, 0xff, cElems * 4); }
        hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 0>), dim3(grid), dim3(256), 0, 0, pr.d, 1, pr.blocks, q + 8 * (it % 25), nq);
        if (it % 25 == 0) {
            CK(hipMemcpy(it == 0 ? ref.data() : got.data(), Cdev, cElems * 4, hipMemcpyDeviceToHost));
            if (it > 0 && memcmp(ref.data(), got.data(), cElems * 4) != 0) {
                size_t nd = 0, first = 0;
                for (size_t i = 0; i < cElems; ++i) if (memcmp(&ref[i], &got[i], 4)) { if (!nd) first = i; ++nd; }
                printf("    launch %d: %zu values differ from launch 0 (first at %zu: %g vs %g)\n", it, nd, first, ref[first], got[first]);
                ++bad;
            }
        }
    }
    CK(hipDeviceSynchronize());
    hipFree(q);
    printf("  stress: %d launches, %d of %d compared results differ\n", launches, bad, (launches + 24) / 25 - 1);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    if (argc > 1 && (!strcmp(argv[1], "qk") || !strcmp(argv[1], "conv") || !strcmp(argv[1], "qkv"))) {
        // one shape, a few launches: for rocprofv3 --pmc passes (FETCH_SIZE, TCC hits) of exactly this problem
        constexpr int BM = 128, BN = 64, WM = 2, WN = 2;
        const bool qk = !strcmp(argv[1], "qk"), qkv = !strcmp(argv[1], "qkv");
        Prob pr = qk ? make_dense<BM, BN>(4800, 4800, 960) : qkv ? make_dense<BM, BN>(72000, 768, 256) : make<BM, BN>(15, 256, 2304, false, true);
        const int nq = argc > 2 ? atoi(argv[2]) : (qk || qkv ? 1 : 8);
        const float ms = time_v3<BM, BN, WM, WN, 0>(pr, 0, 6, nq);
        printf("%s, %d queue(s): %.1f us  %.1f TF\n", argv[1], nq, ms * 1e3, pr.gflop / ms);
        pr.free_all();
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "stress")) {
        constexpr int BM = 128, BN = 64, WM = 2, WN = 2;
        {
            Prob pr = make<BM, BN>(15, 256, 2304, false, true);
            GGProblem hp; CK(hipMemcpy(&hp, pr.d, sizeof(hp), hipMemcpyDeviceToHost));
            printf("== stress, conv T=15\n"); fflush(stdout);
            stress<BM, BN, WM, WN>(pr, (size_t)15 * 34 * 164 * 256, hp.C, 1500, 8);
            pr.free_all();
        }
        {
            Prob pr = make_dense<BM, BN>(72000, 768, 256);
            GGProblem hp; CK(hipMemcpy(&hp, pr.d, sizeof(hp), hipMemcpyDeviceToHost));
            printf("== stress, QKV shape\n"); fflush(stdout);
            stress<BM, BN, WM, WN>(pr, (size_t)72000 * 768, hp.C, 1500, 1);
            pr.free_all();
        }
        {
            Prob pr = make_dense<BM, BN>(4800, 4800, 960);
            GGProblem hp; CK(hipMemcpy(&hp, pr.d, sizeof(hp), hipMemcpyDeviceToHost));
            printf("== stress, QK^T shape\n"); fflush(stdout);
            stress<BM, BN, WM, WN>(pr, (size_t)4800 * 4800, hp.C, 1500, 1);
            pr.free_all();
        }
        return 0;
    }
    constexpr int BM = 128, BN = 64, WM = 2, WN = 2;
    const int T = 15, K = 2304;
#define RUN(label, pr, abl, occ) { const float ms = time_v3<BM, BN, WM, WN, abl>(pr, occ); printf("  %-58s %8.1f us  %6.1f TF\n", label, ms * 1e3, pr.gflop / ms); fflush(stdout); }
    for (int pass = 0; pass < 2; ++pass) {
        printf("== pass %d: 128x64 tile, T=15 conv (84.9 GFLOP), 3 workgroups / CU\n", pass);
        struct { int cs, ks; } cfgs[] = {{256, 2304}};
        for (auto c : cfgs) {
            Prob pr = make<BM, BN>(T, c.cs, c.ks, false, true);
            char lab[96]; snprintf(lab, sizeof lab, "pixel stride %d floats, weight row stride %d floats", c.cs, c.ks);
            RUN(lab, pr, 0, 0);
            pr.free_all();
        }
        {
            Prob pr = make<BM, BN>(T, 256, 2304, false, true);
            RUN("(again) shipped kernel", pr, 0, 0);
            RUN("shipped strides, 2 workgroups / CU", pr, 0, 2);
            RUN("  no tile pipelining (plain claim / tables / first chunk sequence per tile)", pr, 2048, 0);
            RUN("(again) shipped kernel", pr, 0, 0);
            RUN("  no epilogue", pr, 128, 0);
            RUN("  no operand DMA in the loop", pr, 2, 0);
            RUN("  no operand DMA, no epilogue", pr, 130, 0);
            RUN("  no chunk barrier (racy)", pr, 1, 0);
            pr.free_all();
            Prob nr = make<BM, BN>(T, 256, 2304, false, false);
            RUN("shipped strides, no residual read", nr, 0, 0);
            nr.free_all();
            Prob hot = make<BM, BN>(T, 256, 2304, true, true);
            RUN("every operand row = one hot row (perfect caching)", hot, 0, 0);
            RUN("  hot rows, no epilogue", hot, 128, 0);
            hot.free_all();
        }
    }
    for (int which = 0; which < 2; ++which) {
        // the short-K GEMMs of the step: QKV (K = 256: 8 chunks per tile) and the 4800-token QK^T (K = 960: 30 chunks)
        const int M_ = which == 0 ? 72000 : 4800, N_ = which == 0 ? 768 : 4800, K_ = which == 0 ? 256 : 960;
        Prob pr = make_dense<BM, BN>(M_, N_, K_);
        printf("== dense %d x %d x %d (%s), %d tiles of %d chunks, %.1f GFLOP\n", M_, N_, K_, which == 0 ? "QKV" : "QK^T 4800 tokens", pr.blocks, K_ / 32, pr.gflop);
        for (int rep = 0; rep < 2; ++rep) {
            { const float ms = time_v3<BM, BN, WM, WN, 0>(pr, 0, 24, 1); printf("  shipped kernel, one global queue                     %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
            { const float ms = time_v3<BM, BN, WM, WN, 0>(pr, 0, 24, 8); printf("  shipped kernel, per-XCD queues                       %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
            { const float ms = time_v3<BM, BN, WM, WN, 0>(pr, 0, 24, 0x108); printf("  shipped kernel, per-XCD queues, grouped tile order   %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
            { const float ms = time_v3<BM, BN, WM, WN, 0>(pr, 0, 24, 0x101); printf("  shipped kernel, one global queue, grouped tile order %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
            { const float ms = time_v3<BM, BN, WM, WN, 2048>(pr, 0, 24, 1); printf("  no tile pipelining, one global queue                 %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
            { const float ms = time_v3<BM, BN, WM, WN, 128>(pr, 0, 24, 1); printf("  no epilogue, one global queue                        %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
            { const float ms = time_v3<BM, BN, WM, WN, 2>(pr, 0, 24, 1); printf("  no operand DMA in the loop, one global queue         %8.1f us  %6.1f TF\n", ms * 1e3, pr.gflop / ms); }
        }
        trace<BM, BN, WM, WN>(pr, K_, 1);
        pr.free_all();
    }
    {
        Prob pr = make<BM, BN>(T, 256, 2304, false, true);
        printf("== timeline, shipped strides\n");
        trace<BM, BN, WM, WN>(pr, K);
        phases<BM, BN, WM, WN>(pr, 3);
        phases<BM, BN, WM, WN>(pr, 2);
        phases<BM, BN, WM, WN>(pr, 1);
        pr.free_all();
        Prob p2 = make<BM, BN>(T, 288, 2336, false, true);
        printf("== timeline, strides 288 / 2336\n");
        trace<BM, BN, WM, WN>(p2, K);
        p2.free_all();
        Prob hot = make<BM, BN>(T, 256, 2304, true, true);
        printf("== timeline, hot rows\n");
        trace<BM, BN, WM, WN>(hot, K);
        hot.free_all();
    }
    return 0;
}
