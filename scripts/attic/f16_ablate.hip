// Ablation microbenchmark of the fp16-operand gather-GEMM (gather_gemm_f16_v6), built HERE and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/f16_ablate.hip -o video-subtitle-remover_amd/build/f16_ablate
// Conv-shaped problem of the STTN feed-forward (M = 72000 rows of a [15,34,164,256] halo'd NHWC tensor in split format,
// N = 256, K = 2304, channel-major chunk order) timed with single mechanisms switched off (results are then wrong on
// purpose): what bounds the kernel -- operand fetch (L2 -> LDS), barriers, fragment reads or the MFMAs?
#define GG_ABLATE 1
#include "../video-subtitle-remover_amd/csrc/gather_gemm.hip"
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BM, int BN, int WM, int WN, int ST, int ABL>
static float run_v6(const GGProblem* d, int blocks, int iters, int residentPerCU)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f16_v6<BM, BN, WM, WN, ST, ABL>, 256, 0);
    if (residentPerCU > 0 && residentPerCU < occ) occ = residentPerCU;
    const int grid = blocks < 256 * occ ? blocks : 256 * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f16_v6<BM, BN, WM, WN, ST, ABL>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q + 8 * i, 8, (unsigned int*)nullptr);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    printf("    [%d resident/CU] ", occ);
    return best;
}

template <int BM, int BN, int WM, int WN, int ST>
static int sweep(int T, int N)
{
    const int H = 30, W = 160, C = 256, halo = 2, Hp = H + 2 * halo, Wp = W + 2 * halo;
    const int M = T * H * W, K = 9 * C;
    const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
    std::vector<int32_t> rowA(tilesM * BM), colA(K / 32), rowB(tilesN * BN), colB(K / 32), rowC(tilesM * BM), colC(tilesN * BN / 32);
    for (int m = 0; m < tilesM * BM; ++m) {
        const int mm = m < M ? m : 0;
        const int t = mm / (H * W), y = (mm / W) % H, x = mm % W;
        rowA[m] = ((t * Hp + y + halo) * Wp + x + halo) * C;
        rowC[m] = rowA[m];
    }
    int i = 0;
    for (int c0 = 0; c0 < C; c0 += 32)
        for (int ky = -1; ky <= 1; ++ky)
            for (int kx = -1; kx <= 1; ++kx) colA[i++] = (ky * Wp + kx) * C + c0;
    for (int n = 0; n < tilesN * BN; ++n) rowB[n] = (n < N ? n : 0) * K;
    for (int k = 0; k < K / 32; ++k) colB[k] = 32 * k;
    for (int n = 0; n < tilesN * BN / 32; ++n) colC[n] = 32 * n;
    const size_t actElems = (size_t)T * Hp * Wp * C;
    float *A, *B, *Cc, *bias;
    int32_t *dRowA, *dColA, *dRowB, *dColB, *dRowC, *dColC;
    CK(hipMalloc(&A, actElems * 4)); CK(hipMalloc(&Cc, actElems * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&bias, N * 4));
    std::vector<unsigned short> hA(actElems * 2), hB((size_t)N * K * 2);     // halves: small finite values
    unsigned s = 12345;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2c00 + ((s >> 8) & 0x3ff) + ((s >> 20) & 1) * 0x8000); }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2000 + ((s >> 8) & 0x3ff) + ((s >> 20) & 1) * 0x8000); }
    CK(hipMemcpy(A, hA.data(), actElems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Cc, 0, actElems * 4)); CK(hipMemset(bias, 0, N * 4));
#define UP(d, h) CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice))
    UP(dRowA, rowA); UP(dColA, colA); UP(dRowB, rowB); UP(dColB, colB); UP(dRowC, rowC); UP(dColC, colC);
    GGProblem p{};
    p.A = A; p.B = B; p.C = Cc; p.bias = bias; p.R = A;
    p.rowA = dRowA; p.colA = dColA; p.rowB = dRowB; p.colB = dColB; p.rowC = dRowC; p.colC = dColC; p.rowR = dRowA;
    p.M = M; p.N = N; p.K = K; p.tilesM = tilesM; p.tilesN = tilesN; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 1 | VSR_ACT_OUT_SPLIT; p.alpha = 1.f; p.splitStride = 0;
    GGProblem* d;
    CK(hipMalloc(&d, sizeof(p))); CK(hipMemcpy(d, &p, sizeof(p), hipMemcpyHostToDevice));
    const int blocks = tilesM * tilesN;
    const double gf = 2.0 * M * N * (double)K / 1e9;
    const int it = 10;
    printf("v6 tile %dx%d stages %d, M=%d N=%d K=%d: %d tiles, %.1f GFLOP\n", BM, BN, ST, M, N, K, blocks, gf);
#define R6(abl, rpc, what) { float ms = run_v6<BM, BN, WM, WN, ST, abl>(d, blocks, it, rpc); printf("abl=%2d %-52s %8.1f us  %7.1f TF\n", abl, what, ms * 1e3, gf / ms); }
    R6(0, 0, "full kernel");
    R6(0, 1, "full kernel, 1 workgroup per CU");
    R6(8, 0, "operands from ONE hot chunk (TCP/L2 hits only)");
    R6(2, 0, "no operand fetch (fragment reads + MFMA + barriers)");
    R6(3, 0, "no fetch, no barriers (fragment reads + MFMA)");
    R6(18, 0, "no fetch, register operands (MFMA + barriers)");
    R6(19, 0, "MFMA only");
    R6(4, 0, "fetch + barriers, no fragment reads / MFMA");
    R6(12, 0, "fetch of one hot chunk + barriers, no MFMA");
    R6(5, 0, "fetch only (no barriers, no MFMA)");
    R6(32, 0, "full main loop, no output stores");
    R6(64, 0, "full main loop, no residual read");
    R6(96, 0, "full main loop, no residual read, no output stores");
    R6(19 + 96, 0, "MFMA only, no residual read, no output stores");
    R6(7 + 96, 0, "nothing (tile bookkeeping only)");
    for (int abl : {256}) {   // timeline of wave 0 of a few workgroups
        std::vector<unsigned long long> z(1024 * 256, 0), h(1024 * 256);
        hipMemcpyToSymbol(HIP_SYMBOL(gg_trace), z.data(), z.size() * 8);
        unsigned int* q; hipMalloc(&q, 32); hipMemset(q, 0, 32);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f16_v6<BM, BN, WM, WN, ST, 256>, 256, 0);
        const int grid = blocks < 256 * occ ? blocks : 256 * occ;
        if (abl == 256) hipLaunchKernelGGL((gather_gemm_f16_v6<BM, BN, WM, WN, ST, 256>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q, 8, (unsigned int*)nullptr);
        else hipLaunchKernelGGL((gather_gemm_f16_v6<BM, BN, WM, WN, ST, 256 + 103>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q, 8, (unsigned int*)nullptr);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_trace), h.size() * 8);
        hipFree(q);
        unsigned long long t0 = ~0ull, t1 = 0;
        const int nw = grid < 1024 ? grid : 1024;
        for (int w = 0; w < nw; ++w) for (int i = 0; i < 256; ++i) { unsigned long long v = h[w * 256 + i]; if (v) { if (v < t0) t0 = v; if (v > t1) t1 = v; } }
        printf("  trace abl=%d: span %.1f us (first stamp to last, 100 MHz clock)\n", abl, (t1 - t0) / 100.0);
        double sum[4] = {0, 0, 0, 0}; long ntile = 0;
        for (int w = 0; w < nw; ++w) {
            const unsigned long long* s = &h[w * 256];
            for (int tile = 0; tile < 60; ++tile) {
                const unsigned long long* p = s + tile * 4;
                if (!p[0] || !p[3]) break;
                sum[0] += (double)(p[1] - p[0]); sum[1] += (double)(p[2] - p[1]); sum[2] += (double)(p[3] - p[2]);
                if (tile > 0) sum[3] += (double)(p[0] - p[-1]);
                ++ntile;
            }
        }
        printf("    %ld tiles traced: tables %.2f us, main loop %.2f us, epilogue %.2f us, gap to next tile %.2f us (averages)\n", ntile,
               sum[0] / ntile / 100, sum[1] / ntile / 100, sum[2] / ntile / 100, sum[3] / ntile / 100);
        {   // end time of every workgroup: histogram (10 us bins) + the last finisher's timeline
            int hist[64] = {0}; int lastW = 0; unsigned long long lastT = 0; int maxTiles = 0, minTiles = 1000;
            for (int w = 0; w < nw; ++w) {
                const unsigned long long* s = &h[w * 256];
                int nt = 0; while (nt < 60 && s[nt * 4] && s[nt * 4 + 3]) ++nt;
                if (!nt) continue;
                const unsigned long long e = s[nt * 4 - 1];
                int b = (int)((e - t0) / 1000); if (b > 63) b = 63; hist[b]++;
                if (e > lastT) { lastT = e; lastW = w; }
                if (nt > maxTiles) maxTiles = nt; if (nt < minTiles) minTiles = nt;
            }
            printf("    tiles per workgroup %d..%d; workgroup end times (10 us bins):", minTiles, maxTiles);
            for (int b = 0; b < 64; ++b) if (hist[b]) printf(" %d-%dus:%d", b * 10, b * 10 + 10, hist[b]);
            printf("\n    last finisher wg %d:", lastW);
            const unsigned long long* s = &h[lastW * 256];
            for (int tile = 0; tile < 8; ++tile) {
                const unsigned long long* p = s + tile * 4;
                if (!p[0] || !p[3]) break;
                printf(" [@%.1f t%.1f l%.1f e%.1f]", (p[0] - t0) / 100.0, (p[1] - p[0]) / 100.0, (p[2] - p[1]) / 100.0, (p[3] - p[2]) / 100.0);
            }
            printf("\n");
        }
        for (int w : {0, 1, 300, 511}) {
            const unsigned long long* s = &h[w * 256];
            printf("    wg %3d:", w);
            for (int tile = 0; tile < 8; ++tile) {
                const unsigned long long* p = s + tile * 4;
                if (!p[0] || !p[3]) break;
                printf(" [@%.1f t%.1f l%.1f e%.1f]", (p[0] - t0) / 100.0, (p[1] - p[0]) / 100.0, (p[2] - p[1]) / 100.0, (p[3] - p[2]) / 100.0);
            }
            printf("\n");
        }
    }
    hipFree(A); hipFree(B); hipFree(Cc); hipFree(bias); hipFree(d);
    hipFree(dRowA); hipFree(dColA); hipFree(dRowB); hipFree(dColB); hipFree(dRowC); hipFree(dColC);
    return 0;
}

int main()
{
    if (sweep<128, 64, 2, 2, 3>(15, 256)) return 1;
    if (sweep<128, 128, 2, 2, 2>(15, 256)) return 1;
    return 0;
}
