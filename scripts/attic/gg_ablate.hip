// Ablation microbenchmark of the gather-GEMM kernel (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/gg_ablate.hip -o /tmp/gg_ablate && /tmp/gg_ablate
// Conv-shaped problem (M = 72000 rows of a [15,34,164,256] halo'd NHWC tensor, N = 256, K = 2304,
// channel-major chunk order) timed with single mechanisms switched off (results are then wrong on
// purpose): what do barriers / global loads / LDS stores / LDS reads / table s_loads cost?
#define GG_ABLATE 1
#include "../video-subtitle-remover_amd/csrc/gather_gemm.hip"
#include "../video-subtitle-remover_amd/csrc/gather_gemm_v3.h"
#include "../video-subtitle-remover_amd/csrc/gather_gemm_v4.h"
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BM, int BN, int WM, int WN, int ABL>
static float run(const GGProblem* d, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gather_gemm_f32<BM, BN, WM, WN, VSR_BMODE_NK, ABL>), dim3(blocks), dim3(256), 0, 0, d, 1);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((gather_gemm_f32<BM, BN, WM, WN, VSR_BMODE_NK, ABL>), dim3(blocks), dim3(256), 0, 0, d, 1);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

template <int BM, int BN, int WM, int WN, int V>
static float run_v2(const GGProblem* d, int blocks, int iters, int residentPerCU)
{

    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 0>, 256, 0);
    if (residentPerCU > 0 && residentPerCU < occ) occ = residentPerCU;
    const int grid = blocks < 256 * occ ? blocks : 256 * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 0>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q + 8 * i, 8);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    printf("  v%d persistent, %d resident/CU (grid %d)                 %8.1f us  ", V, occ, grid, best * 1e3);
    return best;
}

template <int BM, int BN, int WM, int WN, int ABL>
static float run_v4(const GGProblem* d, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v4<BM, BN, WM, WN, VSR_BMODE_NK, false, ABL>, 256, 0);
    const int grid = blocks < 256 * occ ? blocks : 256 * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f32_v4<BM, BN, WM, WN, VSR_BMODE_NK, false, ABL>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q + 8 * i, 8, (unsigned int*)nullptr);
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
static int sweep(int T)
{
    const int H = 30, W = 160, C = 256, halo = 2, Hp = H + 2 * halo, Wp = W + 2 * halo;
    const int M = T * H * W, N = 256, K = 9 * C;
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
    std::vector<float> hA(actElems), hB((size_t)N * K);
    unsigned s = 12345;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 32768.f - 1.f) * 0.02f; }
    CK(hipMemcpy(A, hA.data(), actElems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Cc, 0, actElems * 4)); CK(hipMemset(bias, 0, N * 4));
#define UP(d, h) CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice))
    UP(dRowA, rowA); UP(dColA, colA); UP(dRowB, rowB); UP(dColB, colB); UP(dRowC, rowC); UP(dColC, colC);
    GGProblem p{};
    p.A = A; p.B = B; p.C = Cc; p.bias = bias; p.R = A;
    p.rowA = dRowA; p.colA = dColA; p.rowB = dRowB; p.colB = dColB; p.rowC = dRowC; p.colC = dColC; p.rowR = dRowA;
    p.M = M; p.N = N; p.K = K; p.tilesM = tilesM; p.tilesN = tilesN; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 1; p.alpha = 1.f; p.splitStride = 0;
    GGProblem* d;
    CK(hipMalloc(&d, sizeof(p))); CK(hipMemcpy(d, &p, sizeof(p), hipMemcpyHostToDevice));
    const int blocks = tilesM * tilesN;
    const double gf = 2.0 * M * N * (double)K / 1e9;
    const int it = 10;
    printf("tile %dx%d T=%d: %d workgroups, %.1f GFLOP\n", BM, BN, T, blocks, gf);
#define R(abl, what) { float ms = run<BM, BN, WM, WN, abl>(d, blocks, it); printf("  abl=%2d %-46s %8.1f us  %6.1f TF\n", abl, what, ms * 1e3, gf / ms); }
    R(0, "full kernel");
    { float ms = run_v2<BM, BN, WM, WN, 3>(d, blocks, it, 0); printf("%6.1f TF\n", gf / ms); }
    { float ms = run_v2<BM, BN, WM, WN, 3>(d, blocks, it, 2); printf("%6.1f TF\n", gf / ms); }
    { float ms = run_v2<BM, BN, WM, WN, 3>(d, blocks, it, 1); printf("%6.1f TF\n", gf / ms); }
#define R4(abl, what) { float ms = run_v4<BM, BN, WM, WN, abl>(d, blocks, it); printf("  v4 abl=%3d %-44s %8.1f us  %6.1f TF\n", abl, what, ms * 1e3, gf / ms); }
    R4(0, "split-half full kernel");
    R4(128, "no split arithmetic (bit casts)");
    R4(1, "no barriers");
    R4(2, "no global loads in loop");
    R4(4, "no LDS stores (and no conversion)");
    R4(6, "no loads, no stores");
    R4(7, "no loads/stores/barriers (LDS reads + MFMA)");
    R4(15, "MFMA only");
    if (BM == 128 && BN == 64 && false) {   // v3 timeline of a few workgroups: per-chunk durations, prologue, epilogue
        std::vector<unsigned long long> z(1024 * 256, 0), h(1024 * 256);
        hipMemcpyToSymbol(HIP_SYMBOL(gg_trace), z.data(), z.size() * 8);
        unsigned int* q; hipMalloc(&q, 32); hipMemset(q, 0, 32);
        hipLaunchKernelGGL((gather_gemm_f32_v3<BM, BN, WM, WN, VSR_BMODE_NK, 64>), dim3(768), dim3(256), 0, 0, d, 1, blocks, q, 8);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_trace), h.size() * 8);
        hipFree(q);
        const int nch = K / 32, per = nch + 3;           // stamps per tile: start, prologue, nch chunks, epilogue
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int w = 0; w < 768; ++w) for (int i = 0; i < 256; ++i) { unsigned long long v = h[w * 256 + i]; if (v) { if (v < t0) t0 = v; if (v > t1) t1 = v; } }
        printf("  v3 trace: kernel span %.1f us (100 MHz timer -> cycles are 10 ns?) raw span %llu\n", (t1 - t0) / 100.0, t1 - t0);
        for (int w : {0, 1, 300, 767}) {
            const unsigned long long* s = &h[w * 256];
            for (int tile = 0; tile < 3; ++tile) {
                const unsigned long long* p = s + tile * per;
                if (!p[0] || !p[per - 1]) break;
                double cmin = 1e18, cmax = 0, csum = 0;
                for (int c = 0; c < nch; ++c) { double dd = (double)(p[2 + c] - p[1 + c]); csum += dd; if (dd < cmin) cmin = dd; if (dd > cmax) cmax = dd; }
                printf("    wg %3d tile %d: start@%llu prologue %llu  chunks avg %.0f min %.0f max %.0f  epilogue %llu  total %llu\n", w, tile,
                       p[0] - t0, p[1] - p[0], csum / nch, cmin, cmax, p[per - 1] - p[per - 2], p[per - 1] - p[0]);
            }
        }
    }
    {   // phase timing of wave 0 of every workgroup (s_memtime), one launch
        std::vector<unsigned long long> z(4096 * 8, 0), h(4096 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(gg_dbg), z.data(), z.size() * 8);
        hipLaunchKernelGGL((gather_gemm_f32<BM, BN, WM, WN, VSR_BMODE_NK, 32>), dim3(blocks), dim3(256), 0, 0, d, 1);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_dbg), h.size() * 8);
        const char* names[5] = {"issue loads", "ds_read+MFMA", "barrier1", "vmcnt+ds_write", "barrier2"};
        const int nb = blocks < 4096 ? blocks : 4096;
        for (int grp = 0; grp < 2; ++grp) {       // first resident round vs the rest
            const int lo = grp == 0 ? 0 : 1024, hi = grp == 0 ? (nb < 768 ? nb : 768) : nb;
            if (hi <= lo) continue;
            double tot[5] = {0, 0, 0, 0, 0};
            for (int b = lo; b < hi; ++b) for (int k = 0; k < 5; ++k) tot[k] += (double)h[b * 8 + k];
            double all = 0; for (int k = 0; k < 5; ++k) all += tot[k];
            printf("  phases (wave 0, workgroups %d..%d, cycles per chunk):", lo, hi - 1);
            for (int k = 0; k < 5; ++k) printf(" %s %.0f (%.0f%%)", names[k], tot[k] / (hi - lo) / (K / 32), 100 * tot[k] / all);
            printf("\n");
        }
    }
    R(16, "chunk offsets by arithmetic (no s_load)");
    R(1, "no barriers");
    R(2, "no global loads in loop");
    R(4, "no LDS stores in loop");
    R(6, "no global loads, no LDS stores");
    R(7, "no loads/stores/barriers (LDS reads + MFMA)");
    R(15, "MFMA only (no LDS reads either)");
    R(31, "MFMA only, no s_load");
    hipFree(A); hipFree(B); hipFree(Cc); hipFree(bias); hipFree(d);
    hipFree(dRowA); hipFree(dColA); hipFree(dRowB); hipFree(dColB); hipFree(dRowC); hipFree(dColC);
    return 0;
}

int main()
{
    if (sweep<128, 128, 2, 2>(15)) return 1;
    if (sweep<128, 64, 2, 2>(15)) return 1;
    if (sweep<128, 128, 2, 2>(10)) return 1;   // 750 workgroups: a single resident round
    if (sweep<128, 64, 2, 2>(10)) return 1;
    if (sweep<256, 64, 4, 1>(15)) return 1;
    return 0;
}
