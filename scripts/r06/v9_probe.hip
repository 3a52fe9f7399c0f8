// Probe of gather_gemm_f16_v9 (scripts/r06/gather_gemm_v9.h -- NOT part of the library: the experiment lost, profiles/r06_v9_probe.log; v7's 256 x 256 tile with a four-stage ring of 32-deep stages and complementary roles of
// the two waves of a SIMD) against gather_gemm_f16_v7 on the shapes of the STTN fp16-operand mode; built HERE, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/r06/v9_probe.hip -o video-subtitle-remover_amd/build/v9_probe
// For each shape: v7, v9 with roles, v9 without roles (every wave issues first: the ring alone) on the same split-format operands, outputs
// compared word for word with v7's (same MFMAs in the same k order: identical bits expected), best-of-3 time of 10 launches each.
#define GG_ABLATE 1
#define GG_WITH_V9_PROBE 1
#include "../../video-subtitle-remover_amd/csrc/gather_gemm.hip"
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int g_cus = 256;
static int g_order = 1;          // tile order of the 256 x 256 kernel: 1 XCD-aware static slots, 0 plain (argv[1])

// how the host cuts one problem for the 256 x 256 kernel: whole rounds of 256-row tiles, then the remaining rows spread over
// one short tile per CU (tile height = roundup32(ceil(M / tilesM)), derived again in the kernel)
static void v7_split(const GGProblem& p, std::vector<GGProblem>& out)
{
    const int tilesN = (p.N + 255) / 256;
    const int perRound = g_cus / tilesN > 0 ? g_cus / tilesN : 1;       // M tiles of one round
    const int full = p.M / 256;                                         // whole 256-row tiles
    const int body = full / perRound * perRound;                        // ... in whole rounds
    GGProblem a = p;
    a.tilesN = tilesN;
    if (body > 0) {
        a.M = body * 256; a.tilesM = body;
        out.push_back(a);
    }
    const int rem = p.M - body * 256;
    if (rem > 0) {
        GGProblem b = p;
        b.tilesN = tilesN;
        b.M = rem;
        b.rowA = p.rowA + body * 256; b.rowC = p.rowC + body * 256; b.rowR = p.rowR ? p.rowR + body * 256 : nullptr;
        int rows = (rem + perRound - 1) / perRound;                     // rows per tile when every CU takes one
        rows = (rows + 31) & ~31;
        if (rows > 256) rows = 256;
        b.tilesM = (rem + rows - 1) / rows;
        out.push_back(b);
    }
}

template <int SPLIT, int ABL>
static float time_v7(const GGProblem* d, int nprobs, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * sizeof(unsigned int));
    const int grid = blocks < g_cus ? blocks : g_cus;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f16_v7<SPLIT, ABL>), dim3(grid), dim3(512), 0, 0, d, nprobs, blocks, q + i, (unsigned int*)nullptr, g_order);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    return best;
}

template <int ROLES, int ABL>
static float time_v9(const GGProblem* d, int nprobs, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * sizeof(unsigned int));
    const int grid = blocks < g_cus ? blocks : g_cus;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f16_v9<ROLES, ABL>), dim3(grid), dim3(512), 0, 0, d, nprobs, blocks, q + i, (unsigned int*)nullptr, g_order);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    return best;
}

static float time_v6(const GGProblem* d, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f16_v6<128, 64, 2, 2, 3, 0>, 256, 0);
    const int grid = blocks < g_cus * occ ? blocks : g_cus * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f16_v6<128, 64, 2, 2, 3, 0>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q + 8 * i, 8, (unsigned int*)nullptr);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    return best;
}

static float time_v5(const GGProblem* d, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v5<128, 64, 2, 2, VSR_BMODE_NK, 2, false, 0>, 256, 0);
    const int grid = blocks < g_cus * occ ? blocks : g_cus * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f32_v5<128, 64, 2, 2, VSR_BMODE_NK, 2, false, 0>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q + 8 * i, 8, (unsigned int*)nullptr);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    return best;
}

static void timeline(const GGProblem* d, int nprobs, int blocks)
{
    std::vector<unsigned long long> z(1024 * 256, 0), h(1024 * 256);
    hipMemcpyToSymbol(HIP_SYMBOL(gg_trace), z.data(), z.size() * 8);
    unsigned int* q; hipMalloc(&q, 32); hipMemset(q, 0, 32);
    const int grid = blocks < g_cus ? blocks : g_cus;
    hipLaunchKernelGGL((gather_gemm_f16_v7<1, 256>), dim3(grid), dim3(512), 0, 0, d, nprobs, blocks, q, (unsigned int*)nullptr, g_order);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_trace), h.size() * 8);
    hipFree(q);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < grid; ++w) for (int i = 0; i < 256; ++i) { unsigned long long v = h[w * 256 + i]; if (v) { if (v < t0) t0 = v; if (v > t1) t1 = v; } }
    printf("  timeline: span %.1f us (100 MHz clock)\n", (t1 - t0) / 100.0);
    double sum[3] = {0, 0, 0}; long ntile = 0;
    int hist[64] = {0};
    for (int w = 0; w < grid; ++w) {
        const unsigned long long* s = &h[w * 256];
        int nt = 0;
        for (int tile = 0; tile < 60; ++tile) {
            const unsigned long long* p = s + tile * 4;
            if (!p[0] || !p[3]) break;
            sum[0] += (double)(p[1] - p[0]); sum[1] += (double)(p[2] - p[1]); sum[2] += (double)(p[3] - p[2]);
            ++ntile; ++nt;
        }
        if (nt) { int b = (int)((s[nt * 4 - 1] - t0) / 1000); if (b > 63) b = 63; hist[b]++; }
    }
    printf("    %ld tiles: tables %.2f us, main loop %.2f us, epilogue %.2f us (averages); workgroup end times (10 us bins):", ntile,
           sum[0] / ntile / 100, sum[1] / ntile / 100, sum[2] / ntile / 100);
    for (int b = 0; b < 64; ++b) if (hist[b]) printf(" %d:%d", b * 10, hist[b]);
    printf("\n");
    for (int w : {0, 1, 100, 255}) {
        const unsigned long long* s = &h[w * 256];
        printf("    wg %3d:", w);
        for (int tile = 0; tile < 4; ++tile) {
            const unsigned long long* p = s + tile * 4;
            if (!p[0] || !p[3]) break;
            printf(" [@%.1f t%.1f l%.1f e%.1f]", (p[0] - t0) / 100.0, (p[1] - p[0]) / 100.0, (p[2] - p[1]) / 100.0, (p[3] - p[2]) / 100.0);
        }
        printf("\n");
    }
}

static unsigned g_seed = 12345;
static void fill_halves(std::vector<unsigned short>& v, unsigned short expo)
{
    for (auto& x : v) { g_seed = g_seed * 1664525u + 1013904223u; x = (unsigned short)(expo + ((g_seed >> 8) & 0x3ff) + ((g_seed >> 20) & 1) * 0x8000); }
}

#define UP(d, h) CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice))

static int compare(const float* c6, const float* c7, size_t n, const char* what)
{
    std::vector<unsigned> a(n), b(n);
    CK(hipMemcpy(a.data(), c6, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), c7, n * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nz = 0, first = 0;
    for (size_t i = 0; i < n; ++i) { if (a[i] != b[i]) { if (!diff) first = i; ++diff; } if (a[i]) ++nz; }
    printf("  %s: %zu of %zu words differ (v6 output has %zu non-zero words)%s\n", what, diff, n, nz, diff ? "   <-- MISMATCH" : "   identical");
    if (diff) printf("    first at %zu: v6 %08x v7 %08x\n", first, a[first], b[first]);
    return diff ? 1 : 0;
}

static int conv_case(int T)
{
    const int H = 30, W = 160, C = 256, halo = 2, Hp = H + 2 * halo, Wp = W + 2 * halo, N = 256;
    const int M = T * H * W, K = 9 * C;
    const int padM = (M + 255) / 256 * 256;
    std::vector<int32_t> rowA(padM), colA(K / 32), rowB(256), colB(K / 32), colC(256 / 32);
    for (int m = 0; m < padM; ++m) {
        const int mm = m < M ? m : 0;
        const int t = mm / (H * W), y = (mm / W) % H, x = mm % W;
        rowA[m] = ((t * Hp + y + halo) * Wp + x + halo) * C;
    }
    int i = 0;
    for (int c0 = 0; c0 < C; c0 += 32)
        for (int ky = -1; ky <= 1; ++ky)
            for (int kx = -1; kx <= 1; ++kx) colA[i++] = (ky * Wp + kx) * C + c0;
    for (int n = 0; n < 256; ++n) rowB[n] = (n < N ? n : 0) * K;
    for (int k = 0; k < K / 32; ++k) colB[k] = 32 * k;
    for (int n = 0; n < 256 / 32; ++n) colC[n] = 32 * n;
    const size_t actElems = (size_t)T * Hp * Wp * C;
    float *A, *B, *C6, *C7, *bias;
    int32_t *dRowA, *dColA, *dRowB, *dColB, *dColC;
    CK(hipMalloc(&A, actElems * 4)); CK(hipMalloc(&C6, actElems * 4)); CK(hipMalloc(&C7, actElems * 4));
    CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&bias, N * 4));
    std::vector<unsigned short> hA(actElems * 2), hB((size_t)N * K * 2);
    fill_halves(hA, 0x2c00); fill_halves(hB, 0x2000);
    std::vector<float> hb(N);
    for (int n = 0; n < N; ++n) hb[n] = 0.01f * (n % 17 - 8);
    CK(hipMemcpy(A, hA.data(), actElems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C6, 0, actElems * 4)); CK(hipMemset(C7, 0, actElems * 4));
    UP(dRowA, rowA); UP(dColA, colA); UP(dRowB, rowB); UP(dColB, colB); UP(dColC, colC);
    GGProblem p{};
    p.A = A; p.B = B; p.C = C6; p.bias = bias; p.R = A;
    p.rowA = dRowA; p.colA = dColA; p.rowB = dRowB; p.colB = dColB; p.rowC = dRowA; p.colC = dColC; p.rowR = dRowA;
    p.M = M; p.N = N; p.K = K; p.tilesM = (M + 127) / 128; p.tilesN = (N + 63) / 64; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 1 | VSR_ACT_OUT_SPLIT; p.alpha = 1.f; p.splitStride = 0;
    GGProblem* d6;
    CK(hipMalloc(&d6, sizeof(p))); CK(hipMemcpy(d6, &p, sizeof(p), hipMemcpyHostToDevice));
    const int blocks6 = p.tilesM * p.tilesN;
    std::vector<GGProblem> v7;
    GGProblem p7 = p; p7.C = C7;
    v7_split(p7, v7);
    int blocks7 = 0;
    for (auto& q : v7) { q.tileStart = blocks7; blocks7 += q.tilesM * q.tilesN; }
    GGProblem* d7;
    CK(hipMalloc(&d7, sizeof(p) * v7.size())); CK(hipMemcpy(d7, v7.data(), sizeof(p) * v7.size(), hipMemcpyHostToDevice));
    const double gf = 2.0 * M * N * (double)K / 1e9;
    printf("conv T=%d: M=%d N=%d K=%d, %.1f GFLOP; v6 %d tiles; v7 %d tiles in %zu problems:", T, M, N, K, gf, blocks6, blocks7, v7.size());
    for (auto& q : v7) printf(" [M=%d tilesM=%d]", q.M, q.tilesM);
    printf("\n");
    // v7 writes C7 (through d7); v9 writes C6: a second descriptor set that differs in C only
    std::vector<GGProblem> v9 = v7;
    for (auto& q : v9) q.C = C6 + (q.C - C7);
    GGProblem* d9;
    CK(hipMalloc(&d9, sizeof(p) * v9.size())); CK(hipMemcpy(d9, v9.data(), sizeof(p) * v9.size(), hipMemcpyHostToDevice));
    float ms7 = time_v7<0, 0>(d7, (int)v7.size(), blocks7, 10);
    printf("  v7 256x256 (two 64-deep stages)             %8.1f us  %7.1f TF\n", ms7 * 1e3, gf / ms7);
    float ms9 = time_v9<1, 0>(d9, (int)v9.size(), blocks7, 10);
    printf("  v9 ring + complementary roles               %8.1f us  %7.1f TF\n", ms9 * 1e3, gf / ms9);
    CK(hipDeviceSynchronize());
    int bad = compare(C7, C6, actElems, "conv output (split format), v9 roles vs v7");
    CK(hipMemset(C6, 0, actElems * 4));
    ms9 = time_v9<0, 0>(d9, (int)v9.size(), blocks7, 10);
    printf("  v9 ring, every wave issues first            %8.1f us  %7.1f TF\n", ms9 * 1e3, gf / ms9);
    CK(hipDeviceSynchronize());
    bad |= compare(C7, C6, actElems, "conv output (split format), v9 ring vs v7");
    {
        float ms = time_v9<1, 2>(d9, (int)v9.size(), blocks7, 10);      // 2: ablation hook of dma (no operand fetch) is not wired in v9: same kernel
        (void)ms;
        ms = time_v9<1, 4>(d9, (int)v9.size(), blocks7, 10);
        printf("  v9 roles: fetch + barriers only (no MFMA)   %8.1f us\n", ms * 1e3);
        ms = time_v9<1, 32 + 64>(d9, (int)v9.size(), blocks7, 10);
        printf("  v9 roles: no residual read, no stores       %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
        ms = time_v7<0, 32 + 64>(d7, (int)v7.size(), blocks7, 10);
        printf("  v7:       no residual read, no stores       %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
    }
    hipFree(d9);
    hipFree(A); hipFree(B); hipFree(C6); hipFree(C7); hipFree(bias); hipFree(d6); hipFree(d7);
    hipFree(dRowA); hipFree(dColA); hipFree(dRowB); hipFree(dColB); hipFree(dColC);
    return bad;
}

static int qk_case(int T)
{
    const int Ntok = T * 320, D = 960;                      // tokens x patch dimension (64 channels x 5 x 3)
    const int M = Ntok, N = Ntok, K = D;
    std::vector<int32_t> rowQ(M + 256), colK(K / 32), colC((N + 255) / 256 * 256 / 32), rowC(M + 256);
    for (int m = 0; m < M + 256; ++m) { rowQ[m] = (m < M ? m : 0) * D; rowC[m] = (m < M ? m : 0) * N; }
    for (int k = 0; k < K / 32; ++k) colK[k] = 32 * k;
    for (size_t n = 0; n < colC.size(); ++n) colC[n] = 32 * (int)n;
    float *Q, *Kt, *C6, *C7;
    int32_t *dRowQ, *dColK, *dColC, *dRowC;
    CK(hipMalloc(&Q, (size_t)M * D * 4)); CK(hipMalloc(&Kt, (size_t)N * D * 4));
    CK(hipMalloc(&C6, (size_t)M * N * 4)); CK(hipMalloc(&C7, (size_t)M * N * 4));
    std::vector<unsigned short> hQ((size_t)M * D * 2), hK((size_t)N * D * 2);
    fill_halves(hQ, 0x2c00); fill_halves(hK, 0x2c00);
    CK(hipMemcpy(Q, hQ.data(), hQ.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Kt, hK.data(), hK.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(C6, 0, (size_t)M * N * 4)); CK(hipMemset(C7, 0, (size_t)M * N * 4));
    UP(dRowQ, rowQ); UP(dColK, colK); UP(dColC, colC); UP(dRowC, rowC);
    GGProblem p{};
    p.A = Q; p.B = Kt; p.C = C6; p.bias = nullptr; p.R = nullptr;
    p.rowA = dRowQ; p.colA = dColK; p.rowB = dRowQ; p.colB = dColK; p.rowC = dRowC; p.colC = dColC; p.rowR = nullptr;
    p.M = M; p.N = N; p.K = K; p.tilesM = (M + 127) / 128; p.tilesN = (N + 63) / 64; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 0; p.alpha = 0.0466f; p.splitStride = 0;
    GGProblem* d6;
    CK(hipMalloc(&d6, sizeof(p))); CK(hipMemcpy(d6, &p, sizeof(p), hipMemcpyHostToDevice));
    const int blocks6 = p.tilesM * p.tilesN;
    std::vector<GGProblem> v7;
    GGProblem p7 = p; p7.C = C7;
    v7_split(p7, v7);
    int blocks7 = 0;
    for (auto& q : v7) { q.tileStart = blocks7; blocks7 += q.tilesM * q.tilesN; }
    GGProblem* d7;
    CK(hipMalloc(&d7, sizeof(p) * v7.size())); CK(hipMemcpy(d7, v7.data(), sizeof(p) * v7.size(), hipMemcpyHostToDevice));
    const double gf = 2.0 * M * N * (double)K / 1e9;
    printf("qk T=%d: M=N=%d K=%d, %.1f GFLOP; v6 %d tiles; v7 %d tiles in %zu problems:", T, M, K, gf, blocks6, blocks7, v7.size());
    for (auto& q : v7) printf(" [M=%d tilesM=%d tilesN=%d]", q.M, q.tilesM, q.tilesN);
    printf("\n");
    std::vector<GGProblem> v9 = v7;
    for (auto& q : v9) q.C = C6 + (q.C - C7);
    GGProblem* d9;
    CK(hipMalloc(&d9, sizeof(p) * v9.size())); CK(hipMemcpy(d9, v9.data(), sizeof(p) * v9.size(), hipMemcpyHostToDevice));
    float ms7 = time_v7<0, 0>(d7, (int)v7.size(), blocks7, 10);
    printf("  v7 256x256 (two 64-deep stages)             %8.1f us  %7.1f TF\n", ms7 * 1e3, gf / ms7);
    float ms9 = time_v9<1, 0>(d9, (int)v9.size(), blocks7, 10);
    printf("  v9 ring + complementary roles               %8.1f us  %7.1f TF\n", ms9 * 1e3, gf / ms9);
    CK(hipDeviceSynchronize());
    int bad = compare(C7, C6, (size_t)M * N, "scores (fp32), v9 roles vs v7");
    CK(hipMemset(C6, 0, (size_t)M * N * 4));
    ms9 = time_v9<0, 0>(d9, (int)v9.size(), blocks7, 10);
    printf("  v9 ring, every wave issues first            %8.1f us  %7.1f TF\n", ms9 * 1e3, gf / ms9);
    CK(hipDeviceSynchronize());
    bad |= compare(C7, C6, (size_t)M * N, "scores (fp32), v9 ring vs v7");
    hipFree(d9);
    hipFree(Q); hipFree(Kt); hipFree(C6); hipFree(C7); hipFree(d6); hipFree(d7);
    hipFree(dRowQ); hipFree(dColK); hipFree(dColC); hipFree(dRowC);
    return bad;
}

int main(int argc, char** argv)
{
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (argc > 1) g_order = atoi(argv[1]);
    printf("CUs: %d, tile order %d\n", g_cus, g_order);
    int bad = 0;
    for (int T : {15, 10, 3}) bad |= conv_case(T);
    for (int T : {15, 10}) bad |= qk_case(T);
    printf(bad ? "RESULT: MISMATCH\n" : "RESULT: all outputs identical\n");
    return bad;
}
