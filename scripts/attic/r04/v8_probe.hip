// Probe of gather_gemm_f32_v8 (exact fp32, up to 288 x 256 per 8-wave workgroup, one workgroup per CU) against gather_gemm_f32_v3
// (128 x 64, three workgroups per CU) on the 3x3 256 -> 256 convolutions of the STTN step; built HERE, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/r04/v8_probe.hip -o video-subtitle-remover_amd/build/v8_probe
// For each T: both kernels on the same fp32 operands (bias + LeakyReLU + residual), outputs compared word for word (same k order,
// same MFMA, same epilogue arithmetic -> identical bits expected), best-of-3 time of 10 launches each, ablations of v8 and a tile
// timeline (wave 0 of every workgroup).
#define GG_ABLATE 1
#include "../../video-subtitle-remover_amd/csrc/gather_gemm.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static int g_cus = 256;

template <int ABL>
static float time_v8(const GGProblem* d, int nprobs, int blocks, int iters)
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
            hipLaunchKernelGGL((gather_gemm_f32_v8<9, ABL>), dim3(grid), dim3(512), 0, 0, d, nprobs, blocks, q + i);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / iters < best) best = ms / iters;
    }
    hipFree(q);
    return best;
}

static float time_v3(const GGProblem* d, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned int* q;
    hipMalloc(&q, 64 * 8 * sizeof(unsigned int));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_gemm_f32_v3<128, 64, 2, 2, VSR_BMODE_NK, 0>, 256, 0);
    const int grid = blocks < g_cus * occ ? blocks : g_cus * occ;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(q, 0, 64 * 8 * sizeof(unsigned int));
        hipEventRecord(a, 0);
        for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL((gather_gemm_f32_v3<128, 64, 2, 2, VSR_BMODE_NK, 0>), dim3(grid), dim3(256), 0, 0, d, 1, blocks, q + 8 * i, 8);
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
    hipLaunchKernelGGL((gather_gemm_f32_v8<9, 256>), dim3(grid), dim3(512), 0, 0, d, nprobs, blocks, q);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gg_trace), h.size() * 8);
    hipFree(q);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < grid; ++w) for (int i = 0; i < 256; ++i) { unsigned long long v = h[w * 256 + i]; if (v) { if (v < t0) t0 = v; if (v > t1) t1 = v; } }
    {
        std::vector<unsigned long long> dbg(4096 * 8);
        hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(gg_dbg), dbg.size() * 8);
        std::vector<double> mhz;
        for (int w = 0; w < grid; ++w) if (dbg[w * 8 + 6]) mhz.push_back((double)dbg[w * 8 + 5] / (double)dbg[w * 8 + 6] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        if (!mhz.empty()) printf("  shader clock over the workgroups' lifetimes (s_memtime / s_memrealtime): median %.0f MHz [%.0f .. %.0f]\n", mhz[mhz.size() / 2], mhz.front(), mhz.back());
    }
    printf("  timeline: span %.1f us (100 MHz clock)\n", (t1 - t0) / 100.0);
    double sum[3] = {0, 0, 0}; long ntile = 0;
    int hist[128] = {0};
    for (int w = 0; w < grid; ++w) {
        const unsigned long long* s = &h[w * 256];
        int nt = 0;
        for (int tile = 0; tile < 60; ++tile) {
            const unsigned long long* p = s + tile * 4;
            if (!p[0] || !p[3]) break;
            sum[0] += (double)(p[1] - p[0]); sum[1] += (double)(p[2] - p[1]); sum[2] += (double)(p[3] - p[2]);
            ++ntile; ++nt;
        }
        if (nt) { int b = (int)((s[nt * 4 - 1] - t0) / 1000); if (b > 127) b = 127; hist[b]++; }
    }
    printf("    %ld tiles: tables %.2f us, main loop %.2f us, epilogue %.2f us (averages); workgroup end times (10 us bins):", ntile,
           sum[0] / ntile / 100, sum[1] / ntile / 100, sum[2] / ntile / 100);
    for (int b = 0; b < 128; ++b) if (hist[b]) printf(" %d:%d", b * 10, hist[b]);
    printf("\n");
    for (int w : {0, 1, 100, 249}) {
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
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return (float)((g_seed >> 8) & 0xffff) / 32768.f - 1.f; }

#define UP(d, h) CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice))

static int compare(const float* c3, const float* c8, size_t n, const char* what)
{
    std::vector<unsigned> a(n), b(n);
    CK(hipMemcpy(a.data(), c3, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), c8, n * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nz = 0, first = 0;
    for (size_t i = 0; i < n; ++i) { if (a[i] != b[i]) { if (!diff) first = i; ++diff; } if (a[i]) ++nz; }
    printf("  %s: %zu of %zu words differ (v3 output has %zu non-zero words)%s\n", what, diff, n, nz, diff ? "   <-- MISMATCH" : "   identical");
    if (diff) printf("    first at %zu: v3 %08x v8 %08x\n", first, a[first], b[first]);
    return diff ? 1 : 0;
}

static int conv_case(int T, bool full)
{
    const int H = 30, W = 160, C = 256, halo = 2, Hp = H + 2 * halo, Wp = W + 2 * halo, N = 256;
    const int M = T * H * W, K = 9 * C;
    const int padM = (M + 511) / 512 * 512;
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
    float *A, *B, *C3, *C8, *bias;
    int32_t *dRowA, *dColA, *dRowB, *dColB, *dColC;
    CK(hipMalloc(&A, actElems * 4)); CK(hipMalloc(&C3, actElems * 4)); CK(hipMalloc(&C8, actElems * 4));
    CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&bias, N * 4));
    {
        std::vector<float> hA(actElems, 0.f), hB((size_t)N * K);
        for (int t = 0; t < T; ++t)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float* p = &hA[(((size_t)t * Hp + y + halo) * Wp + x + halo) * C];
                    for (int c = 0; c < C; ++c) p[c] = frand();
                }
        for (auto& x : hB) x = frand() * 0.02f;
        std::vector<float> hb(N);
        for (int n = 0; n < N; ++n) hb[n] = 0.01f * (n % 17 - 8);
        CK(hipMemcpy(A, hA.data(), actElems * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(C3, 0, actElems * 4)); CK(hipMemset(C8, 0, actElems * 4));
    UP(dRowA, rowA); UP(dColA, colA); UP(dRowB, rowB); UP(dColB, colB); UP(dColC, colC);
    GGProblem p{};
    p.A = A; p.B = B; p.C = C3; p.bias = bias; p.R = A;
    p.rowA = dRowA; p.colA = dColA; p.rowB = dRowB; p.colB = dColB; p.rowC = dRowA; p.colC = dColC; p.rowR = dRowA;
    p.M = M; p.N = N; p.K = K; p.tilesM = (M + 127) / 128; p.tilesN = (N + 63) / 64; p.splitK = 1; p.chunksPerSplit = K / 32; p.tileStart = 0;
    p.act = 1; p.alpha = 1.f; p.splitStride = 0;
    GGProblem* d3;
    CK(hipMalloc(&d3, sizeof(p))); CK(hipMemcpy(d3, &p, sizeof(p), hipMemcpyHostToDevice));
    const int blocks3 = p.tilesM * p.tilesN;
    GGProblem p8 = p; p8.C = C8;
    vsr_v8_split(&p8, g_cus, &p8);
    p8.tileStart = 0;
    const int blocks8 = p8.tilesM * p8.tilesN;
    GGProblem* d8;
    CK(hipMalloc(&d8, sizeof(p))); CK(hipMemcpy(d8, &p8, sizeof(p), hipMemcpyHostToDevice));
    const double gf = 2.0 * M * N * (double)K / 1e9;
    const int R = (((M + p8.tilesM - 1) / p8.tilesM) + 31) & ~31;
    printf("conv T=%d: M=%d N=%d K=%d, %.1f GFLOP; v3 %d tiles; v8 %d tiles of %d rows\n", T, M, N, K, gf, blocks3, blocks8, R);
    float ms3 = time_v3(d3, blocks3, 10);
    printf("  v3 128x64 x3            %8.1f us  %7.1f TF\n", ms3 * 1e3, gf / ms3);
    float ms8 = time_v8<0>(d8, 1, blocks8, 10);
    printf("  v8 288x256              %8.1f us  %7.1f TF\n", ms8 * 1e3, gf / ms8);
    CK(hipDeviceSynchronize());
    int bad = compare(C3, C8, actElems, "conv output");
    {
        float msp = time_v8<512>(d8, 1, blocks8, 10);
        printf("  v8, barrier at the chunk boundary (plain sequence)   %8.1f us  %7.1f TF\n", msp * 1e3, gf / msp);
    }
    if (full) {
        float ms = time_v8<2>(d8, 1, blocks8, 10);
        printf("  v8, no operand fetch                      %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
        ms = time_v8<8>(d8, 1, blocks8, 10);
        printf("  v8, one hot chunk (L2 hits)               %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
        ms = time_v8<4>(d8, 1, blocks8, 10);
        printf("  v8, fetch + barriers only                 %8.1f us\n", ms * 1e3);
        ms = time_v8<128>(d8, 1, blocks8, 10);
        printf("  v8, no epilogue                           %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
        ms = time_v8<32 + 64>(d8, 1, blocks8, 10);
        printf("  v8, no residual read, no stores           %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
        ms = time_v8<1>(d8, 1, blocks8, 10);
        printf("  v8, no barriers (results invalid)         %8.1f us  %7.1f TF\n", ms * 1e3, gf / ms);
    }
    timeline(d8, 1, blocks8);
    hipFree(A); hipFree(B); hipFree(C3); hipFree(C8); hipFree(bias); hipFree(d3); hipFree(d8);
    hipFree(dRowA); hipFree(dColA); hipFree(dRowB); hipFree(dColB); hipFree(dColC);
    return bad;
}

int main(int argc, char** argv)
{
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev);
    printf("CUs: %d\n", g_cus);
    int bad = 0;
    bad |= conv_case(15, true);
    for (int T : {14, 10, 3}) bad |= conv_case(T, false);
    printf(bad ? "RESULT: MISMATCH\n" : "RESULT: all outputs identical\n");
    return bad;
}
