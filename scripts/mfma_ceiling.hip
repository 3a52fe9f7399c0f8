// What is the fp32-MFMA ceiling of this board, and what sets it?  ("Next round" item 3.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_ceiling.hip -o video-subtitle-remover_amd/build/mfma_ceiling
// Every workgroup (256 threads = one wave per SIMD) runs "chunks" of 32 v_mfma_f32_32x32x2_f32 per wave -- the MFMA
// work of one 32-deep K chunk of a 64x32 wave tile -- with nothing else (bare), with the fragment reads of the real
// kernel beside them (NDS ds_read_b128 per chunk and wave), or with fragment reads plus the LDS-DMA operand stream
// (NDMA global_load_lds_dwordx4 pieces per chunk and wave, double-buffered, one barrier per chunk: the loop structure
// of gather_gemm_f32_v3).  Per-wave rates of the tile shapes:
//     128x64  tile, 4 waves of 64x32 :  12 reads, 6 pieces per 32 MFMAs   (the shipped kernel, 3 workgroups / CU)
//     128x128 tile, 4 waves of 64x64 :   8 reads, 4 pieces per 32 MFMAs   (2 workgroups / CU)
//     256x128 tile, 8 waves of 64x64 :   8 reads, 3 pieces per 32 MFMAs   (8 waves / CU)
// Wave 0 of every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop, so the
// shader clock the board HOLDS under each load is measured, not assumed:  sclk = d(memtime) / d(realtime) * 100 MHz.
// Operands are random floats or zeros (data toggling changes the power drawn, hence the clock the DVFS grants).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const float __attribute__((address_space(1)))* gcf32;
typedef __attribute__((address_space(3))) void* lds_vptr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int BUF_FLOATS = 6144;          // 24 KB per operand buffer, two buffers
constexpr int REGION_FLOATS = 32768;      // 128 KB of source per workgroup: L2-resident, larger than the 32 KB vector L1

__device__ __forceinline__ void glds16(gcf32 src, lds_vptr dst)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

template <int NDS, int NDMA>
__global__ void __launch_bounds__(256)
k_ceiling(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* region = src + (size_t)blockIdx.x * REGION_FLOATS;
    for (int i = tid; i < 2 * BUF_FLOATS; i += 256) smem[i] = region[i];
    f32x4 ra = *reinterpret_cast<const f32x4*>(region + tid * 4);
    f32x4 rb = *reinterpret_cast<const f32x4*>(region + 1024 + tid * 4);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    constexpr int NR = NDS / 4;           // fragment reads per group of 8 MFMAs
    unsigned long long c0 = 0, r0 = 0;
    if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int c = 0; c < chunks; ++c) {
        if constexpr (NDMA > 0) {
            float* dst = smem + ((c + 1) & 1) * BUF_FLOATS;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int off = (((c * NDMA + i) * 1024) + tid * 4) & (REGION_FLOATS - 1);
                glds16((gcf32)(region + off), (lds_vptr)(dst + (wave * NDMA + i) * 256));
            }
        }
        const float* cur = smem + (c & 1) * BUF_FLOATS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 fr[NR > 0 ? NR : 1];
            if constexpr (NR > 0) {
#pragma unroll
                for (int k = 0; k < NR; ++k)
                    fr[k] = *reinterpret_cast<const f32x4*>(cur + (((g * NR + k) * 64 + lane) * 4));
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int j = m >> 1;
                float a, b;
                if constexpr (NR > 0) { a = fr[(m & 1) % NR][j]; b = fr[NR - 1][j]; }
                else { a = ra[j]; b = rb[j]; }
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            }
        }
        if constexpr (NDMA > 0 || NDS > 0) __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[blockIdx.x * 2 + 0] = c1 - c0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int NDS, int NDMA>
static void run(const char* name, const float* src, const char* data, int occ, float* out, unsigned long long* stamps, int chunks)
{
    // dynamic LDS sized so that exactly `occ` workgroups fit the 160 KB of a CU
    const int lds = occ == 1 ? 120 * 1024 : occ == 2 ? 72 * 1024 : 50 * 1024;
    CK(hipFuncSetAttribute((const void*)k_ceiling<NDS, NDMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int got = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, k_ceiling<NDS, NDMA>, 256, lds));
    const int blocks = 256 * occ;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ~0.4 s of the same load first: the DVFS loop settles over tens of milliseconds
    for (int i = 0; i < 150; ++i) hipLaunchKernelGGL((k_ceiling<NDS, NDMA>), dim3(blocks), dim3(256), lds, 0, src, out, stamps, chunks);
    CK(hipDeviceSynchronize());
    const int reps = 40;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_ceiling<NDS, NDMA>), dim3(blocks), dim3(256), lds, 0, src, out, stamps, chunks);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<unsigned long long> h(blocks * 2);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz(blocks), busy(blocks);
    for (int b = 0; b < blocks; ++b) {
        mhz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
        busy[b] = (double)occ * chunks * 32 * 64 / (double)h[2 * b];        // MFMA issue cycles of the SIMD / cycles the loop took
    }
    std::sort(mhz.begin(), mhz.end()); std::sort(busy.begin(), busy.end());
    const double flops = (double)blocks * 4 * chunks * 32 * 4096.0;
    printf("%-34s data=%-6s wg/CU=%d (occupancy api %d)  %7.3f ms  %6.1f TF  sclk median %4.0f MHz [%4.0f..%4.0f]  pipe busy (in-loop) %.3f\n",
           name, data, occ, got, ms, flops / (ms * 1e-3) * 1e-12, mhz[blocks / 2], mhz[0], mhz[blocks - 1], busy[blocks / 2]);
    fflush(stdout);
}

int main()
{
    const int maxBlocks = 256 * 3;
    const size_t nsrc = (size_t)maxBlocks * REGION_FLOATS;
    std::vector<float> h(nsrc);
    srand(1);
    for (size_t i = 0; i < nsrc; ++i) h[i] = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    float *rnd, *zero, *out;
    unsigned long long* stamps;
    CK(hipMalloc(&rnd, nsrc * 4)); CK(hipMalloc(&zero, nsrc * 4)); CK(hipMalloc(&out, (size_t)maxBlocks * 256 * 4));
    CK(hipMalloc(&stamps, maxBlocks * 2 * 8));
    CK(hipMemcpy(rnd, h.data(), nsrc * 4, hipMemcpyHostToDevice));
    CK(hipMemset(zero, 0, nsrc * 4));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs %d  clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int chunks = 700;      // 700 x 32 MFMAs x 64 cycles = 1.43 M cycles per wave
    for (int pass = 0; pass < 2; ++pass) {
        printf("-- pass %d\n", pass);
        for (int occ = 1; occ <= 3; ++occ) {
            run<0, 0>("bare MFMA", zero, "zero", occ, out, stamps, chunks);
            run<0, 0>("bare MFMA", rnd, "random", occ, out, stamps, chunks);
        }
        run<12, 0>("MFMA + 12 ds_read_b128 / chunk", rnd, "random", 3, out, stamps, chunks);
        run<8, 0>("MFMA +  8 ds_read_b128 / chunk", rnd, "random", 2, out, stamps, chunks);
        run<12, 6>("128x64-like: 12 reads + 6 DMA", rnd, "random", 3, out, stamps, chunks);
        run<12, 6>("128x64-like: 12 reads + 6 DMA", zero, "zero", 3, out, stamps, chunks);
        run<8, 4>("128x128-like: 8 reads + 4 DMA", rnd, "random", 2, out, stamps, chunks);
        run<8, 3>("256x128-like: 8 reads + 3 DMA", rnd, "random", 2, out, stamps, chunks);
        run<8, 4>("128x128-like at 3 wg/CU", rnd, "random", 3, out, stamps, chunks);
    }
    return 0;
}
