// Cycles per v_mfma_f32_32x32x16_bf16 with one wave per SIMD on every CU, random operands in registers:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
// NACC accumulators in rotation (2 = the policy kernel's pair per row tile, 16 = its first layer), with FILL plain vector
// instructions behind every MFMA.  Cycles by s_memtime around the loop (first wave of each workgroup), clock by
// s_memrealtime; prints the median over the workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// COPY: every 8 MFMAs each wave moves 1 KiB of a 64 KiB table (L2-resident) into LDS: 1 = global_load_lds_dwordx4 (M0 + lane
// * 16), 2 = global_load_dwordx4 into registers, stored to LDS with ds_write_b128 eight MFMAs later.
template <int NACC, int FILL, int COPY = 0>
__global__ __launch_bounds__(256) void k(unsigned long long *out, float *sink, int iters, unsigned seed, const uint4 *table = nullptr)
{
    __shared__ uint4 lbuf[4 * 64 * 4];
    [[maybe_unused]] const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    [[maybe_unused]] uint4 stage = {0, 0, 0, 0};
    [[maybe_unused]] int cnt = 0;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    bf16x8 A, B;
    unsigned x = seed * (threadIdx.x + 1) + blockIdx.x * 7919u;
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; A[j] = (__bf16)((float)(x >> 20) * 1e-3f - 2.f); x = x * 1664525u + 1013904223u; B[j] = (__bf16)((float)(x >> 20) * 1e-3f - 2.f); }
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = (float)j + (float)(x & 15);
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j & 7]) : "v"(1.0001f));
            if (COPY != 0 && (a & 7) == 7) {
                const int piece = (cnt++ * 4 + wave) & 63;
                if (COPY == 1) {
                    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(lbuf + (wave * 4 + (cnt & 3)) * 64);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                 :: "s"(__builtin_amdgcn_readfirstlane(dst)), "v"((unsigned)(lane * 16)), "s"(table + piece * 64) : "memory");
                } else {
                    lbuf[(wave * 4 + (cnt & 3)) * 64 + lane] = stage;
                    stage = table[piece * 64 + lane];
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    if (COPY != 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); s += (float)lbuf[threadIdx.x].x + (float)stage.y; }
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
    for (int j = 0; j < 8; ++j) s += f[j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NACC, int FILL, int COPY = 0>
void run(unsigned long long *d, float *sink, const uint4 *table = nullptr)
{
    const int iters = 4096 / NACC, nwg = 256;
    std::vector<unsigned long long> h(2 * nwg);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NACC, FILL, COPY>), dim3(nwg), dim3(256), 0, 0, d, sink, iters, 12345u + rep, table);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int w = 0; w < nwg; ++w) { cyc.push_back((double)h[2 * w] / (iters * NACC)); ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    printf("accumulators %2d  fillers %2d  copy %d : %.1f cycles per MFMA (median; min %.1f max %.1f), clock %.2f GHz\n", NACC, FILL, COPY, cyc[nwg / 2], cyc[0], cyc[nwg - 1], ghz[nwg / 2]);
}

int main()
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 2 * 256 * 8); hipMalloc(&sink, 256 * 256 * 4);
    run<16, 0>(d, sink); run<2, 0>(d, sink); run<1, 0>(d, sink);
    uint4 *table; hipMalloc(&table, 65536); hipMemset(table, 1, 65536);
    run<16, 0, 1>(d, sink, table); run<16, 0, 2>(d, sink, table); run<16, 2, 1>(d, sink, table); run<16, 2, 2>(d, sink, table);
    run<4, 2>(d, sink); run<4, 3>(d, sink); run<4, 4>(d, sink); run<4, 5>(d, sink); run<8, 3>(d, sink); run<8, 4>(d, sink); run<8, 5>(d, sink);
    run<16, 3>(d, sink); run<16, 4>(d, sink); run<16, 5>(d, sink); run<4, 4, 1>(d, sink, table); run<8, 4, 1>(d, sink, table);
    run<2, 2>(d, sink); run<2, 4>(d, sink); run<2, 6>(d, sink); run<2, 8>(d, sink); run<16, 6>(d, sink);
    return 0;
}
