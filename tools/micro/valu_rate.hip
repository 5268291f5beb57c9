// Issue rate of the vector instructions the cell pass is made of, per SIMD, at 1, 2 and 4 waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
// Each wave runs N iterations of 8 independent chains of one instruction kind; the kernel is launched with one
// workgroup per CU (256 CUs) of 256/512/1024 threads (1/2/4 waves per SIMD).  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned *out, int iters, unsigned seed)
{
    unsigned a[8]; f32x2 p[8]; double d[8];
    for (int q = 0; q < 8; ++q) { a[q] = seed * (threadIdx.x + 1) + q * 7919u; p[q] = f32x2{(float)(a[q] & 1023), (float)q}; d[q] = 1.0 + q * 1e-3 + (a[q] & 7); }
    const f32x2 c = {1.0001f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (KIND == 0) a[q] = __builtin_amdgcn_alignbit(a[q], a[(q + 1) & 7], 31);
            if (KIND == 1) asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(a[q]) : "v"(a[q]), "v"(a[(q + 1) & 7]), "v"(seed));
            if (KIND == 2) a[q] = a[q] < a[(q + 1) & 7] ? a[q] : a[(q + 1) & 7];
            if (KIND == 3) p[q] = p[q] * c + c;                              // v_pk_fma_f32 (contraction on for this file)
            if (KIND == 4) p[q] = p[q] - c;                                  // v_pk_add_f32
            if (KIND == 5) a[q] = (a[q] & ~31u) | (unsigned)q;               // v_and_or_b32
            if (KIND == 6) p[q].x = __builtin_fmaf(p[q].x, c.x, c.y);        // v_fma_f32
            if (KIND == 7) d[q] = __builtin_fma(d[q], 1.0000001, 1e-9);      // v_fma_f64
            if (KIND == 8) d[q] = d[q] + 1e-9;                               // v_add_f64
            if (KIND == 9) d[q] = d[q] < d[(q + 1) & 7] ? d[q] : d[(q + 1) & 7] + 1.0;   // v_cmp_f64 + 2 v_cndmask + v_add_f64
            if (KIND == 10) d[q] = 1.0 / d[q] + 1.5;                         // double division sequence
            if (KIND == 11) d[q] = sqrt(d[q]) + 2.0;                         // double sqrt sequence
            if (KIND == 12) p[q].x = __builtin_amdgcn_sqrtf(p[q].x);         // v_sqrt_f32
            if (KIND == 13) p[q].x = __builtin_amdgcn_cosf(p[q].x);          // v_cos_f32
            if (KIND == 14) p[q].x = __builtin_amdgcn_rcpf(p[q].x);          // v_rcp_f32
            if (KIND == 15) asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(a[q]) : "v"(a[q]), "v"(a[(q + 1) & 7]), "v"(seed));
            if (KIND == 16) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a[q]) : "v"(a[(q + 1) & 7]), "v"(a[q]));
            if (KIND == 17) a[q] = (unsigned)__builtin_ctz(a[q] | 256u) + a[(q + 1) & 7];   // v_ffbl_b32 (+ or, add)
        }
    }
    unsigned r = 0;
    for (int q = 0; q < 8; ++q) r ^= a[q] ^ __float_as_uint(p[q].x) ^ __float_as_uint(p[q].y) ^ (unsigned)__double_as_longlong(d[q]);
    if (r == 0x12345678u) out[0] = r;
}

template <int KIND>
static void run(const char *name)
{
    unsigned *d; hipMalloc(&d, 4);
    const int iters = KIND >= 10 ? 2000 : 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, d, 100, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, d, iters, 3u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * 8 * wps;                // wave-instructions issued on each SIMD
        printf("%-14s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, wps, ms,
               ms * 1e6 / insts_per_simd);
    }
    hipFree(d);
}

int main()
{
    run<0>("v_alignbit"); run<1>("v_med3_u32"); run<2>("v_min_u32"); run<3>("v_pk_fma_f32"); run<4>("v_pk_add_f32");
    run<5>("v_and_or_b32"); run<6>("v_fma_f32"); run<7>("v_fma_f64"); run<8>("v_add_f64"); run<9>("f64 cmp+sel+add");
    run<10>("f64 1/x + add"); run<11>("f64 sqrt + add");
    run<12>("v_sqrt_f32"); run<13>("v_cos_f32"); run<14>("v_rcp_f32"); run<15>("v_min3_u32"); run<16>("v_bcnt_u32_b32");
    run<17>("ffbl+or+add");
    return 0;
}
