// Store bandwidth of the observation block's write patterns on MI355X: 4096 environments x 23 040 B (192 rows of 30
// floats), one 256-thread workgroup per pair of environments, nothing but stores.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/store_rate.hip -o store_rate && ./store_rate
// Patterns: rows  = one dword per lane, lanes 0-29 / 32-61 write a 120-byte row of each environment (the step kernel's);
//           align = the same instruction count, every row at a 128-byte boundary (wrong addresses, request shape only);
//           lines = lanes 0-31 / 32-63 write one whole 128-byte line of each environment;
//           flat4 = the pair's 46 080 B as 16 bytes per lane, consecutive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ROWS = 192, NA = 30, ENV_B = ROWS * NA * 4;
template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k_store(float *obs, float v)
{
    const int tid = threadIdx.x, sub = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, es = l >> 5, i = l & 31;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obs + (size_t)blockIdx.x * 2 * ROWS * NA, 0, 0x7fffffff, 0x00020000);
    if constexpr (PAT == 0) {
        if (i < NA)
#pragma unroll 8
            for (int r = sub * 48; r < sub * 48 + 48; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, es * ENV_B + i * 4, r * NA * 4, NT ? 2 : 0);
    } else if constexpr (PAT == 1) {
        if (i < NA)
#pragma unroll 8
            for (int r = sub * 48; r < sub * 48 + 48; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, es * ENV_B + i * 4, (r * NA * 4) & ~127, NT ? 2 : 0);
    } else if constexpr (PAT == 2) {
#pragma unroll 9
        for (int q = sub * 45; q < sub * 45 + 45; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, es * ENV_B + i * 4, q * 128, NT ? 2 : 0);
    } else {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 w = {__float_as_uint(v), __float_as_uint(v), __float_as_uint(v), __float_as_uint(v)};
        for (int q = sub; q < 45; q += 4)
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, l * 16, q * 1024, NT ? 2 : 0);
    }
}
template <int PAT, bool NT>
static void run(const char *name, float *obs, int E)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((k_store<PAT, NT>), dim3(E / 2), dim3(256), 0, 0, obs, 1.0f);
    CHECK(hipEventRecord(a));
    const int reps = 50;
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((k_store<PAT, NT>), dim3(E / 2), dim3(256), 0, 0, obs, (float)k);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)E * ENV_B * (PAT == 1 ? 1.0 : 1.0);
    printf("%-6s %s: %7.2f us per launch, %6.2f TB/s\n", name, NT ? "nt   " : "plain", ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e12);
}
int main()
{
    const int E = 4096;
    float *obs; CHECK(hipMalloc(&obs, (size_t)E * ENV_B));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("rows", obs, E); run<0, true>("rows", obs, E);
        run<1, false>("align", obs, E); run<1, true>("align", obs, E);
        run<2, false>("lines", obs, E); run<2, true>("lines", obs, E);
        run<3, false>("flat4", obs, E); run<3, true>("flat4", obs, E);
    }
    return 0;
}
