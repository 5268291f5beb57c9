// HIP kernels (gfx950) for the assembly-environment step path.
//
// Reference algorithm (cited per block):
//   CPP = cus_gym/gym/envs/customized_envs/envs_cplus/src/AssemblyEnv.cpp
//   ENV = cus_gym/gym/envs/customized_envs/assembly.py
//
// Mapping: one lane per agent.  An environment owns SL consecutive lanes
// (SL = power of two >= n_a); a workgroup holds EPB = block/SL environments
// (two 30-agent environments per 64-lane wavefront; one 1024-agent environment per
// 16-wave workgroup).  Per environment the workgroup keeps positions, velocities,
// the transformed target grid, one sensed-cell bitmask per agent and two per-cell
// occupancy bitmaps in LDS, so HBM sees each input once and each output once.
//
// Numerics: the state is stored as S (float for the batched path, double for the
// five reference-compatible entry points).  Every DECISION of the reference is a
// strict comparison on a double-precision distance (CPP:663, 894, 907, 166, 190,
// 487, 552, 1171; ENV:446; CPP:849).  They are evaluated here in double on the
// stored state, un-fused (this file is compiled with -ffp-contract=off), with
// sqrt(s) < t rewritten as s < T(t) where T(t) is the exact preimage boundary of the
// correctly rounded square root (computed on the host, asm_capi.cpp).  Fed the same
// numbers, the kernels therefore take the same branches and produce the same
// indices as the reference; values are rounded once to S on output.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <algorithm>

#include "asm_kernels.h"

namespace asmenv {

#define ASM_PI 3.14159265358979323846

// Phase accounting build (make exits; tools/phase_valu.sh): the step kernel returns at the phase boundary
// named by MARL_AMD_DBG_EXIT, so that counter differences between runs give per-phase instruction counts.
#ifdef ASM_EXITS
__device__ int g_exit_at = 0;
#define EXIT_AT(k) do { if (xat == (k)) return; } while (0)
#else
#define EXIT_AT(k) do { } while (0)
#endif

// Diagnostic build only (make diag): cycle stamps at phase boundaries, written by lane 0
// of each workgroup's first wave to a buffer of its own.  The product build has no stamps.
#ifdef ASM_DIAG
__device__ unsigned long long *g_diag = nullptr;          // [grid][32]
__device__ unsigned long long *g_diag_only = nullptr;           // k_step_fast: stamp only launches on this position array (NULL: all)
#define DIAG_COUNT(k, v) do { if (g_diag && threadIdx.x == 0) g_diag[(size_t)blockIdx.x * 32 + (k)] = (unsigned long long)(v); } while (0)
#define STAMP(k) do { if (g_diag && threadIdx.x == 0) { g_diag[(size_t)blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); \
    if ((k) == 0) g_diag[(size_t)blockIdx.x * 32 + 10] = wall_clock64(); if ((k) == 9) g_diag[(size_t)blockIdx.x * 32 + 11] = wall_clock64(); } } while (0)
// (the policy kernel's stamps: slots 0..29 in program order, wall clocks of the first and the last in slots 30 and 31)
#define PSTAMP(k, last) do { if (g_diag && threadIdx.x == 0) { g_diag[(size_t)blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); \
    if ((k) == 0) g_diag[(size_t)blockIdx.x * 32 + 30] = wall_clock64(); if (last) g_diag[(size_t)blockIdx.x * 32 + 31] = wall_clock64(); } } while (0)
#else
#define STAMP(k) do { } while (0)
#define PSTAMP(k, last) do { } while (0)
#define DIAG_COUNT(k, v) do { } while (0)
#endif

__device__ __forceinline__ void wrap_rel(double &x, double &y, double hw, double hh)
{   // CPP:705-720
    if (x < -hw) x += 2 * hw; else if (x > hw) x -= 2 * hw;
    if (y < -hh) y += 2 * hh; else if (y > hh) y -= 2 * hh;
}

#include "kernels_common.h"
#include "kernels_lattice.h"
#include "kernels_generic.h"
#include "kernels_fast.h"
#include "kernels_aux.h"
#include "kernels_compat.h"
#include "kernels_policy.h"

// ---------------------------------------------------------------------------------
// launchers
#define LAUNCH_CHECK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return (int)_e; } while (0)

template <typename K>
static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

template <typename S>
int launch_observe(const DevParams &P, const S *p, const S *dp, const S *heading, const S *grid, const int *n_g,
                   const double *l_cell, int *nbr, int *inflags, int *nearest, S *obs, S *rew, int *sensed,
                   int *occupied, hipStream_t st)
{
    if (int rc = set_lds(k_observe<S>, P.lds_block_bytes)) return rc;
    hipLaunchKernelGGL(k_observe<S>, dim3(P.grid), dim3(P.block), P.lds_block_bytes, st, P, p, dp, heading, grid, n_g,
                       l_cell, nbr, inflags, nearest, obs, rew, sensed, occupied);
    LAUNCH_CHECK();
    return 0;
}

template <int T, bool PER>
static int launch_step_fast_t(const DevParams &P, const LatArgs &lat, float *p, float *dp, const float *act, const float *grid,
                              const int *n_g, const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs,
                              float *rew, float *prior, int *sensed, int *occupied, double *stats, hipStream_t st)
{
    // three instantiations: the index-list outputs are compiled in only where they are asked for
    // MARL_AMD_LDS_PAD (measurement knob): extra bytes of dynamic LDS per workgroup, to find where a CU stops taking one more workgroup
    static const unsigned lds_pad = [] { const char *v = std::getenv("MARL_AMD_LDS_PAD"); return v ? (unsigned)std::atoi(v) : 0u; }();
    const unsigned lds = (P.lat ? P.fast_lds_bytes_lat : P.fast_lds_bytes) + lds_pad;
    auto go = [&](auto kernel) {
        if (int rc = set_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, dim3((P.E + 1) / 2), dim3(64 * T), lds, st, P, lat, p, dp, act, grid, n_g,
                           l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats);
        LAUNCH_CHECK();
        return 0;
    };
    if constexpr (T == 4)                                           // the lattice path exists for this schedule only
        if (P.lat) {
            if (P.obs_nt)
                return occupied ? go(k_step_fast<T, PER, 2, true, true>) : sensed ? go(k_step_fast<T, PER, 1, true, true>) : go(k_step_fast<T, PER, 0, true, true>);
            return occupied ? go(k_step_fast<T, PER, 2, true>) : sensed ? go(k_step_fast<T, PER, 1, true>) : go(k_step_fast<T, PER, 0, true>);
        }
    return occupied ? go(k_step_fast<T, PER, 2, false>) : sensed ? go(k_step_fast<T, PER, 1, false>) : go(k_step_fast<T, PER, 0, false>);
}

int launch_step_fast(const DevParams &P, const LatArgs &lat, float *p, float *dp, const float *act, const float *grid, const int *n_g,
                     const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs, float *rew, float *prior,
                     int *sensed, int *occupied, double *stats, hipStream_t st)
{
#ifdef ASM_EXITS
    { const char *x = std::getenv("MARL_AMD_DBG_EXIT"); const int v = x ? std::atoi(x) : 0;
      static int set_to = -1;                                       // (copied when it changes: a copy per launch would serialise the streams)
      if (v != set_to) { if (hipMemcpyToSymbol(HIP_SYMBOL(g_exit_at), &v, sizeof v) != hipSuccess) return -1; set_to = v; } }
#endif
    if (P.periodic)                                                 // the host offers the periodic arena at T = 4 only
        return launch_step_fast_t<4, true>(P, lat, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    switch (P.fast_T) {
    case 1: return launch_step_fast_t<1, false>(P, lat, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    case 2: return launch_step_fast_t<2, false>(P, lat, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    default: return launch_step_fast_t<4, false>(P, lat, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    }
}

// The large swarm's own instantiation of the step kernel (k_step<float, true>) serves exactly these launches.
bool uses_swarm_kernel(const DevParams &P)
{
    static const bool off = std::getenv("MARL_AMD_NO_SWARM_KERNEL") != nullptr;          // measurement knob: the generic instantiation
    return P.lat == 2 && P.bins && P.parts <= 1 && !P.periodic && P.EPB == 1 && P.cartesian && P.W <= 32 && P.n_obs >= 2 &&
           (P.n_obs & 1) == 0 && P.MW >= 12 && !off;     // (MW: neighbour lists and nearest cells pass through the word keys' area, 24 bytes per agent)
}

template <typename S>
int launch_step(const DevParams &P, const LatArgs &lat, S *p, S *dp, const float *act, const S *grid, const int *n_g,
                const double *l_cell, int *nbr, int *inflags, int *nearest, S *obs, S *rew, S *prior, int *sensed,
                int *occupied, double *stats, hipStream_t st)
{
#ifdef ASM_EXITS
    { const char *x = std::getenv("MARL_AMD_DBG_EXIT"); const int v = x ? std::atoi(x) : 0;
      static int set_to = -1;                                       // (copied when it changes: a copy per launch would serialise the streams)
      if (v != set_to) { if (hipMemcpyToSymbol(HIP_SYMBOL(g_exit_at), &v, sizeof v) != hipSuccess) return -1; set_to = v; } }
#endif
    if constexpr (sizeof(S) == 4) {
        // the large swarm's own instantiation: walls, lattice description, cell lists, the whole environment in one workgroup
        if (uses_swarm_kernel(P)) {
            if (int rc = set_lds(k_step<S, true>, P.lds_block_bytes)) return rc;
            hipLaunchKernelGGL((k_step<S, true>), dim3(P.grid), dim3(P.block), P.lds_block_bytes, st, P, lat, p, dp, act, grid, n_g, l_cell,
                               nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats);
            LAUNCH_CHECK();
            return 0;
        }
    }
    if (int rc = set_lds(k_step<S>, P.lds_block_bytes)) return rc;
    hipLaunchKernelGGL(k_step<S>, dim3(P.grid), dim3(P.block), P.lds_block_bytes, st, P, lat, p, dp, act, grid, n_g, l_cell,
                       nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats);
    LAUNCH_CHECK();
    return 0;
}

int launch_step_split(const DevParams &P, const DevParams &Q, const LatArgs &lat, float *p, float *dp, const float *act, const float *grid,
                      const int *n_g, const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs, float *rew,
                      float *prior, int *sensed, int *occupied, double *stats, hipStream_t st)
{
    if (int rc = launch_step<float>(P, lat, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st)) return rc;
    if (int rc = set_lds(k_observe_part, Q.lds_block_bytes)) return rc;
    hipLaunchKernelGGL(k_observe_part, dim3((unsigned)P.E * Q.parts), dim3(Q.rows), Q.lds_block_bytes, st, Q, lat, p, dp, grid, n_g, l_cell,
                       nbr, inflags, nearest, obs, rew, sensed, occupied, stats);
    LAUNCH_CHECK();
    if (stats) {
        hipLaunchKernelGGL(k_swarm_stats, dim3((unsigned)P.E), dim3(P.SL), 0, st, P, p, grid, nbr, inflags, nearest, rew, stats);
        LAUNCH_CHECK();
    }
    return 0;
}

template <typename S>
int launch_reset(const DevParams &P, uint64_t seed, uint32_t episode, int64_t env0, int n_shapes,
                 const double *shape_grids, const int *shape_off, const int *shape_ng, const double *shape_lcell,
                 int *shape_id, S *p, S *dp, S *grid, int *n_g, double *l_cell, int *nbr, int *inflags, int *nearest,
                 S *obs, int *sensed, int *occupied, double *xform, hipStream_t st)
{
    if (int rc = set_lds(k_reset<S>, P.lds_block_bytes)) return rc;
    hipLaunchKernelGGL(k_reset<S>, dim3(P.grid), dim3(P.block), P.lds_block_bytes, st, P, seed, episode, env0,
                       n_shapes, shape_grids, shape_off, shape_ng, shape_lcell, shape_id, p, dp, grid, n_g, l_cell, nbr,
                       inflags, nearest, obs, sensed, occupied, xform);
    LAUNCH_CHECK();
    return 0;
}

template int launch_observe<float>(const DevParams &, const float *, const float *, const float *, const float *,
                                   const int *, const double *, int *, int *, int *, float *, float *, int *, int *,
                                   hipStream_t);
template int launch_observe<double>(const DevParams &, const double *, const double *, const double *, const double *,
                                    const int *, const double *, int *, int *, int *, double *, double *, int *, int *,
                                    hipStream_t);
template int launch_step<float>(const DevParams &, const LatArgs &, float *, float *, const float *, const float *, const int *,
                                const double *, int *, int *, int *, float *, float *, float *, int *, int *, double *,
                                hipStream_t);
template int launch_reset<float>(const DevParams &, uint64_t, uint32_t, int64_t, int, const double *, const int *,
                                 const int *, const double *, int *, float *, float *, float *, int *, double *, int *,
                                 int *, int *, float *, int *, int *, double *, hipStream_t);

int launch_check_lattice(int E, int n_g_max, const float *grid, const int *n_g, const LatArgs &lat, double tol, int *n_bad,
                         hipStream_t st)
{
    hipError_t e = hipMemsetAsync(n_bad, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_check_lattice, dim3((unsigned)E), dim3(64), 0, st, E, n_g_max, grid, n_g, lat, tol, n_bad);
    LAUNCH_CHECK();
    return 0;
}

int launch_random_actions(uint64_t seed, uint32_t step, int64_t env0, int E, int n_a, float *act, hipStream_t st)
{
    const size_t n = (size_t)E * 2 * n_a;
    hipLaunchKernelGGL(k_random_actions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, step, env0, E, n_a, act);
    LAUNCH_CHECK();
    return 0;
}

int launch_replay_draw(uint64_t seed, uint32_t counter, const uint32_t *counter_dev, int64_t lo, int64_t hi, int N, int64_t *out,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k_replay_draw, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, seed, counter, counter_dev, lo, hi, N, out);
    LAUNCH_CHECK();
    return 0;
}

int launch_replay_gather(int E, int n_a, int obs_dim, int T, const float *obs_ring, const float *act_ring,
                         const float *prior_ring, const float *rew_ring, const int64_t *index, int N, float *obs_out,
                         float *act_out, float *rew_out, float *next_out, float *prior_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_replay_gather, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, E, n_a, obs_dim, T, obs_ring,
                       act_ring, prior_ring, rew_ring, index, N, obs_out, act_out, rew_out, next_out, prior_out);
    LAUNCH_CHECK();
    return 0;
}

int launch_obs_rows(int E, int n_a, int obs_dim, const float *obs, float *rows, hipStream_t st)
{
    const int tiles = (n_a + ASM_ROWS_TILE - 1) / ASM_ROWS_TILE;
    const size_t lds = (size_t)obs_dim * 33 * sizeof(float);
    if (int rc = set_lds(k_obs_rows, lds)) return rc;
    hipLaunchKernelGGL(k_obs_rows, dim3((unsigned)E * tiles), dim3(256), lds, st, n_a, obs_dim, tiles, obs, rows);
    LAUNCH_CHECK();
    return 0;
}

int launch_policy_mlp(int E, int n_a, int obs_dim, int hidden, int act_dim, const float *obs, const void *w1, const void *w2,
                      const void *w3, const void *w4, const float *bias, float *act, float noise_scale, float epsilon,
                      uint64_t seed, uint32_t step, int64_t env0, hipStream_t st)
{
    const int TPE = (n_a + 31) / 32, ntiles = E * TPE, K1 = (obs_dim + 15) / 16;
    const dim3 grid((unsigned)((ntiles + 7) / 8)), block(256);
    const int HT = hidden / 32;
    // two LDS buffers of this many 1 KiB fragments: half the first layer, half a hidden layer, the whole output layer
    const int half_frags = HT * std::max({(K1 + 1) / 2, HT, 2});
    const size_t bias_bytes = 4 * (3 * (size_t)hidden + 32), lds_max = 160 * 1024;
    size_t lds = bias_bytes + (size_t)2048 * half_frags;
    if (lds > lds_max) return -1;
    // Observation slabs through LDS where a k-step's 16 rows of an environment are one aligned contiguous block and at
    // least two k-steps of slabs fit behind the first layer's weights.
    const int ps = n_a > 16 ? 4 : 2;                                    // 1 KiB copy instructions per k-step and wave
    const size_t lds_slabs = bias_bytes + (size_t)K1 * HT * 1024 + (size_t)4 * 3 * ps * 1024;
    const bool slabs = TPE == 1 && HT >= 4 && ((size_t)obs_dim * n_a) % 4 == 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0 &&
                       lds_slabs <= lds_max;
    if (slabs) lds = std::max(lds, lds_slabs);
    auto go = [&](auto kernel) {
        if (int rc = set_lds(kernel, lds)) return rc;
        hipLaunchKernelGGL(kernel, grid, block, lds, st, n_a, obs_dim, K1, act_dim, TPE, ntiles, half_frags, obs, (const uint4 *)w1,
                           (const uint4 *)w2, (const uint4 *)w3, (const uint4 *)w4, bias, act, noise_scale, epsilon, seed, step, env0);
        LAUNCH_CHECK();
        return 0;
    };
    switch (hidden * 10 + (slabs ? ps : 0)) {
    case 320: return go(k_policy_mlp<1, 0>);
    case 640: return go(k_policy_mlp<2, 0>);
    case 1280: return go(k_policy_mlp<4, 0>);
    case 1282: return go(k_policy_mlp<4, 2>);
    case 1284: return go(k_policy_mlp<4, 4>);
    case 2560: return go(k_policy_mlp<8, 0>);
    case 2562: return go(k_policy_mlp<8, 2>);
    case 2564: return go(k_policy_mlp<8, 4>);
    default: return -1;
    }
}

#ifdef ASM_DIAG
int diag_set_buffer(unsigned long long *dev)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &dev, sizeof dev);
}
int diag_set_filter(const void *p)
{
    unsigned long long *q = (unsigned long long *)p;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_only), &q, sizeof q);
}
#endif

int launch_cos_f32_err(uint32_t lo, uint32_t hi, double *out_max, hipStream_t st)
{
    const uint64_t n = (uint64_t)hi - lo + 1;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_cos_f32_err, dim3(blocks), dim3(256), 0, st, lo, hi, reinterpret_cast<unsigned long long *>(out_max));
    return (int)hipGetLastError();
}

int launch_rule_actions(const DevParams &P, const float *p, const float *dp, const float *grid, const int *n_g,
                        const int *inflags, const int *nearest, const int *sensed, float *act, hipStream_t st)
{
    const int threads = (P.n_a + 63) / 64 * 64;
    hipLaunchKernelGGL(k_rule_actions, dim3((unsigned)P.E), dim3(threads), 0, st, P, p, dp, grid, n_g, inflags, nearest,
                       sensed, act);
    LAUNCH_CHECK();
    return 0;
}

int launch_metrics(int E, int n_a, int NGP, double r_avoid, double size_a, const float *p, const float *grid,
                   const int *n_g, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_metrics, dim3((unsigned)E), dim3(256), 0, st, n_a, NGP, r_avoid / 2, 2 * size_a, p, grid, n_g, out);
    LAUNCH_CHECK();
    return 0;
}

int launch_sqrt(const double *in, double *out, int n, hipStream_t st)
{
    hipLaunchKernelGGL(k_sqrt_f64, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n);
    LAUNCH_CHECK();
    return 0;
}

int launch_reward_compat(const DevParams &P, const double *p, const double *grid, int n_g, const int *nbr,
                         const int *inflags, const int *sensed, bool c0, bool c3, bool c4, double *reward, hipStream_t st)
{
    hipLaunchKernelGGL(k_reward_compat, dim3((P.n_a + 63) / 64), dim3(64), 0, st, P, p, grid, n_g, nbr, inflags, sensed,
                       c0, c3, c4, reward);
    LAUNCH_CHECK();
    return 0;
}

int launch_sf_b2b_compat(const double *p, const double *d_edge, const bool *collide, const double *d_center, int n_a,
                         double k_ball, bool periodic, double hw, double hh, double *sf, hipStream_t st)
{
    hipLaunchKernelGGL(k_sf_b2b_compat, dim3((n_a + 63) / 64), dim3(64), 0, st, p, d_edge, collide, d_center, n_a,
                       k_ball, periodic, hw, hh, sf);
    LAUNCH_CHECK();
    return 0;
}

int launch_dist_b2w_compat(const double *p, const double *r, int n_a, const double *bpos_host, double *d_b2w,
                           bool *coll, hipStream_t st)
{
    hipLaunchKernelGGL(k_dist_b2w_compat, dim3((n_a + 63) / 64), dim3(64), 0, st, p, r, n_a, bpos_host[0], bpos_host[1],
                       bpos_host[2], bpos_host[3], d_b2w, coll);
    LAUNCH_CHECK();
    return 0;
}

int launch_prior_compat(const DevParams &P, const double *p, const double *dp, const double *grid, int n_g,
                        const int *nbr, double l_cell, double *a_prior, hipStream_t st)
{
    hipLaunchKernelGGL(k_prior_compat, dim3((P.n_a + 63) / 64), dim3(64), 0, st, P, p, dp, grid, n_g, nbr, l_cell,
                       a_prior);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace asmenv
