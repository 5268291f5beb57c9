// Internal interface between the C-ABI layer (asm_capi.cpp) and the HIP kernels
// (asm_kernels.hip).  Not installed; the public interface is include/marl_amd_env.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/marl_amd_env.h"

// k_step_fast keeps a cell's y this many floats after its x in LDS (x rows of the two environments first), a
// compile-time distance (a multiple of 64) so that one address serves both; it must hold two rows of up to 1024 cells
// and the 64 own-position entries that follow them.
#ifndef ASM_GY_OFF
#define ASM_GY_OFF 2112
#endif

namespace asmenv {

constexpr int ASM_MAX_AGENTS = 1024;   // largest n_a the entry points accept

// Cell list over the arena (large swarms): nbx x nby square bins of side 1 / inv_h counted from the arena's lower left
// corner (positions outside fall into the border bins); two agents closer than the radius the list was sized for lie
// at most `reach` bins apart in either direction.
struct BinGeom { int nbx, nby, reach; float inv_h; };

// Kernel parameters, passed by value.  The T_* members are squared-distance
// thresholds: for a threshold t of the reference, T = min{ s : fl(sqrt(s)) >= t } so that
// fl(sqrt(s)) < t  <=>  s < T exactly (T_*_le: min{ s : fl(sqrt(s)) > t }, so that
// !(fl(sqrt(s)) > t) <=> s < T).
struct DevParams {
    int E, n_a, SL, EPB, n_g_max, NGP, W, MW;
    int topo, n_obs, n_occ, obs_dim;
    int periodic, cartesian, con_self, prior, recompute, want_reward, want_stats;
    int observe_only;               // k_step_fast: no dynamics, the state is observed as it is (reset, observe)
    int state_only;                 // k_reset: draw the state and the grid only; the observation comes from another launch
    int n_g_e;                      // set per environment inside the kernels
    int part;                       // set per workgroup inside the part kernel: which part of the environment it observes
    int block, grid;
    unsigned lds_env_bytes, lds_block_bytes;
    int fast_T, capp;               // k_step_fast: waves per workgroup (0: not eligible), padded slot count
    int obs_nt;                     // observation rows stored non-temporal (set per launch from the size of the observation array)
    unsigned fast_lds_bytes, fast_lds_bytes_lat;     // dynamic LDS of k_step_fast: all-cells path, lattice path
    int lat, lat_KO;                // lattice path: 0 off, 1 k_step_fast, 2 k_step (large swarms); lattice rows |k| <= lat_KO can hold an occupied cell
    float lat_cb;                   // band coefficient: storage rounding of the cell coordinates (see kernels_lattice.h)
    int state_rows;                 // state arrays of SL entries kept in LDS: 4 (px, py, vx, vy), or 2 for a part workgroup (velocities stay in global memory)
    int rows;                       // per-agent LDS rows (sensed masks, word keys) of one workgroup: SL, or SL / parts for a part workgroup
    int parts;                      // large swarms: the observation of an environment is split over this many workgroups (0, 1: not split)
    int bins;                       // large swarms: the pair passes (contacts, k nearest neighbours) go through cell lists
    BinGeom bin_c, bin_s;           // ... sized for the contact distance and for the sensing radius
    float bin_x0, bin_y0;           // half_w, half_h in single precision
    double T_sen;                   // norm < d_sen            CPP:663, 907
    double T_half_le;               // !(norm > r_avoid/2)     CPP:190
    double T_half_lt;               // norm < r_avoid/2        WRAP:67
    double T_ravoid;                // norm < r_avoid          CPP:487, 1171
    double T_contact;               // norm - 2*size_a < 0     ENV:445-446
    double T_v005;                  // norm < 0.05             CPP:552
    // single-precision screens (S = float): threshold, margins, inflated thresholds
    float Tsen_f, Tsen_amb_f, Tsen_hi_f, Thalf_hi_f, Thalf_f, Thalf_amb_f, Tcontact_hi_f, inv2dsen_f;
    double d_sen, r_avoid, sizes2, size_a, vel_max, k_ball, k_wall, c_wall, dt, sens, mass, half_w, half_h;
};

// The kernels' view of an asm_lattice (device pointers only).
struct LatArgs {
    const asm_lattice_shape *shapes;
    const int *shape_id;
    const double *xform;
    int n_shapes;
};

template <typename S>
int launch_observe(const DevParams &P, const S *p, const S *dp, const S *heading, const S *grid, const int *n_g,
                   const double *l_cell, int *nbr, int *inflags, int *nearest, S *obs, S *rew, int *sensed,
                   int *occupied, hipStream_t st);
template <typename S>
int launch_step(const DevParams &P, const LatArgs &lat, S *p, S *dp, const float *act, const S *grid, const int *n_g,
                const double *l_cell, int *nbr, int *inflags, int *nearest, S *obs, S *rew, S *prior, int *sensed,
                int *occupied, double *stats, hipStream_t st);
template <typename S>
int launch_reset(const DevParams &P, uint64_t seed, uint32_t episode, int64_t env0, int n_shapes,
                 const double *shape_grids, const int *shape_off, const int *shape_ng, const double *shape_lcell,
                 int *shape_id, S *p, S *dp, S *grid, int *n_g, double *l_cell, int *nbr, int *inflags, int *nearest,
                 S *obs, int *sensed, int *occupied, double *xform, hipStream_t st);
int launch_check_lattice(int E, int n_g_max, const float *grid, const int *n_g, const LatArgs &lat, double tol, int *n_bad,
                         hipStream_t st);
// The large-swarm step as three launches: dynamics (one workgroup per environment), observation + reward (P.parts
// workgroups per environment), statistics.  Q: the parameters of a part workgroup (rows, LDS size).
int launch_step_split(const DevParams &P, const DevParams &Q, const LatArgs &lat, float *p, float *dp, const float *act, const float *grid,
                      const int *n_g, const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs, float *rew,
                      float *prior, int *sensed, int *occupied, double *stats, hipStream_t st);
int launch_step_fast(const DevParams &P, const LatArgs &lat, float *p, float *dp, const float *act, const float *grid, const int *n_g,
                     const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs, float *rew, float *prior,
                     int *sensed, int *occupied, double *stats, hipStream_t st);
int launch_random_actions(uint64_t seed, uint32_t step, int64_t env0, int E, int n_a, float *act, hipStream_t st);
int launch_replay_draw(uint64_t seed, uint32_t counter, const uint32_t *counter_dev, int64_t lo, int64_t hi, int N, int64_t *out,
                       hipStream_t st);
int launch_replay_gather(int E, int n_a, int obs_dim, int T, const float *obs_ring, const float *act_ring,
                         const float *prior_ring, const float *rew_ring, const int64_t *index, int N, float *obs_out,
                         float *act_out, float *rew_out, float *next_out, float *prior_out, hipStream_t st);
int launch_rule_actions(const DevParams &P, const float *p, const float *dp, const float *grid, const int *n_g,
                        const int *inflags, const int *nearest, const int *sensed, float *act, hipStream_t st);
int launch_metrics(int E, int n_a, int NGP, double r_avoid, double size_a, const float *p, const float *grid,
                   const int *n_g, double *out, hipStream_t st);
bool uses_swarm_kernel(const DevParams &P);              // the step of these parameters runs k_step<float, true>
int launch_obs_rows(int E, int n_a, int obs_dim, const float *obs, float *rows, hipStream_t st);
int launch_policy_mlp(int E, int n_a, int obs_dim, int hidden, int act_dim, const float *obs, const void *w1, const void *w2,
                      const void *w3, const void *w4, const float *bias, float *act, float noise_scale, float epsilon,
                      uint64_t seed, uint32_t step, int64_t env0, hipStream_t st);
int launch_sqrt(const double *in, double *out, int n, hipStream_t st);
int launch_cos_f32_err(uint32_t lo, uint32_t hi, double *out_max, hipStream_t st);
int launch_reward_compat(const DevParams &P, const double *p, const double *grid, int n_g, const int *nbr,
                         const int *inflags, const int *sensed, bool c_periodic, bool c_interaction,
                         bool c_exploration, double *reward, hipStream_t st);
int launch_sf_b2b_compat(const double *p, const double *d_edge, const bool *collide, const double *d_center, int n_a,
                         double k_ball, bool periodic, double hw, double hh, double *sf, hipStream_t st);
int launch_dist_b2w_compat(const double *p, const double *r, int n_a, const double *bpos_host, double *d_b2w,
                           bool *coll, hipStream_t st);
int launch_prior_compat(const DevParams &P, const double *p, const double *dp, const double *grid, int n_g,
                        const int *nbr, double l_cell, double *a_prior, hipStream_t st);

#ifdef ASM_DIAG
int diag_set_buffer(unsigned long long *dev);
int diag_set_filter(const void *p);
#endif

}  // namespace asmenv
