/*
 * TEST INFRASTRUCTURE — CPU oracle for the assembly-environment step path.
 *
 * A plain-C, double-precision restatement of the reference algorithm
 *   CPP = cus_gym/gym/envs/customized_envs/envs_cplus/src/AssemblyEnv.cpp
 *   ENV = cus_gym/gym/envs/customized_envs/assembly.py
 * Every function cites the reference lines it follows.  Arithmetic is kept in the
 * reference's operation order (compile with -ffp-contract=off, no -ffast-math) so
 * that, fed the same doubles, it returns the same doubles and the same indices.
 *
 * Parity pin: checked against the reference itself (oracle/_ref/libAssemblyEnv.so,
 * compiled from the reference sources, and the reference Python env) by
 * tests/test_oracle_vs_reference.py in the build container, and against the
 * golden vectors those produced (tests/golden/) everywhere.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (marl_amd/) never links, loads or calls it.
 */
#ifndef ASM_ORACLE_H
#define ASM_ORACLE_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the five reference entry points, same signatures (AssemblyEnv.h:13-109) ---- */
void oracle_get_observation(double *p, double *dp, double *heading, double *obs,
                            double *boundary_pos, double *grid_center, int *neighbor_index,
                            int *in_flags, int *sensed_index, int *occupied_index, double d_sen,
                            double r_avoid, double l_cell, double Vel_max, int topo_nei_max,
                            int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g,
                            int obs_dim_agent, int dim, bool *condition);
void oracle_get_reward(double *p, double *dp, double *heading, double *act, double *reward,
                       double *boundary_pos, double *grid_center, int *neighbor_index,
                       int *in_flags, int *sensed_index, int *occupied_index, double d_sen,
                       double r_avoid, double l_cell, int topo_nei_max, int num_obs_grid_max,
                       int num_occupied_grid_max, int n_a, int n_g, int dim, bool *condition,
                       bool *is_collide_b2b, bool *is_collide_b2w, double *coefficients);
void oracle_sf_b2b_all(double *p, double *sf_b2b, double *d_b2b_edge, bool *is_collide_b2b,
                       double *boundary_pos, double *d_b2b_center, int n_a, int dim, double k_ball,
                       bool is_periodic);
void oracle_get_dist_b2w(double *p, double *r, double *d_b2w, bool *isCollision, int dim, int n_a,
                         double *boundary_pos);
void oracle_calculateActionPrior(double *p, double *dp, double *a_prior, double *grid_center,
                                 int *neighbor_index, double d_sen, double r_avoid, double l_cell,
                                 int topo_nei_max, int n_a, int n_g, int dim);

/* ---- numpy glue of ENV restated ---- */
/* ENV:437-452 (+771-776, 787-793): centre distances, edge gaps, contact flags */
void oracle_get_dist_b2b(const double *p, int n_a, double size_a, bool is_periodic,
                         double half_w, double half_h, double *d_center, double *d_edge,
                         bool *is_collide);

/* Environment constants of ENV:17-89,194 that the step uses. */
typedef struct {
    int n_a;                 /* ENV:92 */
    int topo_nei_max;        /* ENV:33  = 6 */
    int num_obs_grid_max;    /* ENV:126 = 80 */
    int num_occupied_grid_max; /* ENV:128 = 200 */
    int is_boundary;         /* ENV:97-101; periodic = !is_boundary */
    int is_con_self_state;   /* ENV:105 */
    int llm_rl;              /* ENV:600: training_method == 'llm_rl' */
    double size_a;           /* ENV:43  = 0.035 */
    double d_sen;            /* ENV:194 = 0.4 */
    double r_avoid;          /* ENV:122 */
    double Vel_max;          /* ENV:51  = 0.8 */
    double k_ball, k_wall, c_wall; /* ENV:70-73 = 30, 100, 5 */
    double dt;               /* ENV:78  = 0.1 */
    double sensitivity;      /* ENV:80  = 1 */
    double mass;             /* ENV:39  = 1 */
    double half_w, half_h;   /* ENV:63-64 = 2.4 */
} oracle_cfg;

void oracle_default_cfg(oracle_cfg *c, int n_a, double r_avoid);
int oracle_obs_dim(const oracle_cfg *c); /* ENV:806 */

/* Which implementation of the five entry points the step driver calls:
 * NULL -> this file's restatement; otherwise dlopen(path) and bind the reference
 * names (_get_observation, ...).  Returns 0 on success. */
int oracle_bind(const char *so_path);

/*
 * One environment step, ENV:482-661, agent_strategy 'input', dynamics 'Cartesian'.
 * State (p, dp: (2,n_a) doubles; neighbor_index: (n_a,topo) from the previous
 * observation call) is updated in place.  Outputs: obs (obs_dim,n_a), reward (n_a),
 * a_prior (2,n_a) (zeros unless llm_rl), in_flags (n_a), sensed_index (n_a,80),
 * occupied_index (n_a,200), is_collide_b2b (n_a,n_a) of the PRE-integration state.
 * `act` is (2,n_a) float32 as the learner hands it over (TRAIN:121-123).
 */
void oracle_env_step(const oracle_cfg *c, double *p, double *dp, const float *act,
                     const double *grid_center, int n_g, double l_cell, int *neighbor_index,
                     double *obs, double *reward, double *a_prior, int *in_flags,
                     int *sensed_index, int *occupied_index, bool *is_collide_b2b);

/* The two halves of the step, for teacher-forced checking of an fp32 device path:
 *  dynamics : ENV:486-647 — forces, prior and integration from the OLD state.
 *  observe  : ENV:653-654 — observation + reward from the (given) NEW state.      */
void oracle_env_dynamics(const oracle_cfg *c, double *p, double *dp, const float *act,
                         const double *grid_center, int n_g, double l_cell,
                         const int *neighbor_index, double *a_prior, bool *is_collide_b2b);
void oracle_env_observe(const oracle_cfg *c, const double *p, const double *dp,
                        const double *grid_center, int n_g, double l_cell, int *neighbor_index,
                        double *obs, double *reward, int *in_flags, int *sensed_index,
                        int *occupied_index);

/* ---- formation metrics, WRAP = cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py ---- */
/* ENV:525-596: actions (2, n_a) of the 'rule' expert on the given state */
void oracle_rule_actions(const oracle_cfg *c, const double *p, const double *dp, const double *grid,
                         int n_g, double l_cell, double *act);

double oracle_coverage_rate(const double *p, int n_a, const double *grid, int n_g, double r_avoid); /* WRAP:48-72 */
double oracle_distribution_uniformity(const double *p, int n_a);                                   /* WRAP:74-101 */
double oracle_voronoi_uniformity(const double *p, int n_a, const double *grid, int n_g);           /* WRAP:103-129 */
int    oracle_count_collisions(const bool *is_collide_b2b, int n_a);                               /* WRAP:131-145 */
double oracle_avg_distance_to_target(const double *p, int n_a, const double *grid, int n_g);       /* WRAP:147-165 */

/* ---- counter-based RNG shared with the device reset (Philox4x32-10) ---- */
void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/*
 * Batched reset: domain randomisation of ENV:151-218 with draws taken from
 * Philox keyed by (seed, global env id) and counter (episode, draw index) instead
 * of numpy's global stream (the draw ORDER and ranges are the reference's).
 * shapes: n_shapes canonical grids, grid k at shape_grids + shape_off[k]*2 stored
 * as (n_g_k, 2) rows (the results.pkl 'grid_coords' layout, ENV:115).
 * Outputs for env e (global id env0+e): shape_id[e], p/dp (2,n_a) doubles,
 * grid (2,n_g_max) doubles (first n_g[e] columns valid), n_g[e], l_cell[e].
 */
void oracle_batch_reset(uint64_t seed, uint32_t episode, int64_t env0, int E, int n_a,
                        int n_shapes, const double *shape_grids, const int *shape_off,
                        const int *shape_ng, const double *shape_lcell, int n_g_max,
                        double half_w, double half_h, int *shape_id, double *p, double *dp,
                        double *grid, int *n_g, double *l_cell);

/* ---- multi-env CPU driver used as the timed CPU baseline ---- */
/* Runs `steps` env steps on E independent envs (state arrays as oracle_batch_reset
 * lays them out), actions act[t][e][2][n_a] float32, with `threads` OpenMP threads
 * (envs are split across threads).  Returns the sum of all rewards (a checksum). */
double oracle_run_envs(const oracle_cfg *c, int E, int steps, int threads, double *p, double *dp,
                       const float *act, const double *grid, int n_g_max, const int *n_g,
                       const double *l_cell);

#ifdef __cplusplus
}
#endif
#endif
