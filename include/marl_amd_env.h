/*
 * libAssemblyEnv.so — MI355X (gfx950) implementation of the assembly-environment
 * step path.  C ABI only: plain pointers, sizes and scalars.
 *
 * Reference files cited below (relative to the reference repository root):
 *   HDR  = cus_gym/gym/envs/customized_envs/envs_cplus/src/AssemblyEnv.h
 *   CPP  = cus_gym/gym/envs/customized_envs/envs_cplus/src/AssemblyEnv.cpp
 *   ENV  = cus_gym/gym/envs/customized_envs/assembly.py
 *   CLIB = cus_gym/gym/envs/customized_envs/envs_cplus/c_lib.py
 *
 * Two groups of entry points:
 *
 *  (A) The five symbols the reference environment binds through ctypes
 *      (CLIB:11-45 loads build/libAssemblyEnv.so; ENV:229, 352, 490, 455, 608 call
 *      them).  Same names, same argument lists, same row-major layouts, HOST
 *      pointers to double / int32 / bool, outputs overwritten in place, `void`
 *      return.  Each call copies its inputs to the GPU, runs the double-precision
 *      instantiation of the HIP kernels for one environment and copies the outputs
 *      back.  There is no CPU fallback.  The reference's symbols cannot fail and
 *      never end the host process (assembly.py:144-145 only prints on a bad
 *      mode); a call that cannot run here (no usable HIP device, any HIP error, an
 *      unsupported argument) fills its outputs with NaN / -1 / false, leaves the
 *      message in asm_last_error() and its code in asm_compat_status(), prints the
 *      message once per process and returns.  MARL_AMD_ABORT_ON_ERROR=1 in the
 *      environment turns such a failure into abort().
 *
 *  (B) The batched interface: E independent environments per launch, DEVICE
 *      pointers, fp32 state, a HIP stream argument, `int` status return
 *      (0 = ok; asm_last_error() gives the message).  No reference counterpart —
 *      the reference steps one environment per call (ENV:482-661); these are
 *      that step, the observation-only half used by reset (ENV:216) and the
 *      domain-randomised reset (ENV:151-218) over a batch.
 *      Nothing here allocates, frees or synchronises: every buffer is the
 *      caller's, work is enqueued on `stream`.
 */
#ifndef MARL_AMD_ENV_H
#define MARL_AMD_ENV_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASM_ABI_VERSION 2

/* ------------------------------------------------------------------ (A) -- */

/* replaces HDR:13-34 / CPP:23-356 (called at ENV:229-250) */
void _get_observation(double *p_input, double *dp_input, double *heading_input, double *obs_input,
                      double *boundary_pos_input, double *grid_center_input,
                      int *neighbor_index_input, int *in_flags_input, int *sensed_index_input,
                      int *occupied_index_input, double d_sen, double r_avoid, double l_cell,
                      double Vel_max, int topo_nei_max, int num_obs_grid_max,
                      int num_occupied_grid_max, int n_a, int n_g, int obs_dim_agent, int dim,
                      bool *condition);

/* replaces HDR:35-58 / CPP:359-631 (called at ENV:352-375) */
void _get_reward(double *p_input, double *dp_input, double *heading_input, double *act_input,
                 double *reward_input, double *boundary_pos_input, double *grid_center_input,
                 int *neighbor_index_input, int *in_flags_input, int *sensed_index_input,
                 int *occupied_index_input, double d_sen, double r_avoid, double l_cell,
                 int topo_nei_max, int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g,
                 int dim, bool *condition, bool *is_collide_b2b_input, bool *is_collide_b2w_input,
                 double *coefficients);

/* replaces HDR:64-73 / CPP:740-820 (called at ENV:490-499) */
void _sf_b2b_all(double *p_input, double *sf_b2b_input, double *d_b2b_edge_input,
                 bool *is_collide_b2b_input, double *boundary_pos_input, double *d_b2b_center_input,
                 int n_a, int dim, double k_ball, bool is_periodic);

/* replaces HDR:75-81 / CPP:822-860 (called at ENV:455-461) */
void _get_dist_b2w(double *p_input, double *r_input, double *d_b2w_input, bool *isCollision_input,
                   int dim, int n_a, double *boundary_pos);

/* replaces HDR:98-109 / CPP:1066-1123 (called at ENV:608-619) */
void calculateActionPrior(double *p_input, double *dp_input, double *a_prior_input,
                          double *grid_center_input, int *neighbor_index_input, double d_sen,
                          double r_avoid, double l_cell, int topo_nei_max, int n_a, int n_g, int dim);

/* Status of the calling thread's last call of one of the five symbols above: 0, or the error code whose text
 * asm_last_error() holds.  No counterpart in the reference (its symbols cannot fail); for hosts that want to
 * raise instead of reading NaN. */
int asm_compat_status(void);

/*
 * _get_observation for callers that cannot allocate device memory: tight serving loops that must stay off the
 * allocator, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, layouts and
 * results as _get_observation above (bit for bit); returns 0, or an error code with the text in asm_last_error().
 * The device staging area is a fixed 4 MiB array inside the library (one copy per GPU, held until the process
 * ends) instead of a workspace that grows on demand; a call whose arrays need more (about n_a = 1024 with the
 * default list capacities) is rejected.  The call makes no hipMalloc / hipFree / stream creation; it runs on the
 * legacy default stream of the current device with blocking copies and returns when the outputs are in the caller's
 * arrays (so it cannot be recorded into a HIP graph: a stream capture in progress rejects such copies).  Every call
 * on one GPU uses that GPU's array: calls on the same GPU must not overlap (calls through this entry point are
 * serialised by the library; other work that touches the array does not exist).
 */
int asm_get_observation_noalloc(double *p_input, double *dp_input, double *heading_input, double *obs_input,
                                double *boundary_pos_input, double *grid_center_input,
                                int *neighbor_index_input, int *in_flags_input, int *sensed_index_input,
                                int *occupied_index_input, double d_sen, double r_avoid, double l_cell,
                                double Vel_max, int topo_nei_max, int num_obs_grid_max,
                                int num_occupied_grid_max, int n_a, int n_g, int obs_dim_agent, int dim,
                                bool *condition);

/* ------------------------------------------------------------------ (B) -- */

/* Environment constants: the attributes ENV:17-89, 91-149, 194 that the step reads. */
typedef struct asm_params {
    int32_t n_a;                    /* ENV:92   agents per environment (1..1024) */
    int32_t n_g_max;                /* row length of the per-env grid arrays (>= every n_g[e]) */
    int32_t topo_nei_max;           /* ENV:33   must be 6 */
    int32_t num_obs_grid_max;       /* ENV:126  80 */
    int32_t num_occupied_grid_max;  /* ENV:128  200 */
    int32_t is_boundary;            /* ENV:97-101  1: walls, 0: periodic arena */
    int32_t is_con_self_state;      /* ENV:105  1: observation carries the agent's own p, dp */
    int32_t compute_prior;          /* ENV:600  training_method == 'llm_rl' */
    double size_a;                  /* ENV:43   0.035 */
    double d_sen;                   /* ENV:194  0.4 */
    double r_avoid;                 /* ENV:122 */
    double vel_max;                 /* ENV:51   0.8 */
    double k_ball, k_wall, c_wall;  /* ENV:70-73 30, 100, 5 */
    double dt;                      /* ENV:78   0.1 */
    double sensitivity;             /* ENV:80   1 */
    double mass;                    /* ENV:39   1 */
    double half_w, half_h;          /* ENV:63-64 2.4 */
} asm_params;

/* Fills the defaults listed above (r_avoid must still be set: ENV:121-122). */
void asm_default_params(asm_params *prm, int n_a, int n_g_max);

/* ENV:806: 2*dim*(topo_nei_max + 1 + self) + dim*num_obs_grid_max */
int asm_obs_dim(const asm_params *prm);

int asm_abi_version(void);
/* Tuning / test knob: which kernel asm_batch_*_f32 runs for environments of at most 32 agents: 4 (default), 2 or 1
 * wavefronts per workgroup of the multi-wave step kernel, 0 = the generic kernel, -1 = back to the default (the
 * MARL_AMD_FAST_T environment variable, read once per process, else 4).  + 16: large swarms (256 agents and more) run
 * their pair passes (contacts, nearest neighbours) over all agents instead of through cell lists.  + 32: large swarms
 * of 512 agents and more on the lattice path step in three launches (dynamics; observation split over 256-agent
 * workgroups; statistics) instead of one.  + 128: the multi-wave kernel stores observation rows the ordinary way
 * instead of non-temporal (the default on its lattice path: rows stored the ordinary way push the grids and the state
 * out of the last-level cache between two steps; a reader that wants the rows cached sets this bit); + 64 is accepted
 * and means the default.  All produce the same results. */
void asm_set_step_schedule(int waves);
const char *asm_last_error(void);
/* Number of visible HIP devices, or a negative HIP error code. */
int asm_device_count(void);

/*
 * Target-grid bitmap.  Every shape the reference derives from an image is a subset of a regular lattice: cells
 * are the all-black 36-pixel tiles, listed row by row (assembly_cfg.py:60-80), so cell index order is
 * (lattice row, lattice column) order.  asm_lattice_shape holds that lattice for one shape: per lattice row the
 * occupied columns as bit windows with the index of the first cell at or after the window start, and the list of
 * boundary cells (cells with a missing 4-neighbour: the only cells, besides the four lattice sites around a query
 * point, that can be nearest to it).  Given it, and per environment the shape id and the rigid transform the reset
 * applied (assembly.py:170-182), the step kernel finds an agent's sensed cells by enumerating the lattice rows its
 * sensing disc crosses (each row contributes one contiguous run of cell indices) instead of testing every cell,
 * and its nearest cell among boundary cells + 4 sites.  Every decision stays the reference's: anything within
 * rounding distance of a threshold is re-evaluated in double on the stored cell coordinates.
 *
 * The description is a promise by the caller: grid[e] must hold, rounded to float, the cells of shape
 * shape_id[e] transformed by xform[e]:  x = c*gx + s*gy + ox,  y = -s*gx + c*gy + oy  (c, s, ox, oy = xform[e][0..3]).
 * asm_batch_reset_f32 produces exactly that when given xform / shape_id outputs.  Pass lattice = NULL whenever the
 * grid arrays were written any other way.
 */
#define ASM_LAT_ROWS   48    /* table rows: lattice rows + one empty row below and above (rows <= 46) */
#define ASM_LAT_CHUNKS 3     /* 32-bit column windows per row, starting at columns 0, 16, 32 (columns <= 48) */
#define ASM_LAT_NB     256   /* boundary-list capacity */

typedef struct asm_lattice_shape {
    uint32_t row_tab[ASM_LAT_ROWS][ASM_LAT_CHUNKS][2];  /* [row + 1][j] = { occupancy bits of columns 16j .. 16j+31,
                                                           index of the first cell of the row at a column >= 16j } */
    uint16_t boundary[ASM_LAT_NB];                      /* boundary cell indices, ascending */
    int32_t n_boundary, cols, rows, n_g;
    double x0, y0;                                      /* canonical coordinates of lattice site (column 0, row 0) */
    double l_cell;                                      /* lattice pitch */
    double reserved;
} asm_lattice_shape;

/* Host-side: fills *out from the canonical cell centres grid_xy[n_g][2] (the 'grid_coords' entry of results.pkl,
 * assembly.py:115) and the cell size.  Returns 0, or non-zero with asm_last_error() set when the cells are not a
 * row-major subset of a lattice of pitch l_cell or exceed the table sizes above (then run without a lattice). */
int asm_lattice_build(const double *grid_xy, int n_g, double l_cell, asm_lattice_shape *out);

typedef struct asm_lattice {
    const asm_lattice_shape *shapes;   /* device memory, n_shapes entries */
    int32_t n_shapes;
    int32_t reserved;
    const int32_t *shape_id;           /* device [E]     shape of every environment */
    const double *xform;               /* device [E][4]  cos, sin, offset x, offset y of every environment */
    double l_cell_min;                 /* smallest l_cell among the shapes (host value; the kernels' lattice window
                                          covers d_sen <= 7.4 l_cell_min and r_avoid/2 <= 3.4 l_cell_min; beyond
                                          that the calls run as if lattice were NULL) */
} asm_lattice;

/* Number of cells of grid[e] that differ from the lattice description by more than tol, summed over the E
 * environments, written to *n_bad (one device int32, zeroed by this call's kernel).  A debugging aid for callers
 * that maintain grid / shape_id / xform themselves. */
int asm_batch_check_lattice(const asm_params *prm, int E, const float *grid, const int32_t *n_g,
                            const asm_lattice *lattice, double tol, int32_t *n_bad, void *stream);

/*
 * Array layouts (all contiguous, device memory, E environments):
 *   p, dp, act, a_prior  [E][2][n_a]        float   (component-major, as ENV's (2, n_a))
 *   grid                 [E][2][n_g_max]    float   (ENV's grid_center (2, n_g), padded)
 *   n_g                  [E]                int32
 *   l_cell               [E]                double  (ENV:158)
 *   neighbor_index       [E][n_a][6]        int32   -1 padded   (ENV:223)
 *   in_flags             [E][n_a]           int32               (ENV:224)
 *   nearest_cell         [E][n_a]           int32   index of the nearest grid cell
 *   obs                  [E][obs_dim][n_a]  float   (ENV's (obs_dim, n_a) per env)
 *   reward               [E][n_a]           float
 *   sensed_index         [E][n_a][80]       int32   -1 padded   (ENV:225)  optional (NULL)
 *   occupied_index       [E][n_a][200]      int32   -1 padded   (ENV:226)  optional (NULL)
 *   stats                [E][8]             double  optional (NULL), accumulated (+=):
 *        [0] sum of rewards  [1] sum of squared rewards  [2] coverage_rate
 *        [3] distribution_uniformity  [4] contact pairs (pre-integration state)
 *        [5] average_distance_to_target  [6] 1 (step count)  [7] agents inside the shape
 *        (the per-step quantities TRAIN:128-135 accumulates; metrics of WRAP:48-165).
 *        The step kernel of swarms of at most 32 agents adds with memory-side double-precision
 *        atomic adds that return nothing (one adding lane per entry and launch, so the sums are
 *        those of a load / add / store): stats must lie in ordinary device memory (hipMalloc,
 *        a torch CUDA tensor), not in host-pinned or fine-grained memory.
 */

/*
 * Observation half of the step on a given state: what ENV.reset (ENV:216) and
 * ENV.step (ENV:653-654) compute after the state is final.  Writes obs,
 * neighbor_index, in_flags, nearest_cell (+ optional sensed/occupied, reward).
 */
int asm_batch_observe_f32(const asm_params *prm, int E, const float *p, const float *dp,
                          const float *grid, const int32_t *n_g, const double *l_cell,
                          int32_t *neighbor_index, int32_t *in_flags, int32_t *nearest_cell,
                          float *obs, float *reward /*nullable*/, int32_t *sensed_index /*nullable*/,
                          int32_t *occupied_index /*nullable*/, const asm_lattice *lattice /*nullable*/, void *stream);

/*
 * One full step, ENV:482-661 with agent_strategy 'input', for E environments.
 * In:  p, dp (updated in place), act, grid/n_g/l_cell, and the three arrays the
 *      previous observation call left: neighbor_index, in_flags, nearest_cell
 *      (the prior of CPP:1066-1123 is evaluated on the pre-integration state with
 *      the previous neighbour list, ENV:607-619; its nearest-cell search is the one
 *      the previous observation already did on that same state, so it is reused —
 *      pass recompute_target != 0 if p or grid were changed since).
 * Out: obs, reward, a_prior (NULL or compute_prior == 0: skipped), refreshed
 *      neighbor_index / in_flags / nearest_cell, optional sensed/occupied/stats.
 */
int asm_batch_step_f32(const asm_params *prm, int E, float *p, float *dp, const float *act,
                       const float *grid, const int32_t *n_g, const double *l_cell,
                       int32_t *neighbor_index, int32_t *in_flags, int32_t *nearest_cell,
                       float *obs, float *reward, float *a_prior /*nullable*/,
                       int32_t *sensed_index /*nullable*/, int32_t *occupied_index /*nullable*/,
                       double *stats /*nullable*/, int recompute_target, const asm_lattice *lattice /*nullable*/,
                       void *stream);

/*
 * Batched reset, ENV:151-218: per environment a shape, a rotation, an offset and
 * initial positions / velocities, drawn from Philox4x32-10 keyed by
 * (seed, global environment id = env0 + e) with counter (draw index, episode), so
 * any split of the id range over GPUs yields the same environments.
 * Shapes: n_shapes canonical grids; grid k has shape_ng[k] cells stored as
 * (x, y) rows starting at shape_grids + 2*shape_off[k] (the 'grid_coords' layout
 * ENV:115 loads), cell size shape_lcell[k] — all device memory, doubles / int32.
 * Writes shape_id, p, dp, grid, n_g, l_cell (and, if given, the rotation and offset it drew as xform); then
 * the observation of the new state exactly as asm_batch_observe_f32 (reward is not produced, ENV:216-218).
 */
int asm_batch_reset_f32(const asm_params *prm, uint64_t seed, uint32_t episode, int64_t env0, int E,
                        int n_shapes, const double *shape_grids, const int32_t *shape_off,
                        const int32_t *shape_ng, const double *shape_lcell, int32_t *shape_id,
                        float *p, float *dp, float *grid, int32_t *n_g, double *l_cell,
                        int32_t *neighbor_index, int32_t *in_flags, int32_t *nearest_cell,
                        float *obs, int32_t *sensed_index /*nullable*/,
                        int32_t *occupied_index /*nullable*/, double *xform /*nullable: [E][4] cos, sin, offset x, y */,
                        const asm_lattice *lattice /*nullable; its shape_id / xform must be the arrays written here */,
                        void *stream);

/* U(-1,1) float32 actions from the same generator (ENV:519 'random' strategy):
 * act[e][c][i] = draw (seed, env0+e, counter stream 1, step, c*n_a+i). */
int asm_batch_random_actions_f32(uint64_t seed, uint32_t step, int64_t env0, int E, int n_a,
                                 float *act, void *stream);

/*
 * The in-environment 'rule' expert (assembly.py:525-596; what eval/collect_expert_data.py drives): actions
 * [E][2][n_a] for the state the last observation call saw, from that call's in_flags, nearest_cell and
 * sensed_index (required here: the lists of assembly.py:539-567 are exactly the observation's sensed lists).
 * Evaluated in double in the reference's order, clipped to [-1, 1] and rounded to float (the reference
 * feeds its dynamics the double values: the difference is one float32 rounding of the action).
 */
int asm_batch_rule_actions_f32(const asm_params *prm, int E, const float *p, const float *dp, const float *grid,
                               const int32_t *n_g, const int32_t *in_flags, const int32_t *nearest_cell,
                               const int32_t *sensed_index, float *act, void *stream);

/*
 * Formation metrics of the wrapper (assembly_wrapper.py:48-165, get_formation_metrics :167-187) for the
 * current positions of every environment, evaluated in double with the reference's definitions:
 *   metrics [E][5] double = { coverage_rate (:48-72), distribution_uniformity (:74-101),
 *                             voronoi_based_uniformity (:103-129), contact pairs of the given positions
 *                             (upper triangle of ENV:445-446, what :131-145 counts), average_distance_to_target
 *                             (:147-165) }
 * eval_assembly.py:156-158 evaluates them once per step of an evaluation episode.  The two uniformity
 * values are 0/0 = NaN when all agents are equidistant / own equally many cells, as in the reference.
 */
int asm_batch_metrics_f64(const asm_params *prm, int E, const float *p, const float *grid, const int32_t *n_g,
                          double *metrics, void *stream);

/* ---- Group C: on-device replay of batched rollouts ------------------------------------------
 * Counterpart of ReplayBufferAgent (marl_llm/algorithm/utils/buffer_agent.py:21-177) for the batched
 * environment.  Storage is caller-owned device memory, organised by episode slots of `episode_length`
 * steps so that the step kernel's own outputs are the stored rows (obs is written once and serves as
 * next_obs of the previous step, buffer_agent.py:107-110):
 *   obs_ring   [slots][episode_length + 1][E][obs_dim][n_a]   frame t = observation before step t
 *   act_ring   [slots][episode_length][E][2][n_a]             actions (train_assembly.py:121-125)
 *   prior_ring [slots][episode_length][E][2][n_a]             a_prior of the step (nullable)
 *   rew_ring   [slots][episode_length][E][n_a]                rewards
 * A transition id is ((slot * episode_length + t) * E + e) * n_a + i.  Rows come out agent-major like
 * buffer_agent.py:157-177: obs_out/next_obs_out [N][obs_dim], act_out/prior_out [N][2], rew_out [N].
 * done is all zeros in this environment (assembly.py:475-477) and is not stored. */
int asm_replay_gather_f32(int E, int n_a, int obs_dim, int episode_length, const float *obs_ring,
                          const float *act_ring, const float *prior_ring /*nullable*/, const float *rew_ring,
                          const int64_t *index /*[N], device*/, int N, float *obs_out, float *act_out,
                          float *rew_out, float *next_obs_out, float *prior_out /*nullable*/, void *stream);

/* index[k] = lo + floor(u_k * (hi - lo)) with u_k from the counter-based generator (key = seed,
 * counter = (k, counter), stream 2): uniform transition ids WITH replacement (buffer_agent.py:149-154
 * draws without replacement; with >= 1e7 stored rows and 2048 draws the difference is ~2e-4 expected
 * duplicates per batch).  counter_dev (nullable): one device word added to `counter` when the kernel runs, so that a
 * launch captured into a HIP graph draws new ids on every replay once the caller increments that word. */
int asm_replay_draw_indices(uint64_t seed, uint32_t counter, const uint32_t *counter_dev /*nullable*/, int64_t lo,
                            int64_t hi, int N, int64_t *index, void *stream);

/* Observation frames in the layout the reference hands its networks: rows[e][i][r] = obs[e][r][i], i.e. one row of
 * obs_dim floats per agent (maddpg.py:84 transposes the (obs_dim, n_a) frame for the policy, train_assembly.py:119-120;
 * buffer_agent.py:95-117 stores agent rows).  obs: [E][obs_dim][n_a] as the step writes it, rows: [E][n_a][obs_dim];
 * both device memory, float32.  A tiled transpose through LDS at copy speed: PyTorch's strided copy of the same
 * view takes 3.5 times as long at 4096 x 30 agents. */
int asm_obs_rows_f32(int E, int n_a, int obs_dim, const float *obs, float *rows, void *stream);

/* Rollout inference of the policy network for every agent of every environment, on the matrix cores (bf16 operands, fp32
 * accumulation): what MADDPG.step -> DDPGAgent.step -> MLPNetwork.forward compute one environment at a time
 * (maddpg.py:72-87, agents.py:69-96 without the exploration noise, networks.py:33-44: Linear - leaky_relu three times,
 * Linear - tanh).  obs: [E][obs_dim][n_a] as the step writes it; act: [E][act_dim][n_a] as the step reads it.  Weights:
 * one 16-byte fragment of 8 bf16 per lane, packed by the caller (marl_amd/policy_kernel.py: FusedPolicyBF16.repack):
 *   w1 [K1][H/32][64][8] (k-step s, row tile m, lane l, element j), K1 = ceil(obs_dim / 16):
 *       W1[32 m + (l & 31)][16 s + 8 (l >> 5) + j]
 *   w2, w3 [H/16][H/32][64][8]: W[32 m + (l & 31)][32 (s >> 1) + 16 (s & 1) + 8 (j >> 2) + 4 (l >> 5) + (j & 3)]
 *   w4 [H/16][1][64][8]: the same with the act_dim output rows padded to 32
 * (zero beyond the matrices' bounds), bias: [3 H + 32] floats (b1, b2, b3, b4 padded).  H = 32, 64, 128 or 256,
 * act_dim <= 32.  Values differ from the float32 network by bf16 rounding of weights and activations (about 1e-2). */
int asm_policy_mlp_forward_bf16(int E, int n_a, int obs_dim, int hidden, int act_dim, const float *obs, const void *w1,
                                const void *w2, const void *w3, const void *w4, const float *bias, float *act,
                                void *stream);
/* The same with the exploration of agents.py:86-94 applied in the kernel's epilogue: per environment, with probability
 * epsilon every agent's action is drawn uniformly from [-1, 1) (:87-89), else Gaussian noise of standard deviation
 * noise_scale is added (noise.py:29-30) and the action clamped to [-1, 1] (:91-94).  Counter-based draws (Philox4x32-10)
 * keyed by (seed, env0 + e, agent, output pair, step): the caller advances `step` once per call; the numbers do not
 * depend on how the environments are split over calls or GPUs.  noise_scale = epsilon = 0: no exploration. */
int asm_policy_mlp_act_bf16(int E, int n_a, int obs_dim, int hidden, int act_dim, const float *obs, const void *w1,
                            const void *w2, const void *w3, const void *w4, const float *bias, float *act,
                            float noise_scale, float epsilon, uint64_t seed, uint32_t step, int64_t env0, void *stream);

/* Self-test hook: which kernel asm_batch_step_f32 would launch for these arguments: waves per workgroup of the
 * multi-wave kernel (4, 2, 1) or 0 for the generic kernel, plus 16 if its lattice path is taken, plus 32 if the
 * generic kernel's (large swarms), plus 64 if the pair passes go through cell lists, plus 128 for the three-launch
 * step, plus 256 when the large swarm's own instantiation of the generic kernel runs; negative on bad arguments.
 * Host-side only. */
int asm_debug_step_schedule(const asm_params *prm, int E, const asm_lattice *lattice /*nullable*/);

/* Self-test hook: out[k] = sqrt(in[k]) evaluated on the device in double (the
 * kernels rely on it being correctly rounded); n doubles, device pointers. */
int asm_debug_sqrt_f64(const double *in, double *out, int n, void *stream);

/* Self-test hook: *out_max = max(*out_max, max |v_cos_f32(x) - cos(2 pi x)|) over every float x whose
 * bit pattern lies in [bits_lo, bits_hi] (the hardware cosine takes revolutions).  The single-precision
 * screen of the reward (CPP:515-552) budgets 5e-7 for it.  out_max: one device double, zeroed by the caller. */
int asm_debug_cos_f32_max_err(uint32_t bits_lo, uint32_t bits_hi, double *out_max, void *stream);

#ifdef __cplusplus
}
#endif
#endif
