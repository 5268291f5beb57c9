/*
 * TEST INFRASTRUCTURE — CPU oracle (see asm_oracle.h for the contract).
 *
 * CPP = cus_gym/gym/envs/customized_envs/envs_cplus/src/AssemblyEnv.cpp
 * ENV = cus_gym/gym/envs/customized_envs/assembly.py
 * WRAP = cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py
 *
 * Written as flat loops over the reference's row-major arrays; no per-agent heap
 * objects.  Floating-point expressions keep the reference's operand order.
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile).  Never -ffast-math.
 */
#define _GNU_SOURCE
#include "asm_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* sqrt(x*x + y*y): CPP:641, 162, 1148 and _norm CPP:999-1005 (pow(v,2) == v*v) */
static inline double norm2(double x, double y) { return sqrt(x * x + y * y); }

/* CPP:705-720, is_rel == true branch, one column */
static inline void wrap_rel(double *x, double *y, double half_w, double half_h)
{
    if (*x < -half_w) *x += 2 * half_w;
    else if (*x > half_w) *x -= 2 * half_w;
    if (*y < -half_h) *y += 2 * half_h;
    else if (*y > half_h) *y -= 2 * half_h;
}

/* CPP:1017-1025 */
static double rho_cos_dec(double z, double delta, double r)
{
    if (z < delta * r) return 1.0;
    else if (z < r) return (1.0 / 2.0) * (1.0 + cos(M_PI * (z / r - delta) / (1.0 - delta)));
    else return 0.0;
}

/* ------------------------------------------------------------------------- */
/* CPP:863-913 _get_target_grid_state for one agent.
 * norms[] (n_g) receives every cell distance; returns the first-minimum index.
 * in_flag = min_dist < sqrt(2)*l_cell/2 (CPP:894).  sensed = ascending cells with
 * norm < d_sen (CPP:905-910); count returned through n_sensed. */
static int target_grid_state(int i, const double *p, int n_a, const double *grid, int n_g,
                             double l_cell, double d_sen, double *norms, int *in_flag,
                             int *sensed, int *n_sensed)
{
    const double px = p[i], py = p[n_a + i];
    int min_index = 0;
    for (int j = 0; j < n_g; ++j) {
        double rx = grid[j] - px, ry = grid[n_g + j] - py;       /* CPP:876-877 */
        norms[j] = norm2(rx, ry);                                /* CPP:885 */
        if (norms[j] < norms[min_index]) min_index = j;          /* CPP:889 first minimum */
    }
    *in_flag = norms[min_index] < sqrt(2) * l_cell / 2;          /* CPP:894 */
    int ns = 0;
    if (sensed) {
        for (int j = 0; j < n_g; ++j)
            if (norms[j] < d_sen) sensed[ns++] = j;              /* CPP:906-910 */
        *n_sensed = ns;
    }
    return min_index;
}

/* ------------------------------------------------------------------------- */
/* CPP:23-356 */
void oracle_get_observation(double *p, double *dp, double *heading, double *obs,
                            double *boundary_pos, double *grid, int *neighbor_index,
                            int *in_flags, int *sensed_index, int *occupied_index, double d_sen,
                            double r_avoid, double l_cell, double Vel_max, int topo_nei_max,
                            int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g,
                            int obs_dim_agent, int dim, bool *condition)
{
    (void)Vel_max; (void)dim;                      /* Vel_max unused (CPP:36); dim == 2 */
    const bool is_periodic = condition[0], is_cartesian = condition[1], con_self = condition[2];
    const double half_w = (boundary_pos[2] - boundary_pos[0]) / 2.0;   /* CPP:75 */
    const double half_h = (boundary_pos[1] - boundary_pos[3]) / 2.0;   /* CPP:76 */

    double *norm_a = (double *)malloc(sizeof(double) * (size_t)n_a);
    double *sq_a = (double *)malloc(sizeof(double) * (size_t)n_a);
    double *rel = (double *)malloc(sizeof(double) * 2 * (size_t)n_a);
    int *order = (int *)malloc(sizeof(int) * (size_t)n_a);
    double *norm_g = (double *)malloc(sizeof(double) * (size_t)(n_g > 0 ? n_g : 1));
    int *sensed = (int *)malloc(sizeof(int) * (size_t)(n_g > 0 ? n_g : 1));
    int *sensed0 = (int *)malloc(sizeof(int) * (size_t)(n_g > 0 ? n_g : 1));
    int *occupied = (int *)malloc(sizeof(int) * (size_t)(n_g > 0 ? n_g : 1));

    /* outputs are fully overwritten (CPP:49-55, 329-355) */
    for (int k = 0; k < obs_dim_agent * n_a; ++k) obs[k] = 0.0;
    for (int k = 0; k < n_a * topo_nei_max; ++k) neighbor_index[k] = -1;
    for (int k = 0; k < n_a * num_obs_grid_max; ++k) sensed_index[k] = -1;
    for (int k = 0; k < n_a * num_occupied_grid_max; ++k) occupied_index[k] = -1;

    for (int i = 0; i < n_a; ++i) {
        const double px = p[i], py = p[n_a + i];
        /* --- relative positions (CPP:82-95) and k-NN (_get_focused, CPP:633-703) --- */
        for (int j = 0; j < n_a; ++j) {
            double rx = p[j] - px, ry = p[n_a + j] - py;
            if (is_periodic) wrap_rel(&rx, &ry, half_w, half_h);
            rel[j] = rx; rel[n_a + j] = ry;
            sq_a[j] = rx * rx + ry * ry;
            norm_a[j] = sqrt(sq_a[j]);                             /* CPP:641 */
            order[j] = j;
        }
        /* sort by norm ascending (CPP:646).  std::sort leaves the order of equal norms
         * unspecified (what the compiled reference returns for them is an artefact of
         * libstdc++'s introsort and depends on all n_a norms); here equal norms are ordered by
         * the squared norm and then by the lower index, i.e. the list is sorted by (squared
         * norm, index) -- the same order as by norm wherever the norms differ (insertion sort,
         * stable). */
        for (int a = 1; a < n_a; ++a) {
            int v = order[a]; int b = a - 1;
            while (b >= 0 && sq_a[order[b]] > sq_a[v]) { order[b + 1] = order[b]; --b; }
            order[b + 1] = v;
        }
        /* keep norm < d_sen (CPP:663,672), drop the first kept (remove_self, CPP:677-681),
         * take at most topo_nei_max (CPP:692-700) */
        int kept = 0, n_nei = 0;
        double *col = obs + i;                                    /* obs[row*n_a + i] */
        const int self_rows = con_self ? 4 : 0;                   /* CPP:108-122 */
        if (con_self && is_cartesian) {
            col[0 * n_a] = px; col[1 * n_a] = py; col[2 * n_a] = dp[i]; col[3 * n_a] = dp[n_a + i];
        } else if (con_self) {
            col[0 * n_a] = px; col[1 * n_a] = py; col[2 * n_a] = dp[i]; col[3 * n_a] = dp[n_a + i];
        }
        for (int a = 0; a < n_a; ++a) {
            int j = order[a];
            if (!(norm_a[j] < d_sen)) continue;
            if (kept++ == 0) continue;                            /* the erased first element */
            if (n_nei >= topo_nei_max) break;
            neighbor_index[i * topo_nei_max + n_nei] = j;         /* CPP:103-105 */
            double vx, vy;
            if (is_cartesian) { vx = dp[j] - dp[i]; vy = dp[n_a + j] - dp[n_a + i]; }      /* CPP:86 */
            else { vx = heading[j] - heading[i]; vy = heading[n_a + j] - heading[n_a + i]; } /* CPP:88 */
            /* column-major flatten of rows [pos_x,pos_y,vel_x,vel_y] (CPP:124-131) */
            int r0 = self_rows + 4 * n_nei;
            col[(r0 + 0) * n_a] = rel[j]; col[(r0 + 1) * n_a] = rel[n_a + j];
            col[(r0 + 2) * n_a] = vx;     col[(r0 + 3) * n_a] = vy;
            ++n_nei;
        }
        const int agent_rows = self_rows + 4 * topo_nei_max;

        /* --- target state (CPP:137-142) --- */
        int in_flag, n_sensed = 0;
        int min_index = target_grid_state(i, p, n_a, grid, n_g, l_cell, d_sen, norm_g, &in_flag,
                                          sensed, &n_sensed);
        in_flags[i] = in_flag;
        double tpx, tpy, tvx, tvy;
        if (in_flag) { tpx = px; tpy = py; tvx = dp[i]; tvy = dp[n_a + i]; }   /* CPP:896-897 */
        else { tpx = grid[min_index]; tpy = grid[n_g + min_index]; tvx = 0.0; tvy = 0.0; } /* CPP:900-901 */
        const double t_rel[4] = {tpx - px, tpy - py, tvx - dp[i], tvy - dp[n_a + i]};

        /* --- remove occupied cells (CPP:145-212) --- */
        const int n_sensed0 = n_sensed;
        memcpy(sensed0, sensed, sizeof(int) * (size_t)n_sensed);              /* CPP:148 */
        if (n_sensed0 > 0 && in_flag == 1) {
            for (int j = 0; j < n_a; ++j) {                                   /* ascending j, CPP:165-171 */
                double dx = p[j] - px, dy = p[n_a + j] - py;
                double dn = sqrt(dx * dx + dy * dy);                          /* CPP:162 */
                if (!(dn < (d_sen + r_avoid / 2.0))) continue;                /* CPP:166 */
                int m = 0;
                for (int k = 0; k < n_sensed; ++k) {
                    double gx = grid[sensed[k]] - p[j], gy = grid[n_g + sensed[k]] - p[n_a + j];
                    if (norm2(gx, gy) > r_avoid / 2.0) sensed[m++] = sensed[k]; /* CPP:190 */
                }
                n_sensed = m;
            }
        }
        /* occupied = sensed_before \ sensed_after, order kept (CPP:215-221) */
        int n_occ = 0;
        {
            int k2 = 0;
            for (int k = 0; k < n_sensed0; ++k) {
                if (k2 < n_sensed && sensed[k2] == sensed0[k]) ++k2;          /* both ascending */
                else occupied[n_occ++] = sensed0[k];
            }
        }
        if (n_occ > num_occupied_grid_max) {                                  /* CPP:223-233 */
            double step = (double)(n_occ - 1) / (num_occupied_grid_max - 1);
            for (int k = 0; k < num_occupied_grid_max; ++k)
                occupied_index[i * num_occupied_grid_max + k] = occupied[(int)round(k * step)];
        } else {
            for (int k = 0; k < n_occ; ++k) occupied_index[i * num_occupied_grid_max + k] = occupied[k];
        }
        /* sensed list, uniform subsample above the cap (CPP:243-276) */
        int n_out;
        if (n_sensed > num_obs_grid_max) {
            double step = (double)(n_sensed - 1) / (num_obs_grid_max - 1);    /* CPP:246 */
            for (int k = 0; k < num_obs_grid_max; ++k)
                sensed_index[i * num_obs_grid_max + k] = sensed[(int)round(k * step)]; /* CPP:250 */
            n_out = num_obs_grid_max;
        } else {
            for (int k = 0; k < n_sensed; ++k) sensed_index[i * num_obs_grid_max + k] = sensed[k];
            n_out = n_sensed;
        }

        /* --- write the column (CPP:299-325) --- */
        if (is_cartesian) {
            /* rows: [agent block | target_pos_rel | target_vel_rel | num_obs_grid_max x (x,y)] */
            const int base = obs_dim_agent - (2 + num_obs_grid_max) * 2;      /* == agent_rows */
            (void)agent_rows;
            col[(base + 0) * n_a] = t_rel[0]; col[(base + 1) * n_a] = t_rel[1];
            col[(base + 2) * n_a] = t_rel[2]; col[(base + 3) * n_a] = t_rel[3];
            const int gbase = obs_dim_agent - num_obs_grid_max * 2;
            for (int k = 0; k < n_out; ++k) {                                 /* CPP:279-296 */
                int c = sensed_index[i * num_obs_grid_max + k];
                col[(gbase + 2 * k + 0) * n_a] = grid[c] - px;
                col[(gbase + 2 * k + 1) * n_a] = grid[n_g + c] - py;
            }
        } else {
            /* rows: [agent block | target_pos_rel | target_vel_rel | heading] (CPP:312-325) */
            const int base = obs_dim_agent - 3 * 2;
            col[(base + 0) * n_a] = t_rel[0]; col[(base + 1) * n_a] = t_rel[1];
            col[(base + 2) * n_a] = t_rel[2]; col[(base + 3) * n_a] = t_rel[3];
            col[(base + 4) * n_a] = heading[i]; col[(base + 5) * n_a] = heading[n_a + i];
        }
    }
    free(norm_a); free(sq_a); free(rel); free(order); free(norm_g); free(sensed); free(sensed0); free(occupied);
}

/* ------------------------------------------------------------------------- */
/* CPP:359-631; live part 457-564 and 628-630 */
void oracle_get_reward(double *p, double *dp, double *heading, double *act, double *reward,
                       double *boundary_pos, double *grid, int *neighbor_index, int *in_flags,
                       int *sensed_index, int *occupied_index, double d_sen, double r_avoid,
                       double l_cell, int topo_nei_max, int num_obs_grid_max,
                       int num_occupied_grid_max, int n_a, int n_g, int dim, bool *condition,
                       bool *is_collide_b2b, bool *is_collide_b2w, double *coefficients)
{
    (void)dp; (void)heading; (void)act; (void)occupied_index; (void)l_cell; (void)dim;
    (void)num_occupied_grid_max; (void)is_collide_b2b; (void)is_collide_b2w; (void)coefficients;
    const double half_w = (boundary_pos[2] - boundary_pos[0]) / 2.0;   /* CPP:454 */
    const double half_h = (boundary_pos[1] - boundary_pos[3]) / 2.0;   /* CPP:455 */
    for (int i = 0; i < n_a; ++i) {
        bool is_collision = false, is_uniform = false;
        if (condition[3]) {                                            /* CPP:464-496 */
            for (int k = 0; k < topo_nei_max; ++k) {
                int j = neighbor_index[i * topo_nei_max + k];
                if (j == -1) continue;
                double rx = p[j] - p[i], ry = p[n_a + j] - p[n_a + i];
                if (condition[0]) wrap_rel(&rx, &ry, half_w, half_h);  /* CPP:479-482 */
                if (r_avoid > norm2(rx, ry)) { is_collision = true; break; } /* CPP:487-491 */
            }
        }
        double r = 0.0;
        if (condition[4]) {                                            /* CPP:499-564 */
            if (in_flags[i] == 1) {
                double num0 = 0.0, num1 = 0.0, den = 0.0;
                int any = 0;
                for (int k = 0; k < num_obs_grid_max; ++k) {
                    int c = sensed_index[i * num_obs_grid_max + k];
                    if (c == -1) continue;                             /* CPP:505-509 */
                    any = 1;
                    double gx = grid[c] - p[i], gy = grid[n_g + c] - p[n_a + i]; /* CPP:515-516 */
                    double psi = rho_cos_dec(norm2(gx, gy), 0.0, d_sen);          /* CPP:524,530 */
                    num0 += psi * gx; num1 += psi * gy; den += psi;              /* CPP:537-539 */
                }
                if (any) {
                    if (den == 0) den = 1E-8;                          /* CPP:542-544 */
                    double v0 = 1.0 * num0 / den, v1 = 1.0 * num1 / den; /* CPP:547-548 */
                    if (norm2(v0, v1) < 0.05) is_uniform = true;       /* CPP:550-554 */
                }
            }
            if (in_flags[i] == 1 && !is_collision && is_uniform) r += 1.0; /* CPP:559-561 */
        }
        reward[i] = r;                                                 /* CPP:628-630 */
    }
}

/* ------------------------------------------------------------------------- */
/* CPP:740-820.  Entry (row 2i+c, column k) of the reference's dense matrix:
 *   k <  i : collide[i][k]*d_edge[i][k]*k_ball*(-(delta_c/d_center[i][k])), delta = p_k - p_i
 *   k >  i : minus the (row 2k+c, column i) entry                       (CPP:795-796)
 *   k == i : 0
 * and sf_b2b[c][i] = sum over k ascending (CPP:804-812). */
static inline double sf_entry(const double *p, const double *d_edge, const bool *collide,
                              const double *d_center, int n_a, int i, int j, int c,
                              double k_ball, bool periodic, double half_w, double half_h)
{   /* requires j < i */
    double dx = p[j] - p[i], dy = p[n_a + j] - p[n_a + i];           /* CPP:782-785 */
    if (periodic) wrap_rel(&dx, &dy, half_w, half_h);
    double d = (c == 0 ? dx : dy) / d_center[i * n_a + j];            /* CPP:790-791 */
    return (double)collide[i * n_a + j] * d_edge[i * n_a + j] * k_ball * (-d); /* CPP:792-793 */
}

void oracle_sf_b2b_all(double *p, double *sf_b2b, double *d_b2b_edge, bool *is_collide_b2b,
                       double *boundary_pos, double *d_b2b_center, int n_a, int dim, double k_ball,
                       bool is_periodic)
{
    (void)dim;
    const double half_w = (boundary_pos[2] - boundary_pos[0]) / 2.0;  /* CPP:776 */
    const double half_h = (boundary_pos[1] - boundary_pos[3]) / 2.0;  /* CPP:777 */
    for (int i = 0; i < n_a; ++i)
        for (int c = 0; c < 2; ++c) {
            double sum = 0.0;
            for (int k = 0; k < n_a; ++k) {
                double e;
                if (k < i) e = sf_entry(p, d_b2b_edge, is_collide_b2b, d_b2b_center, n_a, i, k, c,
                                        k_ball, is_periodic, half_w, half_h);
                else if (k > i) e = -sf_entry(p, d_b2b_edge, is_collide_b2b, d_b2b_center, n_a, k, i,
                                              c, k_ball, is_periodic, half_w, half_h);
                else e = 0.0;
                sum += e;
            }
            sf_b2b[c * n_a + i] = sum;
        }
}

/* CPP:822-860 */
void oracle_get_dist_b2w(double *p, double *r, double *d_b2w, bool *isCollision, int dim, int n_a,
                         double *boundary_pos)
{
    (void)dim;
    for (int i = 0; i < n_a; ++i) {
        double d[4];
        d[0] = p[i] - r[i] - boundary_pos[0];                         /* CPP:841 */
        d[1] = boundary_pos[1] - (p[n_a + i] + r[i]);                 /* CPP:842 */
        d[2] = boundary_pos[2] - (p[i] + r[i]);                       /* CPP:843 */
        d[3] = p[n_a + i] - r[i] - boundary_pos[3];                   /* CPP:844 */
        for (int w = 0; w < 4; ++w) {
            isCollision[w * n_a + i] = (d[w] < 0);                    /* CPP:849 */
            d_b2w[w * n_a + i] = fabs(d[w]);                          /* CPP:850 */
        }
    }
}

/* CPP:1066-1123 with robotPolicy CPP:1126-1201 */
void oracle_calculateActionPrior(double *p, double *dp, double *a_prior, double *grid,
                                 int *neighbor_index, double d_sen, double r_avoid, double l_cell,
                                 int topo_nei_max, int n_a, int n_g, int dim)
{
    (void)dim;
    double *norm_g = (double *)malloc(sizeof(double) * (size_t)(n_g > 0 ? n_g : 1));
    const double repulsion_strength = 3.0, attraction_strength = 2.0, sync_strength = 2.0; /* CPP:1133-1135 */
    for (int i = 0; i < n_a; ++i) {
        int in_flag, ns;
        int mi = target_grid_state(i, p, n_a, grid, n_g, l_cell, d_sen, norm_g, &in_flag, NULL, &ns);
        double tx, ty;
        if (in_flag) { tx = p[i]; ty = p[n_a + i]; } else { tx = grid[mi]; ty = grid[n_g + mi]; }
        const double px = p[i], py = p[n_a + i], vx = dp[i], vy = dp[n_a + i];
        double f0 = 0.0, f1 = 0.0;
        double d0 = tx - px, d1 = ty - py;                            /* CPP:1147 */
        double dist = sqrt(d0 * d0 + d1 * d1);                        /* CPP:1148 */
        if (dist > 0) { f0 += attraction_strength * d0 / dist; f1 += attraction_strength * d1 / dist; }
        double av0 = 0.0, av1 = 0.0; int cnt = 0;
        for (int k = 0; k < topo_nei_max; ++k) {
            int j = neighbor_index[i * topo_nei_max + k];
            if (j == -1) continue;                                    /* CPP:1219-1227 */
            double n0 = px - p[j], n1 = py - p[n_a + j];              /* CPP:1167 */
            double dn = norm2(n0, n1);                                /* CPP:1168 */
            if (dn > 0 && dn < r_avoid) {                             /* CPP:1171 */
                double u0 = n0 / dn, u1 = n1 / dn;
                double factor = repulsion_strength * (r_avoid / dn - 1.0); /* CPP:1176 */
                f0 += factor * u0; f1 += factor * u1;
            }
            av0 += dp[j]; av1 += dp[n_a + j]; ++cnt;                  /* CPP:1182-1184 */
        }
        if (cnt > 0) {
            av0 /= cnt; av1 /= cnt;                                   /* CPP:1189-1190 */
            f0 += sync_strength * (av0 - vx); f1 += sync_strength * (av1 - vy);
        }
        f0 = fmax(-1.0, fmin(f0, 1.0)); f1 = fmax(-1.0, fmin(f1, 1.0)); /* CPP:17-19,1197-1198 */
        a_prior[i] = f0; a_prior[n_a + i] = f1;
    }
    free(norm_g);
}

/* ------------------------------------------------------------------------- */
/* ENV:437-452 */
void oracle_get_dist_b2b(const double *p, int n_a, double size_a, bool is_periodic, double half_w,
                         double half_h, double *d_center, double *d_edge, bool *is_collide)
{
    for (int i = 0; i < n_a; ++i)
        for (int j = 0; j < n_a; ++j) {
            double rx = p[j] - p[i], ry = p[n_a + j] - p[n_a + i];   /* ENV:438-441 */
            /* ENV:442-443 hands the whole (2n, n) matrix to _make_periodic, whose is_rel
             * branch (ENV:773-776) only touches rows 0 and 1 — agent 0's relative x and y.
             * Rows of agents >= 1 stay unwrapped; the restatement keeps that. */
            if (is_periodic && i == 0) wrap_rel(&rx, &ry, half_w, half_h);
            double dc = sqrt(rx * rx + ry * ry);                      /* ENV:444 */
            double sizes = (i == j) ? 0.0 : (size_a + size_a);        /* ENV:787-793 */
            double de = dc - sizes;                                   /* ENV:445 */
            d_center[i * n_a + j] = dc;
            is_collide[i * n_a + j] = de < 0;                         /* ENV:446 */
            d_edge[i * n_a + j] = fabs(de);                           /* ENV:447 */
        }
}

void oracle_default_cfg(oracle_cfg *c, int n_a, double r_avoid)
{
    c->n_a = n_a; c->topo_nei_max = 6; c->num_obs_grid_max = 80; c->num_occupied_grid_max = 200;
    c->is_boundary = 1; c->is_con_self_state = 1; c->llm_rl = 1;
    c->size_a = 0.035; c->d_sen = 0.4; c->r_avoid = r_avoid; c->Vel_max = 0.8;
    c->k_ball = 30; c->k_wall = 100; c->c_wall = 5; c->dt = 0.1; c->sensitivity = 1; c->mass = 1;
    c->half_w = 2.4; c->half_h = 2.4;
}

int oracle_obs_dim(const oracle_cfg *c)
{   /* ENV:806 */
    return 2 * 2 * (c->topo_nei_max + 1 + (c->is_con_self_state ? 1 : 0)) + 2 * c->num_obs_grid_max;
}

/* ---- binding of the five entry points used by the step driver ---- */
typedef void (*fn_obs)(double *, double *, double *, double *, double *, double *, int *, int *, int *,
                       int *, double, double, double, double, int, int, int, int, int, int, int, bool *);
typedef void (*fn_rew)(double *, double *, double *, double *, double *, double *, double *, int *,
                       int *, int *, int *, double, double, double, int, int, int, int, int, int,
                       bool *, bool *, bool *, double *);
typedef void (*fn_sf)(double *, double *, double *, bool *, double *, double *, int, int, double, bool);
typedef void (*fn_b2w)(double *, double *, double *, bool *, int, int, double *);
typedef void (*fn_prior)(double *, double *, double *, double *, int *, double, double, double, int,
                         int, int, int);
static fn_obs F_obs = oracle_get_observation;
static fn_rew F_rew = oracle_get_reward;
static fn_sf F_sf = oracle_sf_b2b_all;
static fn_b2w F_b2w = oracle_get_dist_b2w;
static fn_prior F_prior = oracle_calculateActionPrior;

int oracle_bind(const char *so_path)
{
    if (!so_path) {
        F_obs = oracle_get_observation; F_rew = oracle_get_reward; F_sf = oracle_sf_b2b_all;
        F_b2w = oracle_get_dist_b2w; F_prior = oracle_calculateActionPrior;
        return 0;
    }
    void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "oracle_bind: %s\n", dlerror()); return 1; }
    fn_obs a = (fn_obs)dlsym(h, "_get_observation");
    fn_rew b = (fn_rew)dlsym(h, "_get_reward");
    fn_sf c = (fn_sf)dlsym(h, "_sf_b2b_all");
    fn_b2w d = (fn_b2w)dlsym(h, "_get_dist_b2w");
    fn_prior e = (fn_prior)dlsym(h, "calculateActionPrior");
    if (!a || !b || !c || !d || !e) return 2;
    F_obs = a; F_rew = b; F_sf = c; F_b2w = d; F_prior = e;
    return 0;
}

/* ENV:486-647 */
void oracle_env_dynamics(const oracle_cfg *c, double *p, double *dp, const float *act,
                         const double *grid, int n_g, double l_cell, const int *neighbor_index,
                         double *a_prior, bool *is_collide_b2b)
{
    const int n = c->n_a;
    const bool periodic = !c->is_boundary;
    double bpos[4] = {-c->half_w, c->half_h, c->half_w, -c->half_h};  /* ENV:188-191 */
    double *d_center = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *d_edge = (double *)malloc(sizeof(double) * (size_t)n * n);
    bool *coll_local = is_collide_b2b ? NULL : (bool *)malloc((size_t)n * n);
    bool *coll = is_collide_b2b ? is_collide_b2b : coll_local;
    double *sf = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *d_b2w = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    bool *coll_w = (bool *)malloc(4 * (size_t)n);
    double *size = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) size[i] = c->size_a;

    oracle_get_dist_b2b(p, n, c->size_a, periodic, c->half_w, c->half_h, d_center, d_edge, coll); /* ENV:486 */
    F_sf(p, sf, d_edge, coll, bpos, d_center, n, 2, c->k_ball, periodic);                         /* ENV:490-499 */
    if (c->is_boundary) F_b2w(p, size, d_b2w, coll_w, 2, n, bpos);                                /* ENV:511 */
    if (c->llm_rl && a_prior)                                                                    /* ENV:600-619 */
        F_prior(p, dp, a_prior, (double *)grid, (int *)neighbor_index, c->d_sen, c->r_avoid, l_cell,
                c->topo_nei_max, n, n_g, 2);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 2; ++k) {
            double u = (double)((float)c->sensitivity * act[k * n + i]);     /* ENV:627,633: 1*float32 */
            double F = u + sf[k * n + i];                                    /* ENV:633-635 */
            if (c->is_boundary) {
                /* ENV:512: [[1,0,-1,0],[0,-1,0,1]] . (collide*d_b2w) * k_wall
                 * ENV:513: [[-1,0,-1,0],[0,-1,0,-1]] . (collide*[dp;dp]) * c_wall */
                double sfw, dfw, v = dp[k * n + i];
                if (k == 0) {
                    sfw = (1.0 * (coll_w[0 * n + i] * d_b2w[0 * n + i]) + 0.0 * (coll_w[1 * n + i] * d_b2w[1 * n + i])
                           + -1.0 * (coll_w[2 * n + i] * d_b2w[2 * n + i]) + 0.0 * (coll_w[3 * n + i] * d_b2w[3 * n + i])) * c->k_wall;
                    dfw = (-1.0 * (coll_w[0 * n + i] * v) + 0.0 * (coll_w[1 * n + i] * dp[n + i])
                           + -1.0 * (coll_w[2 * n + i] * v) + 0.0 * (coll_w[3 * n + i] * dp[n + i])) * c->c_wall;
                } else {
                    sfw = (0.0 * (coll_w[0 * n + i] * d_b2w[0 * n + i]) + -1.0 * (coll_w[1 * n + i] * d_b2w[1 * n + i])
                           + 0.0 * (coll_w[2 * n + i] * d_b2w[2 * n + i]) + 1.0 * (coll_w[3 * n + i] * d_b2w[3 * n + i])) * c->k_wall;
                    dfw = (0.0 * (coll_w[0 * n + i] * dp[i]) + -1.0 * (coll_w[1 * n + i] * v)
                           + 0.0 * (coll_w[2 * n + i] * dp[i]) + -1.0 * (coll_w[3 * n + i] * v)) * c->c_wall;
                }
                F = F + sfw + dfw;
            }
            double ddp = F / c->mass;                                        /* ENV:638 */
            double v2 = dp[k * n + i] + ddp * c->dt;                         /* ENV:641 */
            if (v2 < -c->Vel_max) v2 = -c->Vel_max;                          /* ENV:642 */
            if (v2 > c->Vel_max) v2 = c->Vel_max;
            dp[k * n + i] = v2;
            p[k * n + i] += v2 * c->dt;                                      /* ENV:645 */
        }
        if (periodic) {                                                      /* ENV:646-647, 778-781 */
            if (p[i] < bpos[0]) p[i] += 2 * c->half_w;
            if (p[i] > bpos[2]) p[i] -= 2 * c->half_w;
            if (p[n + i] < bpos[3]) p[n + i] += 2 * c->half_h;
            if (p[n + i] > bpos[1]) p[n + i] -= 2 * c->half_h;
        }
    }
    free(d_center); free(d_edge); free(coll_local); free(sf); free(d_b2w); free(coll_w); free(size);
}

/* ENV:653-654 (220-250, 346-375) */
void oracle_env_observe(const oracle_cfg *c, const double *p, const double *dp, const double *grid,
                        int n_g, double l_cell, int *neighbor_index, double *obs, double *reward,
                        int *in_flags, int *sensed_index, int *occupied_index)
{
    const int n = c->n_a;
    double bpos[4] = {-c->half_w, c->half_h, c->half_w, -c->half_h};
    double *heading = (double *)calloc(2 * (size_t)n, sizeof(double));       /* ENV:214 */
    bool cond_obs[4] = {!c->is_boundary, true, c->is_con_self_state != 0, false}; /* ENV:227 */
    const int od = oracle_obs_dim(c);
    F_obs((double *)p, (double *)dp, heading, obs, bpos, (double *)grid, neighbor_index, in_flags,
          sensed_index, occupied_index, c->d_sen, c->r_avoid, l_cell, c->Vel_max, c->topo_nei_max,
          c->num_obs_grid_max, c->num_occupied_grid_max, n, n_g, od, 2, cond_obs);
    if (reward) {
        bool cond_rew[5] = {!c->is_boundary, true, true, true, true};         /* ENV:350, 21-23 */
        double coeff[1] = {0.05};                                             /* ENV:349 */
        double *act = (double *)calloc(2 * (size_t)n, sizeof(double));
        bool *cb = (bool *)calloc((size_t)n * n, 1), *cw = (bool *)calloc(4 * (size_t)n, 1);
        F_rew((double *)p, (double *)dp, heading, act, reward, bpos, (double *)grid, neighbor_index,
              in_flags, sensed_index, occupied_index, c->d_sen, c->r_avoid, l_cell, c->topo_nei_max,
              c->num_obs_grid_max, c->num_occupied_grid_max, n, n_g, 2, cond_rew, cb, cw, coeff);
        free(act); free(cb); free(cw);
    }
    free(heading);
}

void oracle_env_step(const oracle_cfg *c, double *p, double *dp, const float *act,
                     const double *grid, int n_g, double l_cell, int *neighbor_index, double *obs,
                     double *reward, double *a_prior, int *in_flags, int *sensed_index,
                     int *occupied_index, bool *is_collide_b2b)
{
    if (a_prior) memset(a_prior, 0, sizeof(double) * 2 * (size_t)c->n_a);
    oracle_env_dynamics(c, p, dp, act, grid, n_g, l_cell, neighbor_index, a_prior, is_collide_b2b);
    oracle_env_observe(c, p, dp, grid, n_g, l_cell, neighbor_index, obs, reward, in_flags,
                       sensed_index, occupied_index);
}

/* ------------------------------------------------------------------------- */
/* ENV:525-596, the 'rule' expert: entering + exploration + interaction velocities of
 * every agent on the current state, clipped to [-1, 1].  Sums run left to right
 * (numpy sums pairwise: compare with a tolerance). */
void oracle_rule_actions(const oracle_cfg *c, const double *p, const double *dp, const double *grid,
                         int n_g, double l_cell, double *act)
{
    const int n = c->n_a, cap = c->num_obs_grid_max;
    const double k_1 = 1, k_2 = 15, k_3 = 17;                               /* ENV:527 */
    double *norm_g = (double *)malloc(sizeof(double) * (size_t)(n_g > 0 ? n_g : 1));
    int *sensed = (int *)malloc(sizeof(int) * (size_t)(n_g > 0 ? n_g : 1));
    double *norm_a = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double px = p[i], py = p[n + i];
        int in_flag, ns = 0;
        const int mi = target_grid_state(i, p, n, grid, n_g, l_cell, c->d_sen, norm_g, &in_flag, sensed, &ns);  /* ENV:833-849 */
        double v0 = 0, v1 = 0;
        if (!in_flag) {                                                     /* ENV:531-536 */
            const double tx = grid[mi] - px, ty = grid[n_g + mi] - py;
            const double tn = norm2(tx, ty) + 1e-8;
            v0 = k_1 * (tx / tn) + (0.0 - dp[i]); v1 = k_1 * (ty / tn) + (0.0 - dp[n + i]);
        }
        for (int j = 0; j < n; ++j) norm_a[j] = norm2(p[j] - px, p[n + j] - py);
        if (ns > 0 && in_flag) {                                            /* ENV:542-553 */
            for (int j = 0; j < n; ++j) {
                if (!(norm_a[j] < c->d_sen + c->r_avoid / 2)) continue;
                int m = 0;
                for (int k = 0; k < ns; ++k) {
                    const int g = sensed[k];
                    if (norm2(grid[g] - p[j], grid[n_g + g] - p[n + j]) > c->r_avoid / 2) sensed[m++] = g;
                }
                ns = m;
            }
        }
        const int n_out = ns > cap ? cap : ns;                              /* ENV:556-567 */
        const double step = ns > cap ? (double)(ns - 1) / (cap - 1) : 1.0;
        double e0 = 0, e1 = 0;
        if (n_out > 0) {                                                    /* ENV:569-579 */
            double num0 = 0, num1 = 0, den = 0;
            for (int k = 0; k < n_out; ++k) {
                const int g = sensed[ns > cap ? (int)nearbyint(k * step) : k];          /* np.round: half to even */
                const double gx = grid[g] - px, gy = grid[n_g + g] - py;
                const double psi = rho_cos_dec(norm2(gx, gy), 0, c->d_sen);
                num0 += psi * gx; num1 += psi * gy; den += psi;
            }
            if (den == 0) den = 1e-8;
            e0 = k_2 * num0 / den; e1 = k_2 * num1 / den;
        }
        int near = 0;                                                       /* ENV:582-593 */
        for (int j = 0; j < n; ++j) near += j != i && norm_a[j] < c->d_sen;
        double w0 = 0, w1 = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i || !(norm_a[j] < c->d_sen)) continue;
            const double rx = p[j] - px, ry = p[n + j] - py;
            if (norm_a[j] < c->r_avoid) {
                const double f = -k_3 * (c->r_avoid / norm_a[j] - 1);
                w0 += f * rx; w1 += f * ry;
            }
            w0 += 5 * (dp[j] - dp[i]) / near; w1 += 5 * (dp[n + j] - dp[n + i]) / near;
        }
        const double a0 = v0 + e0 + w0, a1 = v1 + e1 + w1;                  /* ENV:595-596 */
        act[i] = a0 < -1 ? -1 : a0 > 1 ? 1 : a0; act[n + i] = a1 < -1 ? -1 : a1 > 1 ? 1 : a1;
    }
    free(norm_g); free(sensed); free(norm_a);
}

/* ------------------------------------------------------------------------- */
/* WRAP:48-72 */
double oracle_coverage_rate(const double *p, int n_a, const double *grid, int n_g, double r_avoid)
{
    int occ = 0;
    for (int g = 0; g < n_g; ++g) {
        int any = 0;
        for (int j = 0; j < n_a && !any; ++j)
            any = norm2(p[j] - grid[g], p[n_a + j] - grid[n_g + g]) < r_avoid / 2;
        occ += any;
    }
    return (double)occ / n_g;
}

/* numpy var (population variance): mean of squared deviations from the mean,
 * both means by numpy's pairwise summation, which for n < 8 ... is plain; the
 * oracle uses plain left-to-right sums, so compare with a 1e-12 tolerance. */
static double pop_var(const double *x, int n)
{
    double m = 0.0; for (int i = 0; i < n; ++i) m += x[i]; m /= n;
    double v = 0.0; for (int i = 0; i < n; ++i) v += (x[i] - m) * (x[i] - m);
    return v / n;
}

/* WRAP:74-101 */
double oracle_distribution_uniformity(const double *p, int n_a)
{
    double *md = (double *)malloc(sizeof(double) * (size_t)n_a);
    double lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < n_a; ++i) {
        double m = INFINITY;
        for (int j = 0; j < n_a; ++j) {
            double d = norm2(p[j] - p[i], p[n_a + j] - p[n_a + i]);
            if (d != 0 && d < m) m = d;                                       /* WRAP:93-94 */
        }
        md[i] = m; if (m < lo) lo = m; if (m > hi) hi = m;
    }
    double u = pop_var(md, n_a);
    free(md);
    return (u - lo) / (hi - lo);                                              /* WRAP:99 */
}

/* WRAP:103-129 */
double oracle_voronoi_uniformity(const double *p, int n_a, const double *grid, int n_g)
{
    double *cnt = (double *)calloc((size_t)n_a, sizeof(double));
    for (int g = 0; g < n_g; ++g) {
        int mi = 0; double md = INFINITY;
        for (int j = 0; j < n_a; ++j) {
            double d = norm2(p[j] - grid[g], p[n_a + j] - grid[n_g + g]);
            if (d < md) { md = d; mi = j; }                                   /* argmin: first minimum */
        }
        cnt[mi] += 1;
    }
    double lo = INFINITY, hi = -INFINITY;
    for (int j = 0; j < n_a; ++j) { if (cnt[j] < lo) lo = cnt[j]; if (cnt[j] > hi) hi = cnt[j]; }
    double u = pop_var(cnt, n_a);
    free(cnt);
    return (u - lo) / (hi - lo);                                              /* WRAP:127 */
}

/* WRAP:131-145 */
int oracle_count_collisions(const bool *is_collide_b2b, int n_a)
{
    int c = 0;
    for (int i = 0; i < n_a; ++i) for (int j = i + 1; j < n_a; ++j) c += is_collide_b2b[i * n_a + j];
    return c;
}

/* WRAP:147-165 */
double oracle_avg_distance_to_target(const double *p, int n_a, const double *grid, int n_g)
{
    double tot = 0.0;
    for (int i = 0; i < n_a; ++i) {
        double m = INFINITY;
        for (int g = 0; g < n_g; ++g) {
            double d = norm2(grid[g] - p[i], grid[n_g + g] - p[n_a + i]);
            if (d < m) m = d;
        }
        tot += m;
    }
    return tot / n_a;
}

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11). */
void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* draw number d of (seed, env, episode): word d%4 of Philox(ctr = {d/4, episode, 0, 0},
 * key = {env id low 32 bits, seed low 32 bits ^ env id high 32 bits}) */
static uint32_t draw_u32(uint64_t seed, uint64_t env, uint32_t episode, uint32_t d)
{
    uint32_t ctr[4] = {d >> 2, episode, 0u, 0u};
    uint32_t key[2] = {(uint32_t)env, (uint32_t)seed ^ (uint32_t)(env >> 32)};
    uint32_t out[4];
    oracle_philox4x32(ctr, key, out);
    return out[d & 3u];
}
static double draw_uniform(uint64_t seed, uint64_t env, uint32_t ep, uint32_t d, double lo, double hi)
{   /* lo + (hi-lo)*u, u = x * 2^-32 in [0,1) */
    double u = (double)draw_u32(seed, env, ep, d) * (1.0 / 4294967296.0);
    return lo + (hi - lo) * u;
}

/* ENV:151-218 with the draw order of the reference reset: shape, angle, 2 discarded, offset x/y, coin,
 * 2*n_a position draws, 2 centre draws, 2*n_a velocity draws. */
void oracle_batch_reset(uint64_t seed, uint32_t episode, int64_t env0, int E, int n_a, int n_shapes,
                        const double *shape_grids, const int *shape_off, const int *shape_ng,
                        const double *shape_lcell, int n_g_max, double half_w, double half_h,
                        int *shape_id, double *p, double *dp, double *grid, int *n_g, double *l_cell)
{
    for (int e = 0; e < E; ++e) {
        const uint64_t env = (uint64_t)(env0 + e);
        uint32_t d = 0;
        int s = (int)(((uint64_t)draw_u32(seed, env, episode, d++) * (uint64_t)n_shapes) >> 32); /* ENV:155 */
        shape_id[e] = s; n_g[e] = shape_ng[s]; l_cell[e] = shape_lcell[s];
        double ang = M_PI * draw_uniform(seed, env, episode, d++, -1.0, 1.0);  /* ENV:170 */
        d += 2;                                                                /* ENV:177 discarded */
        double ox = draw_uniform(seed, env, episode, d++, -half_w + 1, half_w - 1); /* ENV:179 */
        double oy = draw_uniform(seed, env, episode, d++, -half_h + 1, half_h - 1); /* ENV:180 */
        double ca = cos(ang), sa = sin(ang);
        const double *g = shape_grids + 2 * (size_t)shape_off[s];
        double *ge = grid + (size_t)e * 2 * n_g_max;
        for (int k = 0; k < n_g_max; ++k) { ge[k] = 0.0; ge[n_g_max + k] = 0.0; }
        for (int k = 0; k < shape_ng[s]; ++k) {                                /* ENV:172-173,182 */
            double gx = g[2 * k], gy = g[2 * k + 1];
            ge[k] = (ca * gx + sa * gy) + ox;
            ge[n_g_max + k] = (-sa * gx + ca * gy) + oy;
        }
        double coin = draw_uniform(seed, env, episode, d++, -1.0, 1.0);        /* ENV:197 */
        double *pe = p + (size_t)e * 2 * n_a, *ve = dp + (size_t)e * 2 * n_a;
        const uint32_t dpos = d, dctr = d + 2 * (uint32_t)n_a, dvel = dctr + 2;
        if (coin > 0) {                                                        /* ENV:198-200 */
            for (int i = 0; i < n_a; ++i) {
                pe[i] = draw_uniform(seed, env, episode, dpos + (uint32_t)i, -half_w, half_w);
                pe[n_a + i] = draw_uniform(seed, env, episode, dpos + (uint32_t)(n_a + i), -half_h, half_h);
            }
        } else {                                                               /* ENV:202-203 */
            double cx = draw_uniform(seed, env, episode, dctr, -half_w + 1, half_w - 1);
            double cy = draw_uniform(seed, env, episode, dctr + 1, -half_h + 1, half_h - 1);
            for (int i = 0; i < n_a; ++i) {
                pe[i] = draw_uniform(seed, env, episode, dpos + (uint32_t)i, -1.0, 1.0) + cx;
                pe[n_a + i] = draw_uniform(seed, env, episode, dpos + (uint32_t)(n_a + i), -1.0, 1.0) + cy;
            }
        }
        for (int i = 0; i < 2 * n_a; ++i)                                      /* ENV:210 */
            ve[i] = draw_uniform(seed, env, episode, dvel + (uint32_t)i, -0.5, 0.5);
    }
}

/* ------------------------------------------------------------------------- */
double oracle_run_envs(const oracle_cfg *c, int E, int steps, int threads, double *p, double *dp,
                       const float *act, const double *grid, int n_g_max, const int *n_g,
                       const double *l_cell)
{
    const int n = c->n_a, od = oracle_obs_dim(c);
    double total = 0.0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel reduction(+ : total)
    {
        double *obs = (double *)malloc(sizeof(double) * (size_t)od * n);
        double *rew = (double *)malloc(sizeof(double) * (size_t)n);
        double *prior = (double *)malloc(sizeof(double) * 2 * (size_t)n);
        int *nbr = (int *)malloc(sizeof(int) * (size_t)n * c->topo_nei_max);
        int *inf = (int *)malloc(sizeof(int) * (size_t)n);
        int *sen = (int *)malloc(sizeof(int) * (size_t)n * c->num_obs_grid_max);
        int *occ = (int *)malloc(sizeof(int) * (size_t)n * c->num_occupied_grid_max);
        double *gc = (double *)malloc(sizeof(double) * 2 * (size_t)n_g_max);
#pragma omp for schedule(static)
        for (int e = 0; e < E; ++e) {
            double *pe = p + (size_t)e * 2 * n, *ve = dp + (size_t)e * 2 * n;
            const int ng = n_g[e];
            /* compact (2,n_g) grid as the reference holds it */
            for (int k = 0; k < ng; ++k) {
                gc[k] = grid[(size_t)e * 2 * n_g_max + k];
                gc[ng + k] = grid[(size_t)e * 2 * n_g_max + n_g_max + k];
            }
            oracle_env_observe(c, pe, ve, gc, ng, l_cell[e], nbr, obs, NULL, inf, sen, occ); /* reset's obs */
            for (int t = 0; t < steps; ++t) {
                const float *a = act + ((size_t)t * E + e) * 2 * n;
                oracle_env_step(c, pe, ve, a, gc, ng, l_cell[e], nbr, obs, rew, prior, inf, sen, occ, NULL);
                for (int i = 0; i < n; ++i) total += rew[i];
            }
        }
        free(obs); free(rew); free(prior); free(nbr); free(inf); free(sen); free(occ); free(gc);
    }
    return total;
}
