// Double-precision kernels behind the reference-named entry points that are not the observation
// (_get_reward, _sf_b2b_all, _get_dist_b2w, calculateActionPrior).
// Part of the single translation unit asm_kernels.hip (included there, inside namespace asmenv, after the
// diagnostic macros); not a stand-alone header.

// ---- the three small reference-compatible kernels (double, one environment) ----

// CPP:359-631 on caller-supplied index arrays
__global__ void k_reward_compat(DevParams P, const double *p, const double *grid, int n_g, const int *nbr,
                                const int *inflags, const int *sensed, bool c_periodic, bool c_interaction,
                                bool c_exploration, double *reward)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_a = P.n_a;
    if (i >= n_a) return;
    const double px = p[i], py = p[n_a + i];
    bool collision = false, uniform = false;
    if (c_interaction) {                                                         // CPP:464-496
        for (int q = 0; q < P.topo; ++q) {
            int j = nbr[i * P.topo + q];
            if (j == -1) continue;
            double rx = p[j] - px, ry = p[n_a + j] - py;
            if (c_periodic) wrap_rel(rx, ry, P.half_w, P.half_h);
            if (rx * rx + ry * ry < P.T_ravoid) { collision = true; break; }     // CPP:487
        }
    }
    double r = 0.0;
    if (c_exploration) {                                                         // CPP:499-564
        if (inflags[i] == 1) {
            double num0 = 0, num1 = 0, den = 0; bool any = false;
            for (int k = 0; k < P.n_obs; ++k) {
                int c = sensed[i * P.n_obs + k];
                if (c == -1) continue;
                any = true;
                double gx = grid[c] - px, gy = grid[n_g + c] - py;
                double z = sqrt(gx * gx + gy * gy), psi;
                if (z < 0.0 * P.d_sen) psi = 1.0;
                else if (z < P.d_sen) psi = (1.0 / 2.0) * (1.0 + cos(ASM_PI * (z / P.d_sen - 0.0) / (1.0 - 0.0)));
                else psi = 0.0;
                num0 += psi * gx; num1 += psi * gy; den += psi;
            }
            if (any) {
                if (den == 0) den = 1E-8;
                double v0 = 1.0 * num0 / den, v1 = 1.0 * num1 / den;
                uniform = v0 * v0 + v1 * v1 < P.T_v005;
            }
        }
        if (inflags[i] == 1 && !collision && uniform) r += 1.0;
    }
    reward[i] = r;
}

// CPP:740-820 on caller-supplied distance / contact matrices
__global__ void k_sf_b2b_compat(const double *p, const double *d_edge, const bool *collide, const double *d_center,
                                int n_a, double k_ball, bool periodic, double hw, double hh, double *sf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    double sx = 0.0, sy = 0.0;
    for (int k = 0; k < n_a; ++k) {
        if (k == i) { sx += 0.0; sy += 0.0; continue; }
        const int a = k < i ? i : k, b = k < i ? k : i;                          // entry (a, b), b < a
        double dx = p[b] - p[a], dy = p[n_a + b] - p[n_a + a];                   // CPP:782-785
        if (periodic) wrap_rel(dx, dy, hw, hh);
        const double dc = d_center[(size_t)a * n_a + b];
        const double m = (double)collide[(size_t)a * n_a + b] * d_edge[(size_t)a * n_a + b] * k_ball;
        double ex = m * (-(dx / dc)), ey = m * (-(dy / dc));                     // CPP:790-793
        if (k > i) { ex = -ex; ey = -ey; }                                       // CPP:795-796
        sx += ex; sy += ey;                                                      // CPP:804-812
    }
    sf[i] = sx; sf[n_a + i] = sy;
}

// CPP:822-860
__global__ void k_dist_b2w_compat(const double *p, const double *r, int n_a, double b0, double b1, double b2,
                                  double b3, double *d_b2w, bool *coll)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    double d[4];
    d[0] = p[i] - r[i] - b0; d[1] = b1 - (p[n_a + i] + r[i]);
    d[2] = b2 - (p[i] + r[i]); d[3] = p[n_a + i] - r[i] - b3;
#pragma unroll
    for (int w = 0; w < 4; ++w) { coll[w * n_a + i] = d[w] < 0; d_b2w[w * n_a + i] = fabs(d[w]); }
}

// CPP:1066-1123: own nearest-cell search, then robotPolicy
__global__ void k_prior_compat(DevParams P, const double *p, const double *dp, const double *grid, int n_g,
                               const int *nbr, double l_cell, double *a_prior)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_a = P.n_a;
    if (i >= n_a) return;
    const double px = p[i], py = p[n_a + i];
    double smin = __builtin_inf(); int imin = 0;
    for (int c = 0; c < n_g; ++c) {
        double rx = grid[c] - px, ry = grid[n_g + c] - py;
        double s = rx * rx + ry * ry;
        if (s < smin) { smin = s; imin = c; }
    }
    const bool inf = sqrt(smin) < sqrt(2.0) * l_cell / 2;
    double tx = inf ? px : grid[imin], ty = inf ? py : grid[n_g + imin];
    int nb[6];
    for (int q = 0; q < 6; ++q) nb[q] = q < P.topo ? nbr[i * P.topo + q] : -1;
    double f0, f1;
    prior_policy<double>(P, px, py, dp[i], dp[n_a + i], tx, ty, nb, p, p + n_a, dp, dp + n_a, f0, f1);
    a_prior[i] = f0; a_prior[n_a + i] = f1;
}

