// The generic kernels: observation + reward of one environment per (part of a) workgroup for any n_a <= 1024 and
// both state types (observe<S>), k_observe, k_step, k_reset.
// Part of the single translation unit asm_kernels.hip (included there, inside namespace asmenv, after the
// diagnostic macros); not a stand-alone header.

// Per-environment step statistics of a large swarm (TRAIN:128-135; WRAP:48-165), one environment per workgroup of SL
// threads: wave shuffles, then the wave partials through `part` ([SL / 64][8] doubles of LDS); two barriers.  The
// variance of the nearest-neighbour distances (WRAP:96) is taken as E[m^2] - mean^2 from one pass (relative error
// ~1e-13 at these magnitudes).  cov (nullable): the environment's coverage bitmap, counted into st[2]; contacts < 0:
// st[4] is left alone.
__device__ __forceinline__ void swarm_stats(const DevParams &P, double *part, int i, bool active, int e, double reward,
                                            double dmin, bool in_flag, double nn_min, const unsigned *cov, double contacts,
                                            double *stats_g)
{
    const int n_a = P.n_a;
    const int wave = i >> 6, nw = P.SL >> 6;
    const double mloc = active ? sqrt(nn_min) : 0.0;
    double v[7] = {active ? reward : 0.0, active ? dmin : 0.0, active && in_flag ? 1.0 : 0.0, mloc,
                   active ? mloc : __builtin_inf(), active ? mloc : -__builtin_inf(), mloc * mloc};
    if (cov && wave == 0 && (i & 63) < P.W) v[2] += 65536.0 * __popc(cov[i & 63]);   // covered cells ride along (W <= 32 words)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        v[0] += __shfl_xor(v[0], m, 64); v[1] += __shfl_xor(v[1], m, 64); v[2] += __shfl_xor(v[2], m, 64);
        v[3] += __shfl_xor(v[3], m, 64); v[6] += __shfl_xor(v[6], m, 64);
        v[4] = fmin(v[4], __shfl_xor(v[4], m, 64)); v[5] = fmax(v[5], __shfl_xor(v[5], m, 64));
    }
    if ((i & 63) == 0) for (int q = 0; q < 7; ++q) part[wave * 8 + q] = v[q];
    __syncthreads();
    if (i == 0 && e < P.E) {
        double rs = 0, ds = 0, ic = 0, ms = 0, sq = 0, lo = __builtin_inf(), hi = -__builtin_inf();
        for (int w = 0; w < nw; ++w) {
            rs += part[w * 8]; ds += part[w * 8 + 1]; ic += part[w * 8 + 2]; ms += part[w * 8 + 3]; sq += part[w * 8 + 6];
            lo = fmin(lo, part[w * 8 + 4]); hi = fmax(hi, part[w * 8 + 5]);
        }
        const double covered = floor(ic / 65536.0), inn = ic - 65536.0 * covered;    // both are small integers
        const double mean = ms / n_a, var = sq / n_a - mean * mean;
        double *st = stats_g + (size_t)e * 8;
        stat_add(&st[0], rs); stat_add(&st[1], rs);                      // rewards are 0 or 1
        if (cov) stat_add(&st[2], P.n_g_e > 0 ? covered / P.n_g_e : 0.0);
        stat_add(&st[3], (var - lo) / (hi - lo));                        // WRAP:98-99
        if (contacts >= 0) stat_add(&st[4], contacts);
        stat_add(&st[5], ds / n_a); stat_add(&st[6], 1.0); stat_add(&st[7], inn);
    }
}

// ---------------------------------------------------------------------------------
// Observation + reward of environment e for agent i on the state held in LDS.
// CPP:23-356 (observation), CPP:359-631 (reward).  Must be reached by every thread.
// The environment's nthr threads are numbered row = 0 .. nthr - 1; row is also the thread's per-agent LDS row (sensed
// mask, word keys).  One workgroup per environment: row = i, nthr = SL.  A part workgroup (P.parts > 1) observes the
// agents i = part * nthr + row on the state of ALL agents: it lists every agent in its bins, and marks the
// occupancy / coverage bits of the other parts' agents too (lattice path only), because the occupied-cell filter
// (CPP:155-212) needs the whole environment's bitmap.
//
// S = float takes single-precision screening paths wherever a decision is certain in
// fp32 and falls back to the exact double evaluation otherwise (cell_pass_fast above;
// the neighbour and reward screens below follow the same rule), so its decisions are
// those of the all-double code; S = double is the all-double code.
// SW: compiled for the large swarm alone -- fp32 state, walls, Cartesian observation, the shapes' lattice description at hand,
// cell lists for the pair passes, the whole environment in this workgroup -- so that the other schedules' code and registers
// are not carried along (the launcher picks it exactly when all of that holds).
template <typename S, bool SW = false>
__device__ void observe(const DevParams &P, const EnvLds<S> &L, bool active, int e, int i, int row, int nthr, double l_cell,
                        const S *heading_g, int *nbr_g, int *inflag_g, int *nearest_g, S *obs_g, S *rew_g,
                        int *sensed_g, int *occup_g, double *stats_g, int contact_pairs, const LatView *LV = nullptr)
{
    constexpr bool F32 = sizeof(S) == 4;
    const int n_a = P.n_a;
#ifdef ASM_EXITS
    const int xat = g_exit_at;
#endif
    // large swarms on fp32 state: pair loops screen 32 agents at a time, the cell pass settles undecided words only
    const bool swarm = SW || (F32 && !P.periodic && P.SL >= 64);
    const bool periodic = !SW && P.periodic, cartesian = SW || P.cartesian, split = !SW && P.parts > 1;
    const double px = active ? (double)L.px[i] : 0.0, py = active ? (double)L.py[i] : 0.0;
    const double vx = active ? (double)L.vx[i] : 0.0, vy = active ? (double)L.vy[i] : 0.0;
    S *ocol = obs_g + (size_t)e * P.obs_dim * n_a + i;                   // obs[row*n_a + agent]
    // (the large swarm stores every observation row non-temporal: written once, read by a later kernel, and 786 KB per
    // environment that would otherwise pass through the cache the grids and the state live in)
    auto oput = [&](size_t r, S v) { if constexpr (SW) __builtin_nontemporal_store(v, ocol + r); else ocol[r] = v; };
    // cell list of the agents for the neighbour search (in the mask area, which the cell pass takes over afterwards)
    [[maybe_unused]] BinView BVs;
    [[maybe_unused]] const BinView *BV = nullptr;
    if constexpr (F32) {
        if (SW || (swarm && P.bins)) {
            BVs = bins_carve(L.mask, P.bin_s, P.SL);
            bins_build(P, P.bin_s, BVs, row, nthr, P.n_a, reinterpret_cast<const float *>(L.px), reinterpret_cast<const float *>(L.py));
            BV = &BVs;
        }
    }

    // ---- k nearest neighbours inside d_sen: _get_focused, CPP:633-703 ----
    double s6[6]; int j6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { s6[q] = __builtin_inf(); j6[q] = -1; }
    double nn_min = __builtin_inf();
    [[maybe_unused]] int perm_a = 0;                                     // large swarm: the agent at this lane's place in the cell list
    if (active) {
        // sorted insert; j arrives in ascending order, so an equal distance already listed stays in front (ties keep
        // the lower index), and once the new entry is placed everything behind it moves down one place
        auto consider = [&](int j, double s) {
            double ks = s; int kj = j; bool placed = false;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (placed || ks < s6[q]) { double ts = s6[q]; int tj = j6[q]; s6[q] = ks; j6[q] = kj; ks = ts; kj = tj; placed = true; }
        };
        if (swarm) {
            if constexpr (F32) {
                // (a function of the agent: the large swarm's kernel runs it for the agent at the lane's place in the cell
                // list, not for the lane's own agent -- see below)
                auto knn_swarm = [&](const int i, const double px, const double py, int (&j6)[6], double &nn_min) {
                const float pxf = L.px[i], pyf = L.py[i];
                // Candidates (fp32 screen) go through a min/max insertion chain on single-precision keys (s32 with
                // the low 10 bits replaced by the index) that keeps the 8 smallest.  The list is read off the keys
                // when every neighbouring pair of the first seven differs by at least two units of 1024 ulps (the
                // double-precision order is then the keys' order) and every listed key lies two units below T_sen.
                // Otherwise the first seven are ordered and filtered exactly, in double: that settles the list as
                // long as the eighth key lies two units above the sixth (nobody beyond the seven can be among the
                // six nearest); if not even that holds, the agent redoes its list with exact insertion of every
                // candidate.  The candidates come from the cell list when there is one (any order: the keys are
                // distinct), else from screens of 32 agents at a time.
                unsigned K[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) K[q] = 0xFFFFFFFFu;
                if (BV) {
                    // Rows of bins nearest first; once eight candidates are held, only keys below the eighth can still
                    // enter the list, so the screen's threshold follows it down and the rows (and the bins of a row)
                    // that lie beyond it are skipped: a row m >= 2 bin rows away holds nobody closer than (m - 1) bin
                    // sides, a bin k >= 1 columns away nobody closer than k - 1 (0.1 % slack for the rounding of the
                    // bin coordinates).
                    const BinGeom &G = P.bin_s;
                    const int bx = bin_coord(pxf, P.bin_x0, G.inv_h, G.nbx), by = bin_coord(pyf, P.bin_y0, G.inv_h, G.nby);
                    const float hs = 0.999f / G.inv_h;
                    const unsigned tsen_bits = __float_as_uint(P.Tsen_hi_f);
                    float thr = P.Tsen_hi_f;
                    for (int t = 0; t <= 2 * G.reach; ++t) {
                        const int m = (t + 1) >> 1, r = by + ((t & 1) ? -m : m);
                        const float dr = m > 1 ? (float)(m - 1) * hs : 0.f, rem = thr - dr * dr;
                        if (r < 0 || r >= G.nby || !(rem > 0.f)) continue;
                        int kx = (int)(__builtin_amdgcn_sqrtf(rem) * G.inv_h * 1.001f) + 1;
                        kx = kx < G.reach ? kx : G.reach;
                        const int xa = bx - kx > 0 ? bx - kx : 0, xb = bx + kx < G.nbx - 1 ? bx + kx : G.nbx - 1;
                        unsigned u = BV->start[r * G.nbx + xa];
                        const unsigned hi = BV->start[r * G.nbx + xb + 1];
                        for (; u < hi; ++u) {
                            const f32x2 c = BV->pos[u];
                            const int j = BV->idx[u];
                            const float rx = c.x - pxf, ry = c.y - pyf, s32 = __builtin_fmaf(ry, ry, rx * rx);
                            if (s32 < thr && j != i) {                                  // self removed CPP:677-681
                                unsigned x = (__float_as_uint(s32) & ~1023u) | (unsigned)j;
#pragma unroll
                                for (int q = 0; q < 8; ++q) { const unsigned lo = K[q] < x ? K[q] : x; x = K[q] < x ? x : K[q]; K[q] = lo; }
                                const unsigned tb = K[7] | 1023u;                       // every key of the eighth's 1024-ulp bucket still passes
                                thr = __uint_as_float(tb < tsen_bits ? tb + 1u : tsen_bits);
                            }
                        }
                    }
                }
                else
                for (int j0 = 0; j0 < P.SL; j0 += 32) {
                    unsigned m = pair_screen32(L.px + j0, L.py + j0, pxf, pyf, P.Tsen_hi_f);
                    if ((unsigned)(i - j0) < 32u) m &= ~(1u << (i - j0));        // self removed CPP:677-681
                    while (m) {
                        const int j = j0 + __builtin_ctz(m); m &= m - 1;
                        const float rx = L.px[j] - pxf, ry = L.py[j] - pyf;
                        unsigned x = (__float_as_uint(__builtin_fmaf(ry, ry, rx * rx)) & ~1023u) | (unsigned)j;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const unsigned lo = K[q] < x ? K[q] : x; x = K[q] < x ? x : K[q]; K[q] = lo; }
                    }
                }
                const unsigned tlo = (__float_as_uint(P.Tsen_f) >> 10) - 2u;
                bool amb = false;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const bool there = K[q] != 0xFFFFFFFFu;
                    const unsigned u = K[q] >> 10;
                    amb |= there && (u >= tlo || (K[q + 1] >> 10) - u < 2u);
                    j6[q] = there ? (int)(K[q] & 1023u) : -1;
                }
                const bool redo = amb && K[5] != 0xFFFFFFFFu && (K[7] >> 10) - (K[5] >> 10) < 2u;
                if (amb) {                                                      // rare paths: exact distances, (s, j) ascending (ties keep the lower index)
                    double e6[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) { e6[q] = __builtin_inf(); j6[q] = -1; }
                    auto insert_exact = [&](int j, double s) {
                        double ks = s; int kj = j; bool placed = false;
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            if (placed || ks < e6[q] || (ks == e6[q] && kj < j6[q])) {
                                const double ts = e6[q]; const int tj = j6[q]; e6[q] = ks; j6[q] = kj; ks = ts; kj = tj; placed = true;
                            }
                    };
                    if (!redo) {                                                    // the first seven, exactly
#pragma unroll
                        for (int q = 0; q < 7; ++q)
                            if (K[q] != 0xFFFFFFFFu) {
                                const int j = (int)(K[q] & 1023u);
                                const double dx = (double)L.px[j] - px, dy = (double)L.py[j] - py;
                                const double s = dx * dx + dy * dy;                 // CPP:641
                                if (s < P.T_sen) insert_exact(j, s);                // CPP:663
                            }
                    }
                    if (redo && BV) {                                               // very rare
                        bins_for_each(P, P.bin_s, *BV, pxf, pyf, [&](f32x2 c, int j) {
                            const float rx = c.x - pxf, ry = c.y - pyf;
                            if (__builtin_fmaf(ry, ry, rx * rx) < P.Tsen_hi_f && j != i) {
                                const double dx = (double)c.x - px, dy = (double)c.y - py;
                                const double s = dx * dx + dy * dy;                 // CPP:641
                                if (s < P.T_sen) insert_exact(j, s);                // CPP:663
                            }
                        });
                    } else if (redo) {
                        for (int j0 = 0; j0 < P.SL; j0 += 32) {
                            unsigned m = pair_screen32(L.px + j0, L.py + j0, pxf, pyf, P.Tsen_hi_f);
                            if ((unsigned)(i - j0) < 32u) m &= ~(1u << (i - j0));
                            while (m) {
                                const int j = j0 + __ffs(m) - 1; m &= m - 1;
                                const double dx = (double)L.px[j] - px, dy = (double)L.py[j] - py;
                                const double s = dx * dx + dy * dy;                 // CPP:641
                                if (s < P.T_sen) insert_exact(j, s);                // CPP:663
                            }
                        }
                    }
                }
                if (P.want_stats) {                                             // nearest other agent (WRAP:81-99)
                    double s0 = 0.0;
                    if (j6[0] >= 0) { const double dx = (double)L.px[j6[0]] - px, dy = (double)L.py[j6[0]] - py; s0 = dx * dx + dy * dy; }
                    if (s0 > 0.0) nn_min = s0;
                    else if (BV) {
                        // nobody in sensing range (or a coincident agent): rings of bins outwards, until the nearest
                        // found so far is closer than anything a further ring can hold (ring m lies more than m - 1
                        // bin sides away)
                        const BinGeom &G = P.bin_s;
                        const int bx = bin_coord(pxf, P.bin_x0, G.inv_h, G.nbx), by = bin_coord(pyf, P.bin_y0, G.inv_h, G.nby);
                        const double hd = 0.999 / (double)G.inv_h;
                        const int mmax = G.nbx > G.nby ? G.nbx : G.nby;
                        auto scan = [&](int r, int xa, int xb) {
                            for (unsigned u = BV->start[r * G.nbx + xa], hi = BV->start[r * G.nbx + xb + 1]; u < hi; ++u) {
                                const f32x2 c = BV->pos[u];
                                const double rx = (double)c.x - px, ry = (double)c.y - py, su = rx * rx + ry * ry;
                                if (su != 0 && su < nn_min) nn_min = su;
                            }
                        };
                        for (int m = 0; m <= mmax; ++m) {
                            if (m >= 2 && nn_min <= ((m - 1) * hd) * ((m - 1) * hd)) break;
                            const int xa = bx - m > 0 ? bx - m : 0, xb = bx + m < G.nbx - 1 ? bx + m : G.nbx - 1;
                            for (int r = by - m; r <= by + m; ++r) {
                                if (r < 0 || r >= G.nby) continue;
                                if (r == by - m || r == by + m) scan(r, xa, xb);
                                else {
                                    if (bx - m >= 0) scan(r, bx - m, bx - m);
                                    if (bx + m < G.nbx) scan(r, bx + m, bx + m);
                                }
                            }
                        }
                    } else
                        for (int j = 0; j < n_a; ++j) {
                            const double rx = (double)L.px[j] - px, ry = (double)L.py[j] - py, su = rx * rx + ry * ry;
                            if (su != 0 && su < nn_min) nn_min = su;
                        }
                }
                };
                if constexpr (SW) {
                    // Lanes in cell-list order: a wave walks as many candidates as its busiest lane, and agents that sit
                    // in one bin have the same candidates, while 64 agents with consecutive indices are anywhere (dense
                    // cluster next to open space).  Lane `row` therefore searches for the agent at place `row` of the
                    // list and leaves the result -- six indices + 1 as 16-bit halves, the nearest neighbour's squared
                    // distance -- in LDS (the area of the cell pass's word keys, which the lattice path does not use)
                    // for the agent's own lane, which reads it behind the next barrier.
                    const int a = (int)BV->idx[row];
                    perm_a = a;
                    int ja[6]; double nna = __builtin_inf();
                    knn_swarm(a, (double)L.px[a], (double)L.py[a], ja, nna);
                    unsigned *kj = reinterpret_cast<unsigned *>(L.wkey);
                    kj[a * 3 + 0] = (unsigned)(ja[0] + 1) | ((unsigned)(ja[1] + 1) << 16);
                    kj[a * 3 + 1] = (unsigned)(ja[2] + 1) | ((unsigned)(ja[3] + 1) << 16);
                    kj[a * 3 + 2] = (unsigned)(ja[4] + 1) | ((unsigned)(ja[5] + 1) << 16);
                    const unsigned long long nb = (unsigned long long)__double_as_longlong(nna);
                    kj[3 * P.SL + 2 * a] = (unsigned)nb; kj[3 * P.SL + 2 * a + 1] = (unsigned)(nb >> 32);
                } else
                    knn_swarm(i, px, py, j6, nn_min);
            }
        } else if (F32 && !periodic) {
            const float pxf = (float)L.px[i], pyf = (float)L.py[i];
            float nn32 = __builtin_inff(); int nnj = -1;
            for (int j = 0; j < n_a; ++j) {
                const float rx = (float)L.px[j] - pxf, ry = (float)L.py[j] - pyf;
                const float s32 = __builtin_fmaf(ry, ry, rx * rx);
                if (P.want_stats && s32 < nn32 && s32 > 0.0f) { nn32 = s32; nnj = j; }
                if (j != i && s32 < P.Tsen_hi_f) {                       // candidates only: exact evaluation
                    const double dx = (double)L.px[j] - px, dy = (double)L.py[j] - py;
                    const double s = dx * dx + dy * dy;                  // CPP:641
                    if (s < P.T_sen) consider(j, s);                     // CPP:663, self removed CPP:677-681
                }
            }
            if (nnj >= 0) { const double dx = (double)L.px[nnj] - px, dy = (double)L.py[nnj] - py; nn_min = dx * dx + dy * dy; }
        } else {
            for (int j = 0; j < n_a; ++j) {
                double rx = (double)L.px[j] - px, ry = (double)L.py[j] - py; // CPP:84
                if (P.want_stats) { double su = rx * rx + ry * ry; if (su != 0 && su < nn_min) nn_min = su; }
                if (periodic) wrap_rel(rx, ry, P.half_w, P.half_h);        // CPP:93-95
                const double s = rx * rx + ry * ry;                          // CPP:641
                if (j != i && s < P.T_sen) consider(j, s);
            }
        }
    }
    // agent block, column-major flatten of rows [pos_x,pos_y,vel_x,vel_y] (CPP:108-131)
    auto agent_block = [&]() {
        int r0 = 0;
        if (P.con_self) {
            oput(0, (S)px); oput((size_t)n_a, (S)py); oput((size_t)2 * n_a, (S)vx); oput((size_t)3 * n_a, (S)vy);
            r0 = 4;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int j = j6[q];
            double rx = 0, ry = 0, wx = 0, wy = 0;
            if (j >= 0) {
                rx = (double)L.px[j] - px; ry = (double)L.py[j] - py;
                if (periodic) wrap_rel(rx, ry, P.half_w, P.half_h);
                if (cartesian) { wx = (double)L.vx[j] - vx; wy = (double)L.vy[j] - vy; }   // CPP:86
                else {                                                                       // CPP:88
                    wx = (double)heading_g[j] - (double)heading_g[i];
                    wy = (double)heading_g[n_a + j] - (double)heading_g[n_a + i];
                }
            }
            nbr_g[((size_t)e * n_a + i) * 6 + q] = j;                    // CPP:103-105
            size_t r = (size_t)(r0 + 4 * q) * n_a;
            oput(r, (S)rx); oput(r + n_a, (S)ry); oput(r + 2 * (size_t)n_a, (S)wx); oput(r + 3 * (size_t)n_a, (S)wy);
        }
    };
    if constexpr (!SW) { if (active) agent_block(); }

    STAMP(4);
    EXIT_AT(3);
    // ---- nearest target cell, sensed mask, occupancy bitmaps: CPP:863-913, 145-212 ----
    for (int w = row; w < 2 * P.W; w += nthr) L.occ[w] = 0;             // occ and cov are adjacent
    __syncthreads();
    double smin = 0.0; int imin = 0;
    if constexpr (SW) {
        if (active) {                                                    // the neighbour list some other lane found for this agent
            const unsigned *kj = reinterpret_cast<const unsigned *>(L.wkey);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned v = kj[i * 3 + q];
                j6[2 * q] = (int)(v & 0xFFFFu) - 1; j6[2 * q + 1] = (int)(v >> 16) - 1;
            }
            nn_min = __longlong_as_double((long long)((unsigned long long)kj[3 * P.SL + 2 * i] | ((unsigned long long)kj[3 * P.SL + 2 * i + 1] << 32)));
            agent_block();
            // the cell pass in the same order (agents of one bin reach the same lattice rows); the nearest cell goes
            // back to the agent's lane through LDS like the neighbour list
            const int a = perm_a;
            const int im = lat_cell_pass_lane(P, *LV, true, L.px[a], L.py[a], L.gx, L.gy, L.mask + (size_t)a * P.MW, L.occ, L.cov);
            const_cast<unsigned *>(kj)[5 * P.SL + a] = (unsigned)im;
        }
    } else
    if (active) {
        bool done = false;
        if constexpr (F32) {
            if (swarm) {
                if (SW || LV)                                            // the shapes' lattice description is at hand
                    imin = lat_cell_pass_lane(P, *LV, active, L.px[i], L.py[i], L.gx, L.gy, L.mask + (size_t)row * P.MW, L.occ, L.cov);
                else
                    imin = cell_pass_words(P, active, L.px[i], L.py[i], L.gx, L.gy, L.mask + (size_t)row * P.MW,
                                           L.wkey + (size_t)row * P.MW, L.occ, L.cov);
                done = true;
            } else
                done = cell_pass_fast(P, (float)L.px[i], (float)L.py[i], reinterpret_cast<const float *>(L.gx),
                                      reinterpret_cast<const float *>(L.gy), row, L.mask, L.occ, L.cov, imin);
            if (done) { const double rx = (double)L.gx[imin] - px, ry = (double)L.gy[imin] - py; smin = rx * rx + ry * ry; }
        }
        if (!done) cell_pass<S, true>(P, true, px, py, L.gx, L.gy, row, L.mask, L.occ, L.cov, smin, imin);
    }
    if constexpr (F32) {
        if (split && LV)                                           // the other parts' agents: their occupancy / coverage bits
            for (int a = row; a < n_a; a += nthr)
                if (a != i) lat_occupancy_lane(P, *LV, L.px[a], L.py[a], L.gx, L.gy, L.occ, L.cov);
    }
    STAMP(5);
    EXIT_AT(4);
    if constexpr (SW) { if (row == 0) L.cnt[1] = 0; }                    // length of the emission's queue of lists to reduce
    __syncthreads();
    if constexpr (SW) {
        if (active) {
            imin = (int)reinterpret_cast<const unsigned *>(L.wkey)[5 * P.SL + i];
            const double rx = (double)L.gx[imin] - px, ry = (double)L.gy[imin] - py; smin = rx * rx + ry * ry;
        }
    }
    const double dmin = sqrt(smin);
    const bool in_flag = dmin < sqrt(2.0) * l_cell / 2;                  // CPP:894

    unsigned *mym = L.mask + (size_t)row * P.MW;
    int n_before = 0, n_after = 0, n_occ = 0;
    unsigned nz = 0;                                                     // words of the sensed list that hold cells (W <= 32)
    const unsigned occm = in_flag ? 0xFFFFFFFFu : 0u;                    // CPP:155: occupied cells leave the list only inside the shape
    if (active) {
        for (int w = 0; w < P.W; ++w) {
            const unsigned m = mym[w], o = m & L.occ[w] & occm;
            n_before += __popc(m);
            n_occ += __popc(o);
            nz |= (m != o ? 1u : 0u) << (w & 31);
        }
        n_after = n_before - n_occ;
        inflag_g[(size_t)e * n_a + i] = in_flag ? 1 : 0;                 // CPP:139
        nearest_g[(size_t)e * n_a + i] = imin;

        // target block (CPP:141-142, 896-901)
        double t0, t1, t2, t3;
        if (in_flag) { t0 = px - px; t1 = py - py; t2 = vx - vx; t3 = vy - vy; }
        else { t0 = (double)L.gx[imin] - px; t1 = (double)L.gy[imin] - py; t2 = 0.0 - vx; t3 = 0.0 - vy; }
        const int tb = cartesian ? P.obs_dim - (2 + P.n_obs) * 2 : P.obs_dim - 6;   // CPP:304, 317
        oput((size_t)(tb + 0) * n_a, (S)t0); oput((size_t)(tb + 1) * n_a, (S)t1);
        oput((size_t)(tb + 2) * n_a, (S)t2); oput((size_t)(tb + 3) * n_a, (S)t3);
        if (!cartesian) {                                              // CPP:322-324
            ocol[(size_t)(tb + 4) * n_a] = heading_g[i]; ocol[(size_t)(tb + 5) * n_a] = heading_g[n_a + i];
        }

        // occupied list = sensed_before \ sensed_after (CPP:215-238)
        if (occup_g) {
            int *dst = occup_g + ((size_t)e * n_a + i) * P.n_occ;
            BitIter it; it.init(mym, P.W, L.occ, false);
            emit_list(it, n_occ, P.n_occ, [&](int k, int cell) { dst[k] = cell; });
        }
    }

    STAMP(6);
    EXIT_AT(5);
    // ---- sensed list (CPP:243-296) and exploration reward terms (CPP:499-557) ----
    // psi = rho_cos_dec(z, 0, d_sen), CPP:1017-1025
    auto psi_exact = [&](double gx, double gy) {
        const double z = sqrt(gx * gx + gy * gy);                        // CPP:524
        if (z < 0.0 * P.d_sen) return 1.0;
        else if (z < P.d_sen) return (1.0 / 2.0) * (1.0 + cos(ASM_PI * (z / P.d_sen - 0.0) / (1.0 - 0.0)));
        return 0.0;
    };
    double num0 = 0.0, num1 = 0.0, den = 0.0;
    bool uniform = false, uniform_known = false;
    [[maybe_unused]] bool exact_by_wave = false, exact_in_order = false;
    auto exact_sums = [&]() {                                            // CPP:515-539 in the reference's order (in-shape agents: rew_on)
        num0 = 0.0; num1 = 0.0; den = 0.0;
        BitIter it2; it2.init(mym, P.W, L.occ, true);
        emit_list(it2, n_after, P.n_obs, [&](int, int cell) {
            if (cell >= 0) {
                const double gx = (double)L.gx[cell] - px, gy = (double)L.gy[cell] - py;
                const double psi = psi_exact(gx, gy);
                num0 += psi * gx; num1 += psi * gy; den += psi;
            }
        });
    };
    if constexpr (SW) {
        // Large swarm (W <= 32 words, even cap: checked where the kernel is picked).
        const int lane = threadIdx.x & 63, cap = P.n_obs, gb = P.obs_dim - cap * 2;
        // A sub-sampled list (n > cap, CPP:246-251) is first reduced, in its row of LDS, to the cells it keeps; after that
        // every list is "slot k = k-th cell" and one loop without data-dependent branches serves all lanes.  (A lane that
        // sub-samples inside the slot loop makes its whole wave take the list bits one at a time, and with a few such lanes
        // in most workgroups the others wait for that wave at the next barrier.)  Slot k keeps the cell of rank
        // round(k (n - 1) / (cap - 1)) = floor((2 k (n - 1) + cap - 1) / (2 (cap - 1))): cap - 1 is odd, so the quotient is
        // never half-way and the reference's double arithmetic rounds to the same side.  A wave reduces one list at a
        // time, a lane per byte of the list's bit words: ranks by a prefix sum of the bytes' popcounts, then the cell of
        // rank r stays iff some k has r D <= k inc + cap - 1 < (r + 1) D (inc = 2 (n - 1), D = 2 (cap - 1); inc > D, so
        // there is at most one), i.e. iff k = ceil((r D - cap + 1) / inc) satisfies the right-hand bound and k < cap.
        // The lists to reduce are few and unevenly spread (most waves have none, some eight), so they are queued in the
        // velocity rows of LDS, which nothing reads after the agent block, and dealt out over the workgroup's waves.
        unsigned *q_meta = reinterpret_cast<unsigned *>(L.vx), *q_nz = reinterpret_cast<unsigned *>(L.vy);
        const bool subs = active && n_after > cap;
        const unsigned long long mb = __ballot(subs);
        int qidx = 0;
        if (mb) {
            int b0 = 0;
            if (lane == 0) b0 = atomicAdd(L.cnt + 1, __popcll(mb));
            qidx = __builtin_amdgcn_readfirstlane(b0) +
                   (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0));
            if (subs) q_meta[qidx] = (unsigned)row | ((unsigned)n_after << 10) | (in_flag ? 1u << 21 : 0u);
        }
        __syncthreads();                                                 // (also: every lane is through with its occupied list)
        const int n_queued = L.cnt[1];
        for (int item = __builtin_amdgcn_readfirstlane(row >> 6); item < n_queued; item += nthr >> 6) {
            const unsigned meta = q_meta[item];
            const int row_s = (int)(meta & 1023u), n_s = (int)((meta >> 10) & 2047u);
            const unsigned occ_s = (meta >> 21) ? 0xFFFFFFFFu : 0u;      // CPP:155
            unsigned *ms = L.mask + (size_t)row_s * P.MW;
            const int D = 2 * (cap - 1), inc = 2 * (n_s - 1);
            const float rinc = 1.0f / (float)inc;
            int base = 0;
            unsigned nz_s = 0;
            for (int t0 = 0; t0 < 4 * P.W; t0 += 64) {
                const int t = t0 + lane;
                const bool in = t < 4 * P.W;
                const int w = in ? t >> 2 : 0, q8 = (t & 3) * 8;
                unsigned sub = in ? ((ms[w] & ~(L.occ[w] & occ_s)) >> q8) & 0xFFu : 0u;
                const int cnt = __popc(sub);
                int incl = cnt;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, 64); incl += lane >= m ? o : 0; }
                int r = base + incl - cnt;
                base += __builtin_amdgcn_readlane(incl, 63);
                unsigned keep = 0;
                while (sub) {
                    const unsigned bit = sub & (0u - sub); sub ^= bit;
                    const int A = r * D - (cap - 1);
                    int k = (int)__builtin_ceilf((float)A * rinc);       // within one of ceil(A / inc): all integers below 2^24
                    k = k < 0 ? 0 : k;
                    if (k * inc < A) ++k; else if (k > 0 && (k - 1) * inc >= A) --k;
                    if (k < cap && k * inc + (cap - 1) < (r + 1) * D) keep |= bit;
                    ++r;
                }
                if (in) reinterpret_cast<unsigned char *>(ms)[t] = (unsigned char)keep;   // (every lane has read its word by now)
                const unsigned long long kb = __ballot(keep != 0);
                for (int j = 0; j < 16; ++j) nz_s |= ((kb >> (4 * j)) & 0xFull) ? 1u << ((t0 >> 2) + j) : 0u;
            }
            if (lane == 0) q_nz[item] = nz_s;
        }
        __syncthreads();
        if (subs) { nz = q_nz[qidx]; n_after = cap; }
        if (active) {
            const bool rew_on = in_flag && P.want_reward;
            const int n_out = n_after < cap ? n_after : cap;
            const float pxf = (float)L.px[i], pyf = (float)L.py[i];
            int *sdst = sensed_g ? sensed_g + ((size_t)e * n_a + i) * cap : nullptr;
            float fn0 = 0.f, fn1 = 0.f, fden = 0.f;
            // The cell block's rows go out as buffer stores: resource = the environment's observation block, vector offset =
            // the lane's column, scalar offset = the row, so a row store needs no vector address arithmetic; non-temporal,
            // as the rows of a large swarm are written once and read by a later kernel.
            const __amdgpu_buffer_rsrc_t obs_rs =
                __builtin_amdgcn_make_buffer_rsrc(obs_g + (size_t)e * P.obs_dim * n_a, 0, 0x7fffffff, 0x00020000);
            // (the row offset is the same in every lane; said explicitly, or it lands in a vector register and every store
            // becomes a loop over the distinct values of that register)
            const int row_bytes = __builtin_amdgcn_readfirstlane(n_a * 4), gb_bytes = __builtin_amdgcn_readfirstlane(gb * n_a * 4);
            auto ostore = [&](int r, float v) {                          // r: row within the cell block
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), obs_rs, i * 4,
                                                      __builtin_amdgcn_readfirstlane(gb_bytes + r * row_bytes), 2);
            };
            // Four slots per trip: their cells are taken off the lane's bit words first (a lane's next non-empty word is
            // fetched one word ahead, so running out of bits costs a register move, not an LDS round trip), then the eight
            // coordinate reads go out together, then the arithmetic and the eight row stores.  The only branches are uniform
            // across the wave (a lane needs its next word; the wave has no reward term; the lists of all lanes are through).
            constexpr int RB = 4;
#ifdef ASM_EXP_NOREW
            const bool any_rew = false;                                  // timing experiment only: rewards are wrong
#else
            const bool any_rew = __ballot(rew_on) != 0;
#endif
            auto fetch = [&](unsigned &fb, int &fw) {                    // the lane's next non-empty word (none: no bits)
                const int w = __ffs(nz) - 1;                             // -1 when none is left
                nz &= nz - 1;
                const int wc = w < 0 ? 0 : w;
                const unsigned f = mym[wc] & ~(L.occ[wc] & occm);
                fb = w < 0 ? 0u : f; fw = wc * 32;
            };
            unsigned bits, nbits; int wbase, nwbase;
            fetch(bits, wbase); fetch(nbits, nwbase);
            for (int k0 = 0; k0 < cap; k0 += RB) {                       // (k0 stays in a scalar register)
                if (__ballot(k0 < n_out) == 0) {                         // the lists of all lanes are through: zeros (CPP:279-296)
#pragma unroll
                    for (int q = 0; q < RB; ++q)
                        if (k0 + q < cap) { ostore(2 * (k0 + q), 0.f); ostore(2 * (k0 + q) + 1, 0.f); if (sdst) sdst[k0 + q] = -1; }
                    continue;
                }
                int cell[RB]; bool need[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    need[q] = k0 + q < n_out;
                    const bool refill = need[q] && bits == 0;
                    if (__ballot(refill)) {
                        if (refill) { bits = nbits; wbase = nwbase; fetch(nbits, nwbase); }
                    }
                    cell[q] = need[q] ? wbase + __ffs(bits) - 1 : 0;     // bits != 0 where need
                    bits = need[q] ? bits & (bits - 1) : bits;
                }
                float gx[RB], gy[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) { gx[q] = (float)L.gx[cell[q]]; gy[q] = (float)L.gy[cell[q]]; }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int k = k0 + q;
                    if (k < cap) {
                        float rx = gx[q] - pxf, ry = gy[q] - pyf;        // CPP:285-286
                        rx = need[q] ? rx : 0.f; ry = need[q] ? ry : 0.f;
                        if (any_rew) {
                            const float z = __builtin_amdgcn_sqrtf(__builtin_fmaf(ry, ry, rx * rx));
                            float psi = 0.5f + 0.5f * __builtin_amdgcn_cosf(z * P.inv2dsen_f);          // cos(pi z / d_sen)
                            psi = (rew_on && need[q]) ? psi : 0.f;
                            fn0 = __builtin_fmaf(psi, rx, fn0); fn1 = __builtin_fmaf(psi, ry, fn1); fden += psi;
                        }
                        ostore(2 * k, rx); ostore(2 * k + 1, ry);
                        if (sdst) sdst[k] = need[q] ? cell[q] : -1;
                    }
                }
            }
            if (rew_on && n_after > 0 && fden > 0.f) {
                // Single-precision screen.  The weights psi carry an absolute error below 1e-6 (argument rounding
                // 2*pi*1.5e-7, v_cos_f32 below 5e-7, halved, plus one rounding), every term and every accumulation
                // step a relative rounding of 6e-8, so with n terms and |g| < d_sen
                // |v32 - v| < 0.64 n (1e-6 / den + 6e-8) + 1e-7 for the norm v of the mean offset; doubled here.
                const float rden = 1.0f / fden;
                const float v0 = fn0 * rden, v1 = fn1 * rden;
                const float vn = __builtin_amdgcn_sqrtf(__builtin_fmaf(v1, v1, v0 * v0));
                const float margin = 1.28f * (float)n_out * __builtin_fmaf(1e-6f, rden, 6e-8f) + 2e-7f;
                if (__builtin_fabsf(vn - 0.05f) > margin) { uniform = vn < 0.05f; uniform_known = true; }
            }
#ifdef ASM_EXP_NOREW
            uniform_known = true;
#endif
            if (rew_on && n_after > 0 && !uniform_known) exact_by_wave = true;   // borderline: the exact sums, by the wave below
        }
    } else
    if (active) {
        BitIter it; it.init(mym, P.W, in_flag ? L.occ : nullptr, true);
        int *sdst = sensed_g ? sensed_g + ((size_t)e * n_a + i) * P.n_obs : nullptr;
        const int gb = P.obs_dim - P.n_obs * 2;
        const bool cart = cartesian != 0;
        const bool rew_on = in_flag && P.want_reward;
        if constexpr (F32) {
            // values: fl32(cell - p) is the correctly rounded difference either way; reward
            // weights in fp32 (v_cos_f32), decided in fp32 unless |v| is within the error bound of 0.05
            const float pxf = (float)L.px[i], pyf = (float)L.py[i];
            float fn0 = 0.f, fn1 = 0.f, fden = 0.f;
            // The cell block's rows go out as buffer stores: resource = the observation block of the wave's first
            // environment onwards, vector offset = the lane's column (plus its environment's distance from the first),
            // scalar offset = the row, so a row store needs no vector address arithmetic; non-temporal, as the rows of a
            // large swarm are written once and read by a later kernel.  (A run-time choice between the two store
            // policies here, per store or per copy of the whole loop, doubled the step time of a 1024-agent swarm.)
            const int e_first = __builtin_amdgcn_readfirstlane(e);
            const __amdgpu_buffer_rsrc_t obs_rs =
                __builtin_amdgcn_make_buffer_rsrc(obs_g + (size_t)e_first * P.obs_dim * n_a, 0, 0x7fffffff, 0x00020000);
            const int olb = ((e - e_first) * P.obs_dim * n_a + i) * 4;
            auto ostore = [&](int r, float v) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), obs_rs, olb, r * n_a * 4, 2);
            };
            auto slot = [&](int k, int cell) {
                float gx = 0.f, gy = 0.f;
                if (cell >= 0) {
                    gx = (float)L.gx[cell] - pxf; gy = (float)L.gy[cell] - pyf;          // CPP:285-286
                    if (rew_on) {
                        const float z = __builtin_amdgcn_sqrtf(__builtin_fmaf(gy, gy, gx * gx));
                        const float psi = 0.5f + 0.5f * __builtin_amdgcn_cosf(z * P.inv2dsen_f);   // cos(pi z / d_sen)
                        fn0 = __builtin_fmaf(psi, gx, fn0); fn1 = __builtin_fmaf(psi, gy, fn1); fden += psi;
                    }
                }
                ostore(gb + 2 * k, gx); ostore(gb + 2 * k + 1, gy);
                if (sdst) sdst[k] = cell;
            };
            if (P.W <= 32 && ((P.n_obs - 1) & 1)) {
                // The same emission (CPP:243-296) as one loop over the slots that is uniform across the wave: a lane
                // walks the non-empty words of its list (nz) bit by bit and once no lane of the wave has a cell left
                // the slots are zero-filled.  Slot k of a sub-sampled list (n > cap, CPP:246-251) takes bit
                // round(k (n - 1) / (cap - 1)) = floor((2 k (n - 1) + cap - 1) / (2 (cap - 1))): cap - 1 is odd here, so
                // the quotient is never half-way and the reference's double arithmetic rounds to the same side.  The
                // numerator is kept as a running sum; a lane takes one more bit whenever it passes the next multiple
                // of the denominator (exactly once per slot when the list is not sub-sampled).
                const int cap = P.n_obs, n_out = n_after < cap ? n_after : cap;
                {
                const int D = 2 * (cap - 1), inc = n_after > cap ? 2 * (n_after - 1) : D;
                int acc = cap - 1 - inc, thr = 0;
                unsigned bits = 0; int wbase = 0, cell = -1;
                for (int k = 0; k < cap; ++k) {
                    if (__ballot(k < n_out) == 0) { slot(k, -1); continue; }
                    acc += inc;
                    while (k < n_out && acc >= thr) {
                        thr += D;
                        if (bits == 0) {
                            int w = __ffs(nz) - 1; nz &= nz - 1; w = w < 0 ? 0 : w;
                            bits = mym[w] & ~(L.occ[w] & occm); wbase = w * 32;
                        }
                        cell = wbase + __ffs(bits) - 1; bits &= bits - 1;
                    }
                    slot(k, k < n_out ? cell : -1);
                }
                }
            } else
            emit_list(it, n_after, P.n_obs, slot);
            if (rew_on && n_after > 0 && fden > 0.f) {
                // Single-precision screen.  The weights psi carry an absolute error below 1e-6 (argument rounding
                // 2*pi*1.5e-7, v_cos_f32 below 5e-7, halved, plus one rounding), every term and every accumulation
                // step a relative rounding of 6e-8, so with n terms and |g| < d_sen
                // |v32 - v| < 0.64 n (1e-6 / den + 6e-8) + 1e-7 for the norm v of the mean offset; doubled here.
                const float rden = 1.0f / fden;
                const float v0 = fn0 * rden, v1 = fn1 * rden;
                const float vn = __builtin_amdgcn_sqrtf(__builtin_fmaf(v1, v1, v0 * v0));
                const int n_terms = n_after < P.n_obs ? n_after : P.n_obs;
                const float margin = 1.28f * (float)n_terms * __builtin_fmaf(1e-6f, rden, 6e-8f) + 2e-7f;
                if (__builtin_fabsf(vn - 0.05f) > margin) { uniform = vn < 0.05f; uniform_known = true; }
            }
            if (rew_on && n_after > 0 && !uniform_known) {               // borderline: the exact sums
                // A large swarm hands an agent whose list is not sub-sampled to its whole wave below (one lane walking 80
                // cells through the double-precision cosine holds up the other 63, and its workgroup, for longer than the
                // rest of the step takes); everybody else sums here, in the reference's order.
                if (swarm && n_after <= P.n_obs) exact_by_wave = true;
                else exact_in_order = true;
            }
        } else {
            emit_list(it, n_after, P.n_obs, [&](int k, int cell) {
                double gx = 0.0, gy = 0.0;
                if (cell >= 0) {
                    gx = (double)L.gx[cell] - px; gy = (double)L.gy[cell] - py;          // CPP:285-286, 515-516
                    if (rew_on) { const double psi = psi_exact(gx, gy); num0 += psi * gx; num1 += psi * gy; den += psi; }  // CPP:537-539
                }
                if (cart) { ocol[(size_t)(gb + 2 * k) * n_a] = (S)gx; ocol[(size_t)(gb + 2 * k + 1) * n_a] = (S)gy; }
                if (sdst) sdst[k] = cell;
            });
        }
    }

    if constexpr (F32) {
        if (swarm) {
            // Borderline agents of a large swarm: the reference's double-precision sums (CPP:515-552), one agent at a time
            // with the 64 lanes of its wave sharing the bytes of its list's bit words.  The terms are the reference's; only
            // their order of summation differs (relative error below 80 * 2^-53), so a result within 1e-11 of the threshold
            // falls through to the sequential evaluation in the reference's order.
            const int lane = threadIdx.x & 63;
            for (unsigned long long todo = __ballot(exact_by_wave); todo; todo &= todo - 1) {
                const int src = __ffsll(todo) - 1;
                const int row_s = __builtin_amdgcn_readlane(row, src), i_s = __builtin_amdgcn_readlane(i, src);
                const double sx = (double)L.px[i_s], sy = (double)L.py[i_s];
                const unsigned *ms = L.mask + (size_t)row_s * P.MW;
                double a0 = 0.0, a1 = 0.0, ad = 0.0;
                for (int t = lane; t < 4 * P.W; t += 64) {
                    const int w = t >> 2, q8 = (t & 3) * 8;
                    unsigned sub = ((ms[w] & ~L.occ[w]) >> q8) & 0xFFu;  // (inside the shape: occupied cells are off the list, CPP:155)
                    while (sub) {
                        const int cell = w * 32 + q8 + __ffs(sub) - 1; sub &= sub - 1;
                        const double gx = (double)L.gx[cell] - sx, gy = (double)L.gy[cell] - sy;
                        const double psi = psi_exact(gx, gy);
                        a0 += psi * gx; a1 += psi * gy; ad += psi;
                    }
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    a0 += __shfl_xor(a0, m, 64); a1 += __shfl_xor(a1, m, 64); ad += __shfl_xor(ad, m, 64);
                }
                if (lane == src) {
                    if (ad == 0) ad = 1E-8;                              // CPP:542-544
                    const double v0 = 1.0 * a0 / ad, v1 = 1.0 * a1 / ad; // CPP:547-548
                    const double qq = v0 * v0 + v1 * v1;
                    if (fabs(qq - P.T_v005) > 1e-11 * P.T_v005) { uniform = qq < P.T_v005; uniform_known = true; }   // CPP:550-552
                    else exact_in_order = true;
                }
            }
        }
        if (exact_in_order) exact_sums();
    }
    STAMP(7);
    EXIT_AT(6);
    // ---- reward: CPP:457-564 ----
    double reward = 0.0;
    if (active && P.want_reward) {
        bool collision = false;
#pragma unroll
        for (int q = 0; q < 6; ++q) {                                    // CPP:464-496
            int j = j6[q];
            if (j >= 0) {
                double rx = (double)L.px[j] - px, ry = (double)L.py[j] - py;
                if (periodic) wrap_rel(rx, ry, P.half_w, P.half_h);
                if (rx * rx + ry * ry < P.T_ravoid) collision = true;    // CPP:487 r_avoid > norm
            }
        }
        if (in_flag && n_after > 0 && !uniform_known) {
            if (den == 0) den = 1E-8;                                    // CPP:542-544
            double v0 = 1.0 * num0 / den, v1 = 1.0 * num1 / den;         // CPP:547-548
            uniform = v0 * v0 + v1 * v1 < P.T_v005;                      // CPP:550-552
        }
        if (in_flag && !collision && uniform) reward += 1.0;             // CPP:559-561
        if (rew_g) rew_g[(size_t)e * n_a + i] = (S)reward;
    }

    STAMP(8);
#ifdef ASM_DIAG
    if constexpr (SW) { if (g_diag && (threadIdx.x & 63) == 0) g_diag[(size_t)blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter(); }
#endif
    EXIT_AT(7);
    // ---- per-environment step statistics (TRAIN:128-135; WRAP:48-165) ----
    if (P.want_stats && split) {
        // an environment split over several workgroups: the statistics come from a launch of their own (k_swarm_stats);
        // the covered cells (WRAP:48-72) are counted here, by the first part, from the whole environment's bitmap
        if (P.part == 0 && row == 0 && e < P.E) {
            int covered = 0;
            for (int w = 0; w < P.W; ++w) covered += __popc(L.cov[w]);
            stats_g[(size_t)e * 8 + 2] += P.n_g_e > 0 ? (double)covered / P.n_g_e : 0.0;
        }
    } else if (P.want_stats && swarm) {
        __syncthreads();                                                 // masks no longer needed
        swarm_stats(P, reinterpret_cast<double *>(L.mask), i, active, e, reward, dmin, in_flag, nn_min, L.cov,
                    (double)(contact_pairs < 0 ? *L.cnt : contact_pairs), stats_g);
    } else if (P.want_stats) {
        __syncthreads();                                                 // masks no longer needed
        double *scr = reinterpret_cast<double *>(L.mask);                // 4 doubles per agent (MW >= 8 words)
        if (active) {
            scr[(size_t)i * 4 + 0] = reward; scr[(size_t)i * 4 + 1] = sqrt(nn_min);
            scr[(size_t)i * 4 + 2] = dmin;   scr[(size_t)i * 4 + 3] = in_flag ? 1.0 : 0.0;
        }
        __syncthreads();
        if (i == 0 && e < P.E) {
            double rs = 0, rq = 0, ds = 0, inn = 0, ms = 0, lo = __builtin_inf(), hi = -__builtin_inf();
            for (int a = 0; a < n_a; ++a) {
                double r = scr[(size_t)a * 4], m = scr[(size_t)a * 4 + 1];
                rs += r; rq += r * r; ds += scr[(size_t)a * 4 + 2]; inn += scr[(size_t)a * 4 + 3];
                ms += m; lo = fmin(lo, m); hi = fmax(hi, m);
            }
            double mean = ms / n_a, var = 0;
            for (int a = 0; a < n_a; ++a) { double d = scr[(size_t)a * 4 + 1] - mean; var += d * d; }
            var /= n_a;
            int covered = 0;
            for (int w = 0; w < P.W; ++w) covered += __popc(L.cov[w]);
            double *st = stats_g + (size_t)e * 8;
            st[0] += rs; st[1] += rq; st[2] += P.n_g_e > 0 ? (double)covered / P.n_g_e : 0.0;
            st[3] += (var - lo) / (hi - lo);                             // WRAP:98-99
            st[4] += (double)(contact_pairs < 0 ? *L.cnt : contact_pairs); st[5] += ds / n_a; st[6] += 1.0; st[7] += inn;
        }
    }
    STAMP(9);
}

// Loads the environment's grid (padded with far-away cells) and state into LDS.
template <typename S>
__device__ __forceinline__ void load_env(const DevParams &P, const EnvLds<S> &L, bool env_ok, bool active, int e,
                                         int i, int n_g, const S *p_g, const S *dp_g, const S *grid_g)
{
    // The loads do not depend on n_g (cells are read up to the row's capacity and padded by a select afterwards) nor on
    // one another, and the state is requested first: a load inside "c < n_g ?" waits for n_g before it is even issued,
    // and a load inside any branch is waited for inside that branch.
    S sx = (S)1e30, sy = (S)1e30, su = (S)0, sv = (S)0;                     // unused slots: far away, so the chunked
    if (active) {                                                            // pair screens never select them
        const size_t b = (size_t)e * 2 * P.n_a;
        sx = p_g[b + i]; sy = p_g[b + P.n_a + i]; su = dp_g[b + i]; sv = dp_g[b + P.n_a + i];
    }
    const int e_c = e < P.E ? e : P.E - 1;
    const S *gx_g = grid_g + (size_t)e_c * 2 * P.n_g_max, *gy_g = gx_g + P.n_g_max;
    for (int c = i; c < P.NGP; c += P.SL) {
        const int cc = c < P.n_g_max ? c : P.n_g_max - 1;
        const S xr = gx_g[cc], yr = gy_g[cc];
        const bool in = env_ok && c < n_g;
        L.gx[c] = in ? xr : (S)1e30; L.gy[c] = in ? yr : (S)1e30;
    }
    L.px[i] = sx; L.py[i] = sy; L.vx[i] = su; L.vy[i] = sv;
}

// ---------------------------------------------------------------------------------
template <typename S>
__global__ void k_observe(DevParams P, const S *p_g, const S *dp_g, const S *heading_g, const S *grid_g,
                          const int *ng_g, const double *lcell_g, int *nbr_g, int *inflag_g, int *nearest_g,
                          S *obs_g, S *rew_g, int *sensed_g, int *occup_g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int es = threadIdx.x / P.SL, i = threadIdx.x % P.SL, e = blockIdx.x * P.EPB + es;
    const bool env_ok = e < P.E, active = env_ok && i < P.n_a;
    EnvLds<S> L = carve<S>(smem, P, es);
    const int n_g = env_ok ? clamp_ng(ng_g[e], P.n_g_max) : 0;
    const double l_cell = env_ok ? lcell_g[e] : 1.0;
    load_env<S>(P, L, env_ok, active, e, i, n_g, p_g, dp_g, grid_g);
    __syncthreads();
    DevParams Q = P; Q.n_g_e = n_g; Q.want_stats = 0;
    observe<S>(Q, L, active, e, i, i, P.SL, l_cell, heading_g, nbr_g, inflag_g, nearest_g, obs_g, rew_g, sensed_g, occup_g,
               nullptr, 0);
}

// ENV:482-661: forces + prior on the old state, integration, observation + reward.
template <typename S, bool SW = false>     // SW: the large-swarm specialisation (see observe)
__global__ void k_step(DevParams P, LatArgs lat, S *p_g, S *dp_g, const float *act_g, const S *grid_g, const int *ng_g,
                       const double *lcell_g, int *nbr_g, int *inflag_g, int *nearest_g, S *obs_g, S *rew_g,
                       S *prior_g, int *sensed_g, int *occup_g, double *stats_g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(!SW || sizeof(S) == 4, "the large-swarm specialisation is single precision");
    const int es = SW ? 0 : threadIdx.x / P.SL, i = SW ? (int)threadIdx.x : threadIdx.x % P.SL, e = SW ? (int)blockIdx.x : blockIdx.x * P.EPB + es;
    const bool periodic = !SW && P.periodic, lat2 = SW || P.lat == 2;
    const bool env_ok = e < P.E, active = env_ok && i < P.n_a;
    const int n_a = P.n_a;
    EnvLds<S> L = carve<S>(smem, P, es);
    STAMP(0);
#ifdef ASM_EXITS
    const int xat = g_exit_at;
#endif
    const int e_c = e < P.E ? e : P.E - 1;                          // loads with a clamped index instead of loads inside branches
    const int ng_r = ng_g[e_c];
    const double l_cell_r = lcell_g[e_c];
    [[maybe_unused]] int sid_r = 0;
    if constexpr (sizeof(S) == 4) { if (lat2) sid_r = lat.shape_id[e_c]; }
    const int n_g = env_ok ? clamp_ng(ng_r, P.n_g_max) : 0;
    const double l_cell = env_ok ? l_cell_r : 1.0;
    load_env<S>(P, L, env_ok, active, e, i, n_g, p_g, dp_g, grid_g);
    if (i == 0) *L.cnt = 0;
    // lattice path (large swarms, one environment per workgroup): the shape's tables into LDS now, the compacted
    // boundary coordinates after the barrier (they come out of the grid in LDS); the barriers of the dynamics phase
    // publish both before the observation uses them
    LatView LV;
    [[maybe_unused]] int lat_sid = 0;
    if constexpr (sizeof(S) == 4) {
        if (lat2) {
            lat_sid = env_ok ? sid_r : 0;
            lat_sid = lat_sid < 0 ? 0 : (lat_sid >= lat.n_shapes ? lat.n_shapes - 1 : lat_sid);
            const asm_lattice_shape *sh = lat.shapes + lat_sid;
            for (int k = i; k < ASM_LAT_TAB_BYTES / 16; k += P.SL) reinterpret_cast<uint4 *>(L.lat)[k] = reinterpret_cast<const uint4 *>(sh)[k];
            LV.tab = L.lat; LV.cbx = reinterpret_cast<const float *>(L.lat + ASM_LAT_TAB_BYTES); LV.cby = LV.cbx + ASM_LAT_NB;
            LV.cols = sh->cols; LV.rows = sh->rows; LV.nb = env_ok ? sh->n_boundary : 0;
            LV.x0 = sh->x0; LV.y0 = sh->y0; LV.invl = 1.0 / sh->l_cell;
            LV.c = lat.xform[(size_t)e_c * 4]; LV.s = lat.xform[(size_t)e_c * 4 + 1]; LV.ox = lat.xform[(size_t)e_c * 4 + 2]; LV.oy = lat.xform[(size_t)e_c * 4 + 3];
        }
    }
    __syncthreads();
    STAMP(1);
    EXIT_AT(1);
    if constexpr (sizeof(S) == 4) {
        if (lat2)
            for (int k = i; k < ASM_LAT_NB; k += P.SL) {
                float x = 1e30f, y = 1e30f;
                if (k < LV.nb) { const int c = lat.shapes[lat_sid].boundary[k]; x = L.gx[c]; y = L.gy[c]; }
                const_cast<float *>(LV.cbx)[k] = x; const_cast<float *>(LV.cby)[k] = y;
            }
    }

    // large swarms: cell list of the old positions for the contact search
    [[maybe_unused]] BinView BC;
    [[maybe_unused]] bool use_bins = false;
    if constexpr (sizeof(S) == 4) {
        if (SW || (!periodic && P.SL >= 64 && P.bins)) {
            BC = bins_carve(L.mask, P.bin_c, P.SL);
            bins_build(P, P.bin_c, BC, i, P.SL, P.n_a, reinterpret_cast<const float *>(L.px), reinterpret_cast<const float *>(L.py));
            use_bins = true;
        }
    }

    double npx = 0, npy = 0, nvx = 0, nvy = 0;
    int contacts = 0;
    if (active) {
        // requested now, used behind the contact search: the actions and the prior's inputs (previous neighbour list,
        // in-shape flag, nearest cell)
        const float act_xr = act_g[(size_t)e * 2 * n_a + i], act_yr = act_g[(size_t)e * 2 * n_a + n_a + i];
        int nb_r[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) nb_r[q] = nbr_g[((size_t)e * n_a + i) * 6 + q];
        const int inf_r = inflag_g[(size_t)e * n_a + i], near_r = nearest_g[(size_t)e * n_a + i];
        const double px = (double)L.px[i], py = (double)L.py[i], vx = (double)L.vx[i], vy = (double)L.vy[i];
        // ---- ball-ball spring forces: ENV:437-452 + CPP:740-820, row sum in ascending k ----
        double fx = 0.0, fy = 0.0;
        if (SW || (sizeof(S) == 4 && !periodic && P.SL >= 64)) {                 // large swarm
            if constexpr (sizeof(S) == 4) {
                const float pxf = L.px[i], pyf = L.py[i];
                // With a cell list: the agents that pass the single-precision screen are gathered from the bins
                // around, kept as a short list ordered by index (in this agent's row of the word-key area, idle
                // until the cell pass), then evaluated exactly in ascending k as the reference sums them.
                bool listed = false;
                if (use_bins) {
                    unsigned short *hits = L.wkey + (size_t)i * P.MW;
                    int nh = 0; bool over = false;
                    bins_for_each(P, P.bin_c, BC, pxf, pyf, [&](f32x2 c, int k) {
                        const float rx = c.x - pxf, ry = c.y - pyf;
                        if (__builtin_fmaf(ry, ry, rx * rx) < P.Tcontact_hi_f && k != i) {
                            if (nh < P.MW) {
                                int at = nh++;
                                for (; at > 0 && (int)hits[at - 1] > k; --at) hits[at] = hits[at - 1];
                                hits[at] = (unsigned short)k;
                            } else over = true;
                        }
                    });
                    if (!over) {
                        listed = true;
                        for (int t = 0; t < nh; ++t) {
                            const int k = hits[t];
                            const double dx = (double)L.px[k] - px, dy = (double)L.py[k] - py;
                            const double s = dx * dx + dy * dy;
                            if (s < P.T_contact) {
                                const double d = sqrt(s), gap = fabs(d - P.sizes2);
                                fx += 1.0 * gap * P.k_ball * (-(dx / d));
                                fy += 1.0 * gap * P.k_ball * (-(dy / d));
                                if (k > i) ++contacts;
                            }
                        }
                    }
                }
                if (!listed)                                                     // 32 agents per screen
                for (int j0 = 0; j0 < P.SL; j0 += 32) {
                    unsigned m = pair_screen32(L.px + j0, L.py + j0, pxf, pyf, P.Tcontact_hi_f);
                    if ((unsigned)(i - j0) < 32u) m &= ~(1u << (i - j0));
                    while (m) {                                                  // ascending k, CPP:740-820
                        const int k = j0 + __ffs(m) - 1; m &= m - 1;
                        const double dx = (double)L.px[k] - px, dy = (double)L.py[k] - py;
                        const double s = dx * dx + dy * dy;
                        if (s < P.T_contact) {
                            const double d = sqrt(s), gap = fabs(d - P.sizes2);
                            fx += 1.0 * gap * P.k_ball * (-(dx / d));
                            fy += 1.0 * gap * P.k_ball * (-(dy / d));
                            if (k > i) ++contacts;
                        }
                    }
                }
            }
        } else
        for (int k = 0; k < n_a; ++k) {
            if (k == i) continue;
            if (sizeof(S) == 4) {                                                // fp32 screen: certain misses
                const float fx32 = (float)L.px[k] - (float)L.px[i], fy32 = (float)L.py[k] - (float)L.py[i];
                if (!(__builtin_fmaf(fy32, fy32, fx32 * fx32) < P.Tcontact_hi_f) && !(P.want_stats && periodic && i == 0)) continue;
            }
            double dx = (double)L.px[k] - px, dy = (double)L.py[k] - py;         // CPP:782-785 (delta = p_k - p_i)
            double s = dx * dx + dy * dy;                                        // ENV:444 (no wrap for rows >= 1)
            if (s < P.T_contact) {                                               // ENV:445-446 d - sizes < 0
                double d = sqrt(s);
                double gap = fabs(d - P.sizes2);                                 // ENV:447
                fx += 1.0 * gap * P.k_ball * (-(dx / d));                        // CPP:790-793
                fy += 1.0 * gap * P.k_ball * (-(dy / d));
                if (k > i) ++contacts;
            } else if (P.want_stats && periodic && i == 0) {
                // ENV:442-443: only agent 0's row of the distance matrix is wrapped; it is
                // never read by the force loop but is counted by WRAP:131-145.
                double wx = dx, wy = dy; wrap_rel(wx, wy, P.half_w, P.half_h);
                if (wx * wx + wy * wy < P.T_contact) ++contacts;
            }
        }
        // ---- walls: CPP:822-860, ENV:510-513 ----
        double wfx = 0.0, wfy = 0.0, dfx = 0.0, dfy = 0.0;
        if (!periodic) {
            const double r = P.size_a;
            double d0 = px - r - (-P.half_w);                                    // CPP:841 (boundary_pos[0] = -half_w)
            double d1 = P.half_h - (py + r);                                     // CPP:842
            double d2 = P.half_w - (px + r);                                     // CPP:843
            double d3 = py - r - (-P.half_h);                                    // CPP:844
            const bool c0 = d0 < 0, c1 = d1 < 0, c2 = d2 < 0, c3 = d3 < 0;       // CPP:849
            const double a0 = c0 ? fabs(d0) : 0.0, a1 = c1 ? fabs(d1) : 0.0;
            const double a2 = c2 ? fabs(d2) : 0.0, a3 = c3 ? fabs(d3) : 0.0;
            wfx = (a0 - a2) * P.k_wall;                                          // ENV:512 row 0: [1,0,-1,0]
            wfy = (-a1 + a3) * P.k_wall;                                         //         row 1: [0,-1,0,1]
            dfx = (-(c0 ? vx : 0.0) + -(c2 ? vx : 0.0)) * P.c_wall;              // ENV:513 row 0: [-1,0,-1,0]
            dfy = (-(c1 ? vy : 0.0) + -(c3 ? vy : 0.0)) * P.c_wall;              //         row 1: [0,-1,0,-1]
        }
        // ---- prior on the OLD state with the PREVIOUS neighbour list: ENV:600-619 ----
        if (P.prior && prior_g) {
            int nb[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) nb[q] = nb_r[q];
            double tx, ty;
            // (handed over through an empty asm statement, so that the flag's comparison is not computed where the flag
            // was requested, which would wait for it there)
            int inf_o = inf_r, near = near_r;
            asm volatile("" : "+v"(inf_o), "+v"(near));
            bool inf = inf_o != 0;
            if (P.recompute) {
                double smin; cell_pass<S, false>(P, true, px, py, L.gx, L.gy, i, nullptr, nullptr, nullptr, smin, near);
                inf = sqrt(smin) < sqrt(2.0) * l_cell / 2;
            }
            if (inf) { tx = px; ty = py; } else { tx = (double)L.gx[near]; ty = (double)L.gy[near]; }  // CPP:896-901
            double f0, f1;
            prior_policy<S>(P, px, py, vx, vy, tx, ty, nb, L.px, L.py, L.vx, L.vy, f0, f1);
            prior_g[(size_t)e * 2 * n_a + i] = (S)f0; prior_g[(size_t)e * 2 * n_a + n_a + i] = (S)f1;
        }
        // ---- integration: ENV:626-647 ----
        const float sens = (float)P.sens;
        double ux = (double)(sens * act_xr);                                     // ENV:627,633 (1 * float32)
        double uy = (double)(sens * act_yr);
        double Fx = ux + fx, Fy = uy + fy;
        if (!periodic) { Fx = Fx + wfx + dfx; Fy = Fy + wfy + dfy; }           // ENV:633
        nvx = vx + (Fx / P.mass) * P.dt; nvy = vy + (Fy / P.mass) * P.dt;        // ENV:638-641
        nvx = fmin(fmax(nvx, -P.vel_max), P.vel_max);                            // ENV:642
        nvy = fmin(fmax(nvy, -P.vel_max), P.vel_max);
        npx = px + nvx * P.dt; npy = py + nvy * P.dt;                            // ENV:645
        if (periodic) {                                                        // ENV:646-647, 778-781
            if (npx < -P.half_w) npx += 2 * P.half_w;
            if (npx > P.half_w) npx -= 2 * P.half_w;
            if (npy < -P.half_h) npy += 2 * P.half_h;
            if (npy > P.half_h) npy -= 2 * P.half_h;
        }
    }
    STAMP(2);
    // contact pairs of the environment (upper triangle, WRAP:141-145): summed into the environment's counter (zeroed
    // before the first barrier); it is read in the statistics phase, several barriers on
    if (P.want_stats && active && contacts) atomicAdd(L.cnt, contacts);
    __syncthreads();                                                             // all reads of the old state done
    if (active) {
        const S a = (S)npx, b = (S)npy, c = (S)nvx, d = (S)nvy;
        L.px[i] = a; L.py[i] = b; L.vx[i] = c; L.vy[i] = d;
        const size_t o = (size_t)e * 2 * n_a;
        p_g[o + i] = a; p_g[o + n_a + i] = b; dp_g[o + i] = c; dp_g[o + n_a + i] = d;
    }
    __syncthreads();
    STAMP(3);
    EXIT_AT(2);
    if (!SW && P.parts > 1) {                                        // the observation of the new state is the part launch's
        if (P.want_stats && i == 0 && env_ok) stats_g[(size_t)e * 8 + 4] += (double)*L.cnt;
        return;
    }
    DevParams Q = P; Q.n_g_e = n_g;
    observe<S, SW>(Q, L, active, e, i, i, P.SL, l_cell, nullptr, nbr_g, inflag_g, nearest_g, obs_g, rew_g, sensed_g, occup_g,
                   stats_g, -1, (sizeof(S) == 4 && lat2) ? &LV : nullptr);
}

// Observation + reward of the state in p_g / dp_g for large swarms on the lattice path, an environment split over
// P.parts workgroups of SL / P.parts threads (blockIdx = environment * parts + part).  Each part holds the whole
// environment's state, grid and lattice tables in LDS and per-agent rows for its own agents only.
__global__ void k_observe_part(DevParams P, LatArgs lat, const float *p_g, const float *dp_g, const float *grid_g,
                               const int *ng_g, const double *lcell_g, int *nbr_g, int *inflag_g, int *nearest_g,
                               float *obs_g, float *rew_g, int *sensed_g, int *occup_g, double *stats_g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int nthr = P.rows, row = threadIdx.x, e = blockIdx.x / P.parts, part = blockIdx.x % P.parts, i = part * nthr + row;
    const int n_a = P.n_a;
    const bool active = i < n_a;
    EnvLds<float> L = carve<float>(smem, P, 0);                             // (P.state_rows = 2: no velocity arrays)
    STAMP(3);
    const int n_g = clamp_ng(ng_g[e], P.n_g_max);
    const double l_cell = lcell_g[e];
    for (int c = row; c < P.NGP; c += nthr) {
        float x = 1e30f, y = 1e30f;
        if (c < n_g) { x = grid_g[(size_t)e * 2 * P.n_g_max + c]; y = grid_g[(size_t)e * 2 * P.n_g_max + P.n_g_max + c]; }
        L.gx[c] = x; L.gy[c] = y;
    }
    for (int a = row; a < P.SL; a += nthr) {
        float x = 1e30f, y = 1e30f;                                          // unused slots: far away
        if (a < n_a) { const size_t b = (size_t)e * 2 * n_a; x = p_g[b + a]; y = p_g[b + n_a + a]; }
        L.px[a] = x; L.py[a] = y;
    }
    L.vx = const_cast<float *>(dp_g) + (size_t)e * 2 * n_a; L.vy = L.vx + n_a;   // velocities (own and the six neighbours') straight from memory
    int sid = lat.shape_id[e];
    sid = sid < 0 ? 0 : (sid >= lat.n_shapes ? lat.n_shapes - 1 : sid);
    const asm_lattice_shape *sh = lat.shapes + sid;
    for (int k = row; k < ASM_LAT_TAB_BYTES / 16; k += nthr) reinterpret_cast<uint4 *>(L.lat)[k] = reinterpret_cast<const uint4 *>(sh)[k];
    LatView LV;
    LV.tab = L.lat; LV.cbx = reinterpret_cast<const float *>(L.lat + ASM_LAT_TAB_BYTES); LV.cby = LV.cbx + ASM_LAT_NB;
    LV.cols = sh->cols; LV.rows = sh->rows; LV.nb = sh->n_boundary;
    LV.x0 = sh->x0; LV.y0 = sh->y0; LV.invl = 1.0 / sh->l_cell;
    LV.c = lat.xform[(size_t)e * 4]; LV.s = lat.xform[(size_t)e * 4 + 1]; LV.ox = lat.xform[(size_t)e * 4 + 2]; LV.oy = lat.xform[(size_t)e * 4 + 3];
    __syncthreads();
    for (int k = row; k < ASM_LAT_NB; k += nthr) {                           // published by the barriers of the bin build
        float x = 1e30f, y = 1e30f;
        if (k < LV.nb) { const int c = sh->boundary[k]; x = L.gx[c]; y = L.gy[c]; }
        const_cast<float *>(LV.cbx)[k] = x; const_cast<float *>(LV.cby)[k] = y;
    }
    DevParams Q = P; Q.n_g_e = n_g; Q.part = part;
    observe<float>(Q, L, active, e, i, row, nthr, l_cell, nullptr, nbr_g, inflag_g, nearest_g, obs_g, rew_g, sensed_g, occup_g,
                   stats_g, 0, &LV);
}

// The statistics of a split step (see swarm_stats), recomputed per agent from what the part launch wrote: reward,
// in-shape flag, distance to the nearest cell, distance to the nearest other agent (the first listed neighbour, or a
// scan over everybody when none is within sensing range).  One workgroup of SL threads per environment.
__global__ void k_swarm_stats(DevParams P, const float *p_g, const float *grid_g, const int *nbr_g, const int *inflag_g,
                              const int *nearest_g, const float *rew_g, double *stats_g)
{
    __shared__ double part[16 * 8];
    __shared__ float sx[ASM_MAX_AGENTS], sy[ASM_MAX_AGENTS];
    const int e = blockIdx.x, i = threadIdx.x, n_a = P.n_a;
    const bool active = i < n_a;
    double reward = 0.0, dmin = 0.0, nn_min = __builtin_inf();
    bool in_flag = false;
    if (active) { sx[i] = p_g[(size_t)e * 2 * n_a + i]; sy[i] = p_g[(size_t)e * 2 * n_a + n_a + i]; }
    __syncthreads();
    if (active) {
        const double px = (double)sx[i], py = (double)sy[i];
        reward = (double)rew_g[(size_t)e * n_a + i];
        in_flag = inflag_g[(size_t)e * n_a + i] != 0;
        int c = nearest_g[(size_t)e * n_a + i];
        c = c < 0 ? 0 : (c >= P.n_g_max ? P.n_g_max - 1 : c);
        const double rx = (double)grid_g[(size_t)e * 2 * P.n_g_max + c] - px, ry = (double)grid_g[(size_t)e * 2 * P.n_g_max + P.n_g_max + c] - py;
        dmin = sqrt(rx * rx + ry * ry);
        const int j = nbr_g[((size_t)e * n_a + i) * 6];
        double s0 = 0.0;
        if (j >= 0 && j < n_a) { const double dx = (double)sx[j] - px, dy = (double)sy[j] - py; s0 = dx * dx + dy * dy; }
        if (s0 > 0.0) nn_min = s0;
        else
            for (int k = 0; k < n_a; ++k) {
                const double qx = (double)sx[k] - px, qy = (double)sy[k] - py, su = qx * qx + qy * qy;
                if (su != 0 && su < nn_min) nn_min = su;
            }
    }
    swarm_stats(P, part, i, active, e, reward, dmin, in_flag, nn_min, nullptr, -1.0, stats_g);
}

// ENV:151-218 with counter-based draws (draw order: shape, angle, 2 discarded, offset
// x/y, coin, 2*n_a positions, 2 centre, 2*n_a velocities), then the observation.
template <typename S>
__global__ void k_reset(DevParams P, uint64_t seed, uint32_t episode, int64_t env0, int n_shapes,
                        const double *shape_grids, const int *shape_off, const int *shape_ng,
                        const double *shape_lcell, int *shape_id_g, S *p_g, S *dp_g, S *grid_g, int *ng_g,
                        double *lcell_g, int *nbr_g, int *inflag_g, int *nearest_g, S *obs_g, int *sensed_g,
                        int *occup_g, double *xform_g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int es = threadIdx.x / P.SL, i = threadIdx.x % P.SL, e = blockIdx.x * P.EPB + es;
    const bool env_ok = e < P.E, active = env_ok && i < P.n_a;
    const int n_a = P.n_a;
    EnvLds<S> L = carve<S>(smem, P, es);
    const uint64_t env = (uint64_t)(env0 + e);
    int n_g = 0; double l_cell = 1.0;
    if (env_ok) {
        uint32_t d = 0;
        const int s = (int)(((uint64_t)draw_u32(seed, env, episode, 0u, d++) * (uint64_t)n_shapes) >> 32);  // ENV:155
        n_g = shape_ng[s]; l_cell = shape_lcell[s];
        const double ang = ASM_PI * draw_uniform(seed, env, episode, d++, -1.0, 1.0);                      // ENV:170
        d += 2;                                                                                            // ENV:177
        const double ox = draw_uniform(seed, env, episode, d++, -P.half_w + 1, P.half_w - 1);              // ENV:179
        const double oy = draw_uniform(seed, env, episode, d++, -P.half_h + 1, P.half_h - 1);              // ENV:180
        const double ca = cos(ang), sa = sin(ang);
        const double *g = shape_grids + 2 * (size_t)shape_off[s];
        for (int c = i; c < P.NGP; c += P.SL) {
            S x = (S)1e30, y = (S)1e30;
            if (c < n_g) {                                                                                 // ENV:172-173,182
                double gx = g[2 * c], gy = g[2 * c + 1];
                x = (S)((ca * gx + sa * gy) + ox); y = (S)((-sa * gx + ca * gy) + oy);
            }
            L.gx[c] = x; L.gy[c] = y;
            if (c < P.n_g_max) {
                grid_g[(size_t)e * 2 * P.n_g_max + c] = c < n_g ? x : (S)0;
                grid_g[(size_t)e * 2 * P.n_g_max + P.n_g_max + c] = c < n_g ? y : (S)0;
            }
        }
        const double coin = draw_uniform(seed, env, episode, d++, -1.0, 1.0);                              // ENV:197
        const uint32_t dpos = d, dctr = d + 2u * (uint32_t)n_a, dvel = dctr + 2u;
        if (i == 0) {
            shape_id_g[e] = s; ng_g[e] = n_g; lcell_g[e] = l_cell;
            if (xform_g) { xform_g[(size_t)e * 4] = ca; xform_g[(size_t)e * 4 + 1] = sa; xform_g[(size_t)e * 4 + 2] = ox; xform_g[(size_t)e * 4 + 3] = oy; }
        }
        if (active) {
            double x, y;
            if (coin > 0) {                                                                                // ENV:198-200
                x = draw_uniform(seed, env, episode, dpos + (uint32_t)i, -P.half_w, P.half_w);
                y = draw_uniform(seed, env, episode, dpos + (uint32_t)(n_a + i), -P.half_h, P.half_h);
            } else {                                                                                       // ENV:202-203
                const double cx = draw_uniform(seed, env, episode, dctr, -P.half_w + 1, P.half_w - 1);
                const double cy = draw_uniform(seed, env, episode, dctr + 1u, -P.half_h + 1, P.half_h - 1);
                x = draw_uniform(seed, env, episode, dpos + (uint32_t)i, -1.0, 1.0) + cx;
                y = draw_uniform(seed, env, episode, dpos + (uint32_t)(n_a + i), -1.0, 1.0) + cy;
            }
            const double u = draw_uniform(seed, env, episode, dvel + (uint32_t)i, -0.5, 0.5);              // ENV:210
            const double v = draw_uniform(seed, env, episode, dvel + (uint32_t)(n_a + i), -0.5, 0.5);
            const S a = (S)x, b = (S)y, c = (S)u, dd = (S)v;
            L.px[i] = a; L.py[i] = b; L.vx[i] = c; L.vy[i] = dd;
            const size_t o = (size_t)e * 2 * n_a;
            p_g[o + i] = a; p_g[o + n_a + i] = b; dp_g[o + i] = c; dp_g[o + n_a + i] = dd;
        } else {
            L.px[i] = (S)1e30; L.py[i] = (S)1e30; L.vx[i] = (S)0; L.vy[i] = (S)0;
        }
    } else {
        for (int c = i; c < P.NGP; c += P.SL) { L.gx[c] = (S)1e30; L.gy[c] = (S)1e30; }
    }
    if (P.state_only) return;                                       // the caller observes the new state with k_step_fast
    __syncthreads();
    DevParams Q = P; Q.n_g_e = n_g; Q.want_stats = 0; Q.want_reward = 0;
    observe<S>(Q, L, active, e, i, i, P.SL, l_cell, nullptr, nbr_g, inflag_g, nearest_g, obs_g, nullptr, sensed_g, occup_g,
               nullptr, 0);
}

