// Small kernels around the step: random actions, replay draw/gather, the rule expert, formation metrics, self-tests.
// Part of the single translation unit asm_kernels.hip (included there, inside namespace asmenv, after the
// diagnostic macros); not a stand-alone header.

__global__ void k_random_actions(uint64_t seed, uint32_t step, int64_t env0, int E, int n_a, float *act)
{
    const size_t per_env = (size_t)2 * n_a;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)E * per_env) return;
    const uint64_t env = (uint64_t)(env0 + (int64_t)(idx / per_env));
    const uint32_t d = (uint32_t)(idx % per_env);
    const double u = (double)draw_u32(seed, env, step, 1u, d) * (1.0 / 4294967296.0);
    act[idx] = (float)(-1.0 + 2.0 * u);                                          // ENV:519
}

// ---- replay: uniform transition ids and the gather of sampled transitions ----------------
// out[k] = lo + floor(u * (hi - lo)), u from two 32-bit draws of counter stream 2
// (counter_dev: optional device word added to the counter, so that a captured launch draws afresh on every replay)
__global__ void k_replay_draw(uint64_t seed, uint32_t counter, const uint32_t *counter_dev, int64_t lo, int64_t hi, int N, int64_t *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    if (counter_dev) counter += *counter_dev;
    const uint64_t u = ((uint64_t)draw_u32(seed, (uint64_t)k, counter, 2u, 0u) << 32) | draw_u32(seed, (uint64_t)k, counter, 2u, 1u);
    out[k] = lo + (int64_t)__umul64hi(u, (uint64_t)(hi - lo));
}

// One wavefront per sampled transition id = ((slot * T + t) * E + e) * n_a + i.  The observation of step t is
// frame t of the episode slot and the next observation is frame t + 1 of the same slot, so an observation
// is stored once.  An agent's observation is a column of the (obs_dim, n_a) frame: obs_dim strided reads,
// one contiguous row written.
__global__ void k_replay_gather(int E, int n_a, int obs_dim, int T, const float *obs_ring, const float *act_ring,
                                const float *prior_ring, const float *rew_ring, const int64_t *index, int N,
                                float *obs_out, float *act_out, float *rew_out, float *next_out, float *prior_out)
{
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= N) return;
    const int64_t id = index[k];
    const int i = (int)(id % n_a);
    const int64_t q = id / n_a;                                     // (slot * T + t) * E + e
    const int e = (int)(q % E);
    const int64_t st = q / E, slot = st / T, t = st - slot * T;
    const size_t frame = (size_t)obs_dim * n_a;
    const float *o = obs_ring + ((size_t)(slot * (T + 1) + t) * E + e) * frame + i;
    const float *o2 = o + (size_t)E * frame;
    for (int r = lane; r < obs_dim; r += 64) {
        obs_out[(size_t)k * obs_dim + r] = o[(size_t)r * n_a];
        next_out[(size_t)k * obs_dim + r] = o2[(size_t)r * n_a];
    }
    if (lane < 2) {
        act_out[(size_t)k * 2 + lane] = act_ring[((size_t)q * 2 + lane) * n_a + i];
        if (prior_out) prior_out[(size_t)k * 2 + lane] = prior_ring[((size_t)q * 2 + lane) * n_a + i];
    }
    if (lane == 2) rew_out[k] = rew_ring[(size_t)q * n_a + i];
}

// Observation frames as the policy network reads them (maddpg.py:84 `observations.t()`, train_assembly.py:119):
// rows[e][i][r] = obs[e][r][i].  One workgroup per (environment, block of 32 agents): the block's [obs_dim][32]
// tile goes through LDS, read as 128-byte row pieces and written as whole agent rows (obs_dim contiguous floats).
#define ASM_ROWS_TILE 32
__global__ __launch_bounds__(256) void k_obs_rows(int n_a, int obs_dim, int tiles, const float *obs, float *rows)
{
    extern __shared__ float tile[];                                 // [obs_dim][33]: the odd stride keeps the column reads off one bank
    const int e = blockIdx.x / tiles, i0 = (blockIdx.x - e * tiles) * ASM_ROWS_TILE;
    const int ni = n_a - i0 < ASM_ROWS_TILE ? n_a - i0 : ASM_ROWS_TILE;
    const float *src = obs + (size_t)e * obs_dim * n_a + i0;
    for (int k = threadIdx.x; k < obs_dim * ASM_ROWS_TILE; k += 256) {
        const int r = k >> 5, c = k & 31;
        if (c < ni) tile[r * 33 + c] = src[(size_t)r * n_a + c];
    }
    __syncthreads();
    float *dst = rows + ((size_t)e * n_a + i0) * obs_dim;           // the block's agents are consecutive rows: one contiguous run
    for (int k = threadIdx.x; k < ni * obs_dim; k += 256) {
        const int c = k / obs_dim, r = k - c * obs_dim;
        dst[k] = tile[r * 33 + c];
    }
}

// The 'rule' expert of ENV:525-596 for every agent of every environment, on the state the last observation was
// taken of (its in-shape flags, nearest cells and sensed lists are inputs): entering velocity towards the nearest
// cell when outside the shape, exploration velocity = 15 x the psi-weighted mean offset of the listed cells,
// interaction velocity from all agents inside d_sen; the sum clipped to [-1, 1].  One workgroup per environment,
// one lane per agent, double precision in the reference's order of operations.
__global__ void __launch_bounds__(1024) k_rule_actions(DevParams P, const float *p_g, const float *dp_g, const float *grid_g,
                                                       const int *ng_g, const int *inflag_g, const int *nearest_g,
                                                       const int *sensed_g, float *act_g)
{
    __shared__ float spx[ASM_MAX_AGENTS], spy[ASM_MAX_AGENTS], svx[ASM_MAX_AGENTS], svy[ASM_MAX_AGENTS];
    const int e = blockIdx.x, n_a = P.n_a, cap = P.n_obs, NG = P.n_g_max;
    const float *gx = grid_g + (size_t)e * 2 * NG, *gy = gx + NG;
    const size_t o = (size_t)e * 2 * n_a;
    for (int j = threadIdx.x; j < n_a; j += blockDim.x) {
        spx[j] = p_g[o + j]; spy[j] = p_g[o + n_a + j]; svx[j] = dp_g[o + j]; svy[j] = dp_g[o + n_a + j];
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i >= n_a) return;
    const double px = spx[i], py = spy[i], vx = svx[i], vy = svy[i];
    const double k_1 = 1, k_2 = 15, k_3 = 17;                                   // ENV:527
    double a0 = 0.0, a1 = 0.0;
    if (!inflag_g[(size_t)e * n_a + i]) {                                       // ENV:531-536
        const int c = nearest_g[(size_t)e * n_a + i];
        const double tx = (double)gx[c] - px, ty = (double)gy[c] - py;
        const double tn = sqrt(tx * tx + ty * ty) + 1e-8;
        a0 = k_1 * (tx / tn) + (0.0 - vx); a1 = k_1 * (ty / tn) + (0.0 - vy);
    }
    {                                                                           // ENV:569-579
        const int *sl = sensed_g + ((size_t)e * n_a + i) * cap;
        double num0 = 0.0, num1 = 0.0, den = 0.0; int n = 0;
        for (int k = 0; k < cap; ++k) {
            const int c = sl[k];
            if (c < 0) break;
            const double g0 = (double)gx[c] - px, g1 = (double)gy[c] - py;
            const double psi = reward_weight(P, sqrt(g0 * g0 + g1 * g1));
            num0 += psi * g0; num1 += psi * g1; den += psi; ++n;
        }
        if (n > 0) {
            if (den == 0) den = 1e-8;
            a0 = a0 + k_2 * num0 / den; a1 = a1 + k_2 * num1 / den;
        }
    }
    {                                                                           // ENV:582-593
        int near = 0;
        for (int j = 0; j < n_a; ++j) {
            const double rx = (double)spx[j] - px, ry = (double)spy[j] - py;
            near += (j != i && sqrt(rx * rx + ry * ry) < P.d_sen) ? 1 : 0;
        }
        double w0 = 0.0, w1 = 0.0;
        for (int j = 0; j < n_a; ++j) {
            const double rx = (double)spx[j] - px, ry = (double)spy[j] - py;
            const double nr = sqrt(rx * rx + ry * ry);
            if (j == i || !(nr < P.d_sen)) continue;
            if (nr < P.r_avoid) {
                const double f = -k_3 * (P.r_avoid / nr - 1);
                w0 += f * rx; w1 += f * ry;
            }
            w0 += 5 * ((double)svx[j] - vx) / near; w1 += 5 * ((double)svy[j] - vy) / near;
        }
        a0 = a0 + w0; a1 = a1 + w1;
    }
    act_g[o + i] = (float)fmin(fmax(a0, -1.0), 1.0); act_g[o + n_a + i] = (float)fmin(fmax(a1, -1.0), 1.0);   // ENV:596
}

// Formation metrics of WRAP:48-165 for the given positions, one workgroup per environment, all in double with
// the reference's definitions (distances are norms, i.e. square roots; argmin/min keep the first minimum; sums
// run left to right).  out[e] = {coverage_rate, distribution_uniformity, voronoi_uniformity, contact pairs,
// average_distance_to_target} (the order of WRAP:181-186).
__global__ void __launch_bounds__(256) k_metrics(int n_a, int NGP, double r_half, double sizes2, const float *p_g,
                                                 const float *grid_g, const int *ng_g, double *out)
{
    __shared__ double px[ASM_MAX_AGENTS], py[ASM_MAX_AGENTS], md[ASM_MAX_AGENTS], mg[ASM_MAX_AGENTS];
    __shared__ int cnt[ASM_MAX_AGENTS];
    __shared__ int covered, contacts;
    const int e = blockIdx.x, tid = threadIdx.x, n_g = clamp_ng(ng_g[e], NGP);
    const float *gx = grid_g + (size_t)e * 2 * NGP, *gy = gx + NGP;
    for (int j = tid; j < n_a; j += 256) {
        px[j] = (double)p_g[(size_t)e * 2 * n_a + j]; py[j] = (double)p_g[(size_t)e * 2 * n_a + n_a + j];
        cnt[j] = 0;
    }
    if (tid == 0) { covered = 0; contacts = 0; }
    __syncthreads();
    for (int g = tid; g < n_g; g += 256) {                          // WRAP:48-72 and 103-129: per cell
        const double cx = (double)gx[g], cy = (double)gy[g];
        double dmin = __builtin_inf(); int jmin = 0; bool any = false;
        for (int j = 0; j < n_a; ++j) {
            const double dx = px[j] - cx, dy = py[j] - cy;
            const double d = sqrt(dx * dx + dy * dy);
            if (d < dmin) { dmin = d; jmin = j; }                   // np.argmin: first minimum
            any |= d < r_half;                                      // WRAP:67
        }
        atomicAdd(&cnt[jmin], 1);
        if (any) atomicAdd(&covered, 1);
    }
    for (int i = tid; i < n_a; i += 256) {                          // WRAP:74-101, 131-145, 147-165: per agent
        double m = __builtin_inf(); int c = 0;
        for (int j = 0; j < n_a; ++j) {
            const double dx = px[j] - px[i], dy = py[j] - py[i];
            const double d = sqrt(dx * dx + dy * dy);
            if (d != 0 && d < m) m = d;                             // WRAP:93-94
            if (j > i && d - sizes2 < 0) ++c;                       // ENV:445-446, upper triangle WRAP:142
        }
        md[i] = m;
        double t = __builtin_inf();
        for (int g = 0; g < n_g; ++g) {
            const double dx = (double)gx[g] - px[i], dy = (double)gy[g] - py[i];
            const double d = sqrt(dx * dx + dy * dy);
            if (d < t) t = d;
        }
        mg[i] = t;
        if (c) atomicAdd(&contacts, c);
    }
    __syncthreads();
    if (tid < 3) {                                                  // three independent left-to-right reductions
        double r;
        if (tid == 2) {
            double tot = 0.0;
            for (int i = 0; i < n_a; ++i) tot += mg[i];
            r = tot / n_a;                                          // WRAP:163
        } else {
            double mean = 0.0, lo = __builtin_inf(), hi = -__builtin_inf();
            for (int i = 0; i < n_a; ++i) {
                const double x = tid == 0 ? md[i] : (double)cnt[i];
                mean += x; lo = fmin(lo, x); hi = fmax(hi, x);
            }
            mean /= n_a;
            double var = 0.0;
            for (int i = 0; i < n_a; ++i) {
                const double x = tid == 0 ? md[i] : (double)cnt[i];
                var += (x - mean) * (x - mean);
            }
            var /= n_a;
            r = (var - lo) / (hi - lo);                             // WRAP:99, 127
        }
        out[(size_t)e * 5 + (tid == 0 ? 1 : tid == 1 ? 2 : 4)] = r;
    }
    if (tid == 3) {
        out[(size_t)e * 5 + 0] = n_g > 0 ? (double)covered / n_g : 0.0;             // WRAP:70
        out[(size_t)e * 5 + 3] = (double)contacts;
    }
}

__global__ void k_sqrt_f64(const double *in, double *out, int n)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = sqrt(in[k]);
}

// max over the float bit patterns [lo, hi] of |v_cos_f32(x) - cos(2 pi x)| (v_cos_f32 takes revolutions)
__global__ void k_cos_f32_err(uint32_t lo, uint32_t hi, unsigned long long *out)
{
    double worst = 0.0;
    for (uint64_t b = (uint64_t)lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= (uint64_t)hi;
         b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const double err = fabs((double)__builtin_amdgcn_cosf(x) - cos(2.0 * ASM_PI * (double)x));
        worst = fmax(worst, err);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) worst = fmax(worst, __shfl_xor(worst, m, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(worst));   // non-negative doubles order like integers
}

