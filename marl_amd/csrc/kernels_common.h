// Device helpers shared by the kernels: the counter-based generator, the LDS layout of the generic kernels, bit-list
// iteration, the prior policy, the agent x cell passes and the sub-sampled list emission.
// Part of the single translation unit asm_kernels.hip (included there, inside namespace asmenv, after the
// diagnostic macros); not a stand-alone header.

// ---- Philox4x32-10 (same generator as the CPU oracle) ------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ uint32_t draw_u32(uint64_t seed, uint64_t env, uint32_t c1, uint32_t c2, uint32_t d)
{
    uint32_t o[4];
    philox4x32(d >> 2, c1, c2, 0u, (uint32_t)env, (uint32_t)seed ^ (uint32_t)(env >> 32), o);
    return o[d & 3u];
}
__device__ __forceinline__ double draw_uniform(uint64_t seed, uint64_t env, uint32_t ep, uint32_t d, double lo, double hi)
{
    double u = (double)draw_u32(seed, env, ep, 0u, d) * (1.0 / 4294967296.0);
    return lo + (hi - lo) * u;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS of the lattice path per environment in the generic kernels: tables as in asm_lattice_shape (row windows, boundary
// list), then x and y of the compacted boundary cells
#define ASM_LAT_TAB_BYTES (8 * ASM_LAT_ROWS * ASM_LAT_CHUNKS + 2 * ASM_LAT_NB)
#define ASM_LAT_LDS_BYTES (ASM_LAT_TAB_BYTES + 8 * ASM_LAT_NB)

// n_g[e] is caller data in device memory: the kernels read cells [0, min(max(n_g, 0), row length)) whatever it holds
__device__ __forceinline__ int clamp_ng(int n_g, int n_g_max) { return n_g < 0 ? 0 : (n_g > n_g_max ? n_g_max : n_g); }

// stats[k] += v as one memory-side add without a returned value: the wave does not wait for the old value, so a
// workgroup ends without a last round trip to memory.  Every statistic has exactly one adding lane per launch (launches
// of a stream are ordered), so the sum is the same IEEE addition, in the same order, as a load / add / store.
__device__ __forceinline__ void stat_add(double *p, double v)
{
    (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
}

// Requests every 64-byte line of the kernel's argument block at once, as the wave's first instructions, and waits for
// them.  A launch's arguments lie at a fresh address, so the first waves of a launch on a CU miss the scalar cache on
// every line; the compiler fetches argument fields where they are first needed, each fetch followed by its own wait, and
// a kernel with several hundred bytes of arguments then starts with as many round trips to memory in a row as it
// touches lines.  With all lines in flight together it is one round trip; the compiler's own loads hit the cache.
// BYTES: the size of the argument block (the explicit arguments; at most 640 bytes are covered).
// The statement is tied to the workgroup index (returned unchanged) instead of being volatile: a volatile statement counts
// as a possible write to memory, after which the compiler no longer reads uniform values through the scalar unit.
template <int BYTES>
__device__ __forceinline__ int kernarg_warm(int block)
{
#ifndef ASM_EXP_NO_KERNARG_WARM
    static_assert(BYTES >= 4 && BYTES <= 640, "ten lines");
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int last = (BYTES - 4) & ~3;
#define ASM_KA_OFF(k) ((k) * 64 < last ? (k) * 64 : last)
    unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    asm("s_load_dword %0, %11, %12\n\t"
                 "s_load_dword %1, %11, %13\n\t"
                 "s_load_dword %2, %11, %14\n\t"
                 "s_load_dword %3, %11, %15\n\t"
                 "s_load_dword %4, %11, %16\n\t"
                 "s_load_dword %5, %11, %17\n\t"
                 "s_load_dword %6, %11, %18\n\t"
                 "s_load_dword %7, %11, %19\n\t"
                 "s_load_dword %8, %11, %20\n\t"
                 "s_load_dword %9, %11, %21\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "+s"(block)
                 : "s"(ka), "i"(ASM_KA_OFF(0)), "i"(ASM_KA_OFF(1)), "i"(ASM_KA_OFF(2)), "i"(ASM_KA_OFF(3)), "i"(ASM_KA_OFF(4)),
                   "i"(ASM_KA_OFF(5)), "i"(ASM_KA_OFF(6)), "i"(ASM_KA_OFF(7)), "i"(ASM_KA_OFF(8)), "i"(ASM_KA_OFF(9)));
#undef ASM_KA_OFF
#endif
    return block;
}

// ---- LDS carve ------------------------------------------------------------------
template <typename S>
struct EnvLds {
    S *px, *py, *vx, *vy, *gx, *gy;
    unsigned *mask, *occ, *cov;
    unsigned short *wkey;          // [SL][MW] upper half of the smallest key of every word (large-swarm cell pass)
    int *cnt;                      // contact pairs of the environment's step (k_step)
    unsigned char *lat;            // lattice path of k_step: row windows + boundary list, then the compacted boundary coordinates
};

template <typename S>
__device__ __forceinline__ EnvLds<S> carve(unsigned char *base, const DevParams &P, int es)
{
    unsigned char *b = base + (size_t)es * P.lds_env_bytes;
    EnvLds<S> L;
    S *f = reinterpret_cast<S *>(b);
    L.px = f; L.py = f + P.SL; L.vx = f + 2 * P.SL; L.vy = f + 3 * P.SL;
    L.gx = f + P.state_rows * P.SL; L.gy = L.gx + P.NGP;
    unsigned *u = reinterpret_cast<unsigned *>(L.gy + P.NGP);
    L.mask = u; L.occ = u + (size_t)P.rows * P.MW; L.cov = L.occ + P.W;
    L.wkey = reinterpret_cast<unsigned short *>(L.cov + P.W);
    L.cnt = reinterpret_cast<int *>(b + P.lds_env_bytes - 16);          // the last 16 bytes of the environment's block
    L.lat = b + P.lds_env_bytes - 16 - ASM_LAT_LDS_BYTES;               // only there when P.lat == 2
    return L;
}

// ---- iteration over the set bits of a per-agent cell mask, ascending ----------
struct BitIter {
    const unsigned *words; int w; int W; unsigned bits; const unsigned *andw; bool inv;
    __device__ __forceinline__ unsigned load(int k) const
    {
        unsigned v = words[k];
        if (andw) v &= inv ? ~andw[k] : andw[k];
        return v;
    }
    __device__ __forceinline__ void init(const unsigned *m, int W_, const unsigned *a, bool inv_)
    { words = m; W = W_; andw = a; inv = inv_; w = 0; bits = load(0); }
    __device__ __forceinline__ int next()
    {
        while (bits == 0) { if (++w >= W) return -1; bits = load(w); }
        int b = __ffs(bits) - 1; bits &= bits - 1;
        return w * 32 + b;
    }
};

// CPP:1126-1201 robotPolicy, given the target position
template <typename S>
__device__ __forceinline__ void prior_policy(const DevParams &P, double px, double py, double vx, double vy,
                                             double tx, double ty, const int *nbr, const S *X, const S *Y,
                                             const S *VX, const S *VY, double &f0, double &f1)
{
    f0 = 0.0; f1 = 0.0;
    double d0 = tx - px, d1 = ty - py;                                   // CPP:1147
    double dist = sqrt(d0 * d0 + d1 * d1);                               // CPP:1148
    if (dist > 0) { f0 += 2.0 * d0 / dist; f1 += 2.0 * d1 / dist; }      // CPP:1150-1153
    double av0 = 0.0, av1 = 0.0; int cnt = 0;
    for (int k = 0; k < P.topo; ++k) {
        int j = nbr[k];
        if (j == -1) continue;                                           // CPP:1219-1227
        double n0 = px - (double)X[j], n1 = py - (double)Y[j];           // CPP:1167
        double s = n0 * n0 + n1 * n1;
        if (s > 0 && s < P.T_ravoid) {                                   // CPP:1171 (dn > 0 && dn < r_avoid)
            double dn = sqrt(s);
            double u0 = n0 / dn, u1 = n1 / dn;
            double factor = 3.0 * (P.r_avoid / dn - 1.0);                // CPP:1176
            f0 += factor * u0; f1 += factor * u1;
        }
        av0 += (double)VX[j]; av1 += (double)VY[j]; ++cnt;               // CPP:1182-1184
    }
    if (cnt > 0) {
        av0 /= cnt; av1 /= cnt;                                          // CPP:1189-1190
        f0 += 2.0 * (av0 - vx); f1 += 2.0 * (av1 - vy);                  // CPP:1192-1193
    }
    f0 = fmax(-1.0, fmin(f0, 1.0)); f1 = fmax(-1.0, fmin(f1, 1.0));      // CPP:1197-1198
}

// Nearest cell (first minimum, CPP:889) of agent at (px,py) over grid arrays; optionally
// builds the sensed mask (CPP:905-910) and the per-cell occupancy bitmaps.
template <typename S, bool MASKS>
__device__ __forceinline__ void cell_pass(const DevParams &P, bool active, double px, double py, const S *gx,
                                          const S *gy, int i, unsigned *mask, unsigned *occ, unsigned *cov,
                                          double &smin, int &imin)
{
    smin = __builtin_inf(); imin = 0;
    for (int w = 0; w < P.W; ++w) {
        unsigned word = 0;
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            const int c = w * 32 + b;
            double rx = (double)gx[c] - px, ry = (double)gy[c] - py;     // CPP:876-877
            double s = rx * rx + ry * ry;                                // CPP:885 (norm squared)
            if (s < smin) { smin = s; imin = c; }
            if (MASKS) {
                word |= (s < P.T_sen ? 1u : 0u) << b;                    // CPP:907  norm < d_sen
                if (active && s < P.T_half_le) {                         // CPP:190  !(norm > r_avoid/2)
                    atomicOr(&occ[w], 1u << b);
                    if (s < P.T_half_lt) atomicOr(&cov[w], 1u << b);     // WRAP:67  norm < r_avoid/2
                }
            }
        }
        if (MASKS) mask[(size_t)i * P.MW + w] = word;
    }
}


// v_med3_u32: the median of three unsigned values
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// v_min3_u32
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Bit b of the result: fl32 |(X[b], Y[b]) - (px, py)|^2 < thr, b = 0..31 (X, Y 16-byte aligned in LDS).
__device__ __forceinline__ unsigned pair_screen32(const float *X, const float *Y, float px, float py, float thr)
{
    unsigned m = 0;
    const f32x2 p2x = {px, px}, p2y = {py, py};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 x4 = reinterpret_cast<const float4 *>(X)[q], y4 = reinterpret_cast<const float4 *>(Y)[q];
        const f32x2 xa = {x4.x, x4.y}, xb = {x4.z, x4.w}, ya = {y4.x, y4.y}, yb = {y4.z, y4.w};
        const f32x2 rxa = xa - p2x, rxb = xb - p2x, rya = ya - p2y, ryb = yb - p2y;
        const f32x2 sa = __builtin_elementwise_fma(rya, rya, rxa * rxa), sb = __builtin_elementwise_fma(ryb, ryb, rxb * rxb);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(sa.x - thr), 31);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(sa.y - thr), 31);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(sb.x - thr), 31);
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(sb.y - thr), 31);
    }
    return __brev(m);
}

// ---- cell list of one environment's agents (large swarms; one environment per workgroup) ----
// Agents in bin order (row-major bins): start[b] .. start[b + 1] are the places of bin b's agents in pos / idx; the
// order inside a bin is whatever the counting pass produced, so users must not depend on it.
struct BinView {
    unsigned *start;               // [nbx * nby + 1] (+ 16 wave totals used while building)
    f32x2 *pos;                    // [SL]
    unsigned short *idx;           // [SL]
};

__device__ __forceinline__ BinView bins_carve(unsigned *region, const BinGeom &G, int SL)
{
    BinView B;
    const int nb1 = G.nbx * G.nby + 1;
    B.start = region;
    B.pos = reinterpret_cast<f32x2 *>(region + ((nb1 + 16 + 1) & ~1));
    B.idx = reinterpret_cast<unsigned short *>(B.pos + SL);
    return B;
}

__device__ __forceinline__ int bin_coord(float x, float x0, float inv_h, int nb)
{
    const float v = __builtin_fminf(__builtin_fmaxf((x + x0) * inv_h, 0.f), (float)(nb - 1));    // NaN -> 0
    return (int)v;
}

// Builds the list of the environment's agents 0 .. n - 1 (positions X, Y in LDS) with the nthr threads t = 0 .. nthr - 1
// of the workgroup (nthr a multiple of 64, at least SL / 4: up to four agents per thread).  Reached by every thread;
// five barriers, the first of which also orders this build after whatever used the region before.
__device__ __forceinline__ void bins_build(const DevParams &P, const BinGeom &G, const BinView &B, int t, int nthr, int n,
                                           const float *X, const float *Y)
{
    const int nb1 = G.nbx * G.nby + 1;
    __syncthreads();
    for (int k = t; k < nb1 + 16; k += nthr) B.start[k] = 0;
    __syncthreads();
    int b[4]; unsigned rank[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int a = t + u * nthr;
        b[u] = -1; rank[u] = 0;
        if (a < n) {
            b[u] = bin_coord(Y[a], P.bin_y0, G.inv_h, G.nby) * G.nbx + bin_coord(X[a], P.bin_x0, G.inv_h, G.nbx);
            rank[u] = atomicAdd(&B.start[b[u]], 1u);
        }
    }
    __syncthreads();
    // exclusive prefix sum in place: thread t owns C consecutive counters (C odd: conflict-free strides)
    const int C = ((nb1 + nthr - 1) / nthr) | 1, k0 = t * C;
    unsigned sum = 0;
    for (int k = 0; k < C; ++k) if (k0 + k < nb1) sum += B.start[k0 + k];
    unsigned incl = sum;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const unsigned v = __shfl_up(incl, m, 64); if ((t & 63) >= m) incl += v; }
    unsigned *wtot = B.start + nb1;
    if ((t & 63) == 63) wtot[t >> 6] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int w = 0; w < (t >> 6); ++w) run += wtot[w];
    for (int k = 0; k < C; ++k) if (k0 + k < nb1) { const unsigned c = B.start[k0 + k]; B.start[k0 + k] = run; run += c; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (b[u] >= 0) {
            const int a = t + u * nthr;
            const unsigned at = B.start[b[u]] + rank[u];
            const f32x2 q = {X[a], Y[a]};
            B.pos[at] = q; B.idx[at] = (unsigned short)a;
        }
    __syncthreads();
}

// f(position, index) for every listed agent within `reach` bins of (x, y), row of bins by row.
template <typename F>
__device__ __forceinline__ void bins_for_each(const DevParams &P, const BinGeom &G, const BinView &B, float x, float y, F f)
{
    const int bx = bin_coord(x, P.bin_x0, G.inv_h, G.nbx), by = bin_coord(y, P.bin_y0, G.inv_h, G.nby);
    const int xa = bx - G.reach > 0 ? bx - G.reach : 0, xb = bx + G.reach < G.nbx - 1 ? bx + G.reach : G.nbx - 1;
    const int ya = by - G.reach > 0 ? by - G.reach : 0, yb = by + G.reach < G.nby - 1 ? by + G.reach : G.nby - 1;
    for (int r = ya; r <= yb; ++r) {
        unsigned t = B.start[r * G.nbx + xa];
        const unsigned hi = B.start[r * G.nbx + xb + 1];
        for (; t < hi; ++t) f(B.pos[t], (int)B.idx[t]);
    }
}

// Occupancy (CPP:190, !(norm > r_avoid/2)) and coverage (WRAP:67, norm < r_avoid/2) bits of one agent for
// one 32-cell word, OR-ed into the environment's bitmaps.  `oc` is the single-precision screen s32 < T_half;
// it is exact for both tests unless some cell lies within the margin of the threshold (smallest |s32 - T_half|
// in `ambo`), in which case this agent settles the whole word in double.
__device__ __forceinline__ void settle_occupancy(const DevParams &P, bool active, unsigned oc, float ambo, double px,
                                                 double py, const float *gx, const float *gy, unsigned *occ_w,
                                                 unsigned *cov_w)
{
    if (!active) oc = 0;
    const bool undecided = active && !(ambo > P.Thalf_amb_f);
    if (__ballot(undecided)) {                                              // rare
        if (undecided) {
            oc = 0;
            unsigned cv = 0;
#pragma nounroll
            for (int b = 0; b < 32; ++b) {
                const double rxd = (double)gx[b] - px, ryd = (double)gy[b] - py;
                const double sd = rxd * rxd + ryd * ryd;
                if (sd < P.T_half_le) oc |= 1u << b;
                if (sd < P.T_half_lt) cv |= 1u << b;
            }
            if (cv) atomicOr(cov_w, cv);
            if (oc) atomicOr(occ_w, oc);
            oc = 0;
        }
    }
    if (oc) { atomicOr(occ_w, oc); atomicOr(cov_w, oc); }
}

// The agent x cell pass of one agent over all W words with single-precision screening and exact
// settlement of whatever the screen cannot decide, word by word (same scheme as k_step_fast below,
// one lane doing all the words): sensed mask, occupancy/coverage bits, nearest cell (first minimum).
__device__ __forceinline__ int cell_pass_words(const DevParams &P, bool active, float pxf, float pyf, const float *gx,
                                               const float *gy, unsigned *mw, unsigned short *wkey, unsigned *occ,
                                               unsigned *cov)
{
    const double px = pxf, py = pyf;
    unsigned G1 = 0xFFFFFFFFu, G2 = 0xFFFFFFFFu, G1w = 0, ambw = 0;
    const f32x2 p2x = {pxf, pxf}, p2y = {pyf, pyf};
    const f32x2 thr = {P.Tsen_f, P.Thalf_f};
    for (int w = 0; w < P.W; ++w) {
        unsigned k1 = 0xFFFFFFFFu, word = 0, ocand = 0;              // k1: the word's smallest s32 (bit pattern)
        float amb = __builtin_inff(), ambo = __builtin_inff();
        const float4 *X = reinterpret_cast<const float4 *>(gx + w * 32);
        const float4 *Y = reinterpret_cast<const float4 *>(gy + w * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x4 = X[q], y4 = Y[q];
            const f32x2 xa = {x4.x, x4.y}, xb = {x4.z, x4.w}, ya = {y4.x, y4.y}, yb = {y4.z, y4.w};
            const f32x2 rxa = xa - p2x, rxb = xb - p2x, rya = ya - p2y, ryb = yb - p2y;
            const f32x2 sa = __builtin_elementwise_fma(rya, rya, rxa * rxa);
            const f32x2 sb = __builtin_elementwise_fma(ryb, ryb, rxb * rxb);
            const float ss[4] = {sa.x, sa.y, sb.x, sb.y};
            k1 = umin3(k1, __float_as_uint(sa.x), __float_as_uint(sa.y));
            k1 = umin3(k1, __float_as_uint(sb.x), __float_as_uint(sb.y));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sv = ss[r];
                const f32x2 s2 = {sv, sv};
                const f32x2 d = s2 - thr;                                   // (s - T_sen, s - T_half)
                word = __builtin_amdgcn_alignbit(word, __float_as_uint(d.x), 31);
                ocand = __builtin_amdgcn_alignbit(ocand, __float_as_uint(d.y), 31);
                amb = __builtin_fminf(amb, __builtin_fabsf(d.x));
                ambo = __builtin_fminf(ambo, __builtin_fabsf(d.y));
            }
        }
        mw[w] = __brev(word);
        wkey[w] = (unsigned short)(k1 >> 16);
        if (!(amb > P.Tsen_amb_f)) ambw |= 1u << w;
        settle_occupancy(P, active, __brev(ocand), ambo, px, py, gx + w * 32, gy + w * 32, &occ[w], &cov[w]);
        if (k1 < G1) G1w = (unsigned)w;                                     // strict: the first word holding the smallest s32
        G2 = umed3(G1, G2, k1);                                             // the two smallest word minima
        G1 = G1 < k1 ? G1 : k1;
    }
    // Nearest cell: the candidates are the cells whose s32 lies within two units of 32 ulps of the smallest one; they
    // sit in its word and, rarely, in words whose minimum is that close (16-bit keys: a coarser, safe test).  Those
    // words are screened again (same arithmetic, same s32) and the candidates compared in double, ascending index +
    // strict '<': the reference's first minimum (CPP:889).
    const unsigned lo = G1 >> 5, lim = (lo << 5) + 64u;
    unsigned minw = active ? 1u << G1w : 0u;                                // padding lanes have no nearest cell
    if (active && !((G2 >> 5) - lo >= 2u))
        for (int w = 0; w < P.W; ++w) if ((unsigned)wkey[w] - (G1 >> 16) < 2u) minw |= 1u << w;
    double best = __builtin_inf(); int imin = 0;
    while (minw) {
        const int w = __ffs(minw) - 1; minw &= minw - 1;
        const float4 *X = reinterpret_cast<const float4 *>(gx + w * 32);
        const float4 *Y = reinterpret_cast<const float4 *>(gy + w * 32);
        unsigned cand = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x4 = X[q], y4 = Y[q];
            const f32x2 xa = {x4.x, x4.y}, xb = {x4.z, x4.w}, ya = {y4.x, y4.y}, yb = {y4.z, y4.w};
            const f32x2 rxa = xa - p2x, rxb = xb - p2x, rya = ya - p2y, ryb = yb - p2y;
            const f32x2 sa = __builtin_elementwise_fma(rya, rya, rxa * rxa);
            const f32x2 sb = __builtin_elementwise_fma(ryb, ryb, rxb * rxb);
            cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(sa.x) - lim, 31);
            cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(sa.y) - lim, 31);
            cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(sb.x) - lim, 31);
            cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(sb.y) - lim, 31);
        }
        cand = __brev(cand);
        while (cand) {
            const int c = w * 32 + __ffs(cand) - 1; cand &= cand - 1;
            const double dx = (double)gx[c] - px, dy = (double)gy[c] - py;
            const double sd = dx * dx + dy * dy;
            if (sd < best) { best = sd; imin = c; }
        }
    }
    while (ambw) {                                                          // rare: sensed bits within the margin of T_sen
        const int w = __ffs(ambw) - 1; ambw &= ambw - 1;
        unsigned word = mw[w];
#pragma unroll 4
        for (int b = 0; b < 32; ++b) {
            const int c = w * 32 + b;
            const float rx = gx[c] - pxf, ry = gy[c] - pyf;
            const float s32 = __builtin_fmaf(ry, ry, rx * rx);
            if (__builtin_fabsf(s32 - P.Tsen_f) <= P.Tsen_amb_f) {
                const double dx = (double)gx[c] - px, dy = (double)gy[c] - py;
                const double sd = dx * dx + dy * dy;
                if (sd < P.T_sen) word |= 1u << b; else word &= ~(1u << b);
            }
        }
        mw[w] = word;
    }
    return imin;
}

// Single-precision screening pass over the cells for one agent (fp32 state only).
//
// s32 = fl32 evaluation of |cell - p|^2 differs from the double-precision value s the
// reference decides on by at most a few 2^-24 relative (one rounding in each difference,
// squared, plus product/fma roundings: < 4e-7).  With the margin EPS = 2e-6 every
// decision below is either certain in fp32 or handed to the exact double evaluation:
//   * sensed bit  s < T_sen      : sign of (s32 - T_sen); the smallest |s32 - T_sen| is
//                                  tracked and, if within EPS*T_sen, the whole pass is redone exactly;
//   * occupancy   s < T_half     : rare; candidates s32 < T_half*(1+EPS) are evaluated exactly in place;
//   * nearest cell (first minimum): the two smallest keys (s32 bits with the low 5 bits replaced
//                                  by the cell's position in its 32-cell word) are tracked; if the
//                                  runner-up is within 2 truncated units (2^-17 relative) the pass is
//                                  redone exactly, otherwise the minimum is certain and unique.
// Returns false when the exact pass is required.
__device__ __forceinline__ bool cell_pass_fast(const DevParams &P, float px, float py, const float *gx,
                                               const float *gy, int i, unsigned *mask, unsigned *occ,
                                               unsigned *cov, int &imin)
{
    unsigned G1 = 0xFFFFFFFFu, G2 = 0xFFFFFFFFu;
    int G1w = 0;
    float amb = __builtin_inff();
    for (int w = 0; w < P.W; ++w) {
        unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, word = 0;
        const float4 *X = reinterpret_cast<const float4 *>(gx + w * 32);
        const float4 *Y = reinterpret_cast<const float4 *>(gy + w * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x4 = X[q], y4 = Y[q];
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = q * 4 + r;
                const float rx = xs[r] - px, ry = ys[r] - py;
                const float s = __builtin_fmaf(ry, ry, rx * rx);
                const unsigned key = (__float_as_uint(s) & ~31u) | (unsigned)b;
                k2 = umed3(k1, k2, key);
                k1 = k1 < key ? k1 : key;
                const float d = s - P.Tsen_f;
                word = __builtin_amdgcn_alignbit(word, __float_as_uint(d), 31);   // (word << 1) | (s < T_sen)
                amb = __builtin_fminf(amb, __builtin_fabsf(d));
                if (s < P.Thalf_hi_f) {                                            // rare: exact evaluation
                    const double rxd = (double)xs[r] - (double)px, ryd = (double)ys[r] - (double)py;
                    const double sd = rxd * rxd + ryd * ryd;
                    if (sd < P.T_half_le) {
                        atomicOr(&occ[w], 1u << b);
                        if (sd < P.T_half_lt) atomicOr(&cov[w], 1u << b);
                    }
                }
            }
        }
        mask[(size_t)i * P.MW + w] = __brev(word);                                 // first cell ended up in bit 31
        if (k1 < G1) G1w = w;
        const unsigned t = G1 > k1 ? G1 : k1;
        const unsigned m = G2 < k2 ? G2 : k2;
        G2 = m < t ? m : t;
        G1 = G1 < k1 ? G1 : k1;
    }
    imin = G1w * 32 + (int)(G1 & 31u);
    return ((G2 >> 5) - (G1 >> 5) >= 2u) && (amb > P.Tsen_amb_f);
}

// Uniform-subsampled emission of a bit list: for k in [0, cap): f(k, cell or -1).
// count <= cap: the k-th set bit; count > cap: bit number round(k*(count-1)/(cap-1))
// (CPP:224-229, 246-251; std::round, half away from zero).
template <typename F>
__device__ __forceinline__ void emit_list(BitIter &it, int count, int cap, F f)
{
    const int n_out = count < cap ? count : cap;
    const double step = count > cap ? (double)(count - 1) / (cap - 1) : 1.0;
    int rank = -1, cell = -1;
    for (int k = 0; k < cap; ++k) {
        if (k < n_out) {
            int target = count > cap ? (int)round(k * step) : k;
            while (rank < target) { cell = it.next(); ++rank; }
            f(k, cell);
        } else {
            f(k, -1);
        }
    }
}

