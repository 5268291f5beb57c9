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
#define DIAG_COUNT(k, v) do { if (g_diag && threadIdx.x == 0) g_diag[(size_t)blockIdx.x * 32 + (k)] = (unsigned long long)(v); } while (0)
#define STAMP(k) do { if (g_diag && threadIdx.x == 0) { g_diag[(size_t)blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); \
    if ((k) == 0) g_diag[(size_t)blockIdx.x * 32 + 10] = wall_clock64(); if ((k) == 9) g_diag[(size_t)blockIdx.x * 32 + 11] = wall_clock64(); } } while (0)
#else
#define STAMP(k) do { } while (0)
#define DIAG_COUNT(k, v) do { } while (0)
#endif

__device__ __forceinline__ void wrap_rel(double &x, double &y, double hw, double hh)
{   // CPP:705-720
    if (x < -hw) x += 2 * hw; else if (x > hw) x -= 2 * hw;
    if (y < -hh) y += 2 * hh; else if (y > hh) y -= 2 * hh;
}

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

// ---- LDS carve ------------------------------------------------------------------
template <typename S>
struct EnvLds {
    S *px, *py, *vx, *vy, *gx, *gy;
    unsigned *mask, *occ, *cov;
    unsigned short *wkey;          // [SL][MW] upper half of the smallest key of every word (large-swarm cell pass)
};

template <typename S>
__device__ __forceinline__ EnvLds<S> carve(unsigned char *base, const DevParams &P, int es)
{
    unsigned char *b = base + (size_t)es * P.lds_env_bytes;
    EnvLds<S> L;
    S *f = reinterpret_cast<S *>(b);
    L.px = f; L.py = f + P.SL; L.vx = f + 2 * P.SL; L.vy = f + 3 * P.SL;
    L.gx = f + 4 * P.SL; L.gy = L.gx + P.NGP;
    unsigned *u = reinterpret_cast<unsigned *>(L.gy + P.NGP);
    L.mask = u; L.occ = u + (size_t)P.SL * P.MW; L.cov = L.occ + P.W;
    L.wkey = reinterpret_cast<unsigned short *>(L.cov + P.W);
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

// ---------------------------------------------------------------------------------
// Observation + reward of environment e for agent i on the state held in LDS.
// CPP:23-356 (observation), CPP:359-631 (reward).  Must be reached by every thread.
//
// S = float takes single-precision screening paths wherever a decision is certain in
// fp32 and falls back to the exact double evaluation otherwise (cell_pass_fast above;
// the neighbour and reward screens below follow the same rule), so its decisions are
// those of the all-double code; S = double is the all-double code.
template <typename S>
__device__ void observe(const DevParams &P, const EnvLds<S> &L, bool active, int e, int i, double l_cell,
                        const S *heading_g, int *nbr_g, int *inflag_g, int *nearest_g, S *obs_g, S *rew_g,
                        int *sensed_g, int *occup_g, double *stats_g, int contact_pairs)
{
    constexpr bool F32 = sizeof(S) == 4;
    const int n_a = P.n_a;
    // large swarms on fp32 state: pair loops screen 32 agents at a time, the cell pass settles undecided words only
    const bool swarm = F32 && !P.periodic && P.SL >= 64;
    const double px = active ? (double)L.px[i] : 0.0, py = active ? (double)L.py[i] : 0.0;
    const double vx = active ? (double)L.vx[i] : 0.0, vy = active ? (double)L.vy[i] : 0.0;
    S *ocol = obs_g + (size_t)e * P.obs_dim * n_a + i;                   // obs[row*n_a + agent]

    // ---- k nearest neighbours inside d_sen: _get_focused, CPP:633-703 ----
    double s6[6]; int j6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { s6[q] = __builtin_inf(); j6[q] = -1; }
    double nn_min = __builtin_inf();
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
                const float pxf = L.px[i], pyf = L.py[i];
                // Candidates (fp32 screen) go through a min/max insertion chain on single-precision keys (s32 with
                // the low 10 bits replaced by the index) that keeps the 7 smallest.  The list is read off the keys
                // when every neighbouring pair differs by at least two units of 1024 ulps and every listed key lies
                // two units below T_sen; otherwise this agent redoes its list with exact double insertion.
                unsigned K[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) K[q] = 0xFFFFFFFFu;
                for (int j0 = 0; j0 < P.SL; j0 += 32) {
                    unsigned m = pair_screen32(L.px + j0, L.py + j0, pxf, pyf, P.Tsen_hi_f);
                    if ((unsigned)(i - j0) < 32u) m &= ~(1u << (i - j0));        // self removed CPP:677-681
                    while (m) {
                        const int j = j0 + __builtin_ctz(m); m &= m - 1;
                        const float rx = L.px[j] - pxf, ry = L.py[j] - pyf;
                        unsigned x = (__float_as_uint(__builtin_fmaf(ry, ry, rx * rx)) & ~1023u) | (unsigned)j;
#pragma unroll
                        for (int q = 0; q < 7; ++q) { const unsigned lo = K[q] < x ? K[q] : x; x = K[q] < x ? x : K[q]; K[q] = lo; }
                    }
                }
                const unsigned tlo = (__float_as_uint(P.Tsen_f) >> 10) - 2u;
                bool redo = false;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const bool there = K[q] != 0xFFFFFFFFu;
                    const unsigned u = K[q] >> 10;
                    redo |= there && (u >= tlo || (K[q + 1] >> 10) - u < 2u);
                    j6[q] = there ? (int)(K[q] & 1023u) : -1;
                }
                if (redo) {                                                     // rare
#pragma unroll
                    for (int q = 0; q < 6; ++q) j6[q] = -1;
                    for (int j0 = 0; j0 < P.SL; j0 += 32) {
                        unsigned m = pair_screen32(L.px + j0, L.py + j0, pxf, pyf, P.Tsen_hi_f);
                        if ((unsigned)(i - j0) < 32u) m &= ~(1u << (i - j0));
                        while (m) {                                             // ascending j: ties keep the lower index
                            const int j = j0 + __ffs(m) - 1; m &= m - 1;
                            const double dx = (double)L.px[j] - px, dy = (double)L.py[j] - py;
                            const double s = dx * dx + dy * dy;                 // CPP:641
                            if (s < P.T_sen) consider(j, s);                    // CPP:663
                        }
                    }
                }
                if (P.want_stats) {                                             // nearest other agent (WRAP:81-99)
                    double s0 = 0.0;
                    if (j6[0] >= 0) { const double dx = (double)L.px[j6[0]] - px, dy = (double)L.py[j6[0]] - py; s0 = dx * dx + dy * dy; }
                    if (s0 > 0.0) nn_min = s0;
                    else
                        for (int j = 0; j < n_a; ++j) {
                            const double rx = (double)L.px[j] - px, ry = (double)L.py[j] - py, su = rx * rx + ry * ry;
                            if (su != 0 && su < nn_min) nn_min = su;
                        }
                }
            }
        } else if (F32 && !P.periodic) {
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
                if (P.periodic) wrap_rel(rx, ry, P.half_w, P.half_h);        // CPP:93-95
                const double s = rx * rx + ry * ry;                          // CPP:641
                if (j != i && s < P.T_sen) consider(j, s);
            }
        }
        // agent block, column-major flatten of rows [pos_x,pos_y,vel_x,vel_y] (CPP:108-131)
        int r0 = 0;
        if (P.con_self) {
            ocol[0] = (S)px; ocol[(size_t)n_a] = (S)py; ocol[(size_t)2 * n_a] = (S)vx; ocol[(size_t)3 * n_a] = (S)vy;
            r0 = 4;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int j = j6[q];
            double rx = 0, ry = 0, wx = 0, wy = 0;
            if (j >= 0) {
                rx = (double)L.px[j] - px; ry = (double)L.py[j] - py;
                if (P.periodic) wrap_rel(rx, ry, P.half_w, P.half_h);
                if (P.cartesian) { wx = (double)L.vx[j] - vx; wy = (double)L.vy[j] - vy; }   // CPP:86
                else {                                                                       // CPP:88
                    wx = (double)heading_g[j] - (double)heading_g[i];
                    wy = (double)heading_g[n_a + j] - (double)heading_g[n_a + i];
                }
            }
            nbr_g[((size_t)e * n_a + i) * 6 + q] = j;                    // CPP:103-105
            size_t r = (size_t)(r0 + 4 * q) * n_a;
            ocol[r] = (S)rx; ocol[r + n_a] = (S)ry; ocol[r + 2 * (size_t)n_a] = (S)wx; ocol[r + 3 * (size_t)n_a] = (S)wy;
        }
    }

    STAMP(4);
    // ---- nearest target cell, sensed mask, occupancy bitmaps: CPP:863-913, 145-212 ----
    for (int w = i; w < 2 * P.W; w += P.SL) L.occ[w] = 0;               // occ and cov are adjacent
    __syncthreads();
    double smin = 0.0; int imin = 0;
    if (active) {
        bool done = false;
        if constexpr (F32) {
            if (swarm) {
                imin = cell_pass_words(P, active, L.px[i], L.py[i], L.gx, L.gy, L.mask + (size_t)i * P.MW,
                                       L.wkey + (size_t)i * P.MW, L.occ, L.cov);
                done = true;
            } else
                done = cell_pass_fast(P, (float)L.px[i], (float)L.py[i], reinterpret_cast<const float *>(L.gx),
                                      reinterpret_cast<const float *>(L.gy), i, L.mask, L.occ, L.cov, imin);
            if (done) { const double rx = (double)L.gx[imin] - px, ry = (double)L.gy[imin] - py; smin = rx * rx + ry * ry; }
        }
        if (!done) cell_pass<S, true>(P, true, px, py, L.gx, L.gy, i, L.mask, L.occ, L.cov, smin, imin);
    }
    STAMP(5);
    const double dmin = sqrt(smin);
    const bool in_flag = dmin < sqrt(2.0) * l_cell / 2;                  // CPP:894
    __syncthreads();

    unsigned *mym = L.mask + (size_t)i * P.MW;
    int n_before = 0, n_after = 0, n_occ = 0;
    if (active) {
        for (int w = 0; w < P.W; ++w) {
            unsigned m = mym[w];
            n_before += __popc(m);
            if (in_flag) { n_occ += __popc(m & L.occ[w]); }              // CPP:155: only when inside the shape
        }
        n_after = n_before - n_occ;
        inflag_g[(size_t)e * n_a + i] = in_flag ? 1 : 0;                 // CPP:139
        nearest_g[(size_t)e * n_a + i] = imin;

        // target block (CPP:141-142, 896-901)
        double t0, t1, t2, t3;
        if (in_flag) { t0 = px - px; t1 = py - py; t2 = vx - vx; t3 = vy - vy; }
        else { t0 = (double)L.gx[imin] - px; t1 = (double)L.gy[imin] - py; t2 = 0.0 - vx; t3 = 0.0 - vy; }
        const int tb = P.cartesian ? P.obs_dim - (2 + P.n_obs) * 2 : P.obs_dim - 6;   // CPP:304, 317
        ocol[(size_t)(tb + 0) * n_a] = (S)t0; ocol[(size_t)(tb + 1) * n_a] = (S)t1;
        ocol[(size_t)(tb + 2) * n_a] = (S)t2; ocol[(size_t)(tb + 3) * n_a] = (S)t3;
        if (!P.cartesian) {                                              // CPP:322-324
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
    if (active) {
        BitIter it; it.init(mym, P.W, in_flag ? L.occ : nullptr, true);
        int *sdst = sensed_g ? sensed_g + ((size_t)e * n_a + i) * P.n_obs : nullptr;
        const int gb = P.obs_dim - P.n_obs * 2;
        const bool cart = P.cartesian != 0;
        const bool rew_on = in_flag && P.want_reward;
        if constexpr (F32) {
            // values: fl32(cell - p) is the correctly rounded difference either way; reward
            // weights in fp32 (v_cos_f32), decided in fp32 unless |v| is within 2e-4 of 0.05
            const float pxf = (float)L.px[i], pyf = (float)L.py[i];
            float fn0 = 0.f, fn1 = 0.f, fden = 0.f;
            emit_list(it, n_after, P.n_obs, [&](int k, int cell) {
                float gx = 0.f, gy = 0.f;
                if (cell >= 0) {
                    gx = (float)L.gx[cell] - pxf; gy = (float)L.gy[cell] - pyf;          // CPP:285-286
                    if (rew_on) {
                        const float z = __builtin_sqrtf(__builtin_fmaf(gy, gy, gx * gx));
                        const float psi = 0.5f + 0.5f * __builtin_amdgcn_cosf(z * P.inv2dsen_f);   // cos(pi z / d_sen)
                        fn0 = __builtin_fmaf(psi, gx, fn0); fn1 = __builtin_fmaf(psi, gy, fn1); fden += psi;
                    }
                }
                ocol[(size_t)(gb + 2 * k) * n_a] = (S)gx; ocol[(size_t)(gb + 2 * k + 1) * n_a] = (S)gy;
                if (sdst) sdst[k] = cell;
            });
            if (rew_on && n_after > 0 && fden > 0.f) {
                const float v0 = fn0 / fden, v1 = fn1 / fden;
                const float vn = __builtin_sqrtf(__builtin_fmaf(v1, v1, v0 * v0));
                if (__builtin_fabsf(vn - 0.05f) > 2e-4f) { uniform = vn < 0.05f; uniform_known = true; }
            }
            if (rew_on && n_after > 0 && !uniform_known) {               // borderline: the exact sums
                BitIter it2; it2.init(mym, P.W, L.occ, true);
                emit_list(it2, n_after, P.n_obs, [&](int, int cell) {
                    if (cell >= 0) {
                        const double gx = (double)L.gx[cell] - px, gy = (double)L.gy[cell] - py;
                        const double psi = psi_exact(gx, gy);
                        num0 += psi * gx; num1 += psi * gy; den += psi;
                    }
                });
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

    STAMP(7);
    // ---- reward: CPP:457-564 ----
    double reward = 0.0;
    if (active && P.want_reward) {
        bool collision = false;
#pragma unroll
        for (int q = 0; q < 6; ++q) {                                    // CPP:464-496
            int j = j6[q];
            if (j >= 0) {
                double rx = (double)L.px[j] - px, ry = (double)L.py[j] - py;
                if (P.periodic) wrap_rel(rx, ry, P.half_w, P.half_h);
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
    // ---- per-environment step statistics (TRAIN:128-135; WRAP:48-165) ----
    if (P.want_stats) {
        __syncthreads();                                                 // masks no longer needed
        double *scr = reinterpret_cast<double *>(L.mask);                // 4 doubles per agent (MW >= 8 words)
        if (active) {
            scr[(size_t)i * 4 + 0] = reward; scr[(size_t)i * 4 + 1] = sqrt(nn_min);
            scr[(size_t)i * 4 + 2] = dmin;   scr[(size_t)i * 4 + 3] = in_flag ? 1.0 : 0.0;
        }
        __syncthreads();
        if (swarm) {
            // whole-workgroup reductions (one environment per workgroup here): wave shuffles, then the wave partials
            const int wave = i >> 6, nw = P.SL >> 6;
            double *part = scr + (size_t)4 * P.SL;                           // [nw][8] behind the per-agent values
            double v[7] = {active ? reward : 0.0, active ? dmin : 0.0, active && in_flag ? 1.0 : 0.0,
                           active ? sqrt(nn_min) : 0.0, active ? sqrt(nn_min) : __builtin_inf(),
                           active ? sqrt(nn_min) : -__builtin_inf(), 0.0};
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                v[0] += __shfl_xor(v[0], m, 64); v[1] += __shfl_xor(v[1], m, 64); v[2] += __shfl_xor(v[2], m, 64);
                v[3] += __shfl_xor(v[3], m, 64);
                v[4] = fmin(v[4], __shfl_xor(v[4], m, 64)); v[5] = fmax(v[5], __shfl_xor(v[5], m, 64));
            }
            if ((i & 63) == 0) for (int q = 0; q < 6; ++q) part[wave * 8 + q] = v[q];
            __syncthreads();
            double ms = 0;
            for (int w = 0; w < nw; ++w) ms += part[w * 8 + 3];
            const double mean = ms / n_a, dv = active ? sqrt(nn_min) - mean : 0.0;
            double sq = dv * dv;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
            if ((i & 63) == 0) part[wave * 8 + 6] = sq;
            __syncthreads();
            if (i == 0 && e < P.E) {
                double rs = 0, ds = 0, inn = 0, var = 0, lo = __builtin_inf(), hi = -__builtin_inf();
                for (int w = 0; w < nw; ++w) {
                    rs += part[w * 8]; ds += part[w * 8 + 1]; inn += part[w * 8 + 2]; var += part[w * 8 + 6];
                    lo = fmin(lo, part[w * 8 + 4]); hi = fmax(hi, part[w * 8 + 5]);
                }
                var /= n_a;
                int covered = 0;
                for (int w = 0; w < P.W; ++w) covered += __popc(L.cov[w]);
                double *st = stats_g + (size_t)e * 8;
                st[0] += rs; st[1] += rs; st[2] += (double)covered / P.n_g_e;   // rewards are 0 or 1
                st[3] += (var - lo) / (hi - lo);                             // WRAP:98-99
                st[4] += (double)contact_pairs; st[5] += ds / n_a; st[6] += 1.0; st[7] += inn;
            }
        } else if (i == 0 && e < P.E) {
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
            st[0] += rs; st[1] += rq; st[2] += (double)covered / P.n_g_e;
            st[3] += (var - lo) / (hi - lo);                             // WRAP:98-99
            st[4] += (double)contact_pairs; st[5] += ds / n_a; st[6] += 1.0; st[7] += inn;
        }
    }
    STAMP(9);
}

// Loads the environment's grid (padded with far-away cells) and state into LDS.
template <typename S>
__device__ __forceinline__ void load_env(const DevParams &P, const EnvLds<S> &L, bool env_ok, bool active, int e,
                                         int i, int n_g, const S *p_g, const S *dp_g, const S *grid_g)
{
    for (int c = i; c < P.NGP; c += P.SL) {
        S x = (S)1e30, y = (S)1e30;
        if (env_ok && c < n_g) {
            x = grid_g[(size_t)e * 2 * P.n_g_max + c];
            y = grid_g[(size_t)e * 2 * P.n_g_max + P.n_g_max + c];
        }
        L.gx[c] = x; L.gy[c] = y;
    }
    if (active) {
        const size_t b = (size_t)e * 2 * P.n_a;
        L.px[i] = p_g[b + i]; L.py[i] = p_g[b + P.n_a + i];
        L.vx[i] = dp_g[b + i]; L.vy[i] = dp_g[b + P.n_a + i];
    } else {                                                                 // unused slots: far away, so the chunked
        L.px[i] = (S)1e30; L.py[i] = (S)1e30; L.vx[i] = (S)0; L.vy[i] = (S)0;   // pair screens never select them
    }
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
    const int n_g = env_ok ? ng_g[e] : 0;
    const double l_cell = env_ok ? lcell_g[e] : 1.0;
    load_env<S>(P, L, env_ok, active, e, i, n_g, p_g, dp_g, grid_g);
    __syncthreads();
    DevParams Q = P; Q.n_g_e = n_g; Q.want_stats = 0;
    observe<S>(Q, L, active, e, i, l_cell, heading_g, nbr_g, inflag_g, nearest_g, obs_g, rew_g, sensed_g, occup_g,
               nullptr, 0);
}

// ENV:482-661: forces + prior on the old state, integration, observation + reward.
template <typename S>
__global__ void k_step(DevParams P, S *p_g, S *dp_g, const float *act_g, const S *grid_g, const int *ng_g,
                       const double *lcell_g, int *nbr_g, int *inflag_g, int *nearest_g, S *obs_g, S *rew_g,
                       S *prior_g, int *sensed_g, int *occup_g, double *stats_g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int es = threadIdx.x / P.SL, i = threadIdx.x % P.SL, e = blockIdx.x * P.EPB + es;
    const bool env_ok = e < P.E, active = env_ok && i < P.n_a;
    const int n_a = P.n_a;
    EnvLds<S> L = carve<S>(smem, P, es);
    STAMP(0);
    const int n_g = env_ok ? ng_g[e] : 0;
    const double l_cell = env_ok ? lcell_g[e] : 1.0;
    load_env<S>(P, L, env_ok, active, e, i, n_g, p_g, dp_g, grid_g);
    __syncthreads();
    STAMP(1);

    double npx = 0, npy = 0, nvx = 0, nvy = 0;
    int contacts = 0;
    if (active) {
        const double px = (double)L.px[i], py = (double)L.py[i], vx = (double)L.vx[i], vy = (double)L.vy[i];
        // ---- ball-ball spring forces: ENV:437-452 + CPP:740-820, row sum in ascending k ----
        double fx = 0.0, fy = 0.0;
        if (sizeof(S) == 4 && !P.periodic && P.SL >= 64) {                       // large swarm: 32 agents per screen
            if constexpr (sizeof(S) == 4) {
                const float pxf = L.px[i], pyf = L.py[i];
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
                if (!(__builtin_fmaf(fy32, fy32, fx32 * fx32) < P.Tcontact_hi_f) && !(P.want_stats && P.periodic && i == 0)) continue;
            }
            double dx = (double)L.px[k] - px, dy = (double)L.py[k] - py;         // CPP:782-785 (delta = p_k - p_i)
            double s = dx * dx + dy * dy;                                        // ENV:444 (no wrap for rows >= 1)
            if (s < P.T_contact) {                                               // ENV:445-446 d - sizes < 0
                double d = sqrt(s);
                double gap = fabs(d - P.sizes2);                                 // ENV:447
                fx += 1.0 * gap * P.k_ball * (-(dx / d));                        // CPP:790-793
                fy += 1.0 * gap * P.k_ball * (-(dy / d));
                if (k > i) ++contacts;
            } else if (P.want_stats && P.periodic && i == 0) {
                // ENV:442-443: only agent 0's row of the distance matrix is wrapped; it is
                // never read by the force loop but is counted by WRAP:131-145.
                double wx = dx, wy = dy; wrap_rel(wx, wy, P.half_w, P.half_h);
                if (wx * wx + wy * wy < P.T_contact) ++contacts;
            }
        }
        // ---- walls: CPP:822-860, ENV:510-513 ----
        double wfx = 0.0, wfy = 0.0, dfx = 0.0, dfy = 0.0;
        if (!P.periodic) {
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
            for (int q = 0; q < 6; ++q) nb[q] = nbr_g[((size_t)e * n_a + i) * 6 + q];
            double tx, ty;
            bool inf; int near;
            if (P.recompute) {
                double smin; cell_pass<S, false>(P, true, px, py, L.gx, L.gy, i, nullptr, nullptr, nullptr, smin, near);
                inf = sqrt(smin) < sqrt(2.0) * l_cell / 2;
            } else {
                inf = inflag_g[(size_t)e * n_a + i] != 0; near = nearest_g[(size_t)e * n_a + i];
            }
            if (inf) { tx = px; ty = py; } else { tx = (double)L.gx[near]; ty = (double)L.gy[near]; }  // CPP:896-901
            double f0, f1;
            prior_policy<S>(P, px, py, vx, vy, tx, ty, nb, L.px, L.py, L.vx, L.vy, f0, f1);
            prior_g[(size_t)e * 2 * n_a + i] = (S)f0; prior_g[(size_t)e * 2 * n_a + n_a + i] = (S)f1;
        }
        // ---- integration: ENV:626-647 ----
        const float sens = (float)P.sens;
        double ux = (double)(sens * act_g[(size_t)e * 2 * n_a + i]);             // ENV:627,633 (1 * float32)
        double uy = (double)(sens * act_g[(size_t)e * 2 * n_a + n_a + i]);
        double Fx = ux + fx, Fy = uy + fy;
        if (!P.periodic) { Fx = Fx + wfx + dfx; Fy = Fy + wfy + dfy; }           // ENV:633
        nvx = vx + (Fx / P.mass) * P.dt; nvy = vy + (Fy / P.mass) * P.dt;        // ENV:638-641
        nvx = fmin(fmax(nvx, -P.vel_max), P.vel_max);                            // ENV:642
        nvy = fmin(fmax(nvy, -P.vel_max), P.vel_max);
        npx = px + nvx * P.dt; npy = py + nvy * P.dt;                            // ENV:645
        if (P.periodic) {                                                        // ENV:646-647, 778-781
            if (npx < -P.half_w) npx += 2 * P.half_w;
            if (npx > P.half_w) npx -= 2 * P.half_w;
            if (npy < -P.half_h) npy += 2 * P.half_h;
            if (npy > P.half_h) npy -= 2 * P.half_h;
        }
    }
    STAMP(2);
    // contact pairs of the environment (upper triangle, WRAP:141-145)
    int contact_pairs = 0;
    if (P.want_stats) {
        __syncthreads();
        int *cnt = reinterpret_cast<int *>(L.occ);
        if (i == 0) cnt[0] = 0;
        __syncthreads();
        if (active && contacts) atomicAdd(&cnt[0], contacts);
        __syncthreads();
        contact_pairs = cnt[0];
    }
    __syncthreads();                                                             // all reads of the old state done
    if (active) {
        const S a = (S)npx, b = (S)npy, c = (S)nvx, d = (S)nvy;
        L.px[i] = a; L.py[i] = b; L.vx[i] = c; L.vy[i] = d;
        const size_t o = (size_t)e * 2 * n_a;
        p_g[o + i] = a; p_g[o + n_a + i] = b; dp_g[o + i] = c; dp_g[o + n_a + i] = d;
    }
    __syncthreads();
    STAMP(3);
    DevParams Q = P; Q.n_g_e = n_g;
    observe<S>(Q, L, active, e, i, l_cell, nullptr, nbr_g, inflag_g, nearest_g, obs_g, rew_g, sensed_g, occup_g,
               stats_g, contact_pairs);
}

// =====================================================================================
// k_step_fast<T>: the step kernel for the common shape of the problem — fp32 state, at most 32
// agents per environment, walls on (no periodic wrap), at most 1024 grid cells, n_g_max a
// multiple of 32.  Same results as k_step<float> (the tests compare them bit for bit); the
// difference is the schedule:
//   * a workgroup = T wavefronts (T = 1, 2, 4) working on TWO environments; lane l of every
//     wavefront is agent (l & 31) of environment (l >> 5), so all T waves see the same agents
//     and every "which wave does what" branch is wave-uniform;
//   * the agent x cell pass is split over the T waves by 32-cell words; the sensed-cell lists
//     are compacted into LDS by words (ranks from per-word popcounts) and the 80 observation
//     slots are then written slot-range by slot-range, every store a full row segment;
//   * pair loops (contacts, k-NN) screen all 32 candidates in fp32 into a bit mask and run the
//     exact double evaluation only on the set bits.
// T is chosen by the host so that a launch has enough waves to fill the chip.
struct FastLds {
    float *px, *py, *vx, *vy;      // [64]   slot = (env half << 5) | agent
    float *g;                      // x rows of the two environments at [0, 2 NGP), their y rows ASM_GY_OFF floats further on
                                   // (a compile-time distance: one address serves both coordinates of a cell); cells >= n_g hold 1e30
    unsigned *mask;                // [64][MW]     sensed-cell bit mask per agent
    unsigned *occ;                 // [2][2W]      per env: occupancy words, then coverage words
    uint4 *part;                   // [T][64]      per-wave partial (min key, runner-up key, word, ambiguity)
    unsigned *fin;                 // [64]         nearest cell | in_flag << 31
    float *rpart;                  // [T][3][64]   per-wave partial reward sums
    unsigned short *sel;           // [64][CAPP]   compacted (sub-sampled) sensed list
};

__device__ __forceinline__ FastLds carve_fast(unsigned char *smem, const DevParams &P, int T)
{
    FastLds L;
    float *f = reinterpret_cast<float *>(smem);
    L.px = f; L.py = f + 64; L.vx = f + 128; L.vy = f + 192;
    L.g = f + 256;
    L.mask = reinterpret_cast<unsigned *>(L.g + ASM_GY_OFF + 2 * P.NGP);
    L.occ = L.mask + 64 * P.MW;
    L.part = reinterpret_cast<uint4 *>(L.occ + 4 * P.W);
    L.fin = reinterpret_cast<unsigned *>(L.part + 64 * T);
    L.rpart = reinterpret_cast<float *>(L.fin + 64);
    L.sel = reinterpret_cast<unsigned short *>(L.rpart + 3 * 64 * T);
    return L;
}

// sum / min / max over the 32 lanes of an environment half
__device__ __forceinline__ double half_sum(double v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
    return v;
}
__device__ __forceinline__ double half_min(double v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 32));
    return v;
}
__device__ __forceinline__ double half_max(double v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 32));
    return v;
}

// psi(z) of the reward's weighted mean (CPP:1017-1025 with delta = 0), the reference's expression
__device__ __forceinline__ double reward_weight(const DevParams &P, double z)
{
    if (z < 0.0 * P.d_sen) return 1.0;
    if (z < P.d_sen) return (1.0 / 2.0) * (1.0 + cos(ASM_PI * (z / P.d_sen - 0.0) / (1.0 - 0.0)));
    return 0.0;
}

#ifndef ASM_W0_LESS
#define ASM_W0_LESS 2        // words wave 0 gives up in the cell pass because it also runs the k-NN pass
#endif
#ifndef ASM_FAST_MINW
#define ASM_FAST_MINW 1
#endif
template <int T>
__global__ __launch_bounds__(64 * T, ASM_FAST_MINW) void k_step_fast(DevParams P, float *p_g, float *dp_g, const float *act_g,
                                                      const float *grid_g, const int *ng_g, const double *lcell_g,
                                                      int *nbr_g, int *inflag_g, int *nearest_g, float *obs_g,
                                                      float *rew_g, float *prior_g, int *sensed_g, int *occup_g,
                                                      double *stats_g)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // sub is the same for every lane of a wave: read through the scalar unit so that everything derived from
    // it (word and slot ranges, row offsets) stays in scalar registers
    const int tid = threadIdx.x, sub = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, es = l >> 5, i = l & 31;
    const int e = blockIdx.x * 2 + es;
    const bool env_ok = e < P.E, active = env_ok && i < P.n_a;
    const int n_a = P.n_a, NGP = P.NGP, W = P.W, MW = P.MW, cap = P.n_obs;
    const FastLds L = carve_fast(smem, P, T);
    float *gxe = L.g + (size_t)es * NGP, *gye = gxe + ASM_GY_OFF;
    unsigned *occ_e = L.occ + es * 2 * W, *cov_e = occ_e + W;
    constexpr int SUB_PRIOR = T > 1 ? 1 : 0;
    STAMP(0);
#ifdef ASM_DIAG
    if (g_diag && l == 0) g_diag[(size_t)blockIdx.x * 32 + 20 + sub] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID of this wave
#endif

#ifdef ASM_EXITS
    const int xat = g_exit_at;
#else
    constexpr int xat = 0;
#endif
    // ---- P0: grid and state into LDS (all global loads issued before the first LDS write) ----
    const int e0 = blockIdx.x * 2;
    const int ng0 = e0 < P.E ? ng_g[e0] : 0, ng1 = e0 + 1 < P.E ? ng_g[e0 + 1] : 0;
    const int n_g = es ? ng1 : ng0;
    const double l_cell = env_ok ? lcell_g[e] : 1.0;
    {
        float sa = -1e30f, sb = -1e30f, sc = 0.f, sd = 0.f;          // unused lanes: far from every agent AND from the padding cells (+1e30)
        if (sub == 0 && active) {
            const size_t o = (size_t)e * 2 * n_a;
            sa = p_g[o + i]; sb = p_g[o + n_a + i]; sc = dp_g[o + i]; sd = dp_g[o + n_a + i];
        }
        // The four coordinate rows (2 environments x {x, y}) are dealt out to the waves, so the row and its n_g are
        // wave-uniform and a lane's float4 index is just l, l + 64, ...; cells beyond n_g become far-away padding.
        for (int row = sub; row < 4; row += T) {
            const int h = row >> 1, ngh = h ? ng1 : ng0;
            const bool row_ok = e0 + h < P.E;
            const float *src = grid_g + ((size_t)(e0 + h) * 2 + (row & 1)) * NGP;
            float *dst = L.g + (size_t)(row & 1) * ASM_GY_OFF + (size_t)h * NGP;
            for (int base = 0; base < NGP; base += 64 * 4 * 4) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = base + (q * 64 + l) * 4;
                    v[q] = make_float4(1e30f, 1e30f, 1e30f, 1e30f);
                    if (base + q * 256 < NGP && c < NGP && row_ok) v[q] = *reinterpret_cast<const float4 *>(src + c);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = base + (q * 64 + l) * 4;
                    if (base + q * 256 < NGP && c < NGP) {
                        float4 u = v[q];
                        if (c + 3 >= ngh) {                         // only the last cells of a row lie beyond n_g
                            if (c + 0 >= ngh) u.x = 1e30f;
                            if (c + 1 >= ngh) u.y = 1e30f;
                            if (c + 2 >= ngh) u.z = 1e30f;
                            u.w = 1e30f;
                        }
                        *reinterpret_cast<float4 *>(dst + c) = u;
                    }
                }
            }
        }
        for (int w = tid; w < 4 * W; w += 64 * T) L.occ[w] = 0;
        if (sub == 0) { L.px[l] = sa; L.py[l] = sb; L.vx[l] = sc; L.vy[l] = sd; }
    }
    __syncthreads();
    STAMP(1);
    EXIT_AT(1);

    int contacts = 0;
    if (!P.observe_only) {                                          // reset / observe: the state is taken as it is
    float npx = 1e30f, npy = 1e30f, nvx = 0.f, nvy = 0.f;
    // ---- P1: forces, prior and integration on the OLD state (ENV:486-647) -------------
    if (sub == 0 && active) {
        const float pxf = L.px[l], pyf = L.py[l];
        const double px = pxf, py = pyf, vx = L.vx[l], vy = L.vy[l];
        unsigned cc = 0;                                            // fp32 screen of the 32 slots
#pragma unroll
        for (int j4 = 0; j4 < 32; j4 += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(L.px + es * 32 + j4);
            const float4 Y = *reinterpret_cast<const float4 *>(L.py + es * 32 + j4);
            const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rx = xs[r] - pxf, ry = ys[r] - pyf;
                const float s = __builtin_fmaf(ry, ry, rx * rx);
                cc = __builtin_amdgcn_alignbit(cc, __float_as_uint(s - P.Tcontact_hi_f), 31);
            }
        }
        cc = __brev(cc) & ~(1u << i);
        double fx = 0.0, fy = 0.0;                                  // ENV:437-452 + CPP:740-820, ascending k
        while (cc) {
            const int k = __ffs(cc) - 1; cc &= cc - 1;
            const double dx = (double)L.px[es * 32 + k] - px, dy = (double)L.py[es * 32 + k] - py;   // CPP:782-785
            const double s = dx * dx + dy * dy;                     // ENV:444
            if (s < P.T_contact) {                                  // ENV:445-446
                const double d = sqrt(s);
                const double gap = fabs(d - P.sizes2);              // ENV:447
                fx += 1.0 * gap * P.k_ball * (-(dx / d));           // CPP:790-793
                fy += 1.0 * gap * P.k_ball * (-(dy / d));
                if (k > i) ++contacts;
            }
        }
        // walls: CPP:822-860, ENV:510-513
        const double r = P.size_a;
        const double d0 = px - r - (-P.half_w), d1 = P.half_h - (py + r);       // CPP:841-842
        const double d2 = P.half_w - (px + r), d3 = py - r - (-P.half_h);       // CPP:843-844
        const bool c0 = d0 < 0, c1 = d1 < 0, c2 = d2 < 0, c3 = d3 < 0;          // CPP:849
        const double a0 = c0 ? fabs(d0) : 0.0, a1 = c1 ? fabs(d1) : 0.0, a2 = c2 ? fabs(d2) : 0.0, a3 = c3 ? fabs(d3) : 0.0;
        const double wfx = (a0 - a2) * P.k_wall, wfy = (-a1 + a3) * P.k_wall;   // ENV:512
        const double dfx = (-(c0 ? vx : 0.0) + -(c2 ? vx : 0.0)) * P.c_wall;    // ENV:513
        const double dfy = (-(c1 ? vy : 0.0) + -(c3 ? vy : 0.0)) * P.c_wall;
        // integration: ENV:626-647
        const float sens = (float)P.sens;
        const double ux = (double)(sens * act_g[(size_t)e * 2 * n_a + i]);       // ENV:627,633
        const double uy = (double)(sens * act_g[(size_t)e * 2 * n_a + n_a + i]);
        const double Fx = ux + fx + wfx + dfx, Fy = uy + fy + wfy + dfy;         // ENV:633
        double ax = Fx, ay = Fy;
        if (P.mass != 1.0) { ax = Fx / P.mass; ay = Fy / P.mass; }               // ENV:638 (x / 1 == x)
        double v0 = vx + ax * P.dt, v1 = vy + ay * P.dt;                         // ENV:641
        v0 = fmin(fmax(v0, -P.vel_max), P.vel_max); v1 = fmin(fmax(v1, -P.vel_max), P.vel_max);   // ENV:642
        npx = (float)(px + v0 * P.dt); npy = (float)(py + v1 * P.dt);            // ENV:645
        nvx = (float)v0; nvy = (float)v1;
    }
    if (sub == SUB_PRIOR && active && P.prior && prior_g) {         // ENV:600-619: old state, previous neighbour list
        const double px = L.px[l], py = L.py[l], vx = L.vx[l], vy = L.vy[l];
        int nb[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) nb[q] = nbr_g[((size_t)e * n_a + i) * 6 + q];
        bool inf; int near;
        if (P.recompute) {
            double smin; cell_pass<float, false>(P, true, px, py, gxe, gye, l, nullptr, nullptr, nullptr, smin, near);
            inf = sqrt(smin) < sqrt(2.0) * l_cell / 2;
        } else {
            inf = inflag_g[(size_t)e * n_a + i] != 0; near = nearest_g[(size_t)e * n_a + i];
        }
        const double tx = inf ? px : (double)gxe[near], ty = inf ? py : (double)gye[near];      // CPP:896-901
        double f0, f1;
        prior_policy<float>(P, px, py, vx, vy, tx, ty, nb, L.px + es * 32, L.py + es * 32, L.vx + es * 32,
                            L.vy + es * 32, f0, f1);
        prior_g[(size_t)e * 2 * n_a + i] = (float)f0; prior_g[(size_t)e * 2 * n_a + n_a + i] = (float)f1;
    }
    STAMP(2);
    __syncthreads();                                                // all reads of the old state done
    if (sub == 0 && active) {
        L.px[l] = npx; L.py[l] = npy; L.vx[l] = nvx; L.vy[l] = nvy;
        const size_t o = (size_t)e * 2 * n_a;
        p_g[o + i] = npx; p_g[o + n_a + i] = npy; dp_g[o + i] = nvx; dp_g[o + n_a + i] = nvy;
    }
    __syncthreads();
    }
    STAMP(3);
    EXIT_AT(2);

    // ---- P3: agent x cell pass on the NEW state, words split over the waves ------------
    const float pxf = L.px[l], pyf = L.py[l];
    const double px = pxf, py = pyf;
    // obs[row * n_a + agent] of this lane's environment, written with buffer stores: resource = the block of the
    // workgroup's two environments, vector offset = the lane's column, scalar offset = the row; a row store then
    // needs no vector address arithmetic
    const __amdgpu_buffer_rsrc_t obs_rs =
        __builtin_amdgcn_make_buffer_rsrc(obs_g + (size_t)e0 * P.obs_dim * n_a, 0, 0x7fffffff, 0x00020000);
    const int olb = (es * P.obs_dim * n_a + i) * 4;
    auto ostore = [&](int row, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), obs_rs, olb, row * n_a * 4, 0);
    };
    double s6[6]; int j6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { s6[q] = __builtin_inf(); j6[q] = -1; }
    double nn_min = __builtin_inf();
    if (sub == 0 && active && xat != 13) {
        // k nearest neighbours inside d_sen (_get_focused, CPP:633-703): fp32 screen, exact on candidates
        unsigned cand = 0, n1 = 0xFFFFFFFFu, n2 = 0xFFFFFFFFu;
#pragma unroll
        for (int j4 = 0; j4 < 32; j4 += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(L.px + es * 32 + j4);
            const float4 Y = *reinterpret_cast<const float4 *>(L.py + es * 32 + j4);
            const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rx = xs[r] - pxf, ry = ys[r] - pyf;
                const float s = __builtin_fmaf(ry, ry, rx * rx);
                cand = __builtin_amdgcn_alignbit(cand, __float_as_uint(s - P.Tsen_hi_f), 31);
                const unsigned key = (__float_as_uint(s) & ~31u) | (unsigned)(j4 + r);   // two smallest: self (0)
                n2 = umed3(n1, n2, key); n1 = n1 < key ? n1 : key;                        // and the nearest neighbour
            }
        }
        cand = __brev(cand) & ~(1u << i);                           // self removed CPP:677-681
        // The candidates' single-precision keys (s32 with the low 5 bits replaced by the index) go through a
        // min/max insertion chain that keeps the 7 smallest in order.  The order of the first six -- among
        // themselves and against the seventh, hence against everything else -- is certain when neighbouring keys
        // differ by at least two units of 32 ulps (the screen is off by a few ulps), and a listed agent is
        // certainly inside d_sen when its key lies two units below T_sen.  Otherwise (ties, near-ties, agents at
        // the sensing range) the lane redoes its list in double below.
        const unsigned cand_all = cand;
        unsigned K[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) K[q] = 0xFFFFFFFFu;
        while (cand) {
            const int j = __builtin_ctz(cand); cand &= cand - 1;
            const float rx = L.px[es * 32 + j] - pxf, ry = L.py[es * 32 + j] - pyf;
            unsigned x = (__float_as_uint(__builtin_fmaf(ry, ry, rx * rx)) & ~31u) | (unsigned)j;
#pragma unroll
            for (int q = 0; q < 7; ++q) { const unsigned lo = K[q] < x ? K[q] : x; x = K[q] < x ? x : K[q]; K[q] = lo; }
        }
        const unsigned tlo = (__float_as_uint(P.Tsen_f) >> 5) - 2u;
        bool redo = false;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const bool there = K[q] != 0xFFFFFFFFu;
            const unsigned u = K[q] >> 5;
            redo |= there && (u >= tlo || (K[q + 1] >> 5) - u < 2u);
            j6[q] = there ? (int)(K[q] & 31u) : -1;
        }
        if (__ballot(redo)) {                                       // rare
            if (redo) {
#pragma unroll
                for (int q = 0; q < 6; ++q) j6[q] = -1;
                cand = cand_all;
                while (cand) {                                      // ascending j: ties keep the lower index
                    const int j = __ffs(cand) - 1; cand &= cand - 1;
                    const double dx = (double)L.px[es * 32 + j] - px, dy = (double)L.py[es * 32 + j] - py;
                    const double s = dx * dx + dy * dy;             // CPP:641
                    if (s < P.T_sen) {                              // CPP:663
                        double ks = s; int kj = j; bool placed = false;     // once placed, the entries behind move down one place
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            if (placed || ks < s6[q]) { const double ts = s6[q]; const int tj = j6[q]; s6[q] = ks; j6[q] = kj; ks = ts; kj = tj; placed = true; }
                    }
                }
            }
        }
        if (P.want_stats && n_a > 1) {
            const int nj = (int)((n1 & 31u) == (unsigned)i ? (n2 & 31u) : (n1 & 31u));
            const double dx = (double)L.px[es * 32 + nj] - px, dy = (double)L.py[es * 32 + nj] - py;
            nn_min = dx * dx + dy * dy;
        }
        // agent block, rows [pos_x,pos_y,vel_x,vel_y] per column (CPP:108-131); fl32(a - b) is the
        // correctly rounded difference, identical to rounding the double difference
        const float vxf = L.vx[l], vyf = L.vy[l];
        int r0 = 0;
        if (P.con_self) { ostore(0, pxf); ostore(1, pyf); ostore(2, vxf); ostore(3, vyf); r0 = 4; }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int j = j6[q];
            float rx = 0.f, ry = 0.f, wx = 0.f, wy = 0.f;
            if (j >= 0) {
                rx = L.px[es * 32 + j] - pxf; ry = L.py[es * 32 + j] - pyf;
                wx = L.vx[es * 32 + j] - vxf; wy = L.vy[es * 32 + j] - vyf;      // CPP:86
            }
            nbr_g[((size_t)e * n_a + i) * 6 + q] = j;               // CPP:103-105
            const int r = r0 + 4 * q;
            ostore(r, rx); ostore(r + 1, ry); ostore(r + 2, wx); ostore(r + 3, wy);
        }
    }
    STAMP(4);
    {
        // single-precision screening pass (see cell_pass_fast for the error argument).  Per word only the smallest
        // s32 is tracked (two cells per v_min3_u32); which cell it belongs to, and whether another cell is too
        // close to tell in single precision, is settled in P4 by screening the one word that holds it once more.
        unsigned G1 = 0xFFFFFFFFu, G2 = 0xFFFFFFFFu, G1w = 0, ambw = 0;   // ambw: words with an undecided sensed bit
        unsigned *wkey = reinterpret_cast<unsigned *>(L.sel);       // [64][MW], aliases the slot lists (written later)
        const f32x2 p2x = {pxf, pxf}, p2y = {pyf, pyf};
        const f32x2 thr = {P.Tsen_f, P.Thalf_f};
        // wave 0 also runs the k-NN pass above, so it takes only the first W0 words; the others share the rest
        const int W0 = T == 1 ? W : (W / T > ASM_W0_LESS ? W / T - ASM_W0_LESS : 0);
        for (int w = (T == 1 || sub == 0) ? 0 : W0 + sub - 1; w < ((T == 1 || sub != 0) ? W : W0); w += (T == 1 || sub == 0) ? 1 : T - 1) {
            unsigned k1 = 0xFFFFFFFFu, word = 0, ocand = 0;
            float amb = __builtin_inff(), ambo = __builtin_inff();
            const float4 *X = reinterpret_cast<const float4 *>(gxe + w * 32);
            const float4 *Y = reinterpret_cast<const float4 *>(gye + w * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x4 = X[q], y4 = Y[q];
                const f32x2 xa = {x4.x, x4.y}, xb = {x4.z, x4.w}, ya = {y4.x, y4.y}, yb = {y4.z, y4.w};
                const f32x2 rxa = xa - p2x, rxb = xb - p2x, rya = ya - p2y, ryb = yb - p2y;
                const f32x2 sa = __builtin_elementwise_fma(rya, rya, rxa * rxa);
                const f32x2 sb = __builtin_elementwise_fma(ryb, ryb, rxb * rxb);
                const float ss[4] = {sa.x, sa.y, sb.x, sb.y};
                k1 = umin3(k1, __float_as_uint(sa.x), __float_as_uint(sa.y));   // s >= 0: the bit patterns order like the values
                k1 = umin3(k1, __float_as_uint(sb.x), __float_as_uint(sb.y));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = ss[r];
                    const f32x2 s2 = {s, s};
                    const f32x2 d = s2 - thr;                                   // (s - T_sen, s - T_half)
                    word = __builtin_amdgcn_alignbit(word, __float_as_uint(d.x), 31);
                    ocand = __builtin_amdgcn_alignbit(ocand, __float_as_uint(d.y), 31);
                    amb = __builtin_fminf(amb, __builtin_fabsf(d.x));
                    ambo = __builtin_fminf(ambo, __builtin_fabsf(d.y));
                }
            }
            L.mask[(size_t)l * MW + w] = __brev(word);
            wkey[(size_t)l * MW + w] = k1;                           // smallest s32 of the word
            if (!(amb > P.Tsen_amb_f)) ambw |= 1u << w;
            settle_occupancy(P, active, __brev(ocand), ambo, px, py, gxe + w * 32, gye + w * 32, &occ_e[w], &cov_e[w]);
            if (k1 < G1) G1w = (unsigned)w;                         // strict: the first word holding the smallest s32
            G2 = umed3(G1, G2, k1);                                 // the two smallest word minima
            G1 = G1 < k1 ? G1 : k1;
        }
        L.part[sub * 64 + l] = make_uint4(G1, G2, G1w, ambw);
    }
    STAMP(5);
    __syncthreads();
    STAMP(12);
    EXIT_AT(3);

    // ---- P4 (wave 0): nearest cell, in-shape flag, target block -------------------------
    double dmin = 0.0;
    if (sub == 0) {
        unsigned G1 = 0xFFFFFFFFu, G2 = 0xFFFFFFFFu, G1w = 0, ambw = 0;
#pragma unroll
        for (int s = 0; s < T; ++s) {
            const uint4 q = L.part[s * 64 + l];
            if (q.x < G1) G1w = q.z;
            const unsigned t = G1 > q.x ? G1 : q.x, m = G2 < q.y ? G2 : q.y;
            G2 = m < t ? m : t;
            G1 = G1 < q.x ? G1 : q.x;
            ambw |= q.w;
        }
        int imin = 0;
        double smin = __builtin_inf();
        if (active) {
            // Candidates for the nearest cell: s32 within two units of 32 ulps of the smallest one (the screen's error
            // is a few ulps).  They lie in the word of the smallest s32 and, rarely, in other words whose minimum is
            // that close.  Those words are screened again (same arithmetic, so the same s32) and the candidates
            // evaluated in double; ascending index + strict '<' gives the reference's first minimum (CPP:889).
            const unsigned lo = G1 >> 5;
            unsigned minw = 1u << G1w;
            if (!((G2 >> 5) - lo >= 2u)) {
                const unsigned *wkey = reinterpret_cast<const unsigned *>(L.sel) + (size_t)l * MW;
                for (int w = 0; w < W; ++w) if ((wkey[w] >> 5) - lo < 2u) minw |= 1u << w;
            }
            const unsigned lim = (lo << 5) + 64u;                   // candidate <=> bits(s32) < lim
            const f32x2 p2x = {pxf, pxf}, p2y = {pyf, pyf};
            while (minw) {
                const int w = __ffs(minw) - 1; minw &= minw - 1;
                const float4 *X = reinterpret_cast<const float4 *>(gxe + w * 32);
                const float4 *Y = reinterpret_cast<const float4 *>(gye + w * 32);
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
                    const double dx = (double)gxe[c] - px, dy = (double)gye[c] - py;
                    const double sd = dx * dx + dy * dy;
                    if (sd < smin) { smin = sd; imin = c; }
                }
            }
            if (ambw) {
                // Rare: sensed bits within the margin of T_sen, decided in double
                unsigned *mw = L.mask + (size_t)l * MW;
                while (ambw) {
                    const int w = __ffs(ambw) - 1; ambw &= ambw - 1;
                    unsigned word = mw[w];
#pragma unroll 4
                    for (int b = 0; b < 32; ++b) {
                        const int c = w * 32 + b;
                        const float rx = gxe[c] - pxf, ry = gye[c] - pyf;
                        const float s32 = __builtin_fmaf(ry, ry, rx * rx);
                        if (__builtin_fabsf(s32 - P.Tsen_f) <= P.Tsen_amb_f) {
                            const double dx = (double)gxe[c] - px, dy = (double)gye[c] - py;
                            const double sd = dx * dx + dy * dy;
                            if (sd < P.T_sen) word |= 1u << b; else word &= ~(1u << b);
                        }
                    }
                    mw[w] = word;
                }
            }
            dmin = sqrt(smin);
            const bool in_flag = dmin < sqrt(2.0) * l_cell / 2;     // CPP:894
            L.fin[l] = (unsigned)imin | ((unsigned)contacts << 16) | (in_flag ? 0x80000000u : 0u);   // imin < 1024, contacts < 32
            inflag_g[(size_t)e * n_a + i] = in_flag ? 1 : 0;        // CPP:139
            nearest_g[(size_t)e * n_a + i] = imin;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;           // CPP:141-142, 896-901 (x - x == 0)
            if (!in_flag) { t0 = gxe[imin] - pxf; t1 = gye[imin] - pyf; t2 = 0.f - L.vx[l]; t3 = 0.f - L.vy[l]; }
            const int tb = P.obs_dim - (2 + cap) * 2;               // CPP:304
            ostore(tb + 0, t0); ostore(tb + 1, t1);
            ostore(tb + 2, t2); ostore(tb + 3, t3);
        } else {
            L.fin[l] = 0;
        }
        if (T > 1 && P.want_stats)                                  // wave 1 computes the formation statistics in P7
            L.part[l] = make_uint4(__double2loint(dmin), __double2hiint(dmin), __double2loint(nn_min), __double2hiint(nn_min));
    }
    STAMP(13);
    __syncthreads();
    STAMP(14);
    EXIT_AT(4);

    // ---- P5: compaction of the sensed lists into LDS -----------------------------------
    const unsigned fin = L.fin[l];
    const bool in_flag = (fin >> 31) != 0;
    const unsigned *mym = L.mask + (size_t)l * MW;
    const unsigned inm = in_flag ? 0xFFFFFFFFu : 0u;                // CPP:155: occupied cells drop out only inside the shape
    const int k0 = sub * cap / T, k1 = (sub + 1) * cap / T;         // this wave's observation slots (P6)
    // Unused lanes sit at -1e30 (P0), so their masks are empty.  The words whose first rank lies below k0 form a
    // prefix of nb words; this wave starts at the first non-empty word after them, whose first rank is bst.
    int n_after = 0, nb = 0, bst = 0;
    unsigned nz = 0;                                                // words with a cell that stays in the list
    for (int w = 0; w < W; ++w) {
        const unsigned f = mym[w] & ~(occ_e[w] & inm);
        const int pc = __popc(f);
        const bool below = n_after < k0;
        n_after += pc;
        nb += below ? 1 : 0;
        bst = below ? n_after : bst;
        nz |= (unsigned)(pc < 1 ? pc : 1) << w;
    }
    const unsigned nzs = nb >= 32 ? 0u : nz & (0xFFFFFFFFu << nb);
    const bool found = nzs != 0;
    const int wst = found ? __ffs(nzs) - 1 : 0;
    const int n_out = n_after < cap ? n_after : cap;
    unsigned short *mysel = L.sel + (size_t)l * P.capp;
    STAMP(16);
    [[maybe_unused]] int diag_it = 0;
    {
        // Slot k holds the set bit of rank t_k = round(k * (n-1)/(cap-1)) when n > cap (CPP:243-261), else
        // rank k.  One loop per lane over its own set bits; the wave iterates as often as its busiest lane.
        if (__ballot(n_after > cap) == 0) {
            // Common case, no lane of the workgroup sub-samples: slot == rank.  Wave `sub` lists the words
            // whose first rank lies in [k0, k1) (found above, so it starts without skipping anything).  The
            // lane's next QN non-empty words are fetched into registers in one batch, so the loop below runs
            // without waiting on LDS; a lane that needs more words finishes in the second loop.
            constexpr int QN = 6;
            bool busy = found && bst < k1;
            int rank = bst;
            nz &= 0xFFFFFFFFu << wst;
            unsigned fq[QN], wq = 0;
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const bool has = nz != 0;
                const int wi = has ? __ffs(nz) - 1 : 0;
                nz &= nz - 1;                                       // stays 0 once empty
                const unsigned f = mym[wi] & ~(occ_e[wi] & inm);
                fq[q] = has ? f : 0u;
                wq |= (unsigned)wi << (5 * q);
            }
            unsigned bits = fq[0];
            busy = busy && bits != 0;
            unsigned short *wp = mysel + rank, *const wend = mysel + k1;
            unsigned wbase = (wq & 31u) << 5;
            while (busy) {                                          // bits != 0 here; two cells of the word per trip
                const unsigned b = (unsigned)__builtin_ctz(bits); bits &= bits - 1;
                *wp = (unsigned short)(wbase | b);
                // the word's next cell, if it has one, goes to the next slot in the same trip (no branch: without
                // one the first cell is stored once more in place)
                const bool two = bits != 0;
                const unsigned b2 = (unsigned)(__ffs(bits) - 1); bits &= bits - 1;      // bits == 0 stays 0
                wp[two ? 1 : 0] = (unsigned short)(wbase | (two ? b2 : b));
                wp += two ? 2 : 1;
                ++diag_it;
                if (bits == 0) {
#pragma unroll
                    for (int q = 0; q + 1 < QN; ++q) fq[q] = fq[q + 1];
                    fq[QN - 1] = 0; wq >>= 5; wbase = (wq & 31u) << 5;
                    bits = fq[0];
                    if (bits == 0 || wp >= wend) busy = false;
                }
            }
            rank = (int)(wp - mysel);
            busy = found && rank < k1 && nz != 0;                   // rare: more than QN words in this wave's range
            bits = 0;
            int w = 0;
            while (busy) {
                if (bits == 0) {
                    if (nz == 0 || rank >= k1) { busy = false; continue; }
                    w = __ffs(nz) - 1; nz &= nz - 1;
                    bits = mym[w] & ~(occ_e[w] & inm);
                }
                const int b = __ffs(bits) - 1; bits &= bits - 1;
                mysel[rank++] = (unsigned short)(w * 32 + b);
            }
#ifdef ASM_DIAG
            for (int m = 32; m >= 1; m >>= 1) diag_it = max(diag_it, __shfl_xor(diag_it, m, 64));
            DIAG_COUNT(17, diag_it);
#endif
        } else {
            DIAG_COUNT(18, 1);
            // cap-1 is odd here, so k(n-1)/(cap-1) is never a half-integer and the reference's double evaluation
            // rounds like the exact rational: t_k = floor((2k(n-1) + (cap-1)) / (2(cap-1))), advanced incrementally
            // as quotient/remainder (increment exactly 1 for n <= cap).  Wave `sub` fills the slots [k0, k1).
            const unsigned two_c = 2u * (unsigned)(cap - 1);
            const unsigned num = n_after > cap ? 2u * (unsigned)(n_after - 1) : two_c;
            const unsigned inc_q = num / two_c, inc_r = num - inc_q * two_c;
            const unsigned n0 = (unsigned)k0 * num + (unsigned)(cap - 1);       // numerator of t_{k0}
            unsigned tq = n0 / two_c, tr = n0 - tq * two_c, bits = 0, wbase = 0;
            // t_{k0} >= k0, and the words before word nb - 1 end below rank k0: the walk starts at word nb - 1,
            // whose first rank is bst minus its own count
            const int w0 = nb > 0 ? nb - 1 : 0;
            int rank = nb > 0 ? bst - __popc(mym[w0] & ~(occ_e[w0] & inm)) : 0;
            nz &= 0xFFFFFFFFu << w0;
            unsigned short *wp = mysel + k0, *const wend = mysel + (k1 < n_out ? k1 : n_out);
            bool busy = active && k0 < n_out;
            while (busy) {
                if (bits == 0) {
                    if (nz == 0) { busy = false; continue; }
                    const int w = __builtin_ctz(nz); nz &= nz - 1;
                    bits = mym[w] & ~(occ_e[w] & inm);
                    wbase = (unsigned)w << 5;
                    const int pc = __popc(bits);
                    if ((unsigned)(rank + pc) <= tq) { rank += pc; bits = 0; continue; }   // the whole word ranks below the next slot
                    for (int d = (int)tq - rank; d > 0; --d) bits &= bits - 1;             // so do its first tq - rank cells
                    rank = rank < (int)tq ? (int)tq : rank;
                }
                const unsigned b = (unsigned)__builtin_ctz(bits); bits &= bits - 1;
                if ((unsigned)rank == tq) {
                    *wp++ = (unsigned short)(wbase | b);
                    tq += inc_q; tr += inc_r;
                    if (tr >= two_c) { tr -= two_c; ++tq; }
                    if (wp >= wend) busy = false;
                }
                ++rank;
            }
        }
    }
    if (sub == 0 && active && occup_g) {                            // occupied list (CPP:215-238), diagnostics only
        int *dst = occup_g + ((size_t)e * n_a + i) * P.n_occ;
        int n_occ_list = 0;
        for (int w = 0; w < W; ++w) n_occ_list += __popc(mym[w] & occ_e[w]);
        BitIter it; it.init(mym, W, occ_e, false);
        emit_list(it, in_flag ? n_occ_list : 0, P.n_occ, [&](int k, int cell) { dst[k] = cell; });
    }
    STAMP(6);
    if (T > 1) __syncthreads();                                     // the lists are complete
    STAMP(15);
    EXIT_AT(5);

    // ---- P6: the observation's cell block (CPP:279-296) + reward weights, slots split over the waves ----
    const bool rew_on = active && in_flag && n_after > 0;
    float fn0 = 0.f, fn1 = 0.f, fden = 0.f;
    {
        const int gb = P.obs_dim - cap * 2;
        int *sdst = sensed_g ? sensed_g + ((size_t)e * n_a + i) * cap : nullptr;
        constexpr int RB = 10;                                      // slots per round: indices, coordinates, then math + stores
        if (active) {
            for (int kb = k0; kb < k1; kb += RB) {
                int cell[RB]; float cx[RB], cy[RB]; bool have[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) cell[q] = mysel[kb + q];   // one batch of LDS reads; entries past k1 (the list is padded by RB) are masked below
#pragma unroll
                for (int q = 0; q < RB; ++q) { have[q] = kb + q < n_out && kb + q < k1; cell[q] = have[q] ? cell[q] : 0; }
#pragma unroll
                for (int q = 0; q < RB; ++q) { cx[q] = gxe[cell[q]]; cy[q] = gye[cell[q]]; }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int k = kb + q;
                    if (k < k1) {
                        const float gx = have[q] ? cx[q] - pxf : 0.f, gy = have[q] ? cy[q] - pyf : 0.f;   // CPP:285-286
#ifndef ASM_EXP_NOSTORE
                        ostore(gb + 2 * k, gx); ostore(gb + 2 * k + 1, gy);
#endif
                        if (sdst) sdst[k] = have[q] ? cell[q] : -1;
                        // psi = (1 + cos(pi z / d_sen)) / 2 (CPP:1017-1025); v_cos_f32 takes revolutions
                        const float z = __builtin_amdgcn_sqrtf(__builtin_fmaf(gy, gy, gx * gx));
                        const float psi = have[q] ? __builtin_fmaf(0.5f, __builtin_amdgcn_cosf(z * P.inv2dsen_f), 0.5f) : 0.f;
                        fn0 = __builtin_fmaf(psi, gx, fn0); fn1 = __builtin_fmaf(psi, gy, fn1); fden += psi;
                    }
                }
            }
        }
        if (T > 1) { L.rpart[(sub * 3 + 0) * 64 + l] = fn0; L.rpart[(sub * 3 + 1) * 64 + l] = fn1; L.rpart[(sub * 3 + 2) * 64 + l] = fden; }
    }
    STAMP(7);
    if (T > 1) __syncthreads();
    EXIT_AT(6);

    // ---- P7 (wave 0): reward (CPP:457-564) and per-environment statistics ---------------
    if (sub == 0) {
        double reward = 0.0;
        bool collision = false, uniform = false, known = true;      // known: the reward's uniformity test is decided
        if (active) {
            if (T > 1) {
                fn0 = 0.f; fn1 = 0.f; fden = 0.f;
#pragma unroll
                for (int s = 0; s < T; ++s) { fn0 += L.rpart[(s * 3 + 0) * 64 + l]; fn1 += L.rpart[(s * 3 + 1) * 64 + l]; fden += L.rpart[(s * 3 + 2) * 64 + l]; }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {                           // CPP:464-496
                const int j = j6[q];
                if (j >= 0) {
                    const double rx = (double)L.px[es * 32 + j] - px, ry = (double)L.py[es * 32 + j] - py;
                    if (rx * rx + ry * ry < P.T_ravoid) collision = true;   // CPP:487
                }
            }
            if (rew_on) {
                known = false;
                if (fden > 0.f) {
                    // Single-precision screen.  The weights psi carry an absolute error below 1e-6 (argument
                    // rounding 2*pi*1.5e-7, v_cos_f32 below 5e-7, halved, plus one rounding), every term and
                    // every accumulation step a relative rounding of 6e-8, so with n terms and |g| < d_sen
                    // |v32 - v| < 0.64 n (1e-6 / den + 6e-8) + 1e-7 for the norm v of the mean offset; doubled here.
                    const float rden = 1.0f / fden;
                    const float v0 = fn0 * rden, v1 = fn1 * rden;
                    const float vn = __builtin_amdgcn_sqrtf(__builtin_fmaf(v1, v1, v0 * v0));
                    const float margin = 1.28f * (float)n_out * __builtin_fmaf(1e-6f, rden, 6e-8f) + 2e-7f;
                    if (__builtin_fabsf(vn - 0.05f) > margin) { uniform = vn < 0.05f; known = true; }
                }
            }
        }
        // Borderline agents: the reference's double-precision sums (CPP:515-552), one agent at a time with the
        // 64 lanes sharing its slots.  The terms are the reference's; only their order of summation differs
        // (relative error below 80 * 2^-53), so a result within 1e-11 of the threshold falls through to the
        // sequential evaluation in the reference's order.
        for (unsigned long long todo = __ballot(!known); todo; todo &= todo - 1) {
            const int src = __ffsll(todo) - 1;
            const int n_src = __builtin_amdgcn_readlane(n_out, src);
            const double sx = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pxf), src));
            const double sy = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pyf), src));
            const unsigned short *ssel = L.sel + (size_t)src * P.capp;
            const float *sgx = L.g + (size_t)(src >> 5) * NGP, *sgy = sgx + ASM_GY_OFF;
            double num0 = 0.0, num1 = 0.0, den = 0.0;
            for (int k = l; k < n_src; k += 64) {
                const int cell = ssel[k];
                const double gx = (double)sgx[cell] - sx, gy = (double)sgy[cell] - sy;           // CPP:515-516
                const double psi = reward_weight(P, sqrt(gx * gx + gy * gy));                     // CPP:524
                num0 += psi * gx; num1 += psi * gy; den += psi;                                  // CPP:537-539
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                num0 += __shfl_xor(num0, m, 64); num1 += __shfl_xor(num1, m, 64); den += __shfl_xor(den, m, 64);
            }
            if (l == src) {
                if (den == 0) den = 1E-8;                           // CPP:542-544
                const double v0 = 1.0 * num0 / den, v1 = 1.0 * num1 / den;                       // CPP:547-548
                const double q = v0 * v0 + v1 * v1;
                if (fabs(q - P.T_v005) > 1e-11 * P.T_v005) {
                    uniform = q < P.T_v005;                         // CPP:550-552
                } else {
                    num0 = 0.0; num1 = 0.0; den = 0.0;
                    for (int k = 0; k < n_out; ++k) {
                        const int cell = mysel[k];
                        const double gx = (double)gxe[cell] - px, gy = (double)gye[cell] - py;
                        const double psi = reward_weight(P, sqrt(gx * gx + gy * gy));
                        num0 += psi * gx; num1 += psi * gy; den += psi;
                    }
                    if (den == 0) den = 1E-8;
                    const double w0 = 1.0 * num0 / den, w1 = 1.0 * num1 / den;
                    uniform = w0 * w0 + w1 * w1 < P.T_v005;
                }
            }
        }
        if (active) {
            if (in_flag && !collision && uniform) reward += 1.0;    // CPP:559-561
            if (P.want_reward) rew_g[(size_t)e * n_a + i] = (float)reward;
        }
        STAMP(8);
        if (P.want_stats) {                                         // TRAIN:128-131: reward sums, agents inside
            const unsigned long long half = es ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
            const double rs = (double)__popcll(__ballot(active && reward > 0.0) & half), rq = rs;     // rewards are 0 or 1
            const double inn = (double)__popcll(__ballot(active && in_flag) & half);
            if (i == 0 && env_ok) {
                double *st = stats_g + (size_t)e * 8;
                st[0] += rs; st[1] += rq; st[7] += inn;
            }
        }
    }
    // Formation statistics (TRAIN:132-135; WRAP:48-165): they do not depend on the reward, so with several waves the
    // second one computes them while the first evaluates the reward (its inputs were handed over through LDS in P4).
    if (sub == SUB_PRIOR && P.want_stats) {
        double dm = dmin, nn = nn_min; int cn = contacts;
        if (T > 1) {
            const uint4 q = L.part[l];
            dm = __hiloint2double((int)q.y, (int)q.x); nn = __hiloint2double((int)q.w, (int)q.z);
            cn = (int)((fin >> 16) & 0xFFu);
        }
        const double m = active ? sqrt(nn) : 0.0;
        int ci = cn + (i < W ? __popc(cov_e[i]) << 16 : 0);                                           // two small sums in one word
#pragma unroll
        for (int mm = 16; mm >= 1; mm >>= 1) ci += __shfl_xor(ci, mm, 32);
        const double cp = (double)(ci & 0xFFFF), covered = (double)(ci >> 16);
        const double ds = half_sum(active ? dm : 0.0), ms = half_sum(m);
        const double lo = half_min(active ? m : __builtin_inf()), hi = half_max(active ? m : -__builtin_inf());
        const double mean = ms / n_a, dv = active ? m - mean : 0.0;
        const double var = half_sum(dv * dv) / n_a;
        if (i == 0 && env_ok) {
            double *st = stats_g + (size_t)e * 8;
            st[2] += covered / n_g;
            st[3] += (var - lo) / (hi - lo);                        // WRAP:98-99
            st[4] += cp; st[5] += ds / n_a; st[6] += 1.0;
        }
    }
    STAMP(9);
}

// ENV:151-218 with counter-based draws (draw order: shape, angle, 2 discarded, offset
// x/y, coin, 2*n_a positions, 2 centre, 2*n_a velocities), then the observation.
template <typename S>
__global__ void k_reset(DevParams P, uint64_t seed, uint32_t episode, int64_t env0, int n_shapes,
                        const double *shape_grids, const int *shape_off, const int *shape_ng,
                        const double *shape_lcell, int *shape_id_g, S *p_g, S *dp_g, S *grid_g, int *ng_g,
                        double *lcell_g, int *nbr_g, int *inflag_g, int *nearest_g, S *obs_g, int *sensed_g,
                        int *occup_g)
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
        if (i == 0) { shape_id_g[e] = s; ng_g[e] = n_g; lcell_g[e] = l_cell; }
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
    observe<S>(Q, L, active, e, i, l_cell, nullptr, nbr_g, inflag_g, nearest_g, obs_g, nullptr, sensed_g, occup_g,
               nullptr, 0);
}

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
__global__ void k_replay_draw(uint64_t seed, uint32_t counter, int64_t lo, int64_t hi, int N, int64_t *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
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
    const int e = blockIdx.x, tid = threadIdx.x, n_g = ng_g[e];
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
        out[(size_t)e * 5 + 0] = (double)covered / n_g;             // WRAP:70
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

template <int T>
static int launch_step_fast_t(const DevParams &P, float *p, float *dp, const float *act, const float *grid,
                              const int *n_g, const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs,
                              float *rew, float *prior, int *sensed, int *occupied, double *stats, hipStream_t st)
{
    if (int rc = set_lds(k_step_fast<T>, P.fast_lds_bytes)) return rc;
    hipLaunchKernelGGL(k_step_fast<T>, dim3((P.E + 1) / 2), dim3(64 * T), P.fast_lds_bytes, st, P, p, dp, act, grid, n_g,
                       l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats);
    LAUNCH_CHECK();
    return 0;
}

int launch_step_fast(const DevParams &P, float *p, float *dp, const float *act, const float *grid, const int *n_g,
                     const double *l_cell, int *nbr, int *inflags, int *nearest, float *obs, float *rew, float *prior,
                     int *sensed, int *occupied, double *stats, hipStream_t st)
{
#ifdef ASM_EXITS
    { const char *x = std::getenv("MARL_AMD_DBG_EXIT"); const int v = x ? std::atoi(x) : 0;
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_exit_at), &v, sizeof v) != hipSuccess) return -1; }
#endif
    switch (P.fast_T) {
    case 1: return launch_step_fast_t<1>(P, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    case 2: return launch_step_fast_t<2>(P, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    default: return launch_step_fast_t<4>(P, p, dp, act, grid, n_g, l_cell, nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats, st);
    }
}

template <typename S>
int launch_step(const DevParams &P, S *p, S *dp, const float *act, const S *grid, const int *n_g,
                const double *l_cell, int *nbr, int *inflags, int *nearest, S *obs, S *rew, S *prior, int *sensed,
                int *occupied, double *stats, hipStream_t st)
{
    if (int rc = set_lds(k_step<S>, P.lds_block_bytes)) return rc;
    hipLaunchKernelGGL(k_step<S>, dim3(P.grid), dim3(P.block), P.lds_block_bytes, st, P, p, dp, act, grid, n_g, l_cell,
                       nbr, inflags, nearest, obs, rew, prior, sensed, occupied, stats);
    LAUNCH_CHECK();
    return 0;
}

template <typename S>
int launch_reset(const DevParams &P, uint64_t seed, uint32_t episode, int64_t env0, int n_shapes,
                 const double *shape_grids, const int *shape_off, const int *shape_ng, const double *shape_lcell,
                 int *shape_id, S *p, S *dp, S *grid, int *n_g, double *l_cell, int *nbr, int *inflags, int *nearest,
                 S *obs, int *sensed, int *occupied, hipStream_t st)
{
    if (int rc = set_lds(k_reset<S>, P.lds_block_bytes)) return rc;
    hipLaunchKernelGGL(k_reset<S>, dim3(P.grid), dim3(P.block), P.lds_block_bytes, st, P, seed, episode, env0,
                       n_shapes, shape_grids, shape_off, shape_ng, shape_lcell, shape_id, p, dp, grid, n_g, l_cell, nbr,
                       inflags, nearest, obs, sensed, occupied);
    LAUNCH_CHECK();
    return 0;
}

template int launch_observe<float>(const DevParams &, const float *, const float *, const float *, const float *,
                                   const int *, const double *, int *, int *, int *, float *, float *, int *, int *,
                                   hipStream_t);
template int launch_observe<double>(const DevParams &, const double *, const double *, const double *, const double *,
                                    const int *, const double *, int *, int *, int *, double *, double *, int *, int *,
                                    hipStream_t);
template int launch_step<float>(const DevParams &, float *, float *, const float *, const float *, const int *,
                                const double *, int *, int *, int *, float *, float *, float *, int *, int *, double *,
                                hipStream_t);
template int launch_reset<float>(const DevParams &, uint64_t, uint32_t, int64_t, int, const double *, const int *,
                                 const int *, const double *, int *, float *, float *, float *, int *, double *, int *,
                                 int *, int *, float *, int *, int *, hipStream_t);

int launch_random_actions(uint64_t seed, uint32_t step, int64_t env0, int E, int n_a, float *act, hipStream_t st)
{
    const size_t n = (size_t)E * 2 * n_a;
    hipLaunchKernelGGL(k_random_actions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, step, env0, E, n_a, act);
    LAUNCH_CHECK();
    return 0;
}

int launch_replay_draw(uint64_t seed, uint32_t counter, int64_t lo, int64_t hi, int N, int64_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_replay_draw, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, seed, counter, lo, hi, N, out);
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

#ifdef ASM_DIAG
int diag_set_buffer(unsigned long long *dev)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &dev, sizeof dev);
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
