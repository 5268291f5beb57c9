// C ABI of libAssemblyEnv.so (declared in include/marl_amd_env.h): argument
// checking, launch geometry, exact squared-distance thresholds, and the five
// reference-compatible host-pointer entry points.  All arithmetic of the path runs
// in the HIP kernels (asm_kernels.hip); nothing here computes environment results
// on the CPU.
#include "../../include/marl_amd_env.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>

#include "asm_kernels.h"

using asmenv::DevParams;

namespace {

thread_local char g_err[512];                                // fixed storage: reporting an error allocates nothing

int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}

// min{ s >= 0 : fl(sqrt(s)) >= t }:  fl(sqrt(s)) < t  <=>  s < thr_lt(t)
double thr_lt(double t)
{
    if (!(t > 0)) return 0.0;
    double x = t * t;
    while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
    for (;;) {
        double y = std::nextafter(x, 0.0);
        if (y < x && std::sqrt(y) >= t) x = y; else break;
    }
    return x;
}
// min{ s >= 0 : fl(sqrt(s)) > t }:  !(fl(sqrt(s)) > t)  <=>  s < thr_le(t)
double thr_le(double t)
{
    if (t < 0) return 0.0;
    double x = t * t;
    while (std::sqrt(x) <= t) x = std::nextafter(x, INFINITY);
    for (;;) {
        double y = std::nextafter(x, 0.0);
        if (y < x && std::sqrt(y) > t) x = y; else break;
    }
    return x;
}

int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// MARL_AMD_FAST_T (tuning / test knob: waves per workgroup of k_step_fast, 0 = generic kernel), read once per process
// asm_set_step_schedule() overrides it at run time (tests compare the schedules within one process).
// Bit 4 of the value (+16): the large-swarm pair passes without their cell lists; bit 5 (+32): the step of a large
// swarm as three launches with the observation split over 256-agent workgroups (measured slower than the single
// launch at every batch size tried, so it is off unless asked for).
std::atomic<int> g_schedule_override{-1};
int schedule_setting()
{
    static const int v = [] { const char *ov = std::getenv("MARL_AMD_FAST_T"); return ov ? std::atoi(ov) : 4; }();
    const int o = g_schedule_override.load(std::memory_order_relaxed);
    return o >= 0 ? o : v;
}
int fast_t_setting() { return schedule_setting() & 15; }
// Observation rows are stored non-temporal by the multi-wave kernel's lattice path: stored the ordinary way they push the
// grids and the state out of the last-level cache between two steps (4096 environments of 30 agents in three shards:
// 31.5 against 30.8 us per step; 8192: 63.1 against 59.6; 16384 in one launch: 165 against 139).  A consumer that reads
// them straight back gains little from the cache (the float32 rollout's transpose kernel: no difference; the bf16 policy
// kernel 1 % faster with cached rows).  Bit 7 of the schedule setting (+128) stores them the ordinary way, bit 6 (+64) is
// accepted for compatibility (it was "force on").  (The large-swarm kernel always stores its cell block non-temporal.)
int obs_nt_for(const DevParams &P)
{
    (void)P;
    return (schedule_setting() & 128) ? 0 : 1;
}
bool cell_lists_off() { return (schedule_setting() & 16) != 0; }
bool split_on() { return (schedule_setting() & 32) != 0; }

// Launch geometry + thresholds.  state_bytes = sizeof(S).
int derive_dev_params(const asm_params *prm, int E, size_t state_bytes, DevParams *out);

// The derivation walks double-precision neighbours for the exact thresholds; a run steps with the same few
// (parameters, E) combinations over and over, so the last results are kept per thread.
int make_dev_params(const asm_params *prm, int E, size_t state_bytes, DevParams *out)
{
    struct Slot { asm_params prm; int E; size_t sb; int sched; DevParams P; bool used; };
    constexpr int N = 8;
    thread_local Slot cache[N];
    thread_local int next = 0;
    static const bool no_cache = std::getenv("MARL_AMD_NO_PARAM_CACHE") != nullptr;   // measurement knob (tools/host_launch_cost.py)
    if (prm && !no_cache)
        for (int k = 0; k < N; ++k)
            if (cache[k].used && cache[k].E == E && cache[k].sb == state_bytes && cache[k].sched == schedule_setting() &&
                !std::memcmp(&cache[k].prm, prm, sizeof *prm)) {
                *out = cache[k].P;
                return 0;
            }
    if (int rc = derive_dev_params(prm, E, state_bytes, out)) return rc;
    Slot &s = cache[next]; next = (next + 1) % N;
    std::memcpy(&s.prm, prm, sizeof *prm); s.E = E; s.sb = state_bytes; s.sched = schedule_setting(); s.P = *out; s.used = true;
    return 0;
}

// LDS layout of the generic kernels for P->rows per-agent rows per environment (SL, or fewer for a part workgroup),
// and the geometry of the large swarms' cell lists, which live in the per-agent mask area (idle until the cell
// pass): bin counters / starts (one word per bin + 1, + 16 wave totals), then position pairs and 16-bit indices in bin
// order.  Needs SL, EPB, NGP, W, MW, rows, periodic, half_w, half_h, sizes2, d_sen.  False: does not fit the LDS.
bool swarm_layout(DevParams *Pp, size_t state_bytes)
{
    DevParams &P = *Pp;
    size_t env_bytes = state_bytes * ((size_t)P.state_rows * P.SL + 2 * (size_t)P.NGP) + 4 * ((size_t)P.rows * P.MW + 2 * (size_t)P.W);
    if (P.parts <= 1) env_bytes += 2 * (size_t)P.rows * P.MW;            // + 16-bit word keys of the large-swarm cell pass (all-cells path)
    env_bytes = (env_bytes + 15) / 16 * 16 + 16;                         // + the step's contact counter
    P.lds_env_bytes = (unsigned)env_bytes;
    P.lds_block_bytes = (unsigned)(env_bytes * P.EPB);
    if (P.lds_block_bytes > 160 * 1024) return false;
    P.bins = 0;
    if (state_bytes == 4 && !P.periodic && P.EPB == 1 && P.SL >= 256 && !cell_lists_off()) {
        const size_t region = 4 * (size_t)P.rows * P.MW;
        auto geom = [&](double radius, int max_reach, asmenv::BinGeom *G) {
            const double r = radius * 1.001;                        // covers the screens' inflated thresholds and the rounding of the bin coordinates
            double h = r / max_reach;
            for (int it = 0; it < 400; ++it, h *= 1.05) {
                const double nx = std::ceil(2 * P.half_w / h), ny = std::ceil(2 * P.half_h / h);
                if (!(nx >= 1 && ny >= 1) || nx * ny > 60000) continue;
                const size_t need = 4 * ((size_t)(nx * ny) + 1 + 16 + 1) + 10 * (size_t)P.SL;
                if (need > region) continue;
                G->nbx = (int)nx; G->nby = (int)ny; G->inv_h = (float)(1.0 / h);
                G->reach = (int)std::ceil(r / h);
                return true;
            }
            return false;
        };
        static const int reach_s = [] { const char *v = std::getenv("MARL_AMD_BIN_REACH"); const int r = v ? std::atoi(v) : 3; return r < 1 ? 1 : (r > 8 ? 8 : r); }();
        if (P.half_w > 0 && P.half_h > 0 && P.sizes2 > 0 && P.d_sen > 0 && geom(P.sizes2, 1, &P.bin_c) && geom(P.d_sen, reach_s, &P.bin_s)) {
            P.bins = 1; P.bin_x0 = (float)P.half_w; P.bin_y0 = (float)P.half_h;
        }
    }
    return true;
}

int derive_dev_params(const asm_params *prm, int E, size_t state_bytes, DevParams *out)
{
    if (!prm) return fail(1, "asm_params is NULL");
    if (E < 1) return fail(1, "E = %d: need at least one environment", E);
    if (prm->n_a < 1 || prm->n_a > 1024) return fail(1, "n_a = %d: supported range is 1..1024", prm->n_a);
    if (prm->topo_nei_max != 6) return fail(1, "topo_nei_max = %d: only 6 (ENV:33) is supported", prm->topo_nei_max);
    if (prm->n_g_max < 1) return fail(1, "n_g_max = %d", prm->n_g_max);
    if (prm->num_obs_grid_max < 2 || prm->num_occupied_grid_max < 2)
        return fail(1, "num_obs_grid_max / num_occupied_grid_max must be >= 2");
    DevParams P;
    std::memset(&P, 0, sizeof P);
    P.E = E; P.n_a = prm->n_a; P.n_g_max = prm->n_g_max;
    P.SL = pow2_ceil(prm->n_a < 8 ? 8 : prm->n_a);
    if (P.SL < 64) { P.block = 64; P.EPB = 64 / P.SL; } else { P.block = P.SL; P.EPB = 1; }
    P.grid = (E + P.EPB - 1) / P.EPB;
    P.NGP = (prm->n_g_max + 31) / 32 * 32;
    P.W = P.NGP / 32;
    P.MW = (P.W | 1) < 9 ? 9 : (P.W | 1);
    P.rows = P.SL; P.state_rows = 4;
    P.half_w = prm->half_w; P.half_h = prm->half_h; P.sizes2 = prm->size_a + prm->size_a; P.d_sen = prm->d_sen;
    P.periodic = prm->is_boundary ? 0 : 1;
    if (!swarm_layout(&P, state_bytes))
        return fail(1, "n_a = %d with n_g_max = %d needs %u bytes of LDS per workgroup (> 160 KiB)", prm->n_a,
                    prm->n_g_max, P.lds_block_bytes);
    P.topo = 6; P.n_obs = prm->num_obs_grid_max; P.n_occ = prm->num_occupied_grid_max;
    P.con_self = prm->is_con_self_state != 0; P.cartesian = 1;
    P.obs_dim = 2 * 2 * (6 + 1 + (P.con_self ? 1 : 0)) + 2 * P.n_obs;      // ENV:806
    P.periodic = prm->is_boundary ? 0 : 1;
    P.prior = prm->compute_prior != 0;
    P.d_sen = prm->d_sen; P.r_avoid = prm->r_avoid; P.size_a = prm->size_a; P.sizes2 = prm->size_a + prm->size_a;
    P.vel_max = prm->vel_max; P.k_ball = prm->k_ball; P.k_wall = prm->k_wall; P.c_wall = prm->c_wall;
    P.dt = prm->dt; P.sens = prm->sensitivity; P.mass = prm->mass; P.half_w = prm->half_w; P.half_h = prm->half_h;
    P.T_sen = thr_lt(P.d_sen);
    P.T_half_le = thr_le(P.r_avoid / 2.0);
    P.T_half_lt = thr_lt(P.r_avoid / 2);
    P.T_ravoid = thr_lt(P.r_avoid);
    P.T_contact = thr_lt(P.sizes2);
    P.T_v005 = thr_lt(0.05);
    // fp32 screens: the fp32 evaluation of a squared distance (two rounded differences, a product, an fma)
    // is within 4 * 2^-24 = 2.4e-7 relative of the double one, the fp32 threshold within 2^-24 = 6e-8 of
    // the double threshold; EPS = 5e-7 covers both with margin
    const double EPS = 5e-7;
    P.Tsen_f = (float)P.T_sen;
    P.Tsen_amb_f = (float)(EPS * P.T_sen);
    P.Tsen_hi_f = (float)(P.T_sen * (1.0 + EPS));
    P.Thalf_hi_f = (float)(P.T_half_le * (1.0 + EPS));
    P.Thalf_f = (float)P.T_half_le;                                 // decided in fp32 unless within Thalf_amb_f of it
    P.Thalf_amb_f = (float)(EPS * P.T_half_le);
    P.Tcontact_hi_f = (float)(P.T_contact * (1.0 + EPS));
    P.inv2dsen_f = (float)(0.5 / P.d_sen);
    // k_step_fast eligibility and geometry (fp32 state only; see asm_kernels.hip)
    P.capp = P.n_obs + (P.n_obs & 1);                               // even, and an odd number of 32-bit words per
    if (((P.capp / 2) & 1) == 0) P.capp += 2;                       // agent: no LDS bank conflicts across lanes
    P.fast_T = 0;
    if (state_bytes == 4 && prm->n_a <= 32 && P.W <= 32 && prm->n_g_max % 32 == 0 && P.n_obs <= 4096 && ((P.n_obs - 1) & 1)) {
        const int blocks = (E + 1) / 2;
        (void)blocks;
        int T = fast_t_setting();                                    // 4: fastest measured schedule at every batch size from 2048 to 32768 environments
        if (P.periodic && T != 4) T = 0;                            // the periodic arena is built for 4 waves only
        if (T == 1 || T == 2 || T == 4) {
            // state 256 floats; grid: x rows of both environments, y rows ASM_GY_OFF floats further on, 64 own-position entries; masks; occupancy;
            // partials; flags; reward partials; slot lists (+ one emission round of padding) aliased with the word keys
            const size_t common = 4 * (256 + ASM_GY_OFF + 2 * (size_t)P.NGP + 64 + 64 * (size_t)P.MW + 4 * (size_t)P.W);
            const size_t bytes = common + 16 * 64 * (size_t)T + 4 * 128 + 4 * 3 * 64 * (size_t)T +
                                 std::max(2 * 64 * (size_t)P.capp + 32, 4 * 64 * (size_t)P.MW);
            // lattice path (T = 4): no partial rows, four ballot words, the reward partials in the shape tables' area,
            // the compact boundary coordinates aliased with the slot lists
            P.fast_lds_bytes_lat = (unsigned)(common + 4 * 128 + 16 +
                                              std::max(2 * 64 * (size_t)P.capp + 32, (size_t)(4 * 4 * ASM_LAT_NB)));
            if (bytes <= 160 * 1024) { P.fast_T = T; P.fast_lds_bytes = (unsigned)bytes; }
        }
    }
    *out = P;
    return 0;
}

// The lattice path of k_step_fast: decides whether this call may use it and fills the kernel's view of it.
// Conditions: the 4-wave schedule; the tables fit the unused part of the kernel's grid area in LDS;
// the sensing disc fits the kernel's window of 15 lattice rows x 15 columns and the occupancy disc 7 x 7.
bool lattice_usable(const asm_lattice *lat, const asm_params *prm, DevParams *P, asmenv::LatArgs *out)
{
    P->lat = 0;
    if (!lat || !lat->shapes || !lat->shape_id || !lat->xform || lat->n_shapes < 1 || !(lat->l_cell_min > 0)) return false;
    if (P->fast_T != 4) return false;                                   // (the cell computations are never wrapped: either arena)
    const size_t table_floats = 2 * sizeof(uint32_t) * (ASM_LAT_ROWS * ASM_LAT_CHUNKS * 2) / 4 + 2 * sizeof(uint16_t) * ASM_LAT_NB / 4;
    if (2 * (size_t)P->NGP + 64 + table_floats > ASM_GY_OFF) return false;
    const double R = prm->d_sen / lat->l_cell_min, Ro = prm->r_avoid / 2.0 / lat->l_cell_min;
    if (!(R <= 7.4) || !(Ro <= 3.4)) return false;
    P->lat = 1;
    P->lat_KO = (int)std::floor(Ro + 0.5 + 1e-9);                       // lattice rows |k| <= KO can hold an occupied cell
    if (P->lat_KO < 1) P->lat_KO = 1;
    // stored cell coordinates are floats of magnitude below half_w + half_h: half an ulp each; a squared lattice
    // distance near the sensing radius moves by at most 2 sqrt(2) R times that (in lattice units); x 2.5 for safety
    const double cmax = prm->half_w + prm->half_h;
    const double half_ulp = std::ldexp(1.0, (int)std::ceil(std::log2(cmax)) - 24) / 2.0;
    P->lat_cb = (float)(2.5 * 2.8284271247461903 * half_ulp);
    out->shapes = lat->shapes; out->shape_id = lat->shape_id; out->xform = lat->xform; out->n_shapes = lat->n_shapes;
    return true;
}

// The same description for the large-swarm schedule of k_step (one environment per workgroup, a lane per agent):
// walls, fp32 state, and room in LDS for the tables and the compacted boundary cells.
bool lattice_usable_swarm(const asm_lattice *lat, const asm_params *prm, DevParams *P, asmenv::LatArgs *out)
{
    DevParams Q = *P;
    Q.fast_T = 4; Q.NGP = 0;                                            // borrow the checks that do not depend on the schedule
    if (P->SL < 64 || P->EPB != 1 || P->periodic || !lattice_usable(lat, prm, &Q, out)) return false;
    const unsigned extra = 8 * ASM_LAT_ROWS * ASM_LAT_CHUNKS + 2 * ASM_LAT_NB + 8 * ASM_LAT_NB;
    if (P->lds_block_bytes + extra > 160 * 1024) return false;
    P->lds_env_bytes += extra; P->lds_block_bytes += extra;
    P->lat = 2; P->lat_KO = Q.lat_KO; P->lat_cb = Q.lat_cb;
    return true;
}

// A large swarm's step as three launches, the observation split over SL / 256 workgroups per environment (see
// launch_step_split): on the lattice path with cell lists only.  Fills the part workgroups' parameters.
bool split_usable(DevParams *P, DevParams *part)
{
    if (P->lat != 2 || !P->bins || P->SL < 512 || P->W > 32 || !split_on()) return false;
    DevParams Q = *P;
    Q.rows = 256; Q.parts = P->SL / 256; Q.state_rows = 2;
    if (!swarm_layout(&Q, sizeof(float)) || !Q.bins) return false;
    const unsigned extra = 8 * ASM_LAT_ROWS * ASM_LAT_CHUNKS + 2 * ASM_LAT_NB + 8 * ASM_LAT_NB;
    Q.lds_env_bytes += extra; Q.lds_block_bytes += extra;
    P->parts = Q.parts;
    Q.state_rows = 2;
    *part = Q;
    return true;
}

int hip_fail(int rc, const char *what)
{
    return fail(rc, "%s: HIP error %d (%s)", what, rc, hipGetErrorString((hipError_t)rc));
}

// ---- workspace + helpers for the five host-pointer entry points -----------------
std::mutex g_compat_mutex;
unsigned char *g_ws = nullptr;
size_t g_ws_bytes = 0;

// The five reference symbols return void and the reference never ends its host process (assembly.py:144-145 only
// prints on a bad mode).  A failed call here -- no usable HIP device, an unsupported argument -- therefore fills the
// outputs with NaN / -1 / false so that the failure cannot pass for a result, leaves the message in asm_last_error(),
// prints it once per process and returns.  MARL_AMD_ABORT_ON_ERROR=1 turns such a failure into abort().  There is no
// CPU fallback either way.
#define HIP_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) return hip_fail((int)_e, #call); } while (0)

thread_local int g_compat_rc = 0;                            // status of this thread's last call of one of the five (asm_compat_status)
void compat_failed(const char *fn)
{
    static const bool abort_on_error = [] { const char *v = std::getenv("MARL_AMD_ABORT_ON_ERROR"); return v && std::atoi(v) != 0; }();
    static std::atomic<bool> printed{false};
    if (abort_on_error || !printed.exchange(true))
        std::fprintf(stderr, "libAssemblyEnv (HIP): %s failed: %s. This library has no CPU fallback; the outputs of a failed "
                             "call are NaN / -1 (further failures are reported through asm_last_error() only).\n", fn, g_err);
    if (abort_on_error) std::abort();
}
template <typename T> void fill_n(T *dst, size_t n, T v) { if (dst) for (size_t k = 0; k < n; ++k) dst[k] = v; }
size_t count_or_0(long long n) { return n > 0 ? (size_t)n : 0; }

// One call = inputs gathered into a pinned host buffer, ONE host-to-device copy, the kernel, ONE device-to-host copy
// of the outputs, one stream synchronisation: the device workspace and the pinned buffer share a layout
// (inputs first, outputs behind them).
struct Arena {
    size_t off = 0;
    template <typename T> size_t take(size_t n) { size_t o = off; off += (n * sizeof(T) + 255) / 256 * 256; return o; }
};
unsigned char *g_host = nullptr;                             // pinned mirror of g_ws
hipStream_t g_stream = nullptr;
int ensure_ws(size_t bytes)
{
    if (!g_stream) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (bytes <= g_ws_bytes) return 0;
    if (g_ws) { hipError_t e1 = hipFree(g_ws), e2 = hipHostFree(g_host); g_ws = nullptr; g_host = nullptr; g_ws_bytes = 0;
                if (e1 != hipSuccess) return hip_fail((int)e1, "hipFree"); if (e2 != hipSuccess) return hip_fail((int)e2, "hipHostFree"); }
    bytes = bytes * 2 + 4096;                                // head-room: the sizes of a run rarely change
    HIP_TRY(hipMalloc((void **)&g_ws, bytes));
    if (hipError_t e = hipHostMalloc((void **)&g_host, bytes, hipHostMallocDefault); e != hipSuccess) {
        (void)hipFree(g_ws); g_ws = nullptr; g_host = nullptr;
        return hip_fail((int)e, "hipHostMalloc");
    }
    g_ws_bytes = bytes;
    return 0;
}
template <typename T> T *at(size_t off) { return reinterpret_cast<T *>(g_ws + off); }
template <typename T> void put(size_t off, const T *src, size_t n) { std::memcpy(g_host + off, src, n * sizeof(T)); }
template <typename T> void get(T *dst, size_t off, size_t n) { std::memcpy(dst, g_host + off, n * sizeof(T)); }
int upload(size_t end) { HIP_TRY(hipMemcpyAsync(g_ws, g_host, end, hipMemcpyHostToDevice, g_stream)); return 0; }
int download(size_t begin, size_t end)
{
    HIP_TRY(hipMemcpyAsync(g_host + begin, g_ws + begin, end - begin, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return 0;
}
#define TRY(expr) do { if (int _rc = (expr)) return _rc; } while (0)

void compat_params(asm_params *a, int n_a, int n_g, int topo, int n_obs, int n_occ, double d_sen, double r_avoid,
                   const double *bpos)
{
    asm_default_params(a, n_a, n_g);
    a->topo_nei_max = topo; a->num_obs_grid_max = n_obs; a->num_occupied_grid_max = n_occ;
    a->d_sen = d_sen; a->r_avoid = r_avoid;
    if (bpos) { a->half_w = (bpos[2] - bpos[0]) / 2.0; a->half_h = (bpos[1] - bpos[3]) / 2.0; }   // CPP:75-76
}

}  // namespace

extern "C" {

void asm_default_params(asm_params *p, int n_a, int n_g_max)
{
    std::memset(p, 0, sizeof *p);
    p->n_a = n_a; p->n_g_max = n_g_max; p->topo_nei_max = 6; p->num_obs_grid_max = 80;
    p->num_occupied_grid_max = 200; p->is_boundary = 1; p->is_con_self_state = 1; p->compute_prior = 1;
    p->size_a = 0.035; p->d_sen = 0.4; p->r_avoid = 0.26; p->vel_max = 0.8; p->k_ball = 30; p->k_wall = 100;
    p->c_wall = 5; p->dt = 0.1; p->sensitivity = 1; p->mass = 1; p->half_w = 2.4; p->half_h = 2.4;
}

int asm_lattice_build(const double *grid_xy, int n_g, double l_cell, asm_lattice_shape *out)
{
    if (!grid_xy || !out || n_g < 1 || !(l_cell > 0)) return fail(1, "asm_lattice_build: bad arguments");
    if (n_g > 1024) return fail(1, "asm_lattice_build: n_g = %d > 1024", n_g);
    std::memset(out, 0, sizeof *out);
    double xmin = grid_xy[0], ymin = grid_xy[1];
    for (int c = 1; c < n_g; ++c) { xmin = std::min(xmin, grid_xy[2 * c]); ymin = std::min(ymin, grid_xy[2 * c + 1]); }
    static thread_local int col[1024], row[1024];
    int cols = 0, rows = 0;
    for (int c = 0; c < n_g; ++c) {
        const double u = (grid_xy[2 * c] - xmin) / l_cell, v = (grid_xy[2 * c + 1] - ymin) / l_cell;
        const double ur = std::nearbyint(u), vr = std::nearbyint(v);
        if (std::fabs(u - ur) > 1e-9 || std::fabs(v - vr) > 1e-9)
            return fail(2, "asm_lattice_build: cell %d is not on a lattice of pitch %.17g", c, l_cell);
        col[c] = (int)ur; row[c] = (int)vr;
        cols = std::max(cols, col[c] + 1); rows = std::max(rows, row[c] + 1);
        if (c > 0 && !(row[c] > row[c - 1] || (row[c] == row[c - 1] && col[c] > col[c - 1])))
            return fail(2, "asm_lattice_build: cells %d, %d are not in row-major lattice order", c - 1, c);
    }
    if (cols > 16 * ASM_LAT_CHUNKS || rows > ASM_LAT_ROWS - 2)
        return fail(3, "asm_lattice_build: %d x %d lattice exceeds %d columns x %d rows", cols, rows, 16 * ASM_LAT_CHUNKS, ASM_LAT_ROWS - 2);
    static thread_local unsigned char occ[ASM_LAT_ROWS][64];
    std::memset(occ, 0, sizeof occ);
    for (int c = 0; c < n_g; ++c) occ[row[c] + 1][col[c]] = 1;
    int c = 0;                                                          // cells before the current position
    for (int r = 0; r < ASM_LAT_ROWS; ++r) {
        int before[65]; before[0] = c;
        for (int a = 0; a < 64; ++a) { c += occ[r][a]; before[a + 1] = c; }
        for (int j = 0; j < ASM_LAT_CHUNKS; ++j) {
            uint32_t bits = 0;
            for (int b = 0; b < 32 && 16 * j + b < 64; ++b) bits |= (uint32_t)occ[r][16 * j + b] << b;
            out->row_tab[r][j][0] = bits; out->row_tab[r][j][1] = (uint32_t)before[16 * j];
        }
    }
    int nb = 0;
    for (int k = 0; k < n_g; ++k) {
        const int r = row[k] + 1, a = col[k];
        const bool inner = occ[r - 1][a] && occ[r + 1][a] && (a > 0 && occ[r][a - 1]) && occ[r][a + 1];
        if (!inner) {
            if (nb >= ASM_LAT_NB) return fail(3, "asm_lattice_build: more than %d boundary cells", ASM_LAT_NB);
            out->boundary[nb++] = (uint16_t)k;
        }
    }
    out->n_boundary = nb; out->cols = cols; out->rows = rows; out->n_g = n_g;
    out->x0 = xmin; out->y0 = ymin; out->l_cell = l_cell;
    return 0;
}

int asm_obs_dim(const asm_params *p)
{
    return 2 * 2 * (p->topo_nei_max + 1 + (p->is_con_self_state ? 1 : 0)) + 2 * p->num_obs_grid_max;
}

int asm_abi_version(void) { return ASM_ABI_VERSION; }
void asm_set_step_schedule(int waves) { g_schedule_override.store(waves, std::memory_order_relaxed); }
const char *asm_last_error(void) { return g_err; }
int asm_compat_status(void) { return g_compat_rc; }

int asm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail((int)e, "hipGetDeviceCount: %s", hipGetErrorString(e)); return -(int)e; }
    return n;
}

int asm_batch_observe_f32(const asm_params *prm, int E, const float *p, const float *dp, const float *grid,
                          const int32_t *n_g, const double *l_cell, int32_t *neighbor_index, int32_t *in_flags,
                          int32_t *nearest_cell, float *obs, float *reward, int32_t *sensed_index,
                          int32_t *occupied_index, const asm_lattice *lattice, void *stream)
{
    DevParams P;
    if (int rc = make_dev_params(prm, E, sizeof(float), &P)) return rc;
    if (!p || !dp || !grid || !n_g || !l_cell || !neighbor_index || !in_flags || !nearest_cell || !obs)
        return fail(1, "asm_batch_observe_f32: NULL required pointer");
    P.want_reward = reward != nullptr;
    int rc;
    if (P.fast_T) {                                                 // the step kernel without its dynamics phase
        P.observe_only = 1; P.prior = 0;
        asmenv::LatArgs LA{};
        lattice_usable(lattice, prm, &P, &LA);
        rc = asmenv::launch_step_fast(P, LA, const_cast<float *>(p), const_cast<float *>(dp), nullptr, grid, n_g, l_cell,
                                      neighbor_index, in_flags, nearest_cell, obs, reward, nullptr, sensed_index,
                                      occupied_index, nullptr, (hipStream_t)stream);
    } else {
        rc = asmenv::launch_observe<float>(P, p, dp, nullptr, grid, n_g, l_cell, neighbor_index, in_flags,
                                           nearest_cell, obs, reward, sensed_index, occupied_index,
                                           (hipStream_t)stream);
    }
    return rc ? hip_fail(rc, "asm_batch_observe_f32") : 0;
}

int asm_batch_step_f32(const asm_params *prm, int E, float *p, float *dp, const float *act, const float *grid,
                       const int32_t *n_g, const double *l_cell, int32_t *neighbor_index, int32_t *in_flags,
                       int32_t *nearest_cell, float *obs, float *reward, float *a_prior, int32_t *sensed_index,
                       int32_t *occupied_index, double *stats, int recompute_target, const asm_lattice *lattice,
                       void *stream)
{
    DevParams P;
    if (int rc = make_dev_params(prm, E, sizeof(float), &P)) return rc;
    if (!p || !dp || !act || !grid || !n_g || !l_cell || !neighbor_index || !in_flags || !nearest_cell || !obs || !reward)
        return fail(1, "asm_batch_step_f32: NULL required pointer");
    P.want_reward = 1; P.want_stats = stats != nullptr; P.recompute = recompute_target != 0;
    P.obs_nt = obs_nt_for(P);
    if (!a_prior) P.prior = 0;
    asmenv::LatArgs LA{};
    if (P.fast_T) lattice_usable(lattice, prm, &P, &LA);
    else lattice_usable_swarm(lattice, prm, &P, &LA);
    DevParams part;
    int rc;
    if (P.fast_T)
        rc = asmenv::launch_step_fast(P, LA, p, dp, act, grid, n_g, l_cell, neighbor_index, in_flags, nearest_cell, obs, reward,
                                      a_prior, sensed_index, occupied_index, stats, (hipStream_t)stream);
    else if (split_usable(&P, &part))
        rc = asmenv::launch_step_split(P, part, LA, p, dp, act, grid, n_g, l_cell, neighbor_index, in_flags, nearest_cell, obs,
                                       reward, a_prior, sensed_index, occupied_index, stats, (hipStream_t)stream);
    else
        rc = asmenv::launch_step<float>(P, LA, p, dp, act, grid, n_g, l_cell, neighbor_index, in_flags, nearest_cell, obs,
                                        reward, a_prior, sensed_index, occupied_index, stats, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_batch_step_f32") : 0;
}

int asm_batch_reset_f32(const asm_params *prm, uint64_t seed, uint32_t episode, int64_t env0, int E, int n_shapes,
                        const double *shape_grids, const int32_t *shape_off, const int32_t *shape_ng,
                        const double *shape_lcell, int32_t *shape_id, float *p, float *dp, float *grid, int32_t *n_g,
                        double *l_cell, int32_t *neighbor_index, int32_t *in_flags, int32_t *nearest_cell, float *obs,
                        int32_t *sensed_index, int32_t *occupied_index, double *xform, const asm_lattice *lattice,
                        void *stream)
{
    DevParams P;
    if (int rc = make_dev_params(prm, E, sizeof(float), &P)) return rc;
    if (n_shapes < 1) return fail(1, "asm_batch_reset_f32: n_shapes = %d", n_shapes);
    if (!shape_grids || !shape_off || !shape_ng || !shape_lcell || !shape_id || !p || !dp || !grid || !n_g || !l_cell ||
        !neighbor_index || !in_flags || !nearest_cell || !obs)
        return fail(1, "asm_batch_reset_f32: NULL required pointer");
    P.state_only = P.fast_T != 0;                                   // then the step kernel observes the new state
    int rc = asmenv::launch_reset<float>(P, seed, episode, env0, n_shapes, shape_grids, shape_off, shape_ng,
                                         shape_lcell, shape_id, p, dp, grid, n_g, l_cell, neighbor_index, in_flags,
                                         nearest_cell, obs, sensed_index, occupied_index, xform, (hipStream_t)stream);
    if (!rc && P.fast_T) {
        P.observe_only = 1; P.prior = 0; P.want_reward = 0; P.want_stats = 0;
        asmenv::LatArgs LA{};
        if (lattice && (lattice->shape_id != shape_id || lattice->xform != xform || !xform))
            return fail(1, "asm_batch_reset_f32: the lattice description must name the shape_id / xform arrays this call writes");
        lattice_usable(lattice, prm, &P, &LA);
        rc = asmenv::launch_step_fast(P, LA, p, dp, nullptr, grid, n_g, l_cell, neighbor_index, in_flags, nearest_cell, obs,
                                      nullptr, nullptr, sensed_index, occupied_index, nullptr, (hipStream_t)stream);
    }
    return rc ? hip_fail(rc, "asm_batch_reset_f32") : 0;
}

int asm_batch_random_actions_f32(uint64_t seed, uint32_t step, int64_t env0, int E, int n_a, float *act, void *stream)
{
    if (E < 1 || n_a < 1 || !act) return fail(1, "asm_batch_random_actions_f32: bad arguments");
    int rc = asmenv::launch_random_actions(seed, step, env0, E, n_a, act, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_batch_random_actions_f32") : 0;
}

int asm_batch_rule_actions_f32(const asm_params *prm, int E, const float *p, const float *dp, const float *grid,
                               const int32_t *n_g, const int32_t *in_flags, const int32_t *nearest_cell,
                               const int32_t *sensed_index, float *act, void *stream)
{
    DevParams P;
    if (int rc = make_dev_params(prm, E, sizeof(float), &P)) return rc;
    if (!p || !dp || !grid || !n_g || !in_flags || !nearest_cell || !sensed_index || !act)
        return fail(1, "asm_batch_rule_actions_f32: NULL pointer (the sensed-index list is required)");
    int rc = asmenv::launch_rule_actions(P, p, dp, grid, n_g, in_flags, nearest_cell, sensed_index, act, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_batch_rule_actions_f32") : 0;
}

int asm_batch_metrics_f64(const asm_params *prm, int E, const float *p, const float *grid, const int32_t *n_g,
                          double *metrics, void *stream)
{
    if (!prm || E < 1 || !p || !grid || !n_g || !metrics) return fail(1, "asm_batch_metrics_f64: bad arguments");
    if (prm->n_a < 1 || prm->n_a > asmenv::ASM_MAX_AGENTS)
        return fail(1, "n_a = %d: supported range is 1..%d", prm->n_a, asmenv::ASM_MAX_AGENTS);
    if (prm->n_g_max < 1) return fail(1, "asm_batch_metrics_f64: n_g_max = %d", prm->n_g_max);
    int rc = asmenv::launch_metrics(E, prm->n_a, prm->n_g_max, prm->r_avoid, prm->size_a, p, grid, n_g, metrics,
                                    (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_batch_metrics_f64") : 0;
}

int asm_replay_draw_indices(uint64_t seed, uint32_t counter, const uint32_t *counter_dev, int64_t lo, int64_t hi, int N,
                            int64_t *index, void *stream)
{
    if (N < 1 || hi <= lo || !index) return fail(1, "asm_replay_draw_indices: bad arguments");
    int rc = asmenv::launch_replay_draw(seed, counter, counter_dev, lo, hi, N, index, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_replay_draw_indices") : 0;
}

int asm_replay_gather_f32(int E, int n_a, int obs_dim, int episode_length, const float *obs_ring, const float *act_ring,
                          const float *prior_ring, const float *rew_ring, const int64_t *index, int N, float *obs_out,
                          float *act_out, float *rew_out, float *next_obs_out, float *prior_out, void *stream)
{
    if (E < 1 || n_a < 1 || obs_dim < 1 || episode_length < 1 || N < 1 || !obs_ring || !act_ring || !rew_ring || !index ||
        !obs_out || !act_out || !rew_out || !next_obs_out || (prior_out && !prior_ring))
        return fail(1, "asm_replay_gather_f32: bad arguments");
    int rc = asmenv::launch_replay_gather(E, n_a, obs_dim, episode_length, obs_ring, act_ring, prior_ring, rew_ring, index, N,
                                          obs_out, act_out, rew_out, next_obs_out, prior_out, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_replay_gather_f32") : 0;
}

int asm_obs_rows_f32(int E, int n_a, int obs_dim, const float *obs, float *rows, void *stream)
{
    if (E < 1 || n_a < 1 || obs_dim < 1 || !obs || !rows) return fail(1, "asm_obs_rows_f32: bad arguments");
    if ((size_t)obs_dim * 33 * sizeof(float) > 160 * 1024) return fail(1, "asm_obs_rows_f32: obs_dim = %d: a tile does not fit the LDS", obs_dim);
    int rc = asmenv::launch_obs_rows(E, n_a, obs_dim, obs, rows, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_obs_rows_f32") : 0;
}

int asm_policy_mlp_act_bf16(int E, int n_a, int obs_dim, int hidden, int act_dim, const float *obs, const void *w1,
                            const void *w2, const void *w3, const void *w4, const float *bias, float *act, float noise_scale,
                            float epsilon, uint64_t seed, uint32_t step, int64_t env0, void *stream)
{
    if (E < 1 || n_a < 1 || obs_dim < 1 || act_dim < 1 || act_dim > 32 || !obs || !w1 || !w2 || !w3 || !w4 || !bias || !act)
        return fail(1, "asm_policy_mlp_act_bf16: bad arguments");
    if (!(noise_scale >= 0.f) || !(epsilon >= 0.f))
        return fail(1, "asm_policy_mlp_act_bf16: noise_scale and epsilon must be >= 0");
    if (hidden != 32 && hidden != 64 && hidden != 128 && hidden != 256)
        return fail(1, "asm_policy_mlp_act_bf16: hidden = %d (supported: 32, 64, 128, 256)", hidden);
    if ((size_t)(hidden / 32) * std::max(((obs_dim + 15) / 16 + 1) / 2, 2) * 2048 + 4 * (3 * (size_t)hidden + 32) > 160 * 1024)
        return fail(1, "asm_policy_mlp_act_bf16: obs_dim = %d with hidden = %d: the first layer's weights do not fit the LDS", obs_dim, hidden);
    int rc = asmenv::launch_policy_mlp(E, n_a, obs_dim, hidden, act_dim, obs, w1, w2, w3, w4, bias, act, noise_scale, epsilon, seed,
                                       step, env0, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_policy_mlp_act_bf16") : 0;
}

int asm_policy_mlp_forward_bf16(int E, int n_a, int obs_dim, int hidden, int act_dim, const float *obs, const void *w1,
                                const void *w2, const void *w3, const void *w4, const float *bias, float *act, void *stream)
{
    return asm_policy_mlp_act_bf16(E, n_a, obs_dim, hidden, act_dim, obs, w1, w2, w3, w4, bias, act, 0.f, 0.f, 0, 0, 0, stream);
}

int asm_batch_check_lattice(const asm_params *prm, int E, const float *grid, const int32_t *n_g,
                            const asm_lattice *lattice, double tol, int32_t *n_bad, void *stream)
{
    if (!prm || E < 1 || !grid || !n_g || !lattice || !lattice->shapes || !lattice->shape_id || !lattice->xform || !n_bad)
        return fail(1, "asm_batch_check_lattice: bad arguments");
    asmenv::LatArgs LA{lattice->shapes, lattice->shape_id, lattice->xform, lattice->n_shapes};
    int rc = asmenv::launch_check_lattice(E, prm->n_g_max, grid, n_g, LA, tol, n_bad, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_batch_check_lattice") : 0;
}

int asm_debug_step_schedule(const asm_params *prm, int E, const asm_lattice *lattice)
{
    DevParams P;
    if (make_dev_params(prm, E, sizeof(float), &P)) return -1;
    asmenv::LatArgs LA{};
    if (P.fast_T) lattice_usable(lattice, prm, &P, &LA);
    else lattice_usable_swarm(lattice, prm, &P, &LA);
    DevParams part;
    const int split = !P.fast_T && split_usable(&P, &part) ? 1 : 0;
    return P.fast_T + 16 * P.lat + 64 * P.bins + 128 * split + 256 * (!P.fast_T && !split && asmenv::uses_swarm_kernel(P) ? 1 : 0);
}

#ifdef ASM_DIAG
// diagnostic build only: device buffer [grid][32] of cycle stamps, or NULL to stop stamping
int asm_diag_set_buffer(void *dev) { return asmenv::diag_set_buffer((unsigned long long *)dev); }
// k_step_fast stamps only the launches whose position array is p (NULL: every launch)
int asm_diag_set_filter(const void *p) { return asmenv::diag_set_filter(p); }
#endif

int asm_debug_cos_f32_max_err(uint32_t bits_lo, uint32_t bits_hi, double *out_max, void *stream)
{
    if (bits_lo > bits_hi || !out_max) return fail(1, "asm_debug_cos_f32_max_err: bad arguments");
    int rc = asmenv::launch_cos_f32_err(bits_lo, bits_hi, out_max, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_debug_cos_f32_max_err") : 0;
}

int asm_debug_sqrt_f64(const double *in, double *out, int n, void *stream)
{
    if (n < 1 || !in || !out) return fail(1, "asm_debug_sqrt_f64: bad arguments");
    int rc = asmenv::launch_sqrt(in, out, n, (hipStream_t)stream);
    return rc ? hip_fail(rc, "asm_debug_sqrt_f64") : 0;
}

// ------------------------------------------------------------------------------
// The five reference entry points: host pointers in, GPU compute, host pointers out.

void _get_observation(double *p_in, double *dp_in, double *heading_in, double *obs_io, double *bpos, double *grid_in,
                      int *nbr_io, int *inflags_io, int *sensed_io, int *occupied_io, double d_sen, double r_avoid,
                      double l_cell, double Vel_max, int topo_nei_max, int num_obs_grid_max,
                      int num_occupied_grid_max, int n_a, int n_g, int obs_dim_agent, int dim, bool *condition)
{
    (void)Vel_max;                                           // unused by the reference too (CPP:36)
    std::lock_guard<std::mutex> lk(g_compat_mutex);
    const int rc = [&]() -> int {
        if (dim != 2) return fail(1, "dim = %d: only 2 is supported", dim);
        if (n_g < 1) return fail(1, "n_g = %d", n_g);
        asm_params a; compat_params(&a, n_a, n_g, topo_nei_max, num_obs_grid_max, num_occupied_grid_max, d_sen, r_avoid, bpos);
        a.is_boundary = condition[0] ? 0 : 1; a.is_con_self_state = condition[2] ? 1 : 0;
        DevParams P;
        TRY(make_dev_params(&a, 1, sizeof(double), &P));
        P.cartesian = condition[1] ? 1 : 0;
        P.obs_dim = obs_dim_agent;
        const int min_rows = (P.con_self ? 4 : 0) + 24 + (P.cartesian ? 4 + 2 * P.n_obs : 6);
        if (obs_dim_agent < min_rows) return fail(1, "obs_dim_agent = %d < %d", obs_dim_agent, min_rows);
        const size_t na = (size_t)n_a, ng = (size_t)n_g;
        Arena A;
        size_t o_p = A.take<double>(2 * na), o_dp = A.take<double>(2 * na), o_h = A.take<double>(2 * na);
        size_t o_g = A.take<double>(2 * ng), o_ng = A.take<int>(1), o_lc = A.take<double>(1);
        size_t o_obs = A.take<double>((size_t)obs_dim_agent * na), o_nbr = A.take<int>(na * 6), o_in = A.take<int>(na);
        size_t o_near = A.take<int>(na), o_sen = A.take<int>(na * P.n_obs), o_occ = A.take<int>(na * P.n_occ);
        TRY(ensure_ws(A.off));
        put(o_p, p_in, 2 * na); put(o_dp, dp_in, 2 * na); put(o_h, heading_in, 2 * na); put(o_g, grid_in, 2 * ng);
        put(o_ng, &n_g, 1); put(o_lc, &l_cell, 1);
        TRY(upload(o_obs));                                  // the inputs end where the outputs begin
        HIP_TRY(hipMemsetAsync(g_ws + o_obs, 0, (size_t)obs_dim_agent * na * sizeof(double), g_stream));   // CPP:49 zero-initialised
        if (int krc = asmenv::launch_observe<double>(P, at<double>(o_p), at<double>(o_dp), at<double>(o_h), at<double>(o_g),
                                                     at<int>(o_ng), at<double>(o_lc), at<int>(o_nbr), at<int>(o_in),
                                                     at<int>(o_near), at<double>(o_obs), nullptr, at<int>(o_sen),
                                                     at<int>(o_occ), g_stream))
            return hip_fail(krc, "_get_observation kernel launch");
        TRY(download(o_obs, A.off));
        get(obs_io, o_obs, (size_t)obs_dim_agent * na); get(nbr_io, o_nbr, na * 6); get(inflags_io, o_in, na);
        get(sensed_io, o_sen, na * P.n_obs); get(occupied_io, o_occ, na * P.n_occ);
        return 0;
    }();
    g_compat_rc = rc;
    if (rc) {
        const size_t na = count_or_0(n_a);
        fill_n(obs_io, na * count_or_0(obs_dim_agent), std::nan(""));
        fill_n(nbr_io, na * count_or_0(topo_nei_max), -1); fill_n(inflags_io, na, -1);
        fill_n(sensed_io, na * count_or_0(num_obs_grid_max), -1); fill_n(occupied_io, na * count_or_0(num_occupied_grid_max), -1);
        compat_failed("_get_observation");
    }
}

void _get_reward(double *p_in, double *dp_in, double *heading_in, double *act_in, double *reward_io, double *bpos,
                 double *grid_in, int *nbr_in, int *inflags_in, int *sensed_in, int *occupied_in, double d_sen,
                 double r_avoid, double l_cell, int topo_nei_max, int num_obs_grid_max, int num_occupied_grid_max,
                 int n_a, int n_g, int dim, bool *condition, bool *is_collide_b2b, bool *is_collide_b2w,
                 double *coefficients)
{
    // dp, heading, act, occupied_index, l_cell, the contact matrices and the coefficients are
    // copied but never read by the reference either (CPP:384-452 vs 457-564).
    (void)dp_in; (void)heading_in; (void)act_in; (void)occupied_in; (void)l_cell; (void)is_collide_b2b;
    (void)is_collide_b2w; (void)coefficients; (void)dim;
    std::lock_guard<std::mutex> lk(g_compat_mutex);
    const int rc = [&]() -> int {
        if (n_g < 1) return fail(1, "n_g = %d", n_g);
        if (topo_nei_max < 1) return fail(1, "topo_nei_max = %d", topo_nei_max);
        asm_params a; compat_params(&a, n_a, n_g, 6, num_obs_grid_max, num_occupied_grid_max, d_sen, r_avoid, bpos);
        DevParams P;
        TRY(make_dev_params(&a, 1, sizeof(double), &P));
        P.topo = topo_nei_max;
        const size_t na = (size_t)n_a, ng = (size_t)n_g;
        Arena A;
        size_t o_p = A.take<double>(2 * na), o_g = A.take<double>(2 * ng), o_nbr = A.take<int>(na * topo_nei_max);
        size_t o_in = A.take<int>(na), o_sen = A.take<int>(na * P.n_obs), o_r = A.take<double>(na);
        TRY(ensure_ws(A.off));
        put(o_p, p_in, 2 * na); put(o_g, grid_in, 2 * ng); put(o_nbr, nbr_in, na * topo_nei_max);
        put(o_in, inflags_in, na); put(o_sen, sensed_in, na * P.n_obs);
        TRY(upload(o_r));
        if (int krc = asmenv::launch_reward_compat(P, at<double>(o_p), at<double>(o_g), n_g, at<int>(o_nbr), at<int>(o_in),
                                                   at<int>(o_sen), condition[0], condition[3], condition[4], at<double>(o_r),
                                                   g_stream))
            return hip_fail(krc, "_get_reward kernel launch");
        TRY(download(o_r, A.off));
        get(reward_io, o_r, na);
        return 0;
    }();
    g_compat_rc = rc;
    if (rc) { fill_n(reward_io, count_or_0(n_a), std::nan("")); compat_failed("_get_reward"); }
}

void _sf_b2b_all(double *p_in, double *sf_io, double *d_edge_in, bool *collide_in, double *bpos, double *d_center_in,
                 int n_a, int dim, double k_ball, bool is_periodic)
{
    (void)dim;
    std::lock_guard<std::mutex> lk(g_compat_mutex);
    if (n_a < 1) return;
    const int rc = [&]() -> int {
        const size_t na = (size_t)n_a;
        Arena A;
        size_t o_p = A.take<double>(2 * na), o_de = A.take<double>(na * na), o_dc = A.take<double>(na * na);
        size_t o_c = A.take<bool>(na * na), o_sf = A.take<double>(2 * na);
        TRY(ensure_ws(A.off));
        put(o_p, p_in, 2 * na); put(o_de, d_edge_in, na * na); put(o_dc, d_center_in, na * na); put(o_c, collide_in, na * na);
        TRY(upload(o_sf));
        const double hw = (bpos[2] - bpos[0]) / 2.0, hh = (bpos[1] - bpos[3]) / 2.0;               // CPP:776-777
        if (int krc = asmenv::launch_sf_b2b_compat(at<double>(o_p), at<double>(o_de), at<bool>(o_c), at<double>(o_dc), n_a,
                                                   k_ball, is_periodic, hw, hh, at<double>(o_sf), g_stream))
            return hip_fail(krc, "_sf_b2b_all kernel launch");
        TRY(download(o_sf, A.off));
        get(sf_io, o_sf, 2 * na);
        return 0;
    }();
    g_compat_rc = rc;
    if (rc) { fill_n(sf_io, 2 * (size_t)n_a, std::nan("")); compat_failed("_sf_b2b_all"); }
}

void _get_dist_b2w(double *p_in, double *r_in, double *d_b2w_io, bool *coll_io, int dim, int n_a, double *bpos)
{
    (void)dim;
    std::lock_guard<std::mutex> lk(g_compat_mutex);
    if (n_a < 1) return;
    const int rc = [&]() -> int {
        const size_t na = (size_t)n_a;
        Arena A;
        size_t o_p = A.take<double>(2 * na), o_r = A.take<double>(na), o_d = A.take<double>(4 * na), o_c = A.take<bool>(4 * na);
        TRY(ensure_ws(A.off));
        put(o_p, p_in, 2 * na); put(o_r, r_in, na);
        TRY(upload(o_d));
        if (int krc = asmenv::launch_dist_b2w_compat(at<double>(o_p), at<double>(o_r), n_a, bpos, at<double>(o_d), at<bool>(o_c),
                                                     g_stream))
            return hip_fail(krc, "_get_dist_b2w kernel launch");
        TRY(download(o_d, A.off));
        get(d_b2w_io, o_d, 4 * na); get(coll_io, o_c, 4 * na);
        return 0;
    }();
    g_compat_rc = rc;
    if (rc) { fill_n(d_b2w_io, 4 * (size_t)n_a, std::nan("")); fill_n(coll_io, 4 * (size_t)n_a, false); compat_failed("_get_dist_b2w"); }
}

void calculateActionPrior(double *p_in, double *dp_in, double *a_prior_io, double *grid_in, int *nbr_in, double d_sen,
                          double r_avoid, double l_cell, int topo_nei_max, int n_a, int n_g, int dim)
{
    (void)dim;
    std::lock_guard<std::mutex> lk(g_compat_mutex);
    const int rc = [&]() -> int {
        if (n_g < 1) return fail(1, "n_g = %d", n_g);
        if (topo_nei_max < 1 || topo_nei_max > 6) return fail(1, "topo_nei_max = %d: supported 1..6", topo_nei_max);
        asm_params a; compat_params(&a, n_a, n_g, 6, 80, 200, d_sen, r_avoid, nullptr);
        DevParams P;
        TRY(make_dev_params(&a, 1, sizeof(double), &P));
        P.topo = topo_nei_max;
        const size_t na = (size_t)n_a, ng = (size_t)n_g;
        Arena A;
        size_t o_p = A.take<double>(2 * na), o_dp = A.take<double>(2 * na), o_g = A.take<double>(2 * ng);
        size_t o_nbr = A.take<int>(na * topo_nei_max), o_a = A.take<double>(2 * na);
        TRY(ensure_ws(A.off));
        put(o_p, p_in, 2 * na); put(o_dp, dp_in, 2 * na); put(o_g, grid_in, 2 * ng); put(o_nbr, nbr_in, na * topo_nei_max);
        TRY(upload(o_a));
        if (int krc = asmenv::launch_prior_compat(P, at<double>(o_p), at<double>(o_dp), at<double>(o_g), n_g, at<int>(o_nbr),
                                                  l_cell, at<double>(o_a), g_stream))
            return hip_fail(krc, "calculateActionPrior kernel launch");
        TRY(download(o_a, A.off));
        get(a_prior_io, o_a, 2 * na);
        return 0;
    }();
    g_compat_rc = rc;
    if (rc) { fill_n(a_prior_io, 2 * count_or_0(n_a), std::nan("")); compat_failed("calculateActionPrior"); }
}

// ------------------------------------------------------------------------------
// _get_observation for callers that cannot allocate (serving loops that must not touch the allocator, harnesses that
// forbid hipMalloc): the device staging area of the call is a file-scope device array instead of the growing
// hipMalloc'ed workspace above, the copies go straight between the caller's arrays and that array, and the work
// runs on the legacy default stream.  Nothing here allocates, frees or creates a stream.
//
// In host code the array's name is only a handle for hipGetSymbolAddress; the address kernels and copies use is
// looked up once per device (each GPU has its own copy at its own address) and kept in g_scratch_ptr.
#define ASM_NOALLOC_SCRATCH_BYTES (4u << 20)
static __device__ __align__(256) unsigned char scratch[ASM_NOALLOC_SCRATCH_BYTES];
namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES];           // device address of `scratch`, per device
std::mutex g_noalloc_mutex;
}

int asm_get_observation_noalloc(double *p_in, double *dp_in, double *heading_in, double *obs_io, double *bpos,
                                double *grid_in, int *nbr_io, int *inflags_io, int *sensed_io, int *occupied_io,
                                double d_sen, double r_avoid, double l_cell, double Vel_max, int topo_nei_max,
                                int num_obs_grid_max, int num_occupied_grid_max, int n_a, int n_g, int obs_dim_agent,
                                int dim, bool *condition)
{
    (void)Vel_max;
    std::lock_guard<std::mutex> lk(g_noalloc_mutex);
    if (!p_in || !dp_in || !heading_in || !obs_io || !bpos || !grid_in || !nbr_io || !inflags_io || !sensed_io ||
        !occupied_io || !condition)
        return fail(1, "asm_get_observation_noalloc: NULL pointer");
    if (dim != 2) return fail(1, "dim = %d: only 2 is supported", dim);
    if (n_g < 1) return fail(1, "n_g = %d", n_g);
    int dev = -1;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return hip_fail((int)he, "hipGetDevice");
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(1, "device %d: at most %d devices are supported", dev, NOALLOC_MAX_DEVICES);
    if (!g_scratch_ptr[dev]) {                                // first call on this device, which is the current one
        void *ptr = nullptr;
        he = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (he != hipSuccess || !ptr) return hip_fail((int)he, "hipGetSymbolAddress(scratch)");
        g_scratch_ptr[dev] = static_cast<unsigned char *>(ptr);
    }
    unsigned char *const ws = g_scratch_ptr[dev];

    asm_params a; compat_params(&a, n_a, n_g, topo_nei_max, num_obs_grid_max, num_occupied_grid_max, d_sen, r_avoid, bpos);
    a.is_boundary = condition[0] ? 0 : 1; a.is_con_self_state = condition[2] ? 1 : 0;
    DevParams P;
    if (int rc = make_dev_params(&a, 1, sizeof(double), &P)) return rc;
    P.cartesian = condition[1] ? 1 : 0;
    P.obs_dim = obs_dim_agent;
    const int min_rows = (P.con_self ? 4 : 0) + 24 + (P.cartesian ? 4 + 2 * P.n_obs : 6);
    if (obs_dim_agent < min_rows) return fail(1, "obs_dim_agent = %d < %d", obs_dim_agent, min_rows);
    const size_t na = (size_t)n_a, ng = (size_t)n_g;
    Arena A;
    const size_t o_p = A.take<double>(2 * na), o_dp = A.take<double>(2 * na), o_h = A.take<double>(2 * na);
    const size_t o_g = A.take<double>(2 * ng), o_ng = A.take<int>(1), o_lc = A.take<double>(1);
    const size_t o_obs = A.take<double>((size_t)obs_dim_agent * na), o_nbr = A.take<int>(na * 6), o_in = A.take<int>(na);
    const size_t o_near = A.take<int>(na), o_sen = A.take<int>(na * P.n_obs), o_occ = A.take<int>(na * P.n_occ);
    if (A.off > ASM_NOALLOC_SCRATCH_BYTES)
        return fail(1, "asm_get_observation_noalloc: n_a = %d, n_g = %d, obs_dim = %d, list capacities %d / %d need %zu bytes of "
                       "staging; the fixed device array holds %u", n_a, n_g, obs_dim_agent, P.n_obs, P.n_occ, A.off,
                    (unsigned)ASM_NOALLOC_SCRATCH_BYTES);
#define NOALLOC_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) return hip_fail((int)_e, #call); } while (0)
    // blocking copies on the legacy default stream: each is ordered against the kernel below without further calls
    NOALLOC_HIP(hipMemcpy(ws + o_p, p_in, 2 * na * sizeof(double), hipMemcpyHostToDevice));
    NOALLOC_HIP(hipMemcpy(ws + o_dp, dp_in, 2 * na * sizeof(double), hipMemcpyHostToDevice));
    NOALLOC_HIP(hipMemcpy(ws + o_h, heading_in, 2 * na * sizeof(double), hipMemcpyHostToDevice));
    NOALLOC_HIP(hipMemcpy(ws + o_g, grid_in, 2 * ng * sizeof(double), hipMemcpyHostToDevice));
    NOALLOC_HIP(hipMemcpy(ws + o_ng, &n_g, sizeof(int), hipMemcpyHostToDevice));
    NOALLOC_HIP(hipMemcpy(ws + o_lc, &l_cell, sizeof(double), hipMemcpyHostToDevice));
    // the array keeps the previous call's contents: the observation must start at zero (CPP:49)
    NOALLOC_HIP(hipMemsetAsync(ws + o_obs, 0, (size_t)obs_dim_agent * na * sizeof(double), nullptr));
    auto at_ws = [ws](size_t off) { return ws + off; };
    int rc = asmenv::launch_observe<double>(
        P, reinterpret_cast<double *>(at_ws(o_p)), reinterpret_cast<double *>(at_ws(o_dp)),
        reinterpret_cast<double *>(at_ws(o_h)), reinterpret_cast<double *>(at_ws(o_g)), reinterpret_cast<int *>(at_ws(o_ng)),
        reinterpret_cast<double *>(at_ws(o_lc)), reinterpret_cast<int *>(at_ws(o_nbr)), reinterpret_cast<int *>(at_ws(o_in)),
        reinterpret_cast<int *>(at_ws(o_near)), reinterpret_cast<double *>(at_ws(o_obs)), nullptr,
        reinterpret_cast<int *>(at_ws(o_sen)), reinterpret_cast<int *>(at_ws(o_occ)), nullptr);
    if (rc) return hip_fail(rc, "asm_get_observation_noalloc kernel launch");
    NOALLOC_HIP(hipMemcpy(obs_io, ws + o_obs, (size_t)obs_dim_agent * na * sizeof(double), hipMemcpyDeviceToHost));
    NOALLOC_HIP(hipMemcpy(nbr_io, ws + o_nbr, na * 6 * sizeof(int), hipMemcpyDeviceToHost));
    NOALLOC_HIP(hipMemcpy(inflags_io, ws + o_in, na * sizeof(int), hipMemcpyDeviceToHost));
    NOALLOC_HIP(hipMemcpy(sensed_io, ws + o_sen, na * P.n_obs * sizeof(int), hipMemcpyDeviceToHost));
    NOALLOC_HIP(hipMemcpy(occupied_io, ws + o_occ, na * P.n_occ * sizeof(int), hipMemcpyDeviceToHost));
#undef NOALLOC_HIP
    return 0;
}

}  // extern "C"
