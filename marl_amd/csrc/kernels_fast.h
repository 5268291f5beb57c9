// k_step_fast<T>: the multi-wave step kernel for fp32 state, n_a <= 32, walls on.
// Part of the single translation unit asm_kernels.hip (included there, inside namespace asmenv, after the
// diagnostic macros); not a stand-alone header.

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
// fl32 |wrap(rx, ry)|^2 for the single-precision neighbour screen of a periodic arena (CPP:705-720 in float).  Where
// the float and the double wrap could disagree (|r| within rounding of the half-width) the offset is about a
// half-width long either way, far outside the sensing range, so the screen's verdict is the same.
__device__ __forceinline__ float wrapped_s32(float rx, float ry, float hw, float hh)
{
    if (rx < -hw) rx += 2 * hw; else if (rx > hw) rx -= 2 * hw;
    if (ry < -hh) ry += 2 * hh; else if (ry > hh) ry -= 2 * hh;
    return __builtin_fmaf(ry, ry, rx * rx);
}

struct FastLds {
    float *px, *py, *vx, *vy;      // [64]   slot = (env half << 5) | agent
    float *g;                      // x rows of the two environments at [0, 2 NGP), their y rows ASM_GY_OFF floats further on
                                   // (a compile-time distance: one address serves both coordinates of a cell); cells >= n_g hold 1e30.
                                   // Entry 2 NGP + l is lane l's own position: the "cell" an empty observation slot reads, so that
                                   // its relative position comes out as +0 without a select
    unsigned *mask;                // [64][MW]     sensed-cell bit mask per agent
    unsigned *occ;                 // [2][2W]      per env: occupancy words, then coverage words
    uint4 *part;                   // [T][64]      per-wave partial (min key, runner-up key, word, ambiguity), then what the statistics need;
                                   //              absent on the lattice path
    unsigned *fin;                 // [2][64]      nearest cell | in_flag << 31; then the list lengths (sparse compaction, sensed-index output)
    unsigned *anyf;                // [2][2]       lattice path: ballots of the second / third wave -- it may have set sensed bits of the agent
    float *rpart;                  // [T][3][64]   per-wave partial reward sums (lattice path: in the area of the shape tables, dead by then)
    unsigned short *sel;           // [64][CAPP]   compacted (sub-sampled) sensed list
};

__device__ __forceinline__ FastLds carve_fast(unsigned char *smem, const DevParams &P, int T, bool lat)
{
    FastLds L;
    float *f = reinterpret_cast<float *>(smem);
    L.px = f; L.py = f + 64; L.vx = f + 128; L.vy = f + 192;
    L.g = f + 256;
    L.mask = reinterpret_cast<unsigned *>(L.g + ASM_GY_OFF + 2 * P.NGP + 64);   // 64: the y halves of the lanes' own-position entries
    L.occ = L.mask + 64 * P.MW;
    L.part = reinterpret_cast<uint4 *>(L.occ + 4 * P.W);
    L.fin = reinterpret_cast<unsigned *>(L.part + (lat ? 0 : 64 * T));
    L.anyf = L.fin + 128;
    if (lat) {
        L.rpart = L.g + 2 * P.NGP + 64;                             // 3 KB of the 3.25 KB the two shape tables occupy until the cell phase is over
        L.sel = reinterpret_cast<unsigned short *>(L.anyf + 4);
    } else {
        L.rpart = reinterpret_cast<float *>(L.fin + 128);
        L.sel = reinterpret_cast<unsigned short *>(L.rpart + 3 * 64 * T);
    }
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
#ifndef ASM_FAST_MINW_LAT
#define ASM_FAST_MINW_LAT 5  // waves per SIMD the default lattice kernel is compiled for (96 registers; its LDS lets five workgroups share a CU)
#endif
#ifndef ASM_SPARSE_MAX
#define ASM_LAT_ITEM_MAX 384 // lattice path: up to this many (agent, row) items per environment the rows are dealt out item by item
#define ASM_SPARSE_MAX 64    // lattice path: up to this many sensing agents of a workgroup's 64, the lists are compacted agent by agent (64: always;
                             // measured as fast as the lane-by-lane loop even when every agent senses cells)
#endif
template <int T, bool PER, int IDX, bool LAT, bool NT = false>   // PER: periodic arena (no walls; wrapped neighbour offsets, ENV:97-101); IDX: index
                                      // lists written (0 none, 1 sensed_index if given, 2 occupied_index too); LAT: the agent x
                                      // cell work runs on the shapes' lattice description (kernels_lattice.h; T = 4 only);
                                      // NT: the observation rows are stored non-temporal (launches whose observations are
                                      // too large to stay in the last-level cache until a later kernel reads them: then
                                      // they should not push the grids and the state out of it either)
__global__ __launch_bounds__(64 * T, (LAT && IDX == 0 && !PER) ? ASM_FAST_MINW_LAT : ASM_FAST_MINW) void k_step_fast(DevParams P, LatArgs lat, float *p_g, float *dp_g, const float *act_g,
                                                      const float *grid_g, const int *ng_g, const double *lcell_g,
                                                      int *nbr_g, int *inflag_g, int *nearest_g, float *obs_g,
                                                      float *rew_g, float *prior_g, int *sensed_g, int *occup_g,
                                                      double *stats_g)
{
#ifdef ASM_DIAG
    const unsigned long long dg_t0 = __builtin_readcyclecounter();   // the wave's first instruction
    unsigned long long dg_t1 = 0, dg_t2 = 0, dg_b = 0, dg_c = 0;
#endif
    // (two structures and 15 pointers, each aligned to 8 bytes: the explicit arguments, which is all this kernel has)
    const int block = kernarg_warm<((sizeof(DevParams) + 7) & ~7) + ((sizeof(LatArgs) + 7) & ~7) + 15 * 8>((int)blockIdx.x);
    extern __shared__ __align__(16) unsigned char smem[];
    // sub is the same for every lane of a wave: read through the scalar unit so that everything derived from
    // it (word and slot ranges, row offsets) stays in scalar registers
    const int tid = threadIdx.x, sub = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, es = l >> 5, i = l & 31;
    const int e = block * 2 + es;
    const bool env_ok = e < P.E, active = env_ok && i < P.n_a;
    const int n_a = P.n_a, NGP = P.NGP, W = P.W, MW = P.MW, cap = P.n_obs;
    const FastLds L = carve_fast(smem, P, T, LAT);
    float *gxe = L.g + (size_t)es * NGP, *gye = gxe + ASM_GY_OFF;
    unsigned *occ_e = L.occ + es * 2 * W, *cov_e = occ_e + W;
    constexpr int SUB_PRIOR = T > 1 ? 1 : 0;
#ifdef ASM_DIAG
    unsigned long long *const g_diag_all = asmenv::g_diag;
    unsigned long long *const g_diag = (g_diag_only && (const void *)g_diag_only != (const void *)p_g) ? nullptr : g_diag_all;   // shadows the global in this kernel
#endif
    STAMP(0);
#ifdef ASM_DIAG
    if (g_diag && l == 0) g_diag[(size_t)blockIdx.x * 32 + 20 + sub] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID of this wave
    if (g_diag && tid == 0) g_diag[(size_t)blockIdx.x * 32 + 24] = __builtin_amdgcn_s_getreg((3 << 11) | 20);     // XCC_ID
#endif

#ifdef ASM_EXITS
    const int xat = g_exit_at;
#else
    constexpr int xat = 0;
#endif
    // ---- P0: grid and state into LDS (all global loads issued before the first LDS write) ----
    // Every load of this phase that does not depend on another one is issued before anything waits: a load inside a
    // branch ("environment e0 + 1 exists ? load : 0") gets its own wait inside that branch, and a chain of such waits was
    // six round trips to memory long.  Hence indices clamped to the last environment instead of branches, the values
    // discarded afterwards where the environment does not exist, and the widest loads (the grid rows) first.
    const int e0 = block * 2;
    __builtin_assume(e0 < P.E);                                     // the grid has (E + 1) / 2 workgroups: loads guarded by it need no branch
    // Per-agent arrays are addressed as a workgroup-uniform base (the first environment of the pair, scalar registers) plus a
    // small per-lane element offset: no 64-bit vector address arithmetic, and nothing 64 bits wide kept per lane for it.
    const int la = es * n_a + i, la2 = es * 2 * n_a + i;            // lane's agent among the pair's agents; its slot in a [2][n_a] array pair
    float *const p_w = p_g + (size_t)e0 * 2 * n_a, *const dp_w = dp_g + (size_t)e0 * 2 * n_a;
    const float *const act_w = act_g ? act_g + (size_t)e0 * 2 * n_a : nullptr;
    float *const prior_w = prior_g ? prior_g + (size_t)e0 * 2 * n_a : nullptr;
    int *const nbr_w = nbr_g + (size_t)e0 * n_a * 6, *const inflag_w = inflag_g + (size_t)e0 * n_a, *const nearest_w = nearest_g + (size_t)e0 * n_a;
    float *const rew_w = rew_g ? rew_g + (size_t)e0 * n_a : nullptr;
    const int e_last = P.E - 1, e0c = e0 < e_last ? e0 : e_last, e1c = e0 + 1 < e_last ? e0 + 1 : e_last, ec = es ? e1c : e0c;
    [[maybe_unused]] float4 gv[4];
    if constexpr (T == 4) {                                         // one coordinate row per wave: its loads go first
        const float *src = grid_g + ((size_t)((sub >> 1) ? e1c : e0c) * 2 + (sub & 1)) * NGP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = (q * 64 + l) * 4;
            gv[q] = make_float4(1e30f, 1e30f, 1e30f, 1e30f);
            if (q * 256 < NGP && c < NGP) gv[q] = *reinterpret_cast<const float4 *>(src + c);
        }
    }
    // Everything the dynamics phase reads from memory is requested here too: the state and the actions (first wave) and
    // the prior's neighbour list, in-shape flag and nearest cell of the previous observation (second wave).
    float act_x = 0.f, act_y = 0.f;
    int nb_pre[6] = {-1, -1, -1, -1, -1, -1}, inf_pre = 0, near_pre = 0;
    const bool prior_on = !P.observe_only && P.prior && prior_g;
    float sa = -1e30f, sb = -1e30f, sc = 0.f, sd = 0.f;              // unused lanes: far from every agent AND from the padding cells (+1e30)
    if (sub == 0 && active) {
        sa = p_w[la2]; sb = p_w[la2 + n_a]; sc = dp_w[la2]; sd = dp_w[la2 + n_a];
        if (!P.observe_only) { act_x = act_w[la2]; act_y = act_w[la2 + n_a]; }
    }
    if (sub == SUB_PRIOR && active && prior_on) {
#pragma unroll
        for (int q = 0; q < 6; ++q) nb_pre[q] = nbr_w[la * 6 + q];
        inf_pre = inflag_w[la]; near_pre = nearest_w[la];   // (unused by a step that recomputes the target)
    }
    const double l_cell_r = lcell_g[ec];
    [[maybe_unused]] double xf_c = 1, xf_s = 0, xf_ox = 0, xf_oy = 0;
    if constexpr (LAT) { xf_c = lat.xform[(size_t)ec * 4]; xf_s = lat.xform[(size_t)ec * 4 + 1]; xf_ox = lat.xform[(size_t)ec * 4 + 2]; xf_oy = lat.xform[(size_t)ec * 4 + 3]; }
    int ng0r = ng_g[e0c], ng1r = ng_g[e1c];
    [[maybe_unused]] int sid0r = 0, sid1r = 0;
    if constexpr (LAT) { sid0r = lat.shape_id[e0c]; sid1r = lat.shape_id[e1c]; }
    // (one point at which all four are needed: otherwise each pair is fetched where it is first used, a round trip apart)
    // (readfirstlane: free for a value the scalar unit loaded; where the compiler took the vector path it moves it over)
    ng0r = __builtin_amdgcn_readfirstlane(ng0r); ng1r = __builtin_amdgcn_readfirstlane(ng1r);
    sid0r = __builtin_amdgcn_readfirstlane(sid0r); sid1r = __builtin_amdgcn_readfirstlane(sid1r);
    asm("" : "+s"(ng0r), "+s"(ng1r), "+s"(sid0r), "+s"(sid1r));
    const int ng0= e0 < P.E ? clamp_ng(ng0r, P.n_g_max) : 0, ng1 = e0 + 1 < P.E ? clamp_ng(ng1r, P.n_g_max) : 0;
    const int n_g = es ? ng1 : ng0;
#ifdef ASM_DIAG
    asm volatile("" :: "s"(ng0));
    const unsigned long long dg_a = __builtin_readcyclecounter();   // the first loaded value (n_g) is there
#endif
    const double l_cell = env_ok ? l_cell_r : 1.0;
    // lattice path: the two environments' shape tables (row windows + boundary list, 1664 bytes each) go into the unused
    // part of the grid area, between the x rows and the y rows
    static_assert(!LAT || T == 4, "the lattice path is scheduled for four waves");
    constexpr int LAT_TAB_BYTES = 8 * ASM_LAT_ROWS * ASM_LAT_CHUNKS + 2 * ASM_LAT_NB, LAT_TAB_VEC = LAT_TAB_BYTES / 16;
    unsigned char *lat_tabs = reinterpret_cast<unsigned char *>(L.g + 2 * NGP + 64);
    [[maybe_unused]] int lat_cols = 0, lat_rows = 0, lat_nb0 = 0, lat_nb1 = 0;
    [[maybe_unused]] double lat_x0 = 0, lat_y0 = 0, lat_invl = 1;
    [[maybe_unused]] uint4 lat_tv = make_uint4(0, 0, 0, 0);
    // The shape ids first: everything else of the lattice hangs on them.  The dependent loads are issued after the grid
    // loads below and their data is not needed before P3 (the tables reach LDS after the first barrier, the compacted
    // boundary cells are built between the two barriers of the dynamics phase), so this second round trip to memory
    // runs under the dynamics instead of delaying the first barrier.
    [[maybe_unused]] int sid0 = 0, sid1 = 0;
    if constexpr (LAT) {
        sid0 = e0 < P.E ? sid0r : 0; sid1 = e0 + 1 < P.E ? sid1r : 0;
    }
    auto lat_loads = [&]() {
#ifdef ASM_DIAG
        dg_t1 = __builtin_readcyclecounter();                      // grid loads issued
        asm volatile("" :: "s"(sid0));
        dg_b = __builtin_readcyclecounter();                        // shape ids there
#endif
        sid0 = sid0 < 0 ? 0 : (sid0 >= lat.n_shapes ? lat.n_shapes - 1 : sid0);
        sid1 = sid1 < 0 ? 0 : (sid1 >= lat.n_shapes ? lat.n_shapes - 1 : sid1);
        const int nb0r = lat.shapes[sid0].n_boundary, nb1r = lat.shapes[sid1].n_boundary;      // (sid clamped above: always a valid shape)
        lat_nb0 = e0 < P.E ? nb0r : 0; lat_nb1 = e0 + 1 < P.E ? nb1r : 0;
        const asm_lattice_shape *sh = lat.shapes + (es ? sid1 : sid0);
        lat_cols = sh->cols; lat_rows = sh->rows; lat_x0 = sh->x0; lat_y0 = sh->y0; lat_invl = 1.0 / sh->l_cell;
        if (tid < 2 * LAT_TAB_VEC) {
            const int h = tid >= LAT_TAB_VEC ? 1 : 0;
            lat_tv = reinterpret_cast<const uint4 *>(lat.shapes + (h ? sid1 : sid0))[tid - h * LAT_TAB_VEC];
        }
    };
    {
        // The four coordinate rows (2 environments x {x, y}) are dealt out to the waves, so the row and its n_g are
        // wave-uniform and a lane's float4 index is just l, l + 64, ...; cells beyond n_g become far-away padding.
        for (int row = sub; row < 4; row += T) {
            const int h = row >> 1, ngh = h ? ng1 : ng0;
            const bool row_ok = e0 + h < P.E;
            const float *src = grid_g + ((size_t)(e0 + h) * 2 + (row & 1)) * NGP;
            float *dst = L.g + (size_t)(row & 1) * ASM_GY_OFF + (size_t)h * NGP;
            for (int base = 0; base < NGP; base += 64 * 4 * 4) {        // (one trip: a row has at most 1024 cells)
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = base + (q * 64 + l) * 4;
                    if constexpr (T == 4) {
                        v[q] = gv[q];                               // requested at the top (a missing environment has n_g = 0: all padding)
                    } else {
                        v[q] = make_float4(1e30f, 1e30f, 1e30f, 1e30f);
                        if (base + q * 256 < NGP && c < NGP && row_ok) v[q] = *reinterpret_cast<const float4 *>(src + c);
                    }
                }
                if constexpr (LAT) if (base == 0) lat_loads();      // T = 4: every wave passes here exactly once
#ifdef ASM_DIAG
                if (base == 0) { dg_c = __builtin_readcyclecounter(); asm volatile("" :: "v"(v[0].x)); dg_t2 = __builtin_readcyclecounter(); }
#endif
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
        if constexpr (LAT) {
            for (int w = tid; w < 64 * MW; w += 64 * T) L.mask[w] = 0;   // the sensed masks are built by OR-ing runs of cells
            if (tid < 4) L.anyf[tid] = 0;
        }
    }
#ifdef ASM_DIAG
    if (g_diag && tid == 0) {                                       // first wave: its start, loads issued, first grid data there, arrival at the barrier
        g_diag[(size_t)blockIdx.x * 32 + 25] = __builtin_readcyclecounter(); g_diag[(size_t)blockIdx.x * 32 + 26] = dg_t0;
        g_diag[(size_t)blockIdx.x * 32 + 27] = dg_t1; g_diag[(size_t)blockIdx.x * 32 + 28] = dg_t2;
        g_diag[(size_t)blockIdx.x * 32 + 19] = dg_a; g_diag[(size_t)blockIdx.x * 32 + 29] = dg_b; g_diag[(size_t)blockIdx.x * 32 + 30] = dg_c;
    }
#endif
    __syncthreads();
    STAMP(1);
    EXIT_AT(1);
    if constexpr (LAT)
        if (tid < 2 * LAT_TAB_VEC) reinterpret_cast<uint4 *>(lat_tabs)[tid] = lat_tv;   // published by the next barrier
    // lattice path: the boundary cells' coordinates, compacted (the candidates of the nearest-cell pass).  Waves 2 and 3
    // gather them, one environment each, between the two barriers of the dynamics phase (the first publishes the tables).
    [[maybe_unused]] float *lat_cbx = reinterpret_cast<float *>(L.sel), *lat_cby = lat_cbx + 2 * ASM_LAT_NB;   // [2][NB] each, aliasing the slot lists
    auto lat_compact = [&]() {
        if constexpr (LAT) {
            if (sub >= 2) {
                const int h = sub - 2, nbh = h ? lat_nb1 : lat_nb0;
                const unsigned short *bl = reinterpret_cast<const unsigned short *>(lat_tabs + h * LAT_TAB_BYTES + 8 * ASM_LAT_ROWS * ASM_LAT_CHUNKS);
                const float *gxh = L.g + (size_t)h * NGP;
#pragma unroll
                for (int k = l; k < ASM_LAT_NB; k += 64) {
                    float x = 1e30f, y = 1e30f;
                    if (k < nbh) { const int c = bl[k]; x = gxh[c]; y = gxh[c + ASM_GY_OFF]; }
                    lat_cbx[h * ASM_LAT_NB + k] = x; lat_cby[h * ASM_LAT_NB + k] = y;
                }
            }
        }

    };
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
        if constexpr (PER) {
            // ENV:442-443: only agent 0's row of the distance matrix is wrapped; the force loop never reads it, but
            // the contact count (WRAP:131-145) does
            if (P.want_stats && i == 0)
                for (int k = 1; k < n_a; ++k) {
                    double wx = (double)L.px[es * 32 + k] - px, wy = (double)L.py[es * 32 + k] - py;
                    if (wx * wx + wy * wy < P.T_contact) continue;              // counted above
                    wrap_rel(wx, wy, P.half_w, P.half_h);
                    if (wx * wx + wy * wy < P.T_contact) ++contacts;
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
        const double ux = (double)(sens * act_x);                                // ENV:627,633
        const double uy = (double)(sens * act_y);
        const double Fx = PER ? ux + fx : ux + fx + wfx + dfx, Fy = PER ? uy + fy : uy + fy + wfy + dfy;   // ENV:633
        double ax = Fx, ay = Fy;
        if (P.mass != 1.0) { ax = Fx / P.mass; ay = Fy / P.mass; }               // ENV:638 (x / 1 == x)
        double v0 = vx + ax * P.dt, v1 = vy + ay * P.dt;                         // ENV:641
        v0 = fmin(fmax(v0, -P.vel_max), P.vel_max); v1 = fmin(fmax(v1, -P.vel_max), P.vel_max);   // ENV:642
        double qx = px + v0 * P.dt, qy = py + v1 * P.dt;                         // ENV:645
        if constexpr (PER) {                                                     // ENV:646-647, 778-781
            if (qx < -P.half_w) qx += 2 * P.half_w;
            if (qx > P.half_w) qx -= 2 * P.half_w;
            if (qy < -P.half_h) qy += 2 * P.half_h;
            if (qy > P.half_h) qy -= 2 * P.half_h;
        }
        npx = (float)qx; npy = (float)qy;
        nvx = (float)v0; nvy = (float)v1;
    }
    if (sub == SUB_PRIOR && active && prior_on) {                   // ENV:600-619: old state, previous neighbour list
        const double px = L.px[l], py = L.py[l], vx = L.vx[l], vy = L.vy[l];
        int nb[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) nb[q] = nb_pre[q];
        // (the flag is handed over through an empty asm statement: otherwise its comparison is computed where the flag was
        // requested, in the load phase, and every load issued before it is waited for there)
        int inf_o = inf_pre, near = near_pre;
        asm volatile("" : "+v"(inf_o), "+v"(near));
        bool inf = inf_o != 0;
        if (P.recompute) {
            double smin; cell_pass<float, false>(P, true, px, py, gxe, gye, l, nullptr, nullptr, nullptr, smin, near);
            inf = sqrt(smin) < sqrt(2.0) * l_cell / 2;
        }
        const double tx = inf ? px : (double)gxe[near], ty = inf ? py : (double)gye[near];      // CPP:896-901
        double f0, f1;
        prior_policy<float>(P, px, py, vx, vy, tx, ty, nb, L.px + es * 32, L.py + es * 32, L.vx + es * 32,
                            L.vy + es * 32, f0, f1);
        prior_w[la2] = (float)f0; prior_w[la2 + n_a] = (float)f1;
    }
    STAMP(2);
    __syncthreads();                                                // all reads of the old state done
    lat_compact();
    if (sub == 0 && active) {
        L.px[l] = npx; L.py[l] = npy; L.vx[l] = nvx; L.vy[l] = nvy;
        p_w[la2] = npx; p_w[la2 + n_a] = npy; dp_w[la2] = nvx; dp_w[la2 + n_a] = nvy;
    }
    __syncthreads();
    } else if constexpr (LAT) {
        __syncthreads();
        lat_compact();
        __syncthreads();
    }
    STAMP(3);
    EXIT_AT(2);

    // ---- P3: agent x cell pass on the NEW state, words split over the waves ------------
    const float pxf = L.px[l], pyf = L.py[l];
    const double px = pxf, py = pyf;
    if (sub == T - 1) { L.g[2 * NGP + l] = pxf; L.g[ASM_GY_OFF + 2 * NGP + l] = pyf; }   // read in P6, several barriers on
    // obs[row * n_a + agent] of this lane's environment, written with buffer stores: resource = the block of the
    // workgroup's two environments, vector offset = the lane's column, scalar offset = the row; a row store then
    // needs no vector address arithmetic
    const __amdgpu_buffer_rsrc_t obs_rs =
        __builtin_amdgcn_make_buffer_rsrc(obs_g + (size_t)e0 * P.obs_dim * n_a, 0, 0x7fffffff, 0x00020000);
    const int olb = (es * P.obs_dim * n_a + i) * 4;
    auto ostore = [&](int row, float v) {
#ifdef ASM_EXP_ALIGNED
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), obs_rs, olb, (row * n_a * 4) & ~127, NT ? 2 : 0);   // timing experiment only: wrong rows
#else
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), obs_rs, olb, row * n_a * 4, NT ? 2 : 0);
#endif
    };
    int j6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) j6[q] = -1;
    double nn_min = __builtin_inf();
    [[maybe_unused]] double dmin_lat = 0.0;                         // lattice path: the nearest cell's distance (fourth wave)
    // Formation statistics (TRAIN:132-135; WRAP:48-165) in three independent pieces, each added by the wave that holds its
    // inputs (one adding lane per entry and launch either way):
    //   nearest-neighbour uniformity + contact pairs + step count, mean distance to the target cell, coverage.
    auto stats_neighbours = [&](double nn, int cn) {
        if (active && nn == 0.0) {                                  // an agent at this very position: WRAP:93-94 drops exact zeros
            nn = __builtin_inf();
#pragma nounroll
            for (int j = 0; j < n_a; ++j) {
                const double ex = (double)L.px[es * 32 + j] - px, ey = (double)L.py[es * 32 + j] - py, su = ex * ex + ey * ey;
                if (su != 0 && su < nn) nn = su;
            }
        }
        const double m = active ? sqrt(nn) : 0.0;
#pragma unroll
        for (int mm = 16; mm >= 1; mm >>= 1) cn += __shfl_xor(cn, mm, 32);
        const double ms = half_sum(m);
        const double lo = half_min(active ? m : __builtin_inf()), hi = half_max(active ? m : -__builtin_inf());
        const double mean = ms / n_a, dv = active ? m - mean : 0.0;
        const double var = half_sum(dv * dv) / n_a;
        if (i == 0 && env_ok) {
            double *st = stats_g + (size_t)e0 * 8 + es * 8;
            stat_add(&st[3], (var - lo) / (hi - lo));               // WRAP:98-99
            stat_add(&st[4], (double)cn); stat_add(&st[6], 1.0);
        }
    };
    auto stats_target = [&](double dm) {
        const double ds = half_sum(active ? dm : 0.0);
        if (i == 0 && env_ok) stat_add(stats_g + (size_t)e0 * 8 + es * 8 + 5, ds / n_a);
    };
    auto stats_coverage = [&]() {
        int ci = i < W ? __popc(cov_e[i]) : 0;
#pragma unroll
        for (int mm = 16; mm >= 1; mm >>= 1) ci += __shfl_xor(ci, mm, 32);
        if (i == 0 && env_ok) stat_add(stats_g + (size_t)e0 * 8 + es * 8 + 2, n_g > 0 ? (double)ci / n_g : 0.0);   // WRAP:70; an environment without cells covers nothing
    };
    [[maybe_unused]] const float hw32 = (float)P.half_w, hh32 = (float)P.half_h;
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
                // sensing uses the wrapped offset in a periodic arena (CPP:93-95), the uniformity statistic the plain one
                cand = __builtin_amdgcn_alignbit(cand, __float_as_uint((PER ? wrapped_s32(rx, ry, hw32, hh32) : s) - P.Tsen_hi_f), 31);
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
            unsigned x = (__float_as_uint(PER ? wrapped_s32(rx, ry, hw32, hh32) : __builtin_fmaf(ry, ry, rx * rx)) & ~31u) | (unsigned)j;
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
                double s6[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) { s6[q] = __builtin_inf(); j6[q] = -1; }
                cand = cand_all;
                while (cand) {                                      // ascending j: ties keep the lower index
                    const int j = __ffs(cand) - 1; cand &= cand - 1;
                    double dx = (double)L.px[es * 32 + j] - px, dy = (double)L.py[es * 32 + j] - py;
                    if constexpr (PER) wrap_rel(dx, dy, P.half_w, P.half_h);   // CPP:93-95
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
                if constexpr (PER) {
                    double dx = (double)L.px[es * 32 + j] - px, dy = (double)L.py[es * 32 + j] - py;
                    wrap_rel(dx, dy, P.half_w, P.half_h);
                    rx = (float)dx; ry = (float)dy;
                }
                wx = L.vx[es * 32 + j] - vxf; wy = L.vy[es * 32 + j] - vyf;      // CPP:86
            }
            nbr_w[la * 6 + q] = j;                                  // CPP:103-105
            const int r = r0 + 4 * q;
            ostore(r, rx); ostore(r + 1, ry); ostore(r + 2, wx); ostore(r + 3, wy);
        }
    }
    STAMP(4);
    if constexpr (LAT) {
        // Lattice path (kernels_lattice.h).  Four jobs, one per wave, all done when the barrier below is reached:
        // wave 0 the k-NN pass above; waves 1 and 2 the 15 lattice rows the sensing disc can cross (odd / even rows);
        // wave 3 the nearest cell from start to finish -- the four lattice sites around the agent, the screen of the
        // compacted boundary cells, the exact comparison of the candidates -- with the in-shape flag and the target
        // block.  No wave waits for another one's partial result, and nothing runs on one wave alone behind the barrier.
        unsigned lat_any = 0;                                       // this wave may have set sensed bits of the lane
        if (sub != 0) {
            LatLane A;
            {
                const double rxp = px - xf_ox, ryp = py - xf_oy;                        // inverse of assembly.py:172-173, 182
                const double cxl = xf_c * rxp - xf_s * ryp, cyl = xf_s * rxp + xf_c * ryp;
                double ud = (cxl - lat_x0) * lat_invl, vd = (cyl - lat_y0) * lat_invl;  // lattice units
                ud = fmin(fmax(ud, -1000.0), 1000.0); vd = fmin(fmax(vd, -1000.0), 1000.0);   // unused lanes sit at -1e30
                const double ur = rint(ud), vr = rint(vd);
                A.ub = (int)ur; A.vb = (int)vr; A.fu = (float)(ud - ur); A.fv = (float)(vd - vr);
                A.inr = active && A.ub >= -7 && A.ub <= lat_cols + 6 && A.vb >= -7 && A.vb <= lat_rows + 6;
                const double Rd = P.d_sen * lat_invl, Rod = P.r_avoid * 0.5 * lat_invl;
                const float invl_f = (float)lat_invl;
                A.R2 = (float)(Rd * Rd); A.Ro2 = (float)(Rod * Rod);
                A.band = P.lat_cb * (float)Rd * invl_f + 4e-6f * A.R2;
                A.bando = P.lat_cb * (float)Rod * invl_f + 4e-6f * A.Ro2;
            }
            const unsigned char *tab = lat_tabs + es * LAT_TAB_BYTES;
            // The lattice rows an agent can touch at all: rows -7 .. 7 around its nearest site that exist in the shape's
            // table, and none when the agent is out of reach of the bounding box.  Under random actions most agents are
            // out of reach, so the two waves deal out (agent, row) items, one per lane and pass, instead of walking all
            // 15 rows with every lane: per half-wave (one environment each) a prefix sum of the agents' row counts, a
            // binary search of the lane's item number in it, then the row for that agent.  When the items would take more
            // passes than the row loop has rows' worth of work (many agents inside their shape), the row loop runs.
            int it_klo = 0, it_cnt = 0, it_incl = 0, it_n0 = 0, it_n1 = 0;
            if (sub != 3) {
                it_klo = -A.vb > -7 ? -A.vb : -7;
                const int it_khi = lat_rows - 1 - A.vb < 7 ? lat_rows - 1 - A.vb : 7;
                it_cnt = A.inr ? it_khi - it_klo + 1 : 0;
                it_cnt = it_cnt < 0 ? 0 : it_cnt;
                it_incl = it_cnt;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) { const int o = __shfl_up(it_incl, d, 32); it_incl += i >= d ? o : 0; }
                it_n0 = __builtin_amdgcn_readlane(it_incl, 31); it_n1 = __builtin_amdgcn_readlane(it_incl, 63);
            }
            const int it_nmax = it_n0 > it_n1 ? it_n0 : it_n1;
            const bool by_items = sub != 3 && it_nmax <= ASM_LAT_ITEM_MAX && 2 * 64 * P.capp + 32 >= 16 * ASM_LAT_NB + 1536;
            if (by_items) {
                // tables behind the compacted boundary coordinates, in the area of the slot lists (written after the next
                // barrier); both waves write the same values, each reads after its own writes
                int *t_incl = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(L.sel) + 16 * ASM_LAT_NB) + es * 32;
                int *t_off = t_incl + 64;
                uint4 *t_site = reinterpret_cast<uint4 *>(t_incl - es * 32 + 128) + es * 32;
                t_incl[i] = it_incl; t_off[i] = it_klo - (it_incl - it_cnt);
                t_site[i] = make_uint4((unsigned)A.ub, (unsigned)A.vb, __float_as_uint(A.fu), __float_as_uint(A.fv));
                const int it_n = es ? it_n1 : it_n0;
                for (int t0 = 0; t0 < it_nmax; t0 += 64) {
                    const int t = t0 + (sub - 1) * 32 + i;
                    const bool valid = t < it_n;
                    int a = 0;
#pragma unroll
                    for (int step = 16; step >= 1; step >>= 1) a += t_incl[a + step - 1] <= t ? step : 0;   // agents whose items end at or before t
                    const int k = t + t_off[a];
                    const uint4 site = t_site[a];
                    LatLane B = A;
                    B.ub = (int)site.x; B.vb = (int)site.y; B.fu = __uint_as_float(site.z); B.fv = __uint_as_float(site.w);
                    B.inr = valid;
                    unsigned *mrow = L.mask + (size_t)(es * 32 + a) * MW;
                    unsigned r = lat_row(P, B, tab, k, k >= -P.lat_KO && k <= P.lat_KO, lat_cols, lat_rows, mrow, occ_e, cov_e);
                    if (r & 5u) atomicOr(&L.anyf[es], 1u << a);     // a run with cells, or a row settled cell by cell below
                    while (__ballot((r & 3u) != 0)) {               // rare: a run end within rounding distance of its circle
                        if (r & 3u) {
                            const double pxa = (double)L.px[es * 32 + a], pya = (double)L.py[es * 32 + a];
                            if (r & 1u) { lat_row_exact<true>(P, tab, B.vb, k, B.ub, 7, lat_cols, lat_rows, pxa, pya, gxe, gye, mrow, nullptr, nullptr); r &= ~1u; }
                            else { lat_row_exact<false>(P, tab, B.vb, k, B.ub, 4, lat_cols, lat_rows, pxa, pya, gxe, gye, nullptr, occ_e, cov_e); r &= ~2u; }
                        }
                    }
                }
            } else
            if (sub != 3) {
                unsigned *mrow = L.mask + (size_t)l * MW;
                unsigned und = 0;                                   // bits 2t, 2t + 1: sensed / occupancy run of row t undecided
                const int kfirst = -7 + (sub - 1), nrows = sub == 1 ? 8 : 7;    // wave-uniform
#pragma unroll 1
                for (int t = 0; t < nrows; ++t) {
                    const int k = kfirst + 2 * t;
                    const unsigned r = lat_row(P, A, tab, k, k >= -P.lat_KO && k <= P.lat_KO, lat_cols, lat_rows, mrow, occ_e, cov_e);
                    und |= (r & 3u) << (2 * t);
                    lat_any |= r;                                   // bit 2 (a run with cells) or bit 0 (a row settled cell by cell)
                }
                lat_any &= 5u;
                while (__ballot(und != 0)) {                        // rare: a run end within rounding distance of its circle
                    if (und) {
                        const int b = __ffs(und) - 1; und &= und - 1;
                        const int k = kfirst + 2 * (b >> 1);
                        if (b & 1) lat_row_exact<false>(P, tab, A.vb, k, A.ub, 4, lat_cols, lat_rows, px, py, gxe, gye, nullptr, occ_e, cov_e);
                        else lat_row_exact<true>(P, tab, A.vb, k, A.ub, 7, lat_cols, lat_rows, px, py, gxe, gye, mrow, nullptr, nullptr);
                    }
                }
            } else {
                // ---- nearest cell (CPP:863-913), in-shape flag, target block ----
                // The four sites around the agent, exactly.  A site that is none of the four is at least one lattice
                // pitch away in one coordinate, so once one of the four holds a cell closer than 0.99 pitches no other
                // cell can be nearer (or as near): such an agent -- inside its shape or at its rim -- needs no screen.
                double best = __builtin_inf(); int imin = 0x7FFFFFFF;
                if (A.inr) {
                    const int du = A.fu < 0.f ? -1 : 1, dw = A.fv < 0.f ? -1 : 1;
                    lat_site_exact(tab, A.ub, A.vb, lat_cols, lat_rows, px, py, gxe, gye, best, imin);
                    lat_site_exact(tab, A.ub + du, A.vb, lat_cols, lat_rows, px, py, gxe, gye, best, imin);
                    lat_site_exact(tab, A.ub, A.vb + dw, lat_cols, lat_rows, px, py, gxe, gye, best, imin);
                    lat_site_exact(tab, A.ub + du, A.vb + dw, lat_cols, lat_rows, px, py, gxe, gye, best, imin);
                }
                const bool scr = active && !(best < 0.98 * (l_cell * l_cell));
                if (__ballot(scr)) {
                    // screen of the compacted boundary cells: per 32-cell word the smallest s32; candidates are the cells
                    // within two units of 32 ulps of the overall smallest (the screen is off by a few ulps), compared in
                    // double in ascending cell index with a strict '<': the reference's first minimum (CPP:889)
                    const int nbw = ((lat_nb0 > lat_nb1 ? lat_nb0 : lat_nb1) + 31) >> 5;    // wave-uniform
                    const f32x2 p2x = {pxf, pxf}, p2y = {pyf, pyf};
                    const float *bxe = lat_cbx + es * ASM_LAT_NB, *bye = lat_cby + es * ASM_LAT_NB;
                    const unsigned short *lat_bl = reinterpret_cast<const unsigned short *>(tab + 8 * ASM_LAT_ROWS * ASM_LAT_CHUNKS);
                    // one word at a time: its smallest s32, the two smallest of those over all words and the first word
                    // that holds the smallest
                    auto word_key = [&](int w) {
                        unsigned k1 = 0xFFFFFFFFu;
                        const float4 *X = reinterpret_cast<const float4 *>(bxe + w * 32);
                        const float4 *Y = reinterpret_cast<const float4 *>(bye + w * 32);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 x4 = X[q], y4 = Y[q];
                            const f32x2 xa = {x4.x, x4.y}, xb = {x4.z, x4.w}, ya = {y4.x, y4.y}, yb = {y4.z, y4.w};
                            const f32x2 rxa = xa - p2x, rxb = xb - p2x, rya = ya - p2y, ryb = yb - p2y;
                            const f32x2 sa = __builtin_elementwise_fma(rya, rya, rxa * rxa);
                            const f32x2 sb = __builtin_elementwise_fma(ryb, ryb, rxb * rxb);
                            k1 = umin3(k1, __float_as_uint(sa.x), __float_as_uint(sa.y));
                            k1 = umin3(k1, __float_as_uint(sb.x), __float_as_uint(sb.y));
                        }
                        return k1;
                    };
                    unsigned G1 = 0xFFFFFFFFu, G2 = 0xFFFFFFFFu, G1w = 0;
#pragma unroll 1
                    for (int w = 0; w < nbw; ++w) {
                        const unsigned k1 = word_key(w);
                        if (k1 < G1) G1w = (unsigned)w;             // strict: the first word holding the smallest s32
                        G2 = umed3(G1, G2, k1);
                        G1 = G1 < k1 ? G1 : k1;
                    }
                    const unsigned lo = G1 >> 5, lim = (lo << 5) + 64u;     // candidate <=> bits(s32) < lim
                    unsigned minw = 1u << G1w;
                    if (__ballot(scr && !((G2 >> 5) - lo >= 2u))) {         // rare: another word's minimum is that close
#pragma unroll 1
                        for (int w = 0; w < nbw; ++w) if ((word_key(w) >> 5) - lo < 2u) minw |= 1u << w;
                    }
                    minw = scr ? minw : 0u;
                    double bs = __builtin_inf(); int bi = 0x7FFFFFFF;
                    while (minw) {
                        const int w = __ffs(minw) - 1; minw &= minw - 1;
                        const float4 *X = reinterpret_cast<const float4 *>(bxe + w * 32);
                        const float4 *Y = reinterpret_cast<const float4 *>(bye + w * 32);
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
                            const double dx = (double)bxe[c] - px, dy = (double)bye[c] - py;
                            const double sd = dx * dx + dy * dy;
                            if (sd < bs) { bs = sd; bi = (int)lat_bl[c]; }      // the list ascends in cell index
                        }
                    }
                    // smaller distance wins, equal distances go to the lower cell index
                    if (bs < best || (bs == best && bi < imin)) { best = bs; imin = bi; }
                }
                if (active) {
                    imin = imin == 0x7FFFFFFF ? 0 : imin;
                    const double dmin3 = sqrt(best);
                    const bool in_flag = dmin3 < sqrt(2.0) * l_cell / 2;    // CPP:894
                    L.fin[l] = (unsigned)imin | (in_flag ? 0x80000000u : 0u);   // imin < 1024
                    inflag_w[la] = in_flag ? 1 : 0;                         // CPP:139
                    nearest_w[la] = imin;
                    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;           // CPP:141-142, 896-901 (x - x == 0)
                    if (!in_flag) { t0 = gxe[imin] - pxf; t1 = gye[imin] - pyf; t2 = 0.f - L.vx[l]; t3 = 0.f - L.vy[l]; }
                    const int tb = P.obs_dim - (2 + cap) * 2;               // CPP:304
                    ostore(tb + 0, t0); ostore(tb + 1, t1);
                    ostore(tb + 2, t2); ostore(tb + 3, t3);
                    dmin_lat = dmin3;
                } else {
                    L.fin[l] = 0;
                }
                L.fin[64 + l] = 0;                                  // list lengths of the sparse compaction (P5); empty lists stay 0
                if (P.want_stats) stats_target(dmin_lat);
            }
        } else if (P.want_stats) {
            stats_neighbours(nn_min, contacts);
        }
        if (sub == 1 || sub == 2) {                                  // (the item passes OR their agents' bits into words 0 and 1)
            const unsigned long long any = __ballot(lat_any != 0);
            if (l == 0) { atomicOr(&L.anyf[(sub - 1) * 2], (unsigned)any); atomicOr(&L.anyf[(sub - 1) * 2 + 1], (unsigned)(any >> 32)); }
        }
    } else {
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

    // ---- P4 (wave 0; all-cells path only -- on the lattice path the fourth wave has done this in P3) ----
    // nearest cell, in-shape flag, target block
    double dmin = 0.0;
    if constexpr (!LAT) {
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
                    if (sd < smin) { smin = sd; imin = c; }         // ascending cell index
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
            L.fin[l] = (unsigned)imin | (in_flag ? 0x80000000u : 0u);   // imin < 1024
            inflag_w[la] = in_flag ? 1 : 0;                         // CPP:139
            nearest_w[la] = imin;
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;           // CPP:141-142, 896-901 (x - x == 0)
            if (!in_flag) { t0 = gxe[imin] - pxf; t1 = gye[imin] - pyf; t2 = 0.f - L.vx[l]; t3 = 0.f - L.vy[l]; }
            const int tb = P.obs_dim - (2 + cap) * 2;               // CPP:304
            ostore(tb + 0, t0); ostore(tb + 1, t1);
            ostore(tb + 2, t2); ostore(tb + 3, t3);
        } else {
            L.fin[l] = 0;
        }
        if (T > 1 && P.want_stats)                                  // wave 1 computes the formation statistics in P7
            L.part[l] = make_uint4((unsigned)contacts, 0u, (unsigned)__double2loint(nn_min), (unsigned)__double2hiint(nn_min));
        if (T > 1 && P.want_stats)
            L.part[(T - 1) * 64 + l] = make_uint4((unsigned)__double2loint(dmin), (unsigned)__double2hiint(dmin), 0u, 0u);
    }
    STAMP(13);
    __syncthreads();
    }
    STAMP(14);
    EXIT_AT(4);

    // ---- P5: compaction of the sensed lists into LDS -----------------------------------
    const unsigned fin = L.fin[l];
    const bool in_flag = (fin >> 31) != 0;
    const unsigned *mym = L.mask + (size_t)l * MW;
    const unsigned inm = in_flag ? 0xFFFFFFFFu : 0u;                // CPP:155: occupied cells drop out only inside the shape
    const int k0 = sub * cap / T, k1 = (sub + 1) * cap / T;         // this wave's observation slots (P6)
    unsigned short *mysel = L.sel + (size_t)l * P.capp;
    int n_after = 0, n_out = 0;
    // Lattice path: the waves of P3 know which agents sense anything at all.  When those are few (most agents are
    // far from their shape under random actions) the lists are compacted agent by agent instead of lane by lane:
    // the wave's cost then follows the number of cells actually listed, not the longest list among its 64 lanes.
    bool sparse = false;
    [[maybe_unused]] unsigned long long nzm = 0;
    if constexpr (LAT) {
        nzm = (unsigned long long)(L.anyf[0] | L.anyf[2]) | ((unsigned long long)(L.anyf[1] | L.anyf[3]) << 32);
        sparse = __popcll(nzm) <= ASM_SPARSE_MAX;
    }
    if (sparse) {
        // The non-empty agents are dealt out to the waves.  For one agent the 64 lanes test 64 cells (two mask words)
        // per pass; a listed cell's slot is its rank among the listed cells (ballot prefix count), or, when the list
        // is sub-sampled (n > cap, CPP:243-261), the slot k with t_k = rank, t_k = round(k (n-1) / (cap-1)).
        const unsigned lanebit = 1u << (l & 31);
        int turn = 0;
        for (unsigned long long m = nzm; m; m &= m - 1) {
            const int a = __builtin_ctzll(m);                       // wave-uniform
            if ((turn++ & (T - 1)) != sub) continue;
            const unsigned inma = (L.fin[a] >> 31) ? 0xFFFFFFFFu : 0u;          // CPP:155
            const unsigned *ma = L.mask + (size_t)a * MW, *oa = L.occ + (a >> 5) * 2 * W;
            const unsigned f = l < W ? ma[l] & ~(oa[l] & inma) : 0u;
            const unsigned nzw = (unsigned)__ballot(f != 0);        // words with a cell that stays in the list (W <= 32)
            unsigned short *sa = L.sel + (size_t)a * P.capp;
            int n = 0;
            for (unsigned pw = nzw; pw; ) {
                const int q = __builtin_ctz(pw) >> 1;               // word pair
                pw &= ~(3u << (2 * q));
                const unsigned w0 = __builtin_amdgcn_readlane(f, 2 * q), w1 = __builtin_amdgcn_readlane(f, 2 * q + 1);
                const bool has = ((l < 32 ? w0 : w1) & lanebit) != 0;
                const unsigned long long bal = __ballot(has);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, (unsigned)n));
                if (has && rank < cap) sa[rank] = (unsigned short)(q * 64 + l);
                n += __popcll(bal);
            }
            if (n > cap) {                                          // sub-sampled list: slot k holds rank t_k
                // (2r-1)(cap-1) <= 2k(n-1) < (2r+1)(cap-1)  <=>  t_k = r; at most one k per rank since n > cap
                const int den = 2 * (n - 1), capm = cap - 1;
                const float rden = 1.0f / (float)den;
                int r0 = 0;
                for (unsigned pw = nzw; pw; ) {
                    const int q = __builtin_ctz(pw) >> 1;
                    pw &= ~(3u << (2 * q));
                    const unsigned w0 = __builtin_amdgcn_readlane(f, 2 * q), w1 = __builtin_amdgcn_readlane(f, 2 * q + 1);
                    const bool has = ((l < 32 ? w0 : w1) & lanebit) != 0;
                    const unsigned long long bal = __ballot(has);
                    const int r = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, (unsigned)r0));
                    const int N = (2 * r - 1) * capm + den - 1;     // k = ceil((2r-1)(cap-1) / den) = floor(N / den), r >= 1
                    int k = (int)((float)N * rden);                 // exact integers below 2^24; the quotient is off by one at most
                    const int rem = N - k * den;
                    k += rem < 0 ? -1 : (rem >= den ? 1 : 0);
                    k = r == 0 ? 0 : k;
                    if (has && k < cap && 2 * k * (n - 1) < (2 * r + 1) * capm) sa[k] = (unsigned short)(q * 64 + l);
                    r0 += __popcll(bal);
                }
            }
            if (l == 0) L.fin[64 + a] = (unsigned)n;                // read back by the agent's own lanes after the barrier
        }
    } else {
        // Unused lanes sit at -1e30 (P0), so their masks are empty.  The words whose first rank lies below k0 form a
        // prefix of nb words; this wave starts at the first non-empty word after them, whose first rank is bst.
        int nb = 0, bst = 0;
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
        n_out = n_after < cap ? n_after : cap;
        if (IDX >= 1 && sub == 0 && sensed_g) L.fin[64 + l] = (unsigned)n_out;      // for the row-wise copy of the lists in P6
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
                constexpr int QN = 5;
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
    }
    STAMP(6);
    if (T > 1) __syncthreads();                                     // the lists are complete
    if (sparse) { n_after = (int)L.fin[64 + l]; n_out = n_after < cap ? n_after : cap; }
    STAMP(15);
    EXIT_AT(5);

    // ---- P6: the observation's cell block (CPP:279-296) + reward weights, slots split over the waves ----
    const bool rew_on = active && in_flag && n_after > 0;
    float fn0 = 0.f, fn1 = 0.f, fden = 0.f;
    {
        f32x2 fn2 = {0.f, 0.f};
        const int gb = P.obs_dim - cap * 2;
        constexpr int RB = 10;                                      // slots per round: indices, coordinates, then math + stores
        if (active) {
            const int own = (2 - es) * NGP + l;                     // an empty slot reads the lane's own position: x - x = +0 (CPP:329-333 zeros)
            const f32x2 p2 = {pxf, pyf};
            for (int kb = k0; kb < k1; kb += RB) {
                if (__ballot(kb < n_out) == 0) {                    // no list of the wave's agents reaches this round: rows of +0
#pragma unroll
                    for (int q = 0; q < RB; ++q)
                        if (kb + q < k1) { ostore(gb + 2 * (kb + q), 0.f); ostore(gb + 2 * (kb + q) + 1, 0.f); }
                    continue;
                }
                int cell[RB]; f32x2 c2[RB]; bool have[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) cell[q] = mysel[kb + q];   // one batch of LDS reads; entries past k1 (the list is padded by RB) are masked below
#pragma unroll
                for (int q = 0; q < RB; ++q) { have[q] = kb + q < n_out && kb + q < k1; cell[q] = have[q] ? cell[q] : own; }
#pragma unroll
                for (int q = 0; q < RB; ++q) { c2[q].x = gxe[cell[q]]; c2[q].y = gye[cell[q]]; }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int k = kb + q;
                    if (k < k1) {
                        const f32x2 g2 = c2[q] - p2;                // CPP:285-286
#ifndef ASM_EXP_NOSTORE
                        ostore(gb + 2 * k, g2.x); ostore(gb + 2 * k + 1, g2.y);
#endif
                        // psi = (1 + cos(pi z / d_sen)) / 2 (CPP:1017-1025); v_cos_f32 takes revolutions
                        const float z = __builtin_amdgcn_sqrtf(__builtin_fmaf(g2.y, g2.y, g2.x * g2.x));
                        const float psi = have[q] ? __builtin_fmaf(0.5f, __builtin_amdgcn_cosf(z * P.inv2dsen_f), 0.5f) : 0.f;
                        const f32x2 psi2 = {psi, psi};
                        fn2 = __builtin_elementwise_fma(psi2, g2, fn2); fden += psi;
                    }
                }
            }
        }
        fn0 = fn2.x; fn1 = fn2.y;
        if (IDX >= 1 && sensed_g) {
            // sensed_index rows (CPP:341-345), copied from the LDS lists agent by agent so that the 64 lanes write
            // 64 consecutive entries; the agents are dealt out to the waves
            for (int a = sub; a < 64; a += T) {
                const int ea = e0 + (a >> 5), ia = a & 31;
                if (ea >= P.E || ia >= n_a) continue;
                const int na = (int)L.fin[64 + a];
                const unsigned short *src = L.sel + (size_t)a * P.capp;
                int *dst = sensed_g + ((size_t)ea * n_a + ia) * cap;
                for (int k = l; k < cap; k += 64) dst[k] = k < na ? (int)src[k] : -1;
            }
        }
        if constexpr (IDX == 2) {
            // occupied_index rows (CPP:215-238): the sensed cells some agent occupies, for agents inside the shape.  One
            // agent at a time per wave; the 64 lanes test 64 cells (two mask words) per pass and the set ones are
            // written in ascending order at ranks from a ballot prefix count; the rest of the row is -1.
            for (int a = sub; a < 64; a += T) {
                const int ea = e0 + (a >> 5), ia = a & 31;
                if (ea >= P.E || ia >= n_a) continue;
                const unsigned *ma = L.mask + (size_t)a * MW, *oc = L.occ + (a >> 5) * 2 * W;
                int *dst = occup_g + ((size_t)ea * n_a + ia) * P.n_occ;
                int cnt = 0;
                if (L.fin[a] >> 31) {                                // CPP:155: only inside the shape
                    int n = 0;
                    for (int w = 0; w < W; ++w) n += __popc(ma[w] & oc[w]);
                    if (n > P.n_occ) {                               // sub-sampled (CPP:224-229): rare, one lane walks the bits
                        if (l == 0) {
                            BitIter it; it.init(ma, W, oc, false);
                            emit_list(it, n, P.n_occ, [&](int k, int cell) { dst[k] = cell; });
                        }
                        cnt = P.n_occ;
                    } else {
                        for (int w0 = 0; w0 < W; w0 += 2) {
                            const int w = w0 + (l >> 5);
                            const unsigned f = w < W ? ma[w] & oc[w] : 0u;
                            const bool has = (f >> (l & 31)) & 1u;
                            const unsigned long long bal = __ballot(has);
                            const int rank = cnt + __popcll(bal & ((1ull << l) - 1ull));
                            if (has) dst[rank] = w * 32 + (l & 31);
                            cnt += __popcll(bal);
                        }
                    }
                }
                for (int k = cnt + l; k < P.n_occ; k += 64) dst[k] = -1;
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
                    double rx = (double)L.px[es * 32 + j] - px, ry = (double)L.py[es * 32 + j] - py;
                    if constexpr (PER) wrap_rel(rx, ry, P.half_w, P.half_h);
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
            if (P.want_reward) rew_w[la] = (float)reward;
        }
        STAMP(8);
        if (P.want_stats) {                                         // TRAIN:128-131: reward sums, agents inside
            const unsigned long long half = es ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
            const double rs = (double)__popcll(__ballot(active && reward > 0.0) & half), rq = rs;     // rewards are 0 or 1
            const double inn = (double)__popcll(__ballot(active && in_flag) & half);
            if (i == 0 && env_ok) {
                double *st = stats_g + (size_t)e0 * 8 + es * 8;
                stat_add(&st[0], rs); stat_add(&st[1], rq); stat_add(&st[7], inn);
            }
        }
    }
    // Formation statistics: they do not depend on the reward, so with several waves the second one adds them while the first
    // evaluates the reward (all-cells path: their inputs were handed over through LDS in P4; lattice path: only the coverage
    // is left, the other two pieces were added in P3 by the waves that computed their inputs).
    if (sub == SUB_PRIOR && P.want_stats) {
        if constexpr (LAT) {
            stats_coverage();
        } else {
            double dm = dmin, nn = nn_min; int cn = contacts;
            if (T > 1) {
                const uint4 q = L.part[l], q3 = L.part[(T - 1) * 64 + l];
                dm = __hiloint2double((int)q3.y, (int)q3.x); nn = __hiloint2double((int)q.w, (int)q.z);
                cn = (int)q.x;
            }
            stats_neighbours(nn, cn);
            stats_target(dm);
            stats_coverage();
        }
    }
    STAMP(9);
}

