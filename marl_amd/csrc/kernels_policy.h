// Rollout inference of the shared policy network on the matrix cores: the four-layer MLP of networks.py:6-44
// (Linear - leaky_relu x 3, Linear - tanh) applied to every agent's observation column, bf16 operands, fp32 accumulation.
// Part of the single translation unit asm_kernels.hip (included there, inside namespace asmenv); not a stand-alone header.
//
// Orientation.  The environment writes observations as obs[e][feature][agent] (ENV.step's (obs_dim, n_a) per
// environment), which is X^T with the agents contiguous.  Every layer is computed as  H^T = W . X^T : the weight matrix
// [out][in] is the A operand of v_mfma_f32_32x32x16_bf16, a tile of 32 agents the B operand.  The 32 x 32 result has
// its column (agent) on the lane and its rows (features) in the 16 accumulator registers, and the next layer sums over
// exactly that row index, so the activated accumulator, converted to bf16 pairwise, IS the next layer's B operand:
// registers 8s .. 8s+7 of row tile m form the fragment of k-step 2m + s, holding the features
// 32m + 16s + 8(j >> 2) + 4(lane >> 5) + (j & 3), j = 0..7.  The weights are packed on the host in that k order (and in
// natural order for the first layer), one 16-byte fragment per lane, k-step major, so a wave's weight read is 1 KiB
// contiguous and half a layer (the first or the second half of its k-steps) is one contiguous block.  No activation
// ever touches LDS or memory.
//
// A wave owns two tiles of 32 agents (one environment of up to 32 agents per tile; larger swarms take several tiles
// per environment) so that every weight fragment feeds two MFMAs; with hidden width 256 that is 2 x 8 x 16 accumulator
// registers plus 2 x 16 x 4 operand registers per lane: one wave per SIMD, the unified 512-register file.
//
// Weights reach the four waves of a workgroup through LDS, copied by global_load_lds_dwordx4 (LDS destination =
// wave-uniform base + 16 bytes per lane, i.e. the packed layout itself; no registers): the whole first layer before
// the first MFMA, the hidden layers half a layer at a time in two buffers, the copy of one half running under the
// MFMAs of the other, one barrier per half.  A wave alone on its SIMD cannot hide a weight fetch from memory per MFMA
// pair (the first version of this kernel did that and ran at the speed of the PyTorch path), nor an LDS read per MFMA
// pair: the fragments of a k-step are read one group ahead (two register sets, scheduling barriers keep the reads in
// front of the MFMAs they overlap with).
//
// The first layer's operands come from the observation rows.  Swarms of at most 32 agents (a tile = an environment)
// with 16-byte aligned environments: a k-step's 16 rows are one contiguous slab, copied by the same instruction into
// per-wave LDS slots three k-steps deep and read from there as 8 dwords per lane and tile (template parameter PS).
// Otherwise 8 strided rows per lane and tile straight from memory into registers, four k-steps ahead.
//
// What bounds the phases of a workgroup (cycle stamps, hidden width 256, 192 features, 30 agents): layer 1 and the
// prologue by the observation stream (every workgroup of the chip is in that phase at the same time); the hidden
// layers by the matrix core plus the issue cost of the weight copies; the activations by their ~900 vector
// instructions (256 accumulator reads, bias add and 0.01 product packed in pairs, 256 max, 128 packs).

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Two floats to a bf16 pair (round to nearest even), at this point of the program: written as a conversion of vector
// elements the compiler forms the pair only where the operand is first used, a layer later, and keeps all 256
// activated floats alive (and spilled) until then.
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (f32x2: kernels_common.h)

// HT: hidden width / 32.  w1: [K1][HT][64] fragments (natural k order, K1 = ceil(obs_dim / 16)); w2, w3: [2 HT][HT][64],
// w4: [2 HT][1][64] (output rows padded to 32), all in the accumulator's k order; bias: [3 * 32 HT + 32] floats.
// Dynamic LDS: the biases, then two buffers of half_frags KiB, half_frags = HT x max(ceil(K1 / 2), HT, 2).
// PS != 0 (swarms of at most 32 agents whose environments start on 16-byte boundaries; PS = 2 up to 16 agents, else 4):
// the 16 observation rows of a k-step are one contiguous slab of 64 n_a bytes per environment, PS / 2 copy instructions
// of 1 KiB; every wave copies the slabs of its two tiles into slots of its own behind the first layer's weights, three
// k-steps deep, and reads its operands from there (LDS: the weights of layer 1, then 4 waves x 3 slots x PS KiB).
template <int HT, int PS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_policy_mlp(int n_a, int D, int K1, int A, int TPE, int ntiles, int half_frags, const float *obs_g, const uint4 *w1,
                  const uint4 *w2, const uint4 *w3, const uint4 *w4, const float *bias, float *out_g, float noise_scale,
                  float epsilon, uint64_t seed, uint32_t step, int64_t env0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int H = 32 * HT, KS = 2 * HT, KH = HT, PF = 4;           // KH: k-steps per half of a hidden layer
    float *const bl = reinterpret_cast<float *>(smem);
    uint4 *const buf0 = reinterpret_cast<uint4 *>(smem + 4 * (3 * H + 32)), *const buf1 = buf0 + (size_t)half_frags * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    const int t0 = (blockIdx.x * 4 + wave) * 2;
    int env[2], agent[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = t0 + u < ntiles ? t0 + u : 0;
        env[u] = t / TPE; agent[u] = (t % TPE) * 32 + r;
        ok[u] = t0 + u < ntiles && agent[u] < n_a;
    }
    // nfrag fragments from w into dst, every wave a quarter of them, all in flight at once.  The copy instruction is
    // written out (LDS destination = M0 + 16 bytes per lane): issued through the compiler's builtin it is accounted as an
    // access that may return through either memory counter, and until the next full drain every wait for an LDS read
    // becomes a wait for ALL LDS reads, the weight fragments just requested for the next MFMA group included.  The
    // compiler therefore does not know of these copies: copies_landed() before the barrier that publishes them.  (M0 is
    // a reserved register that the compiler sets immediately before each use of its own; nothing else here uses it.)
    auto lds_addr = [](const void *p) __attribute__((always_inline)) {
        return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
    };
    // (LDS addresses as 32-bit integers off two scalar bases: a generic pointer per piece costs a 64-bit add, a null
    // check and, unrolled, registers)
    const unsigned a_buf0 = __builtin_amdgcn_readfirstlane(lds_addr(buf0)), a_buf1 = __builtin_amdgcn_readfirstlane(lds_addr(buf1));
    auto dma_piece = [&](const uint4 *src, unsigned dst) __attribute__((always_inline)) {         // src per lane, dst per wave
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(src) : "memory");
    };
    auto dma = [&](const uint4 *w, int nfrag, unsigned dst) __attribute__((always_inline)) {
        for (int f = wave; f < nfrag; f += 4) dma_piece(w + f * 64 + lane, dst + f * 1024);
    };
    auto dma_bias = [&]() __attribute__((always_inline)) {
        for (int k = wave * 64; k < 3 * H + 32; k += 256)
            if (k + lane < 3 * H + 32)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                             :: "s"(__builtin_amdgcn_readfirstlane(lds_addr(bl + k))), "v"(bias + k + lane) : "memory");
    };
    auto copies_landed = []() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    PSTAMP(0, 0);

    f32x16 acc[2][HT];
    f32x16 zero;
#pragma unroll
    for (int q = 0; q < 16; ++q) zero[q] = 0.f;
    // weight fragments: a k-step's HT fragments are read from LDS in groups of GM, one group ahead of the MFMAs that use
    // them (two register sets; 2 GM MFMAs = 64 GM cycles cover the read)
    constexpr int GM = HT >= 2 ? HT / 2 : 1, NG = HT / GM;
    bf16x8 Aq[2][GM];
    // the MFMAs of k-step s of the first layer on the operands B, the fragments of the next group read ahead
    auto layer1_step = [&](int s, int parity, const bf16x8 (&B)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int cur = (parity * NG + g) & 1;
            const int nf = (g + 1 < NG) ? s * HT + (g + 1) * GM : (s + 1 < K1 ? (s + 1) * HT : s * HT);
#pragma unroll
            for (int m = 0; m < GM; ++m) Aq[cur ^ 1][m] = __builtin_bit_cast(bf16x8, buf0[(nf + m) * 64 + lane]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < GM; ++m) {
                acc[0][g * GM + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[cur][m], B[0], acc[0][g * GM + m], 0, 0, 0);
                acc[1][g * GM + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[cur][m], B[1], acc[1][g * GM + m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (PS != 0) {
        // ---- layer 1, operands through LDS slabs ----
        // A k-step of this loop is bound by the number of instructions its one wave issues, not by the matrix core (16
        // MFMAs are 512 cycles; 200 other instructions are twice that), so everything per k-step is pared down: slots at
        // a fixed pitch of one or two copy instructions per tile (the lanes beyond the slab's end fetch its last piece
        // again and land in the slot's padding; no lane is ever switched off), the copy's source as a scalar base plus
        // a 32-bit lane offset, operand addresses that differ between the tiles by an immediate.  In a short last slab
        // the lanes beyond its end copy its last 16-byte piece once more into the slot's padding, so the rows beyond
        // obs_dim hold observation values -- finite, from this or an earlier copy -- and meet zero weights (the packed
        // first layer is zero beyond obs_dim).
        static_assert(NG % 2 == 0 && HT >= 4, "the fragment parity must carry over a k-step; eight chunks over HT MFMA pairs");
        constexpr int PFD = 3, TILEB = PS == 4 ? 2048 : 1024, SLOTB = 2 * TILEB;
        const int slab_bytes = 64 * n_a;                                // a full slab: 16 rows of n_a floats
        const unsigned a_slab = a_buf0 + (unsigned)(K1 * HT) * 1024u + (unsigned)(wave * PFD * SLOTB);
        const char *srcb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) srcb[u] = reinterpret_cast<const char *>(obs_g + (size_t)env[u] * D * n_a);
        unsigned soff = 0, loff[2] = {0, 0};                            // of the slab being issued: byte offset, lane offsets
        auto slab_begin = [&](int q) __attribute__((always_inline)) {    // slab q, or the last one again beyond the end
            const int qc = q < K1 ? q : K1 - 1;
            const int rows = D - 16 * qc < 16 ? D - 16 * qc : 16;
            const unsigned last16 = (unsigned)(rows * n_a / 4 - 1) * 16u;
            soff = (unsigned)qc * (unsigned)slab_bytes;
#pragma unroll
            for (int p = 0; p < PS / 2; ++p) { const unsigned o = (unsigned)lane * 16u + 1024u * p; loff[p] = o < last16 ? o : last16; }
        };
        auto slab_piece = [&](int c, int slot) __attribute__((always_inline)) {       // copy instruction c of PS
            const int u = PS == 4 ? c >> 1 : c, p = PS == 4 ? c & 1 : 0;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(a_slab + (unsigned)(slot * SLOTB + u * TILEB + p * 1024)), "v"(loff[p]), "s"(srcb[u] + soff) : "memory");
        };
        // this lane's 8 operand addresses in tile 0 of slot 0 (tile 1: + TILEB, slot n: + n SLOTB)
        unsigned vrow[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vrow[j] = a_slab + (unsigned)(((8 * h + j) * n_a + (r < n_a ? r : 0)) * 4);   // (lanes beyond the agents read column 0: inside the tile; their MFMA columns are dead)
        typedef __attribute__((address_space(3))) const float lds_cf;
        auto read_tile = [&](unsigned slot_off, int u, float (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *(lds_cf *)(uintptr_t)(vrow[j] + slot_off + (unsigned)(u * TILEB));
        };
        auto pack_tile = [&](const float (&x)[8], u32x4 &B) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) B[j >> 1] = pack_bf16x2(x[j], x[j + 1]);
        };
        slab_begin(0);
#pragma unroll
        for (int c = 0; c < PS; ++c) slab_piece(c, 0);
        dma(w1, K1 * HT, a_buf0);                                       // the whole first layer (it spans both buffers)
        dma_bias();
#pragma unroll
        for (int q = 1; q < PFD; ++q) {
            slab_begin(q);
#pragma unroll
            for (int c = 0; c < PS; ++c) slab_piece(c, q);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PFD - 1) * PS) : "memory");      // slab 0, the weights, the biases
        __syncthreads();
        PSTAMP(1, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < HT; ++m) acc[u][m] = zero;
#pragma unroll
        for (int m = 0; m < GM; ++m) Aq[0][m] = __builtin_bit_cast(bf16x8, buf0[m * 64 + lane]);
        // The operands of k-step s + 1 are read and packed between the MFMAs of k-step s, from the slab that landed
        // PFD - 1 k-steps after its issue; the slot of slab s, in registers since the k-step before, takes slab s + PFD.
        // Eight chunks of that work (the copy instructions, wait + reads of tile 0, reads of tile 1, the two packs)
        // follow the k-step's HT MFMA pairs, one or two a pair.
        u32x4 Bn[2];
        {
            float x[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) { read_tile(0, u, x); pack_tile(x, Bn[u]); }
        }
        int slot_wr = 0, slot_rd = 1;
        for (int s = 0; s < K1; ++s) {
            const bf16x8 B0 = __builtin_bit_cast(bf16x8, Bn[0]), B1 = __builtin_bit_cast(bf16x8, Bn[1]);
            float xr[2][8];
            auto chunk = [&](int c) __attribute__((always_inline)) {
                if (c == 0) slab_begin(s + PFD);
                if (c < PS) {
                    slab_piece(c, slot_wr);
                } else if (c == 4) {
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PFD - 1) * PS) : "memory");
                    read_tile((unsigned)(slot_rd * SLOTB), 0, xr[0]);
                } else if (c == 5) {
                    read_tile((unsigned)(slot_rd * SLOTB), 1, xr[1]);
                } else if (c == 6) {
                    pack_tile(xr[0], Bn[0]);
                } else if (c == 7) {
                    pack_tile(xr[1], Bn[1]);
                }
            };
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = g & 1;
                const int nf = (g + 1 < NG) ? s * HT + (g + 1) * GM : (s + 1 < K1 ? (s + 1) * HT : s * HT);
#pragma unroll
                for (int m = 0; m < GM; ++m) Aq[cur ^ 1][m] = __builtin_bit_cast(bf16x8, buf0[(nf + m) * 64 + lane]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < GM; ++m) {
                    acc[0][g * GM + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[cur][m], B0, acc[0][g * GM + m], 0, 0, 0);
                    acc[1][g * GM + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[cur][m], B1, acc[1][g * GM + m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = (g * GM + m) * (8 / HT); c < (g * GM + m + 1) * (8 / HT); ++c) chunk(c);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            slot_wr = slot_rd;
            slot_rd = slot_rd + 1 == PFD ? 0 : slot_rd + 1;
        }
    } else {
        // ---- layer 1: operands straight from the observation rows (8 consecutive features of one agent per lane) ----
        const float *col[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) col[u] = obs_g + (size_t)env[u] * D * n_a + agent[u];
        // xb[p]: the 16 observation values (2 tiles x 8 features) of a k-step, PF k-steps in flight
        float xb[PF][16];
#define POLICY_FETCH(S, P)                                                                            \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                   \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
                const int k = 16 * (S) + 8 * h + j;                                                     \
                xb[P][u * 8 + j] = (ok[u] && k < D) ? col[u][(size_t)k * n_a] : 0.f;                    \
            }
#pragma unroll
        for (int p = 0; p < PF; ++p) { POLICY_FETCH(p, p) }             // (k-steps beyond K1 read nothing: k >= D)
        // The first layer's weights (the whole layer: it spans both buffers) and the biases follow the observation
        // rows into flight, all by the copy path: an ordinary load here would have to wait for every copy issued
        // before it.
        dma(w1, K1 * HT, a_buf0);
        dma_bias();
        copies_landed();
        __syncthreads();                                               // the first layer's weights and the biases are in LDS
        PSTAMP(1, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u)                                    // (cleared after the wait: nothing to keep alive across it)
#pragma unroll
            for (int m = 0; m < HT; ++m) acc[u][m] = zero;
#pragma unroll
        for (int m = 0; m < GM; ++m) Aq[0][m] = __builtin_bit_cast(bf16x8, buf0[m * 64 + lane]);
        for (int s0 = 0; s0 < K1; s0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int s = s0 + p;
                if (s < K1) {
                    bf16x8 B[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) B[u][j] = (__bf16)xb[p][u * 8 + j];
                    POLICY_FETCH(s + PF, p)
                    layer1_step(s, p, B);                              // (PF * NG is even: the parity carries over the outer loop)
                }
            }
        }
#undef POLICY_FETCH
    }
    PSTAMP(2, 0);

    // bias + leaky_relu (slope 0.01, F.leaky_relu's default: max(v, 0.01 v)) on the accumulators, packed as the next
    // layer's operands.  (Starting the sums from the bias instead, as the first MFMA's C operand, costs more registers
    // than the adds it saves: measured slower.)
    u32x4 F[2][KS];                                                    // 8 bf16 per k-step and tile
    auto activate = [&](const float *b) __attribute__((always_inline)) {
        // (the four biases of a register group are read one group ahead: a read waited for on the spot costs an LDS
        // round trip per group, as much again as the group's arithmetic)
        f32x4 bq[2];
        bq[0] = *reinterpret_cast<const f32x4 *>(b + 4 * h);
#pragma unroll
        for (int m = 0; m < HT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cur = g & 1, nx = (4 * m + g + 1 < 4 * HT) ? 4 * m + g + 1 : 4 * m + g;
                bq[cur ^ 1] = *reinterpret_cast<const f32x4 *>(b + 8 * nx + 4 * h);                // rows 8g + 4h .. + 3 of tile m
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 b4 = bq[cur];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    // (bias add and the 0.01 product as packed pairs: v_pk_add_f32 / v_pk_mul_f32)
                    const f32x2 ta = f32x2{acc[u][m][4 * g], acc[u][m][4 * g + 1]} + f32x2{b4[0], b4[1]};
                    const f32x2 tb = f32x2{acc[u][m][4 * g + 2], acc[u][m][4 * g + 3]} + f32x2{b4[2], b4[3]};
                    const f32x2 sa = ta * 0.01f, sb = tb * 0.01f;
                    const float v[4] = {__builtin_fmaxf(ta.x, sa.x), __builtin_fmaxf(ta.y, sa.y), __builtin_fmaxf(tb.x, sb.x), __builtin_fmaxf(tb.y, sb.y)};
                    // registers 4g .. 4g + 3 are elements (4g & 7) .. + 3 of k-step 2m + (g >> 1)
                    F[u][2 * m + (g >> 1)][(g & 1) * 2] = pack_bf16x2(v[0], v[1]);
                    F[u][2 * m + (g >> 1)][(g & 1) * 2 + 1] = pack_bf16x2(v[2], v[3]);
                }
                __builtin_amdgcn_sched_barrier(0);                     // (keeps the live temporaries of this pass few)
            }
    };
    // k-steps [sa, sa + KH) of a hidden layer from the fragments in wb; the copy of the next block of weights (nfrag
    // fragments from nw to ndst, a quarter per wave) is issued a piece per MFMA group, behind the group's second MFMA
    // pair (scalar base + lane offset: two scalar adds and the copy, nothing for the vector ALU): 1 to 2 % faster than
    // all pieces in one burst in front of the MFMAs.
    const unsigned lane_piece = (unsigned)(wave * 1024 + lane * 16);
    auto hidden_part = [&](const uint4 *wb, int sa, const uint4 *nw, int nfrag, unsigned ndst) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < GM; ++m) Aq[0][m] = __builtin_bit_cast(bf16x8, wb[m * 64 + lane]);
#pragma unroll
        for (int t = 0; t < KH; ++t)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = (t * NG + g) & 1;
                if (t * NG + g + 1 < KH * NG) {
#pragma unroll
                    for (int m = 0; m < GM; ++m)
                        Aq[cur ^ 1][m] = __builtin_bit_cast(bf16x8, wb[((t * NG + g + 1) * GM + m) * 64 + lane]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < GM; ++m) {                          // the layer's first k-step starts the sums
                    const bool first = sa == 0 && t == 0;
                    acc[0][g * GM + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[cur][m], __builtin_bit_cast(bf16x8, F[0][sa + t]), first ? zero : acc[0][g * GM + m], 0, 0, 0);
                    acc[1][g * GM + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[cur][m], __builtin_bit_cast(bf16x8, F[1][sa + t]), first ? zero : acc[1][g * GM + m], 0, 0, 0);
                    if (m == (GM > 1 ? 1 : 0)) {
                        const int i = t * NG + g;                       // piece i of this wave: fragment wave + 4 i
                        __builtin_amdgcn_sched_barrier(0);
                        if (wave + 4 * i < nfrag)
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                         :: "s"(ndst + (unsigned)(wave * 1024 + i * 4096)), "v"(lane_piece),
                                            "s"(reinterpret_cast<const char *>(nw) + (size_t)i * 4096) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    copies_landed();
    __syncthreads();                                                   // everybody is done with the first layer's weights
    PSTAMP(3, 0);
    dma(w2, KH * HT, a_buf0);
    activate(bl);
    PSTAMP(4, 0);
    copies_landed();
    __syncthreads();                                                   // w2's first half is in
    PSTAMP(5, 0);
    hidden_part(buf0, 0, w2 + (size_t)KH * HT * 64, KH * HT, a_buf1);
    PSTAMP(6, 0);
    copies_landed();
    __syncthreads();                                                   // w2's second half is in; buf0 is free
    PSTAMP(7, 0);
    hidden_part(buf1, KH, w3, KH * HT, a_buf0);
    PSTAMP(8, 0);
    activate(bl + H);
    PSTAMP(9, 0);
    copies_landed();
    __syncthreads();
    PSTAMP(10, 0);
    hidden_part(buf0, 0, w3 + (size_t)KH * HT * 64, KH * HT, a_buf1);
    PSTAMP(11, 0);
    copies_landed();
    __syncthreads();
    PSTAMP(12, 0);
    hidden_part(buf1, KH, w4, KS, a_buf0);
    PSTAMP(13, 0);
    activate(bl + 2 * H);
    PSTAMP(14, 0);
    copies_landed();
    __syncthreads();                                                   // the output layer's weights are in
    PSTAMP(15, 0);

    // ---- output layer (rows padded to one tile) and tanh ----
    f32x16 o[2] = {zero, zero};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 Aw = __builtin_bit_cast(bf16x8, buf0[s * 64 + lane]);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aw, __builtin_bit_cast(bf16x8, F[0][s]), o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aw, __builtin_bit_cast(bf16x8, F[1][s]), o[1], 0, 0, 0);
    }
    PSTAMP(16, 0);
    // Exploration (agents.py:86-94), when asked for: per environment one coin -- with probability epsilon the whole team
    // acts uniformly at random, else Gaussian noise is added and the action clamped to [-1, 1].  Counter-based draws
    // (Philox4x32-10, stream 3) keyed by the GLOBAL environment id, the agent, the output pair and the caller's step
    // counter: the same numbers whatever the batch is split into.
    // The code below is straight-line per register pair, for every lane, with only the stores predicated, and leaves at the
    // first pair whose rows all lie beyond the action dimension: a branch around each unused pair is a taken branch to
    // code not yet fetched, a dozen instruction-cache misses per workgroup.
    const bool explore = noise_scale > 0.f || epsilon > 0.f;
    uint32_t k0[2], k1[2];
    bool all_uniform[2] = {false, false};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint64_t eg = (uint64_t)(env0 + env[u]);
        k0[u] = (uint32_t)eg; k1[u] = (uint32_t)seed ^ (uint32_t)(eg >> 32) ^ (uint32_t)(seed >> 32) * 0x9E3779B9u;
    }
    if (explore && epsilon > 0.f) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t c4[4];
            philox4x32(0xFFFFFFFFu, step, 3u, 0u, k0[u], k1[u], c4);
            all_uniform[u] = (float)c4[0] * (1.0f / 4294967296.0f) < epsilon;      // agents.py:87
        }
    }
#pragma unroll
    for (int q = 0; q < 16; q += 2) {                                   // registers q, q + 1 are rows row, row + 1
        if ((q & 3) + 8 * (q >> 2) >= A) break;                         // (the pairs' first rows increase with q)
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        const float b0 = bl[3 * H + row], b1 = bl[3 * H + row + 1];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float a0 = tanhf(o[u][q] + b0), a1 = tanhf(o[u][q + 1] + b1);
            if (explore) {
                uint32_t d[4];
                philox4x32((uint32_t)agent[u] * 16u + (uint32_t)(row >> 1), step, 3u, 1u, k0[u], k1[u], d);
                // agents.py:88: the whole team uniformly at random
                const float r0 = -1.f + 2.f * ((float)d[2] * (1.0f / 4294967296.0f)), r1 = -1.f + 2.f * ((float)d[3] * (1.0f / 4294967296.0f));
                // agents.py:91-94, Box-Muller on two of the draws
                const float u1 = ((float)(d[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(d[1] >> 8) * (1.0f / 16777216.0f);
                const float rad = __builtin_sqrtf(-2.0f * __logf(u1)) * noise_scale;
                float sn, cs;
                __sincosf(6.28318530717958647692f * u2, &sn, &cs);
                const float g0 = __builtin_fminf(__builtin_fmaxf(a0 + rad * cs, -1.f), 1.f);
                const float g1 = __builtin_fminf(__builtin_fmaxf(a1 + rad * sn, -1.f), 1.f);
                a0 = all_uniform[u] ? r0 : g0;
                a1 = all_uniform[u] ? r1 : g1;
            }
            if (ok[u] && row < A) out_g[((size_t)env[u] * A + row) * n_a + agent[u]] = a0;
            if (ok[u] && row + 1 < A) out_g[((size_t)env[u] * A + row + 1) * n_a + agent[u]] = a1;
        }
    }
    PSTAMP(17, 1);
}
