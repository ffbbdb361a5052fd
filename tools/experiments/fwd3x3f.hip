// Forward of a stride-1 3x3 convolution with 32 input and 32 output channels (bf16) with the BatchNorm + LeakyReLU load prologue,
// the residual add and the next BatchNorm's statistics (wideresnet.py:27-35,46-49: the 32-channel body of WideResNet-28-2) on the
// block architecture of bwd3x3f.hip.  gfx950.
//
// conv3x3p_kernel runs these layers at 4.2-4.3 TB/s of their 335-402 MB (two 256-thread blocks per CU, every wave loads, transforms,
// multiplies and stores; its prefetch queue is drained once per tile -- docs/lab_notes_r06.md).  Here: one 512-thread block per CU
// (248 per launch: bwd3x3f.hip on why not 256), eight identical waves of 16 output pixels x 32 channels each (weights register-
// resident: 72 registers; v_mfma_f32_16x16x32_bf16 in conv3x3p's accumulation order: outputs BIT-EQUAL to it), the transformed halo
// tile and the raw residual tile of the NEXT tile staged into the other LDS stage while this one is multiplied, two register stages
// (operands requested two tiles ahead; every request unconditional, whole pairs of tiles in the loop: counted waits), an LDS-only
// barrier per tile, interleaved tile order.  BatchNorm finalisation folded (sv_igemm_args::fold_*) as in conv3x3p.
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a fast path inside it (SV_K_FWD3X3F disables).
#include "common.h"
#include "epilogue.h"

namespace {

constexpr int LDF = 48;     // LDS row of the halo: 32 channels + 16 (96 bytes: conflict-free 16-byte fragment reads, conv3x3.hip)
constexpr int LDR = 40;     // LDS row of the raw residual (80 bytes: the epilogue's 8-byte reads of 16 pixels hit 16 bank pairs)
constexpr int CH = 32;

template <int NS, bool RES>
struct fwdf_stage {
    bf16x8 hv[NS];
    bf16x8 rv[RES ? 1 : 1];
};

template <int WLOG, bool RES>
__global__ __launch_bounds__(512) void fwd3x3f_kernel(const sv_geom g, const sv_igemm_args_g A) {
    const sv_igemm_args& a = A.g[blockIdx.y];
    sv_start_signal(a);
    typedef bf16x8 V;
    typedef bf16x4 Q;
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    // LDS halo rows: row 0 / the last row are the vertical halo; when a tile holds two whole images (W = 8) a zero spacer row
    // separates them -- zero padding is DATA in LDS, the nine taps need no masks (conv3x3p_kernel)
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1, HP = LROWS * WP;
    constexpr int HV = LROWS * W * 4;                  // halo vectors (8 channels each; the two padding columns are zeroed once)
    static_assert(HV > 512 && HV <= 768, "slots: two halo vectors on threads 0..255, one on the others");
    constexpr int SHA = HP * LDF, SRS = RES ? 128 * LDR : 0, STG = SHA + SRS;      // elements per LDS stage
    static_assert((SHA * 2) % 16 == 0 && (STG * 2) % 16 == 0, "16-byte aligned LDS images");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const st0 = reinterpret_cast<bf16*>(smem);                     // [2][STG]: transformed halo | raw residual
    double* const ssum = reinterpret_cast<double*>(st0 + 2 * STG);       // [2][32]
    float* const cf = reinterpret_cast<float*>(ssum + 2 * CH);           // [2][32]: prologue scale, shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    const int NC = gridDim.x;
    const int tstep = NC, t_begin = sv_window_slot(NC, blockIdx.y, blockIdx.x);
    const sv_phase& P = g.phase[0];
    const char* __restrict__ X = reinterpret_cast<const char*>(a.x);
    const char* __restrict__ R = RES ? reinterpret_cast<const char*>(a.residual) : nullptr;
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    float slope = a.pro_slope;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));

    // the BatchNorm of the prologue: finalised by this launch (every block derives the coefficients from the raw statistics, block 0 of
    // a group stores the four vectors for the backward pass) or given
    if (a.fold_stats) {
        float* const c2 = reinterpret_cast<float*>(smem) + 4096;         // [32] pairs {scale, shift}; scratch: 2 * 512 doubles in front
        sv_bn_fold_block512(a, CH, reinterpret_cast<double*>(smem), c2, blockIdx.x == 0);
        if (tid < CH) {
            cf[tid] = c2[2 * tid];
            cf[CH + tid] = c2[2 * tid + 1];
        }
        __syncthreads();
    } else if (tid < CH) {
        cf[tid] = a.pro_scale[tid];
        cf[CH + tid] = a.pro_shift[tid];
    }
    if (tid < 2 * CH) ssum[tid] = 0.0;
    // the two padding columns of every halo row are zero for the kernel's lifetime (no staging slot covers them)
    for (int idx = tid; idx < 2 * LROWS * 2 * 4; idx += 512) {
        const int vv = idx & 3, side = (idx >> 2) & 1, row = (idx >> 3) % LROWS, stg = (idx >> 3) / LROWS;
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        *reinterpret_cast<bf16x8*>(st0 + stg * STG + (row * WP + side * (WP - 1)) * LDF + 8 * vv) = z;
    }

    // ---- staging slots (bwd3x3f.hip): kind 0 = always zero, 1 = image row of this tile, 2 = row above, 3 = row below ---------------
    const int v = tid & 3;
    int hlds[2];                      // LDS element offset (a multiple of 8) | kind in the two low bits; -1: no slot
    uint32_t hoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i;
        const int pix = min(idx, HV - 1) >> 2;
        const int lr = pix >> WLOG, xx = pix & (W - 1);
        const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
        int kind = 1, rel = lr - 1 - seg;
        if (off == 0) {
            if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
            else kind = 0;
        }
        hoff[i] = (uint32_t)(((rel + 1) * W + xx) * CH + 8 * v) * 2u;        // bytes from the row ABOVE the tile
        hlds[i] = idx < HV ? (((lr * WP + xx + 1) * LDF + 8 * v) | kind) : -1;
    }
    const uint32_t hsafe = (uint32_t)(W * CH + 8 * v) * 2u;
    const int cp = tid >> 2;                                                  // this thread's centre pixel (residual)
    const uint32_t coff = (uint32_t)(cp * CH + 8 * v) * 2u;

    auto load_stage = [&](auto& S, int tile) __attribute__((always_inline)) {
        constexpr int NS = sizeof(S.hv) / sizeof(V);
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const int64_t hb = ((int64_t)gr0 - 1) * W * CH * 2;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int kind = hlds[i] & 3;
            const bool ok = hlds[i] >= 0 && (kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok));
            const uint32_t o = ok ? hoff[i] : hsafe;
            S.hv[i] = *reinterpret_cast<const V*>(X + hb + o);
        }
        if constexpr (RES) S.rv[0] = *reinterpret_cast<const V*>(R + (int64_t)gr0 * W * CH * 2 + coff);
    };
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    auto store_stage = [&](auto& S, int tile, int stage) __attribute__((always_inline)) {
        constexpr int NS = sizeof(S.hv) / sizeof(V);
        bf16* sb = st0 + stage * STG;
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cf + 8 * v), s1 = *reinterpret_cast<const f32x4*>(cf + 8 * v + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(cf + CH + 8 * v), t1 = *reinterpret_cast<const f32x4*>(cf + CH + 8 * v + 4);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int kind = hlds[i] & 3;
            const bool ok = kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok);
            V o = bn_act8(S.hv[i], s0, s1, t0, t1, slope);           // (the expression of conv3x3p_kernel's prologue)
            if (!ok) o = zero;
            if (hlds[i] >= 0) *reinterpret_cast<V*>(sb + (hlds[i] & ~7)) = o;
        }
        if constexpr (RES) *reinterpret_cast<V*>(sb + SHA + cp * LDR + 8 * v) = S.rv[0];
    };

    // ---- this wave's 16 output pixels, both 16-channel tiles -----------------------------------------------------------------------------
    const int pp = 16 * wave + fr;
    const int prow = pp >> WLOG, pcol = pp & (W - 1);
    const int hbase = ((prow + 1 + prow / HH) * WP + pcol + 1) * LDF + 8 * fq;
    V wr[9][2];
    {
        const bf16* Wp = reinterpret_cast<const bf16*>(a.w) + P.w_off;          // [n][tap][c]: the layer's forward pack
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) wr[t][i] = *reinterpret_cast<const V*>(Wp + ((16 * i + fr) * 9 + t) * CH + 8 * fq);
    }
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
    auto compute = [&](int tile, int stage) __attribute__((always_inline)) {
        const bf16* hal = st0 + stage * STG;
        const bf16* rs = hal + SHA;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const V af = *reinterpret_cast<const V*>(hal + hbase + ((t / 3 - 1) * WP + (t % 3 - 1)) * LDF);
            mma32(acc[0], wr[t][0], af);
            mma32(acc[1], wr[t][1], af);
        }
        const int gr0 = tile * TR;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = 16 * i + 4 * fq;
            f32x4 vv = acc[i];
            if constexpr (RES) {
                const Q rq = *reinterpret_cast<const Q*>(rs + pp * LDR + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] += to_f(rq[r]);
            }
            Q o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[i][r] += vv[r];
                s2[i][r] += vv[r] * vv[r];
                o[r] = (bf16)vv[r];
            }
            *reinterpret_cast<Q*>(O + ((int64_t)(gr0 + prow) * W + pcol) * CH + c) = o;
        }
    };

    // ---- pipeline (bwd3x3f.hip) -------------------------------------------------------------------------------------------------------------
    auto tile_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int t_last = t_begin + (nT - 1 - t_begin) / tstep * tstep;          // the block's last tile (the launcher guarantees t_begin < nT)
    const int n_tiles = (nT - 1 - t_begin) / tstep + 1;
    // threads 0..255 hold two halo vectors per tile, the others one: two copies of the loop (the slot count is a compile-time constant;
    // one generic lambda over both stage types sent the slot tables to scratch memory)
#define SV_FWDF_LOOP(NSLOTS)                                                                                   \
    {                                                                                                          \
        fwdf_stage<NSLOTS, RES> SA, SB;                                                                        \
        auto iter = [&](int tile, int stage, fwdf_stage<NSLOTS, RES>& S) __attribute__((always_inline)) {      \
            store_stage(S, min(tile + tstep, t_last), stage ^ 1);                                              \
            load_stage(S, min(tile + 3 * tstep, t_last));                                                      \
            compute(tile, stage);                                                                              \
            tile_barrier();                                                                                    \
        };                                                                                                     \
        load_stage(SA, t_begin);                                                                               \
        load_stage(SB, min(t_begin + tstep, t_last));                                                          \
        __syncthreads(); /* the coefficient vectors in LDS */                                                  \
        store_stage(SA, t_begin, 0);                                                                           \
        load_stage(SA, min(t_begin + 2 * tstep, t_last));                                                      \
        __syncthreads(); /* tile t_begin staged */                                                             \
        int tile = t_begin;                                                                                    \
        for (int k = 0; k + 1 < n_tiles; k += 2, tile += 2 * tstep) {                                          \
            iter(tile, 0, SB);                                                                                 \
            iter(tile + tstep, 1, SA);                                                                         \
        }                                                                                                      \
        if (n_tiles & 1) {                                                                                     \
            compute(tile, 0);                                                                                  \
            tile_barrier();                                                                                    \
        }                                                                                                      \
    }
    if (wave < 4) SV_FWDF_LOOP(2) else SV_FWDF_LOOP(1)
#undef SV_FWDF_LOOP

    // ---- the next BatchNorm's statistics: per-lane sums -> 16 pixel lanes -> the eight waves (LDS, doubles) -> one replica ------------
    if (a.stats) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[i][r] = row16_sum(s1[i][r]);
                s2[i][r] = row16_sum(s2[i][r]);
            }
        if (fr == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(&ssum[16 * i + 4 * fq + r], (double)s1[i][r]);
                    atomicAdd(&ssum[CH + 16 * i + 4 * fq + r], (double)s2[i][r]);
                }
        }
        __syncthreads();
        if (tid < 2 * CH) atomicAdd(a.stats + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * CH + tid, ssum[tid]);
    }
}

template <int WLOG, bool RES>
int launch(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int HH = (TR < W) ? TR : W, LROWS = TR + TR / HH + 1, HP = LROWS * WP;
    constexpr size_t lds_stages = (size_t)2 * (HP * LDF + (RES ? 128 * LDR : 0)) * 2 + 2 * CH * 8 + 2 * CH * 4;
    constexpr size_t lds = lds_stages > 4096 * 4 + 2 * CH * 4 + 16 ? lds_stages : 4096 * 4 + 2 * CH * 4 + 16;      // (the fold's scratch)
    const int G = sv_ngroups(a->groups);
    const int nT = g->B * g->Hin / TR;
    // one 512-thread block per CU; NOT 256 blocks per launch (bwd3x3f.hip / Engine.fused_blocks): 31 per XCD.  The persistent-block
    // option counts two blocks per CU: a per-launch budget of b stands for b / 2 CUs.
    int blocks = sv_persistent_blocks() / 2;
    if (blocks > 248) blocks = 248;
    int grid = blocks / G;
    if (grid > nT) grid = nT;
    if (grid < 1) grid = 1;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd3x3f_kernel<WLOG, RES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(fwd3x3f)");
        optin = true;
    }
    sv_igemm_args b = *a;          // this kernel folds the BatchNorm finalisation of its prologue
    if (!sv_fold_claim(b.fold_stats != nullptr)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((fwd3x3f_kernel<WLOG, RES>), dim3(grid, G), dim3(512), lds, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(fwd3x3f)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a 32 -> 32 channel stride-1 3x3 FORWARD (prologue [+ residual] + statistics) this kernel covers.
int sv_fwd3x3f_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_FWD3X3F) || dtype != SV_BF16) return 0;
    if (!a->pro_scale || a->bias || a->ex || a->sparse_out || (a->flags & SV_FLAG_DET) || sv_deterministic()) return 0;
    if (g->nphase != 1 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1 || g->T_orig != 9) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 9 || P.ooy != 0 || P.oox != 0) return 0;
    for (int t = 0; t < 9; ++t)          // geometry.conv_like(k = 3, stride = 1, pad = 1): tap t = 3 ky + kx reads x at (ky - 1, kx - 1)
        if (P.dy[t] != t / 3 - 1 || P.dx[t] != t % 3 - 1 || P.torig[t] != t) return 0;
    if (g->Cin != CH || g->N != CH || g->ldx != CH || g->ldo != CH) return 0;
    if (g->Hin != g->Win || g->Hout != g->Hin || g->Wout != g->Win || g->Hq != g->Hin || g->Wq != g->Win) return 0;
    if (g->Win != 8 && g->Win != 16 && g->Win != 32) return 0;
    if ((g->B * g->Hin) % (128 / g->Win) != 0) return 0;
    if (a->fold_stats && a->fold_replicas > 512) return 0;
    switch (g->Win) {
        case 32: *rc = a->residual ? launch<5, true>(g, a, s) : launch<5, false>(g, a, s); break;
        case 16: *rc = a->residual ? launch<4, true>(g, a, s) : launch<4, false>(g, a, s); break;
        default: *rc = a->residual ? launch<3, true>(g, a, s) : launch<3, false>(g, a, s); break;
    }
    return 1;
}
