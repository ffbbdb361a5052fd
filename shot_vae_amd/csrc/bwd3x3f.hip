// Fused backward of a stride-1 3x3 convolution with 32 input and 32 output channels (bf16): the data gradient with its
// activation-backward epilogue AND the weight gradient in ONE persistent kernel.  gfx950.
//
//   g  [q][c]       = act'(x[q][c] * scale[c] + shift[c]) * sum_{t,n} dy[q + d(t)][n] * Wd[c][t][n]     (+ the two sums of g)
//   dW [n][to(t)][c] += sum_q dy[q + d(t)][n] * act(x[q][c] * scale[c] + shift[c])
//
// replaces the PAIR sv_igemm (ex epilogue) + sv_wgrad_ex of such a layer (wideresnet.py:29-35 under autograd) -- and the
// sv_bn_bwd_apply pass in front of the pair as well, because the gradient dy it works on can be FORMED IN ITS LOAD PATH:
//   MODE 0   dy                                                  (a tensor that exists)
//   MODE 1   dy_scale * dy + dy_scale2 * dy2 + dy_shift          = the BatchNorm backward of the layer behind the convolution
//                                                                  (norm2 between conv1 and conv2 of a unit; sv_bn_bwd_affine)
//   MODE 2   dy_scale * dy + dy_scale2 * dy2 + dy_shift + dy3    = the same plus the residual branch's gradient: the backward of
//                                                                  the NEXT unit's norm1 and the skip connection (wideresnet.py:45-49);
//                                                                  the tile's own rows of it are also written to dy_out, once (the
//                                                                  previous unit's skip gradient needs the tensor)
//
// Why one kernel: the pair reads dy twice and x twice (5 tensor passes; 8 with sv_bn_bwd_apply's 3, 9 with its 4 in the residual
// form); both launches of the pair are HBM-bound at 32 channels, so the sum of their bytes is what the step pays.  Here a tile's
// operands are read once: 3 passes (MODE 0), 4 (MODE 1), 6 (MODE 2: three inputs + x, two outputs).  Both products run from ONE
// LDS image of the tile:
//   * the dy HALO tile [(TR + 2) x (W + 2) pixels][32 n] (zero padding stored as data) serves the nine taps of the data
//     gradient (16-byte pixel fragments, weights register-resident: 72 registers) and -- read k-major with the transposing
//     ds_read_b64_tr_b16 at tap-shifted addresses -- is the A operand of the weight gradient;
//   * the activated input of the tile's 128 CENTER pixels is the weight gradient's B operand (the roles of the two tensors are
//     swapped against wgrad3x3.hip: sum_p dy[p] a[p + fd] = sum_q dy[q + d] a[q] with d = -fd, so one halo serves both products);
//   * the RAW input of the same pixels stays in LDS for the epilogue (activation derivative, xhat).
// Block = 8 waves, one block per CU.  Waves 0-3: the data gradient of 32 pixels each (v_mfma_f32_16x16x32_bf16, the accumulation
// order of conv3x3p_kernel: outputs bit-equal to it).  Waves 4-7: one 16 x 16 quadrant of dW each for all nine taps over the
// tile's 128 pixels.  36 MFMAs per wave and tile on either side; a SIMD hosts one wave of each kind.  ALL waves load and stage
// (weight-gradient waves two halo vectors + one centre vector per tensor and tile, data-gradient waves one + one); the two kinds
// run SEPARATE loops with the same barriers and do their staging at opposite ends of the iteration (data-gradient waves: stage,
// multiply, epilogue; weight-gradient waves: multiply, stage), so one wave's vector work runs under the other's MFMAs.
// Pipeline: LDS double-buffered (one LDS-only barrier per tile), two register stages (a tile's operands are requested two tiles
// ahead; every request is unconditional and the loops run whole pairs of tiles, so the compiler's waits are COUNTED: the other
// stage stays in flight).  Tiles are dealt interleaved (conv3x3p_kernel): the launch sweeps one compact window of the tensors front
// to back, an XCD owns consecutive tiles (their halo rows meet in its L2).
// The weight gradient leaves the block as ONE 36 KB slab per block (plain stores) + sv_slab_reduce.
#include "common.h"
#include "epilogue.h"

void sv_slab_reduce(const float* ws, int nslabs, int64_t n, float* dw, hipStream_t s);      // wgrad3x3.hip
int sv_bwd3x3_64(const sv_geom* g, const sv_bwd3x3_args* a, hipStream_t s);                 // bwd3x3g.hip: 64 channels on 16 x 16 maps

namespace {

constexpr int LDF = 48;     // LDS row of the dy halo and of the activated input: 32 channels + 16 (96 bytes) -- conflict-free for
                            // the 16-byte fragment reads (conv3x3.hip) and for the transposing 8-byte reads (wgrad3x3.hip)
constexpr int LDR = 40;     // LDS row of the raw input (80 bytes: the epilogue's 8-byte reads of 16 pixels hit 16 bank pairs)
constexpr int CH = 32;
// Fragment sets of the weight-gradient waves (compute_g): 1 = all ten fragments of a 32-pixel chunk at once (40 registers), 2 = two
// groups of 5 + 4 taps (24), 3 = two full sets, double-buffered (80).  Measured alone at 4 x 512 images (tools/probes/bwdf_ablate.sh,
// round 6, loader = the weight-gradient waves only): plain 125 / 115 / 121 us with 2 / 1 / 3; two-tensor 143 / 143 / 143.
#ifndef SV_BWDF_FRAGS
#define SV_BWDF_FRAGS 1
#endif
// 16-channel tiles of the data gradient whose weight fragments live in registers (36 each); the others are read from LDS
#ifndef SV_BWDF_WREGS
#define SV_BWDF_WREGS 2
#endif
// timing ablations (tools/probes/bwdf_ablate.sh; results wrong by construction): 1 = no weight-gradient MFMAs, 2 = no data-gradient
// MFMAs / epilogue, 4 = no global loads in the loop, 8 = no staging (transform + LDS stores), 16 = data gradient without its epilogue
// order of an iteration: 1 = the wave stages the next tile BEFORE its MFMAs.  Data-gradient waves: always (computing first: two-tensor
// form 167 vs 149 us, residual form 253 vs 158).  Weight-gradient waves (-1 = by form): behind their MFMAs where dy is a tensor (123 vs
// 126 us), in front of them in the two- and three-tensor forms (138 vs 149, 156 vs 158 us; tools/probes/bwdf_ablate.sh, round 6)
#ifndef SV_BWDF_DFIRST
#define SV_BWDF_DFIRST 1
#endif
#ifndef SV_BWDF_GFIRST
#define SV_BWDF_GFIRST -1
#endif
#ifndef SV_BWDF_ABL
#define SV_BWDF_ABL 0
#endif

struct bwdf_params {
    const void* dy;
    const void* dy2;
    const void* dy3;
    void* dy_out;
    const float* dy_scale;
    const float* dy_scale2;
    const float* dy_shift;
    const void* x;
    const float* x_scale;
    const float* x_shift;
    const float* x_mean;
    const float* x_rstd;
    float x_slope;
    const void* w;
    void* out;
    double* bsums;
    int replicas;
    float* ws;
    const double* fold_bsums;      // sv_bwd3x3_args::fold_* of this group (null: the coefficients are given)
    const float* fold_gamma;
    const float* fold_mean;
    const float* fold_rstd;
    float* fold_dgamma;
    float* fold_dbeta;
    float fold_inv_count;
    int fold_replicas;
};
struct bwdf_g { bwdf_params g[SV_MAX_GROUPS]; };

// 8 consecutive pixels (k = 8 fq + j) of one (shifted) image row, 16 channels starting at col0: a k-major fragment of
// v_mfma_f32_16x16x32_bf16 through two transposing reads (wgrad3x3.hip: frag_tr).  pix_elem_q = element offset of pixel
// 8 fq + q, q = (lane & 15) >> 2.
__device__ __forceinline__ bf16x8 ftr(const bf16* S, int pix_elem_q, int col0, int lane) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const bf16* a0 = S + pix_elem_q + col0 + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * LDF));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

// one register stage of a thread: NS halo vectors of every dy tensor + one centre vector of x
template <int NS, int MODE>
struct bwdf_stage {
    bf16x8 gv[NS];
    bf16x8 yv[MODE >= 1 ? NS : 1];
    bf16x8 rv[MODE == 2 ? NS : 1];
    bf16x8 xv;
};

template <int WLOG, int MODE>
__global__ __launch_bounds__(512) void bwd3x3f_kernel(const sv_geom g, const bwdf_g PG) {
    const bwdf_params& p = PG.g[blockIdx.y];
    typedef bf16x8 V;
    typedef bf16x4 Q;
    constexpr int WREGS = WLOG == 5 ? SV_BWDF_WREGS : (MODE == 0 ? 1 : 0);      // (the variants that do not spill)
    constexpr int FRAGS = WLOG == 5 ? SV_BWDF_FRAGS : 2;                       // fragment sets of the weight-gradient waves
    constexpr bool GFIRST = SV_BWDF_GFIRST < 0 ? MODE >= 1 : SV_BWDF_GFIRST != 0;
    constexpr int WLROWS = (2 - WREGS) * 16 * 9;       // LDS rows of weights [c][tap] of the tiles that are not register-resident
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    // LDS halo rows: row 0 / the last row are the vertical halo; when a tile holds two whole images (W = 8) a zero spacer row
    // separates them -- zero padding is DATA in LDS, the nine taps need no masks (conv3x3p_kernel)
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1, HP = LROWS * WP;
    constexpr int HV = LROWS * W * 4;                  // halo vectors (8 channels each; the two padding columns are zeroed once)
    static_assert(HV > 512 && HV <= 768, "slots: vectors 0..511 on the weight-gradient waves (two each), 512.. on the others (one each)");
    constexpr int SDY = HP * LDF, SAC = 128 * LDF, SXR = 128 * LDR, STG = SDY + SAC + SXR;      // elements per LDS stage
    static_assert((SDY * 2) % 16 == 0 && (SAC * 2) % 16 == 0 && (STG * 2) % 16 == 0, "16-byte aligned LDS images");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const st0 = reinterpret_cast<bf16*>(smem);                     // [2][STG]: dy halo | activated input | raw input
    double* const ssum = reinterpret_cast<double*>(st0 + 2 * STG);       // [2][32]
    float* const cf = reinterpret_cast<float*>(ssum + 2 * CH);           // [7][32]: dy_scale, dy_scale2, dy_shift | x scale, shift, mean, rstd
    bf16* const wl = reinterpret_cast<bf16*>(cf + 8 * CH);               // [WLROWS][LDF]: weights of the channel tiles >= WREGS

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    // interleaved tile order: at its step k the block takes tile k * NC + xsub; an XCD (blocks L, L + 8, ...) owns NC / 8
    // consecutive tiles of every window
    const int NC = gridDim.x;
    const int tstep = NC, t_begin = sv_window_slot(NC, blockIdx.y, blockIdx.x);
    // the taps of a stride-1 3x3 data gradient are FIXED (geometry.convT_like: tap t = 3 ky + kx reads dy at (1 - ky, 1 - kx) and
    // is master tap t; sv_bwd3x3 checks it): as compile-time constants every LDS fragment address is base + immediate -- read from
    // the geometry, the 36 + 18 tap-shifted addresses of the unrolled loops were hoisted and spilled
    const sv_phase& P = g.phase[0];
    const char* __restrict__ DY = reinterpret_cast<const char*>(p.dy);
    const char* __restrict__ DY2 = MODE >= 1 ? reinterpret_cast<const char*>(p.dy2) : nullptr;
    const char* __restrict__ DY3 = MODE == 2 ? reinterpret_cast<const char*>(p.dy3) : nullptr;
    char* __restrict__ DYO = MODE == 2 ? reinterpret_cast<char*>(p.dy_out) : nullptr;
    const char* __restrict__ X = reinterpret_cast<const char*>(p.x);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.out);
    float slope = p.x_slope;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));          // pinned in a vector register (no re-load from the argument segment)

    if (MODE >= 1 && p.fold_bsums) {
        // (ABI 8) the BatchNorm backward's coefficients from its raw sums, in every block; the stage area is free until the first tile is stored
        sv_bn_bwd_affine_block512<CH>(p.fold_bsums, p.fold_replicas, p.fold_inv_count, p.fold_gamma, p.fold_mean, p.fold_rstd,
                                      p.fold_dgamma, p.fold_dbeta, blockIdx.x == 0, reinterpret_cast<double*>(smem), cf);
    }
    if (tid < CH) {
        if (MODE >= 1 && !p.fold_bsums) {
            cf[tid] = p.dy_scale[tid];
            cf[CH + tid] = p.dy_scale2[tid];
            cf[2 * CH + tid] = p.dy_shift[tid];
        }
        cf[3 * CH + tid] = p.x_scale[tid];
        cf[4 * CH + tid] = p.x_shift[tid];
        cf[5 * CH + tid] = p.x_mean[tid];
        cf[6 * CH + tid] = p.x_rstd[tid];
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

    // ---- staging slots: everything that does not depend on the tile --------------------------------------------------------
    // halo slot kind: 0 = always zero (spacer / dummy), 1 = image row of this tile, 2 = row above the tile, 3 = row below it (valid
    // only inside the same image).  A thread's 8-channel group v is the same for all of its slots.  Weight-gradient waves (4-7)
    // hold halo vectors ltid and ltid + 256, data-gradient waves vector 512 + tid (most of them: HV = 768 / 640 / 608).
    // (The first version of this kernel loaded on the weight-gradient waves only -- believing that a wave with stores in flight can
    //  only wait with vmcnt(0).  What drained the queue were REQUESTS BEHIND BRANCHES: with every request unconditional the compiler
    //  counts, stores or not.  Sharing the staging halves the longest wave's vector work.)
    const int v = tid & 3;
    int hlds[2];                      // LDS element offset (a multiple of 8) | kind in the two low bits; -1: no slot
    uint32_t hoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = wave >= 4 ? (tid - 256) + 256 * i : (i == 0 ? 512 + tid : HV);
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
    const uint32_t hsafe = (uint32_t)(W * CH + 8 * v) * 2u;                   // (slots that are zero for this tile read its first pixel)
    const int cp = tid >> 2;                                                  // this thread's centre pixel
    const uint32_t coff = (uint32_t)(cp * CH + 8 * v) * 2u;

    auto load_stage = [&](auto& S, int tile) __attribute__((always_inline)) {
        constexpr int NS = sizeof(S.gv) / sizeof(V);
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const int64_t hb = ((int64_t)gr0 - 1) * W * CH * 2;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int kind = hlds[i] & 3;
            const bool ok = hlds[i] >= 0 && (kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok));
            const uint32_t o = ok ? hoff[i] : hsafe;
            S.gv[i] = *reinterpret_cast<const V*>(DY + hb + o);
            if constexpr (MODE >= 1) S.yv[i] = *reinterpret_cast<const V*>(DY2 + hb + o);
            if constexpr (MODE == 2) S.rv[i] = *reinterpret_cast<const V*>(DY3 + hb + o);
        }
        S.xv = *reinterpret_cast<const V*>(X + (int64_t)gr0 * W * CH * 2 + coff);
    };
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    auto store_stage = [&](auto& S, int tile, int stage) __attribute__((always_inline)) {
        constexpr int NS = sizeof(S.gv) / sizeof(V);
        bf16* sb = st0 + stage * STG;
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const int64_t hb = ((int64_t)gr0 - 1) * W * CH * 2;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int kind = hlds[i] & 3;
            const bool ok = kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok);
            V o = S.gv[i];
            if constexpr (MODE >= 1) {
                // the BatchNorm backward of the layer behind the convolution as ONE expression (two fused multiply-adds, one rounding
                // to bf16); MODE 2 adds the skip connection's gradient in fp32 before that rounding
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 ca = *reinterpret_cast<const f32x4*>(cf + 8 * v + 4 * h);
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(cf + CH + 8 * v + 4 * h);
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(cf + 2 * CH + 8 * v + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = to_f(S.gv[i][4 * h + j]) * ca[j] + (to_f(S.yv[MODE >= 1 ? i : 0][4 * h + j]) * cb[j] + cc[j]);
                        if constexpr (MODE == 2) t += to_f(S.rv[MODE == 2 ? i : 0][4 * h + j]);
                        o[4 * h + j] = (bf16)t;
                    }
                }
                // MODE 2: the tile's own rows of the formed gradient, once (no halo row, no padding column: every element of the
                // tensor belongs to exactly one tile's kind-1 slots)
                if constexpr (MODE == 2) {
                    if (hlds[i] >= 0 && kind == 1) *reinterpret_cast<V*>(DYO + hb + hoff[i]) = o;
                }
            }
            if (!ok) o = zero;
            if (hlds[i] >= 0) *reinterpret_cast<V*>(sb + (hlds[i] & ~7)) = o;
        }
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cf + 3 * CH + 8 * v), s1 = *reinterpret_cast<const f32x4*>(cf + 3 * CH + 8 * v + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(cf + 4 * CH + 8 * v), t1 = *reinterpret_cast<const f32x4*>(cf + 4 * CH + 8 * v + 4);
        *reinterpret_cast<V*>(sb + SDY + cp * LDF + 8 * v) = bn_act8(S.xv, s0, s1, t0, t1, slope);
        *reinterpret_cast<V*>(sb + SDY + SAC + cp * LDR + 8 * v) = S.xv;
    };

    // ---- data-gradient waves (0-3): 32 pixels of the tile each, both 16-channel tiles ------------------------------------------
    const int dwave = wave & 3;
    int hbase[2], prow[2], pcol[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int pp = 32 * dwave + 16 * ms + fr;
        prow[ms] = pp >> WLOG;
        pcol[ms] = pp & (W - 1);
        hbase[ms] = ((prow[ms] + 1 + prow[ms] / HH) * WP + pcol[ms] + 1) * LDF + 8 * fq;
    }
    V wr[9][WREGS > 0 ? WREGS : 1];
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
    {
        const bf16* Wp = reinterpret_cast<const bf16*>(p.w) + P.w_off;          // [c][tap][n]: the layer's data-gradient pack
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < WREGS; ++i) wr[t][i] = *reinterpret_cast<const V*>(Wp + ((16 * i + fr) * 9 + t) * CH + 8 * fq);
        }
        for (int idx = tid; idx < WLROWS * 4; idx += 512)                       // (visible after the prologue's barriers)
            *reinterpret_cast<V*>(wl + (idx >> 2) * LDF + 8 * (idx & 3)) =
                *reinterpret_cast<const V*>(Wp + (WREGS * 16 * 9 + (idx >> 2)) * CH + 8 * (idx & 3));
    }
    auto compute_d = [&](int tile, int stage) __attribute__((always_inline)) {
        const bf16* dyh = st0 + stage * STG;
        const bf16* xr = dyh + SDY + SAC;
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int sh = ((1 - t / 3) * WP + (1 - t % 3)) * LDF;
            const V af0 = *reinterpret_cast<const V*>(dyh + hbase[0] + sh);
            const V af1 = *reinterpret_cast<const V*>(dyh + hbase[1] + sh);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const V wf = i < WREGS ? wr[t][i < WREGS ? i : 0] : *reinterpret_cast<const V*>(wl + (((i - WREGS) * 16 + fr) * 9 + t) * LDF + 8 * fq);
                mma32(acc[i][0], wf, af0);
                mma32(acc[i][1], wf, af1);
            }
        }
        // epilogue: activation backward of the BatchNorm in front of the convolution + its two backward sums (conv3x3p_kernel's
        // arithmetic), raw input from LDS
        const int gr0 = tile * TR;
        if (SV_BWDF_ABL & 16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) s1[i][ms] += acc[i][ms][0] + acc[i][ms][1] + acc[i][ms][2] + acc[i][ms][3];
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = 16 * i + 4 * fq;
            const f32x4 esc = *reinterpret_cast<const f32x4*>(cf + 3 * CH + c), esh = *reinterpret_cast<const f32x4*>(cf + 4 * CH + c);
            const f32x4 emu = *reinterpret_cast<const f32x4*>(cf + 5 * CH + c), ers = *reinterpret_cast<const f32x4*>(cf + 6 * CH + c);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const Q xq = *reinterpret_cast<const Q*>(xr + (32 * dwave + 16 * ms + fr) * LDR + c);
                f32x4 vv = acc[i][ms];
                Q o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xf = to_f(xq[r]);
                    const float gv = vv[r] * act_grad(xf * esc[r] + esh[r], slope);
                    s1[i][r] += gv;
                    s2[i][r] += gv * ((xf - emu[r]) * ers[r]);
                    o[r] = (bf16)gv;
                }
                *reinterpret_cast<Q*>(O + ((int64_t)(gr0 + prow[ms]) * W + pcol[ms]) * CH + c) = o;
            }
        }
    };

    // ---- weight-gradient waves (4-7) -----------------------------------------------------------------------------------------------
    // one 16 x 16 quadrant (wi, wj) of dW for all nine taps over the tile's 128 pixels.  (Round 6 also tried v_mfma_f32_32x32x16_bf16 here
    // -- the whole 32 x 32 tile of five taps over half the pixels per wave: 96 instead of 160 KB of fragment reads per tile -- and
    // it was SLOWER: gonly 56 vs 47 us, launch 129 vs 123 us: these waves wait for LDS latency, not bandwidth; docs/lab_notes_r06.md)
    const int wi = (wave >> 1) & 1, wj = wave & 1;
    f32x4 dacc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) dacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute_g = [&](int stage) __attribute__((always_inline)) {
        const bf16* dyh = st0 + stage * STG;
        const bf16* ac = dyh + SDY;
        // fragments of one 32-pixel chunk: the activated-input fragment + the nine tap-shifted dy fragments, requested together
        // and double-buffered over the chunks (wgrad3x3_kernel)
        // fragments of one 32-pixel chunk: the activated-input fragment + the tap-shifted dy fragments in two groups (5 + 4 taps):
        // the reads of a group are issued right behind the MFMAs of the group before (which have read their operands at issue) and
        // return while the matrix pipe works those off; a full second set (wgrad3x3_kernel) cost 40 registers, all ten fragments 16
        auto frag_b = [&](int kc) __attribute__((always_inline)) {
            const int pq = 32 * kc + 8 * fq + (fr >> 2);          // the lane addresses pixel pq of the tile (and pq + 4)
            return ftr(ac, pq * LDF, 16 * wj, lane);
        };
        auto frag_a = [&](int kc, int t) __attribute__((always_inline)) {
            const int pq = 32 * kc + 8 * fq + (fr >> 2);
            const int jrow = pq >> WLOG, xcol = pq & (W - 1);
            const int hb = ((jrow + 1 + jrow / HH) * WP + xcol + 1) * LDF;
            return ftr(dyh, hb + ((1 - t / 3) * WP + (1 - t % 3)) * LDF, 16 * wi, lane);
        };
        if constexpr (FRAGS == 3) {
            // two fragment sets, double-buffered over the chunks (wgrad3x3_kernel)
            bf16x8 fbA, faA[9], fbB, faB[9];
            auto ld = [&](bf16x8& fb, bf16x8 (&fa)[9], int kc) __attribute__((always_inline)) {
                fb = frag_b(kc);
#pragma unroll
                for (int t = 0; t < 9; ++t) fa[t] = frag_a(kc, t);
            };
            auto mm = [&](const bf16x8& fb, const bf16x8 (&fa)[9]) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < 9; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], fb, dacc[t], 0, 0, 0);
            };
            ld(fbA, faA, 0);
            ld(fbB, faB, 1);
            mm(fbA, faA);
            ld(fbA, faA, 2);
            mm(fbB, faB);
            ld(fbB, faB, 3);
            mm(fbA, faA);
            mm(fbB, faB);
        } else if constexpr (FRAGS == 1) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 fb = frag_b(kc);
                bf16x8 fa[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) fa[t] = frag_a(kc, t);
#pragma unroll
                for (int t = 0; t < 9; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], fb, dacc[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 fb = frag_b(kc);
                bf16x8 fa[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) fa[t] = frag_a(kc, t);
#pragma unroll
                for (int t = 0; t < 5; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], fb, dacc[t], 0, 0, 0);
#pragma unroll
                for (int t = 5; t < 9; ++t) fa[t - 5] = frag_a(kc, t);
#pragma unroll
                for (int t = 5; t < 9; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t - 5], fb, dacc[t], 0, 0, 0);
            }
        }
    };

    // ---- pipeline ---------------------------------------------------------------------------------------------------------------------
    // The barrier between two tiles orders LDS traffic only.  __syncthreads() is also a release fence: behind the epilogue's global
    // stores it waits for their acknowledgements (vmcnt(0)), a memory round trip per tile.
    auto tile_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // The two kinds of wave run SEPARATE loops (the same number of barriers each): in one loop body behind a branch the register
    // allocator kept both kinds' loop-carried state alive together (weights + sums + both register stages + dW: spills).
    // In both: S holds tile + tstep (requested two iterations ago); it goes to the other LDS stage and is re-requested with
    // tile + 3 tstep.  Every request is issued UNCONDITIONALLY (past the end as a harmless re-load of the block's last tile) and the
    // loops run whole PAIRS of tiles, an odd last tile behind them: with a request -- or the second half of a pair -- behind a branch
    // the compiler's wait for S must also be right for the path that skipped it, i.e. it drains the queue.
    const int t_last = t_begin + (nT - 1 - t_begin) / tstep * tstep;          // the block's last tile (the launcher guarantees t_begin < nT)
    const int n_tiles = (nT - 1 - t_begin) / tstep + 1;
    if (wave < 4) {
        bwdf_stage<1, MODE> SA, SB;
        auto iter = [&](int tile, int stage, bwdf_stage<1, MODE>& S) __attribute__((always_inline)) {
            if (!SV_BWDF_DFIRST && !(SV_BWDF_ABL & 2)) compute_d(tile, stage);
            if (!(SV_BWDF_ABL & 8)) store_stage(S, min(tile + tstep, t_last), stage ^ 1);
            if (!(SV_BWDF_ABL & 4)) load_stage(S, min(tile + 3 * tstep, t_last));
            if (SV_BWDF_DFIRST && !(SV_BWDF_ABL & 2)) compute_d(tile, stage);
            tile_barrier();
        };
        load_stage(SA, t_begin);
        load_stage(SB, min(t_begin + tstep, t_last));
        __syncthreads();                                      // the coefficient vectors in LDS
        store_stage(SA, t_begin, 0);
        load_stage(SA, min(t_begin + 2 * tstep, t_last));
        __syncthreads();                                      // tile t_begin staged
        int tile = t_begin;
        for (int k = 0; k + 1 < n_tiles; k += 2, tile += 2 * tstep) {
            iter(tile, 0, SB);
            iter(tile + tstep, 1, SA);
        }
        if (n_tiles & 1) {
            if (!(SV_BWDF_ABL & 2)) compute_d(tile, 0);
            tile_barrier();
        }
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
    } else {
        bwdf_stage<2, MODE> SA, SB;
        auto iter = [&](int tile, int stage, bwdf_stage<2, MODE>& S) __attribute__((always_inline)) {
            if (!GFIRST && !(SV_BWDF_ABL & 1)) compute_g(stage);
            if (!(SV_BWDF_ABL & 8)) store_stage(S, min(tile + tstep, t_last), stage ^ 1);
            if (!(SV_BWDF_ABL & 4)) load_stage(S, min(tile + 3 * tstep, t_last));
            if (GFIRST && !(SV_BWDF_ABL & 1)) compute_g(stage);
            tile_barrier();
        };
        load_stage(SA, t_begin);
        load_stage(SB, min(t_begin + tstep, t_last));
        __syncthreads();
        store_stage(SA, t_begin, 0);
        load_stage(SA, min(t_begin + 2 * tstep, t_last));
        __syncthreads();
        int tile = t_begin;
        for (int k = 0; k + 1 < n_tiles; k += 2, tile += 2 * tstep) {
            iter(tile, 0, SB);
            iter(tile + tstep, 1, SA);
        }
        if (n_tiles & 1) {
            if (!(SV_BWDF_ABL & 1)) compute_g(0);
            tile_barrier();
        }
        // D layout: the lane holds column c = 16 wj + fr, rows n = 16 wi + 4 fq + r
        float* dst = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (9 * CH * CH);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((16 * wi + 4 * fq + r) * 9 + t) * CH + 16 * wj + fr] = dacc[t][r];
        }
    }
    __syncthreads();
    if (tid < 2 * CH) {
        double* dst = p.bsums + (size_t)(blockIdx.x & (p.replicas - 1)) * 2 * CH;
        atomicAdd(dst + tid, ssum[tid]);
    }
}

template <int WLOG, int MODE>
int launch(const sv_geom* g, const bwdf_g& PG, int grid, int groups, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int HH = (TR < W) ? TR : W, LROWS = TR + TR / HH + 1, HP = LROWS * WP;
    constexpr int WREGS = WLOG == 5 ? SV_BWDF_WREGS : (MODE == 0 ? 1 : 0);      // (as in the kernel)
    constexpr size_t lds = (size_t)2 * (HP * LDF + 128 * LDF + 128 * LDR) * 2 + 2 * CH * 8 + 8 * CH * 4 + (size_t)(2 - WREGS) * 16 * 9 * LDF * 2;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd3x3f_kernel<WLOG, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(bwd3x3f)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((bwd3x3f_kernel<WLOG, MODE>), dim3(grid, groups), dim3(512), lds, s, *g, PG);
    sv_prof_end(s);               // (the event bracket times the main kernel only, like the weight-gradient launchers)
    return sv_check_launch("sv_bwd3x3");
}

}  // namespace

extern "C" int sv_bwd3x3(const sv_geom* g, int dtype, const sv_bwd3x3_args* a, void* stream) {
    SV_REQUIRE(g && a, SV_E_ARG, "sv_bwd3x3: null argument");
    SV_REQUIRE(dtype == SV_BF16, SV_E_ARG, "sv_bwd3x3: bf16 only (dtype=%d)", dtype);
    SV_REQUIRE(a->dy && a->x && a->w && a->out && a->dw && a->ws && a->bsums, SV_E_ARG, "sv_bwd3x3: null tensor");
    SV_REQUIRE(a->x_scale && a->x_shift && a->x_mean && a->x_rstd, SV_E_ARG, "sv_bwd3x3: the BatchNorm vectors of x are required");
    SV_REQUIRE(a->x_slope >= 0.f && a->x_slope <= 1.f, SV_E_ARG, "sv_bwd3x3: slope %g outside [0, 1]", (double)a->x_slope);
    SV_REQUIRE(!a->dy2 || a->fold_bsums || (a->dy_scale && a->dy_scale2 && a->dy_shift), SV_E_ARG,
               "sv_bwd3x3: the two-tensor dy operand needs dy_scale, dy_scale2 and dy_shift (or the fold_* sums)");
    SV_REQUIRE(!a->fold_bsums || (a->dy2 && a->fold_gamma && a->fold_mean && a->fold_rstd && a->fold_replicas >= 1 && a->fold_count > 0.f),
               SV_E_ARG, "sv_bwd3x3: incomplete BatchNorm-backward fold (fold_* need dy2, gamma, mean, rstd, replicas >= 1, count > 0)");
    SV_REQUIRE((a->dy3 != nullptr) == (a->dy_out != nullptr) && (!a->dy3 || a->dy2), SV_E_ARG,
               "sv_bwd3x3: the residual form needs dy2 (+ coefficients), dy3 AND dy_out");
    const int groups = sv_ngroups(a->groups);
    SV_REQUIRE(groups <= SV_MAX_GROUPS, SV_E_ARG, "sv_bwd3x3: groups=%d", groups);
    SV_REQUIRE(a->replicas >= 1 && (a->replicas & (a->replicas - 1)) == 0, SV_E_ARG, "sv_bwd3x3: replicas=%d", a->replicas);
    // the geometry is the layer's DATA-gradient geometry (geometry.convT_like of a stride-1 3x3 convolution): one phase of nine taps
    const bool shape_ok = g->nphase == 1 && g->phase[0].ntap == 9 && g->T_orig == 9 && g->sy == 1 && g->sx == 1 && g->osy == 1 &&
                          g->osx == 1 && g->Hq == g->Hin && g->Wq == g->Win && g->Hout == g->Hin && g->Wout == g->Win &&
                          g->Hin == g->Win && g->Cin == g->N && g->ldx == g->Cin && g->ldo == g->N && g->phase[0].ooy == 0 &&
                          g->phase[0].oox == 0 &&
                          ((g->Cin == CH && (g->Win == 8 || g->Win == 16 || g->Win == 32)) || (g->Cin == 64 && g->Win == 16));
    SV_REQUIRE(shape_ok, SV_E_SHAPE, "sv_bwd3x3: stride-1 3x3 layers with 32 -> 32 channels on 8 / 16 / 32-pixel maps or 64 -> 64 channels on 16-pixel maps only");
    for (int t = 0; t < 9; ++t)
        SV_REQUIRE(g->phase[0].dy[t] == 1 - t / 3 && g->phase[0].dx[t] == 1 - t % 3 && g->phase[0].torig[t] == t, SV_E_SHAPE,
                   "sv_bwd3x3: tap %d is not the data-gradient tap of geometry.convT_like(k = 3, stride = 1, pad = 1)", t);
    const int TR = g->Cin == 64 ? 4 : 128 / g->Win;
    SV_REQUIRE((g->B * g->Hin) % TR == 0, SV_E_SHAPE, "sv_bwd3x3: B * H = %d is not a multiple of the %d-row tile", g->B * g->Hin, TR);
    SV_REQUIRE(!sv_deterministic() && !sv_det_stats(), SV_E_ARG, "sv_bwd3x3: not available in deterministic mode (use the pair)");
    if (g->Cin == 64) return sv_bwd3x3_64(g, a, (hipStream_t)stream);
    const int nT = g->B * g->Hin / TR;
    int budget = a->block_budget > 0 ? a->block_budget : 256;
    int grid = budget / groups;
    if (grid > nT) grid = nT;
    if (grid < 1) grid = 1;
    const int64_t slab = 9 * CH * CH;
    SV_REQUIRE(a->ws_elems >= (int64_t)grid * groups * slab, SV_E_ARG, "sv_bwd3x3: workspace of %lld floats, %lld needed",
               (long long)a->ws_elems, (long long)((int64_t)grid * groups * slab));
    hipStream_t s = (hipStream_t)stream;
    bwdf_g PG;
    const int64_t ts = (int64_t)g->B * g->Hin * g->Win * CH * 2;           // bytes of one group's tensor
    for (int64_t grp = 0; grp < SV_MAX_GROUPS; ++grp) {
        const int64_t q = grp < groups ? grp : 0;
        bwdf_params& r = PG.g[grp];
        r.dy = reinterpret_cast<const char*>(a->dy) + q * ts;
        r.dy2 = a->dy2 ? reinterpret_cast<const char*>(a->dy2) + q * ts : nullptr;
        r.dy3 = a->dy3 ? reinterpret_cast<const char*>(a->dy3) + q * ts : nullptr;
        r.dy_out = a->dy_out ? reinterpret_cast<char*>(a->dy_out) + q * ts : nullptr;
        const bool fold = a->dy2 && a->fold_bsums;
        r.dy_scale = a->dy2 && !fold ? a->dy_scale + q * CH : nullptr;
        r.dy_scale2 = a->dy2 && !fold ? a->dy_scale2 + q * CH : nullptr;
        r.dy_shift = a->dy2 && !fold ? a->dy_shift + q * CH : nullptr;
        r.fold_bsums = fold ? a->fold_bsums + q * (int64_t)a->fold_replicas * 2 * CH : nullptr;
        r.fold_gamma = a->fold_gamma;
        r.fold_mean = fold ? a->fold_mean + q * CH : nullptr;
        r.fold_rstd = fold ? a->fold_rstd + q * CH : nullptr;
        r.fold_dgamma = a->fold_dgamma;
        r.fold_dbeta = a->fold_dbeta;
        r.fold_inv_count = fold ? 1.f / a->fold_count : 0.f;
        r.fold_replicas = a->fold_replicas;
        r.x = reinterpret_cast<const char*>(a->x) + q * ts;
        r.x_scale = a->x_scale + q * CH;
        r.x_shift = a->x_shift + q * CH;
        r.x_mean = a->x_mean + q * CH;
        r.x_rstd = a->x_rstd + q * CH;
        r.x_slope = a->x_slope;
        r.w = a->w;
        r.out = reinterpret_cast<char*>(a->out) + q * ts;
        r.bsums = a->bsums + q * (int64_t)a->replicas * 2 * CH;
        r.replicas = a->replicas;
        r.ws = a->ws;
    }
    int rc;
    const int mode = a->dy3 ? 2 : a->dy2 ? 1 : 0;
    switch (g->Win * 4 + mode) {
        case 32 * 4 + 0: rc = launch<5, 0>(g, PG, grid, groups, s); break;
        case 32 * 4 + 1: rc = launch<5, 1>(g, PG, grid, groups, s); break;
        case 32 * 4 + 2: rc = launch<5, 2>(g, PG, grid, groups, s); break;
        case 16 * 4 + 0: rc = launch<4, 0>(g, PG, grid, groups, s); break;
        case 16 * 4 + 1: rc = launch<4, 1>(g, PG, grid, groups, s); break;
        case 16 * 4 + 2: rc = launch<4, 2>(g, PG, grid, groups, s); break;
        case 8 * 4 + 0: rc = launch<3, 0>(g, PG, grid, groups, s); break;
        case 8 * 4 + 1: rc = launch<3, 1>(g, PG, grid, groups, s); break;
        default: rc = launch<3, 2>(g, PG, grid, groups, s); break;
    }
    if (rc != SV_OK) return rc;
    sv_slab_reduce(a->ws, grid * groups, slab, a->dw, s);
    return sv_check_launch("sv_bwd3x3(slab reduce)");
}
