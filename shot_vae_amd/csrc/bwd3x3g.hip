// Fused backward of a stride-1 3x3 convolution with 64 input and 64 output channels on 16 x 16 maps (bf16): bwd3x3f.hip's kernel for the
// second WideResNet stage (wideresnet.py:29-35 at 64 channels).  gfx950.  Same products, same three forms of the dy operand (MODE 0 / 1 /
// 2: a tensor, the BatchNorm backward of the layer behind the convolution formed in the load path, the same plus the residual
// branch with the side output), same pipeline (LDS double-buffered, two register stages, unconditional requests, pairs of tiles,
// LDS-only tile barrier, interleaved tile order) -- read bwd3x3f.hip first.  What differs:
//   * a tile is 64 pixels (4 rows of 16): 768 halo vectors per dy tensor and 512 centre vectors, as in the 32-channel kernel -- but the
//     DATA-gradient threads stage the halo (three vectors per tensor each) and the weight-gradient threads the centre (two each): the
//     144 accumulators of the latter leave room for nothing more (a first split -- two halo slots / one halo + one centre slot, half of
//     the weights in registers -- spilled 109-139 registers);
//   * data-gradient waves 0-3 = (channel half dc) x (pixel half dp): 32 c x 32 pixels, K = 9 taps x 64 n, ALL weights in LDS (83 KB:
//     a wave's half of them is 144 registers): 8 LDS fragment reads per 8 MFMAs;
//   * weight-gradient waves 4-7 = one 32 n x 32 c quadrant of dW each for all nine taps: 36 accumulator tiles, K = the tile's 64
//     pixels (two k-steps); the four activated-input fragments of a tile are read once, the dy fragments per tap.
// Per tile 288 MFMAs on either side, ~450 KB of LDS fragment reads: the LDS port is the bound (1.5 us per tile and CU).
#include "common.h"
#include "epilogue.h"

void sv_slab_reduce(const float* ws, int nslabs, int64_t n, float* dw, hipStream_t s);      // wgrad3x3.hip

namespace {

constexpr int CH = 64, WLOG = 4, W = 16, TR = 4, TP = 64, WP = W + 2;
constexpr int LDF = 72;     // LDS rows of the dy halo, the activated input and the weights: 64 channels + 8 (144 bytes) -- eight 16-byte
                            // fragment reads land on eight different bank groups (9 fr mod 16), the transposing 8-byte reads of four pixels
                            // x four channel quads on 32 different banks
constexpr int LDR = 72;     // LDS row of the raw input (the epilogue's 8-byte reads of 16 pixels: 16 different bank pairs)
constexpr int VPP = CH / 8; // 16-byte vectors per pixel
constexpr int LROWS = TR + 2, HP = LROWS * WP, HV = LROWS * W * VPP;
static_assert(HV == 768 && TP * VPP == 512, "slots: three halo vectors per tensor on the data-gradient threads, two centre vectors on the others");
constexpr int SDY = HP * LDF, SAC = TP * LDF, SXR = TP * LDR, STG = SDY + SAC + SXR;       // elements per LDS stage
constexpr int WLROWS = CH * 9;                                                             // LDS rows of weights: [c][tap]
static_assert((SDY * 2) % 16 == 0 && (SAC * 2) % 16 == 0 && (STG * 2) % 16 == 0, "16-byte aligned LDS images");
constexpr size_t LDS_BYTES = (size_t)2 * STG * 2 + 2 * CH * 8 + 8 * CH * 4 + (size_t)WLROWS * LDF * 2;

#ifndef SV_BWDG_PD
#define SV_BWDG_PD 1          // steps the data-gradient waves' fragment reads run ahead of their MFMAs
#endif
#ifndef SV_BWDG_WREG
#define SV_BWDG_WREG 0        // 16-channel tiles (of a data-gradient wave's two) whose weights live in registers (18 fragments each) in the
                             // residual form; the other forms take SV_BWDG_WREG01
#endif
#ifndef SV_BWDG_WREG01
#define SV_BWDG_WREG01 1        // (2: 177-186 spilled registers)
#endif
#ifndef SV_BWDG_HSTG2
#define SV_BWDG_HSTG2 0        // (1: 135 vs 126 us -- slower)
#endif
#ifndef SV_BWDG_DPP
#define SV_BWDG_DPP 1
#endif
#ifndef SV_BWDG_ABL
#define SV_BWDG_ABL 0        // timing ablations as in bwd3x3f.hip: 1 no weight-gradient MFMAs, 2 no data gradient, 4 no loads, 8 no staging, 16 no epilogue
#endif

struct bwdg_params {
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
struct bwdg_g { bwdg_params g[SV_MAX_GROUPS]; };

// bwd3x3f.hip's ftr with this file's row stride
__device__ __forceinline__ bf16x8 gtr(const bf16* S, int pix_elem_q, int col0, int lane) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const bf16* a0 = S + pix_elem_q + col0 + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * LDF));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

template <int MODE>
struct bwdg_halo {                 // register stage of a data-gradient thread
    bf16x8 gv[3];
    bf16x8 yv[MODE >= 1 ? 3 : 1];
    bf16x8 rv[MODE == 2 ? 3 : 1];
};
struct bwdg_centre {               // register stage of a weight-gradient thread
    bf16x8 xv[2];
};

template <int MODE>
__global__ __launch_bounds__(512) void bwd3x3g_kernel(const sv_geom g, const bwdg_g PG) {
    const bwdg_params& p = PG.g[blockIdx.y];
    typedef bf16x8 V;
    typedef bf16x4 Q;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const st0 = reinterpret_cast<bf16*>(smem);                     // [2][STG]: dy halo | activated input | raw input
    double* const ssum = reinterpret_cast<double*>(st0 + 2 * STG);       // [2][64]
    float* const cf = reinterpret_cast<float*>(ssum + 2 * CH);           // [7][64]: dy_scale, dy_scale2, dy_shift | x scale, shift, mean, rstd
    bf16* const wl = reinterpret_cast<bf16*>(cf + 8 * CH);               // [WLROWS][LDF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    const int NC = gridDim.x;
    const int tstep = NC, t_begin = sv_window_slot(NC, blockIdx.y, blockIdx.x);
    const sv_phase& P = g.phase[0];
    const char* __restrict__ DY = reinterpret_cast<const char*>(p.dy);
    const char* __restrict__ DY2 = MODE >= 1 ? reinterpret_cast<const char*>(p.dy2) : nullptr;
    const char* __restrict__ DY3 = MODE == 2 ? reinterpret_cast<const char*>(p.dy3) : nullptr;
    char* __restrict__ DYO = MODE == 2 ? reinterpret_cast<char*>(p.dy_out) : nullptr;
    const char* __restrict__ X = reinterpret_cast<const char*>(p.x);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(p.out);
    float slope = p.x_slope;
    asm volatile("v_mov_b32 %0, %0" : "+v"(slope));

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
    // the two padding columns of every halo row are zero for the kernel's lifetime
    for (int idx = tid; idx < 2 * LROWS * 2 * VPP; idx += 512) {
        const int vv = idx & (VPP - 1), side = (idx / VPP) & 1, row = (idx / (2 * VPP)) % LROWS, stg = (idx / (2 * VPP)) / LROWS;
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        *reinterpret_cast<bf16x8*>(st0 + stg * STG + (row * WP + side * (WP - 1)) * LDF + 8 * vv) = z;
    }

    // ---- staging slots: halo vector idx = pixel idx / 8 of the 6 x 16 halo rows, 8-channel group idx % 8; data-gradient thread tid
    // holds vectors tid, tid + 256, tid + 512.  kind: 1 = image row of the tile, 2 = the row above it, 3 = the row below it (valid only
    // inside the same image).  Centre vector idx = pixel idx / 8, group idx % 8: weight-gradient thread tid holds tid - 256 and tid.
    const int v = tid & (VPP - 1);
    int hlds[3];
    uint32_t hoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = (tid & 255) + 256 * i;
        const int pix = idx / VPP;
        const int lr = pix >> WLOG, xx = pix & (W - 1);
        const int kind = lr == 0 ? 2 : lr == LROWS - 1 ? 3 : 1;
        hoff[i] = (uint32_t)((lr * W + xx) * CH + 8 * v) * 2u;                // bytes from the row ABOVE the tile
        hlds[i] = ((lr * WP + xx + 1) * LDF + 8 * v) | kind;
    }
    const uint32_t hsafe = (uint32_t)(W * CH + 8 * v) * 2u;
    const int cp0 = (tid & 255) / VPP;                                        // centre pixels cp0 and cp0 + 32

    auto load_halo = [&](bwdg_halo<MODE>& S, int tile) __attribute__((always_inline)) {
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const int64_t hb = ((int64_t)gr0 - 1) * W * CH * 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kind = hlds[i] & 3;
            const bool ok = kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok);
            const uint32_t o = ok ? hoff[i] : hsafe;
            S.gv[i] = *reinterpret_cast<const V*>(DY + hb + o);
            if constexpr (MODE >= 1) S.yv[i] = *reinterpret_cast<const V*>(DY2 + hb + o);
            if constexpr (MODE == 2) S.rv[i] = *reinterpret_cast<const V*>(DY3 + hb + o);
        }
    };
    auto load_centre = [&](bwdg_centre& S, int tile) __attribute__((always_inline)) {
        const char* xb = X + (int64_t)tile * TR * W * CH * 2 + (uint32_t)(cp0 * CH + 8 * v) * 2u;
        S.xv[0] = *reinterpret_cast<const V*>(xb);
        S.xv[1] = *reinterpret_cast<const V*>(xb + 32 * CH * 2);
    };
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    auto store_halo = [&](bwdg_halo<MODE>& S, int tile, int stage) __attribute__((always_inline)) {
        bf16* sb = st0 + stage * STG;
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const int64_t hb = ((int64_t)gr0 - 1) * W * CH * 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kind = hlds[i] & 3;
            const bool ok = kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok);
            V o = S.gv[i];
            if constexpr (MODE >= 1) {
                // the BatchNorm backward of the layer behind the convolution as ONE expression (two fused multiply-adds, one rounding
                // to bf16); MODE 2 adds the skip connection's gradient in fp32 before that rounding (bwd3x3f.hip)
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
                if constexpr (MODE == 2) {
                    if (kind == 1) *reinterpret_cast<V*>(DYO + hb + hoff[i]) = o;
                }
            }
            if (!ok) o = zero;
            *reinterpret_cast<V*>(sb + (hlds[i] & ~7)) = o;
        }
    };
    auto store_centre = [&](bwdg_centre& S, int stage) __attribute__((always_inline)) {
        bf16* sb = st0 + stage * STG;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cf + 3 * CH + 8 * v), s1 = *reinterpret_cast<const f32x4*>(cf + 3 * CH + 8 * v + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(cf + 4 * CH + 8 * v), t1 = *reinterpret_cast<const f32x4*>(cf + 4 * CH + 8 * v + 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cp = cp0 + 32 * i;
            *reinterpret_cast<V*>(sb + SDY + cp * LDF + 8 * v) = bn_act8(S.xv[i], s0, s1, t0, t1, slope);
            *reinterpret_cast<V*>(sb + SDY + SAC + cp * LDR + 8 * v) = S.xv[i];
        }
    };

    // ---- data-gradient waves (0-3): channels 32 dc .. + 31 of pixels 32 dp .. + 31 ----------------------------------------------------
    const int dc = wave & 1, dp = (wave >> 1) & 1;
    int hbase[2], prow[2], pcol[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int pp = 32 * dp + 16 * ms + fr;
        prow[ms] = pp >> WLOG;
        pcol[ms] = pp & (W - 1);
        hbase[ms] = ((prow[ms] + 1) * WP + pcol[ms] + 1) * LDF + 8 * fq;
    }
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
    constexpr int NWR = MODE == 2 ? SV_BWDG_WREG : SV_BWDG_WREG01;
    V wr[NWR > 0 ? NWR : 1][18];                     // [tile][tap, k-step] of channel 32 dc + 16 tile + fr
    {
        const bf16* Wp = reinterpret_cast<const bf16*>(p.w) + P.w_off;          // [c][tap][n]: the layer's data-gradient pack
        for (int idx = tid; idx < WLROWS * VPP; idx += 512)                     // (visible after the prologue's barriers)
            *reinterpret_cast<V*>(wl + (idx / VPP) * LDF + 8 * (idx & (VPP - 1))) = *reinterpret_cast<const V*>(Wp + (idx / VPP) * CH + 8 * (idx & (VPP - 1)));
    }
    auto compute_d = [&](int tile, int stage) __attribute__((always_inline)) {
        const bf16* dyh = st0 + stage * STG;
        const bf16* xr = dyh + SDY + SAC;
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 18 steps (tap t, k-step k) of four fragment reads + four MFMAs, the reads of a step issued PD steps ahead into a ring of
        // PD + 1 fragment sets: left to itself the compiler kept three fragment registers and put every read right in front of its
        // MFMA (s_waitcnt lgkmcnt(0) before each of them: the whole LDS latency per MFMA, 85 us for these waves alone)
        // Pixel fragments: a 16-pixel fragment is a WHOLE image row (W = 16), so the fragments of the taps left and right of the centre
        // column are the centre fragment moved by one lane with ZERO coming in at the end of the row (the convolution's padding):
        // one LDS read + two DPP row shifts per (kernel row, k-step, pixel tile) instead of three reads -- 12 instead of 36 pixel-fragment
        // reads per tile and wave on a kernel that is bound by the LDS port (SV_BWDG_DPP 0: three reads).  Bit-identical operands.
        constexpr int PD = SV_BWDG_PD, NB = PD + 1;
        V fw0[NWR >= 1 ? 1 : NB], fw1[NWR >= 2 ? 1 : NB];
        V pc[2][2][2];                              // [buffer][k-step][pixel tile]: the centre-column fragments of a kernel row
        auto rdw = [&](int s_) __attribute__((always_inline)) {
            const int t = s_ >> 1, k = s_ & 1, b = s_ % NB;
            if (NWR < 1) fw0[NWR >= 1 ? 0 : b] = *reinterpret_cast<const V*>(wl + ((32 * dc + fr) * 9 + t) * LDF + 32 * k + 8 * fq);
            if (NWR < 2) fw1[NWR >= 2 ? 0 : b] = *reinterpret_cast<const V*>(wl + ((32 * dc + 16 + fr) * 9 + t) * LDF + 32 * k + 8 * fq);
        };
        auto rdp = [&](int ty) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) pc[ty & 1][k][ms] = *reinterpret_cast<const V*>(dyh + hbase[ms] + (1 - ty) * WP * LDF + 32 * k);
        };
        auto shifted = [&](const V& c, int tx) __attribute__((always_inline)) -> V {
            if (tx == 1) return c;
            typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ in = __builtin_bit_cast(u32x4_, c);
            u32x4_ o;
#pragma unroll
            for (int j = 0; j < 4; ++j)      // tap column 0 reads pixel x + 1 (the lane above: row_shl), column 2 pixel x - 1 (row_shr); bound_ctrl: 0 comes in
                o[j] = tx == 0 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)in[j], 0x101, 0xf, 0xf, true)
                               : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)in[j], 0x111, 0xf, 0xf, true);
            return __builtin_bit_cast(V, o);
        };
        rdp(0);
#pragma unroll
        for (int s_ = 0; s_ < PD; ++s_) rdw(s_);
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            const int t = s_ >> 1, k = s_ & 1, ty = t / 3, tx = t % 3, b = s_ % NB;
            if (s_ % 6 == 0 && ty < 2) rdp(ty + 1);
            if (s_ + PD < 18) rdw(s_ + PD);
            __builtin_amdgcn_sched_barrier(0);
            V a0, a1;
            if (SV_BWDG_DPP) {
                a0 = shifted(pc[ty & 1][k][0], tx);
                a1 = shifted(pc[ty & 1][k][1], tx);
            } else {
                const int sh = ((1 - ty) * WP + (1 - tx)) * LDF + 32 * k;
                a0 = *reinterpret_cast<const V*>(dyh + hbase[0] + sh);
                a1 = *reinterpret_cast<const V*>(dyh + hbase[1] + sh);
            }
            const V w0 = NWR >= 1 ? wr[0][s_] : fw0[NWR >= 1 ? 0 : b];
            const V w1 = NWR >= 2 ? wr[NWR >= 2 ? 1 : 0][s_] : fw1[NWR >= 2 ? 0 : b];
            mma32(acc[0][0], w0, a0);
            mma32(acc[0][1], w0, a1);
            mma32(acc[1][0], w1, a0);
            mma32(acc[1][1], w1, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int gr0 = tile * TR;
        if (SV_BWDG_ABL & 16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) s1[i][ms] += acc[i][ms][0] + acc[i][ms][1] + acc[i][ms][2] + acc[i][ms][3];
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = 32 * dc + 16 * i + 4 * fq;
            const f32x4 esc = *reinterpret_cast<const f32x4*>(cf + 3 * CH + c), esh = *reinterpret_cast<const f32x4*>(cf + 4 * CH + c);
            const f32x4 emu = *reinterpret_cast<const f32x4*>(cf + 5 * CH + c), ers = *reinterpret_cast<const f32x4*>(cf + 6 * CH + c);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const Q xq = *reinterpret_cast<const Q*>(xr + (32 * dp + 16 * ms + fr) * LDR + c);
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

    // ---- weight-gradient waves (4-7): rows n = 32 wi .., columns c = 32 wj .. of dW, all nine taps ----------------------------------
    const int wi = (wave >> 1) & 1, wj = wave & 1;
    f32x4 dacc[9][2][2];                              // [tap][16-row tile of n][16-column tile of c]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) dacc[t][a_][b_] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute_g = [&](int stage) __attribute__((always_inline)) {
        const bf16* dyh = st0 + stage * STG;
        const bf16* ac = dyh + SDY;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int pq = 32 * kc + 8 * fq + (fr >> 2);              // the lane addresses pixel pq of the tile (and pq + 4)
            const int jrow = pq >> WLOG, xcol = pq & (W - 1);
            const int hb = ((jrow + 1) * WP + xcol + 1) * LDF;
            bf16x8 fb[2];
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) fb[b_] = gtr(ac, pq * LDF, 32 * wj + 16 * b_, lane);
#pragma unroll
            for (int tg = 0; tg < 3; ++tg) {                          // three taps (one kernel row) at a time: six fragments in flight
                bf16x8 fa[3][2];
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
                        fa[tt][a_] = gtr(dyh, hb + ((1 - tg) * WP + (1 - tt)) * LDF, 32 * wi + 16 * a_, lane);
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                        for (int b_ = 0; b_ < 2; ++b_)
                            dacc[3 * tg + tt][a_][b_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tt][a_], fb[b_], dacc[3 * tg + tt][a_][b_], 0, 0, 0);
            }
        }
    };

    auto tile_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const int t_last = t_begin + (nT - 1 - t_begin) / tstep * tstep;
    const int n_tiles = (nT - 1 - t_begin) / tstep + 1;
    if (wave < 4) {
        {                             // (inside this branch: loaded in front of it the registers stayed allocated through the other one)
            const bf16* Wp = reinterpret_cast<const bf16*>(p.w) + P.w_off;
#pragma unroll
            for (int i = 0; i < NWR; ++i)
#pragma unroll
                for (int s_ = 0; s_ < 18; ++s_)
                    wr[i][s_] = *reinterpret_cast<const V*>(Wp + ((32 * dc + 16 * i + fr) * 9 + (s_ >> 1)) * CH + 32 * (s_ & 1) + 8 * fq);
        }
        // ONE register stage (bwd3x3f.hip has two): a tile takes ~3 us here, the request issued at the top of an iteration has a whole
        // iteration to arrive -- and the second stage's 24 / 48 / 72 registers are what the register-resident weights need
        // (SV_BWDG_HSTG2: a second stage in the residual form, whose six tensor passes make it the HBM-bound one)
        constexpr bool TWO = MODE == 2 && SV_BWDG_HSTG2;
        bwdg_halo<MODE> S, S2;
        auto iter = [&](int tile, int stage, bwdg_halo<MODE>& Q_) __attribute__((always_inline)) {
            if (!(SV_BWDG_ABL & 8)) store_halo(Q_, min(tile + tstep, t_last), stage ^ 1);
            if (!(SV_BWDG_ABL & 4)) load_halo(Q_, min(tile + (TWO ? 3 : 2) * tstep, t_last));
            if (!(SV_BWDG_ABL & 2)) compute_d(tile, stage);
            tile_barrier();
        };
        load_halo(S, t_begin);
        if (TWO) load_halo(S2, min(t_begin + tstep, t_last));
        __syncthreads();                                      // the coefficient vectors and the weights in LDS
        store_halo(S, t_begin, 0);
        load_halo(S, min(t_begin + (TWO ? 2 : 1) * tstep, t_last));
        __syncthreads();                                      // tile t_begin staged
        int tile = t_begin;
        for (int k = 0; k + 1 < n_tiles; k += 2, tile += 2 * tstep) {
            iter(tile, 0, TWO ? S2 : S);
            iter(tile + tstep, 1, S);
        }
        if (n_tiles & 1) {
            if (!(SV_BWDG_ABL & 2)) compute_d(tile, 0);
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
                    atomicAdd(&ssum[32 * dc + 16 * i + 4 * fq + r], (double)s1[i][r]);
                    atomicAdd(&ssum[CH + 32 * dc + 16 * i + 4 * fq + r], (double)s2[i][r]);
                }
        }
    } else {
        bwdg_centre S;
        auto iter = [&](int tile, int stage) __attribute__((always_inline)) {
            if (!(SV_BWDG_ABL & 8)) store_centre(S, stage ^ 1);
            if (!(SV_BWDG_ABL & 4)) load_centre(S, min(tile + 2 * tstep, t_last));
            if (!(SV_BWDG_ABL & 1)) compute_g(stage);
            tile_barrier();
        };
        load_centre(S, t_begin);
        __syncthreads();
        store_centre(S, 0);
        load_centre(S, min(t_begin + tstep, t_last));
        __syncthreads();
        int tile = t_begin;
        for (int k = 0; k + 1 < n_tiles; k += 2, tile += 2 * tstep) {
            iter(tile, 0);
            iter(tile + tstep, 1);
        }
        if (n_tiles & 1) {
            if (!(SV_BWDG_ABL & 1)) compute_g(0);
            tile_barrier();
        }
        // D layout: the lane holds column c = 32 wj + 16 b + fr, rows n = 32 wi + 16 a + 4 fq + r
        float* dst = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (9 * CH * CH);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[((32 * wi + 16 * a_ + 4 * fq + r) * 9 + t) * CH + 32 * wj + 16 * b_ + fr] = dacc[t][a_][b_][r];
    }
    __syncthreads();
    if (tid < 2 * CH) {
        double* dst = p.bsums + (size_t)(blockIdx.x & (p.replicas - 1)) * 2 * CH;
        atomicAdd(dst + tid, ssum[tid]);
    }
}

template <int MODE>
int launch_g(const sv_geom* g, const bwdg_g& PG, int grid, int groups, hipStream_t s) {
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd3x3g_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_BYTES) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(bwd3x3g)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((bwd3x3g_kernel<MODE>), dim3(grid, groups), dim3(512), LDS_BYTES, s, *g, PG);
    sv_prof_end(s);
    return sv_check_launch("sv_bwd3x3(64 channels)");
}

}  // namespace

// sv_bwd3x3 (bwd3x3f.hip) for 64 -> 64 channels on 16 x 16 maps: the arguments have been checked there.  Tiles of 4 image rows.
int sv_bwd3x3_64(const sv_geom* g, const sv_bwd3x3_args* a, hipStream_t s) {
    const int groups = sv_ngroups(a->groups);
    const int nT = g->B * g->Hin / TR;
    int budget = a->block_budget > 0 ? a->block_budget : 256;
    int grid = budget / groups;
    if (grid > nT) grid = nT;
    if (grid < 1) grid = 1;
    const int64_t slab = 9 * CH * CH;
    SV_REQUIRE(a->ws_elems >= (int64_t)grid * groups * slab, SV_E_ARG, "sv_bwd3x3: workspace of %lld floats, %lld needed",
               (long long)a->ws_elems, (long long)((int64_t)grid * groups * slab));
    bwdg_g PG;
    const int64_t ts = (int64_t)g->B * g->Hin * g->Win * CH * 2;           // bytes of one group's tensor
    for (int64_t grp = 0; grp < SV_MAX_GROUPS; ++grp) {
        const int64_t q = grp < groups ? grp : 0;
        bwdg_params& r = PG.g[grp];
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
    const int mode = a->dy3 ? 2 : a->dy2 ? 1 : 0;
    const int rc = mode == 2 ? launch_g<2>(g, PG, grid, groups, s) : mode == 1 ? launch_g<1>(g, PG, grid, groups, s) : launch_g<0>(g, PG, grid, groups, s);
    if (rc != SV_OK) return rc;
    sv_slab_reduce(a->ws, grid * groups, slab, a->dw, s);
    return sv_check_launch("sv_bwd3x3(slab reduce)");
}
