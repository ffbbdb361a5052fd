// Stride-1 3x3 convolution (forward and data gradient) with an LDS-resident halo tile.  gfx950.
//
// The generic gather-GEMM (igemm.hip) re-gathers the input once per tap, i.e. moves ~9x the
// algorithmic bytes through L1/L2 -- measured L2-bound on the WRN body convs.  Here a block owns a
// tile of 128 output pixels (TR = 128/W whole image rows), stages the (TR+2) x (W+2) input halo ONCE
// per 32-channel chunk -- BatchNorm-apply + LeakyReLU applied once per element on the way in, zero
// padding written as zeros -- together with the [BN][9][32] weight chunk, and then runs all nine taps
// x BN channels on MFMA straight out of LDS (the tap shift is just an LDS address offset).
// Same sv_geom / packed weights / fused epilogue as sv_igemm: it is a drop-in fast path inside it.
#include <stdlib.h>

#include "common.h"
#include "epilogue.h"

#ifndef SV_C3P_WREG
#define SV_C3P_WREG 1
#endif
#ifndef SV_C3P_INTERLEAVE
#define SV_C3P_INTERLEAVE 1    // tile order of the persistent kernel: 1 = all blocks sweep one moving window, 0 = a contiguous range per block
#endif
#ifndef SV_C3P_DEPTH
#define SV_C3P_DEPTH 2         // register stages of the halo (request distance in tiles); 3: one more stage, operands two tiles ahead
                               // (measured: data gradient 92 -> 96 / 169 us without / with the weights in registers -- spills)
#endif
#ifndef SV_C3P_EOP_AHEAD
#define SV_C3P_EOP_AHEAD 1     // the epilogue operand of a tile is requested one tile ahead (0: inside its own tile)
#endif
#ifndef SV_C3P_MODES
#define SV_C3P_MODES 1         // fusion flags of conv3x3p at compile time for the step's three launch kinds (0: run-time flags only)
#endif
#ifndef SV_C3P_WAVES
#define SV_C3P_WAVES 2          // waves per SIMD the persistent kernel is compiled for (3 => spills, measured slower)
#endif

namespace {

constexpr int CK = 32;          // channel chunk = one MFMA k step
constexpr int LDC = CK + 16;    // LDS pixel / weight-row stride (elements): 96 B (bf16) keeps the b128 fragment reads conflict-free

template <typename T, int NT, int WLOG>
__global__ __launch_bounds__(256) void conv3x3_kernel(const sv_geom g, const sv_igemm_args_g A) {
    const sv_igemm_args& a = A.g[blockIdx.y];
    sv_start_signal(a);
    typedef typename V8<T>::type V;
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int HP = (TR + 2) * WP;           // halo pixels
    constexpr int BN = 16 * NT;
    constexpr int HV = HP * (CK / 8), WV = BN * 9 * (CK / 8);
    constexpr int HI = (HV + 255) / 256, WI = (WV + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = reinterpret_cast<T*>(smem);             // [HP][LDC]
    T* wl = halo + HP * LDC;                          // [BN*9][LDC]
    double* ssum = reinterpret_cast<double*>(wl + BN * 9 * LDC);   // [2][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int H = g.Hin;
    const int BH = g.B * H;                           // global image rows
    const int nT = BH / TR;
    const int nNt = g.N / BN;
    const int L = blockIdx.x;
    int in_i, mt;
    if (nT >= 64) {                                   // XCD-affine: channel tiles of one pixel tile share an L2
        const int xcd = L & 7, slot = L >> 3;
        in_i = slot % nNt;
        mt = (slot / nNt) * 8 + xcd;
        if (mt >= nT) return;
    } else {
        in_i = L % nNt;
        mt = L / nNt;
    }
    const int n0 = in_i * BN;
    const int gr0 = mt * TR;
    const sv_phase& P = g.phase[0];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(a.w) + P.w_off + (int64_t)n0 * 9 * g.Cin;
    const bool has_pro = a.pro_scale != nullptr;

    if (tid < 2 * BN) ssum[tid] = 0.0;

    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;

    // per-thread staging slots (the 8-channel vector index v = tid & 3 is the same for all of them)
    const int v = tid & 3;
    int hoff[HI], hlds[HI];
    bool hok[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int pix = min(idx, HV - 1) >> 2;
        const int j = pix / WP, xx = pix - j * WP;
        const int gr = gr0 + j - 1, x = xx - 1;
        hok[i] = idx < HV && (unsigned)x < (unsigned)W && (unsigned)gr < (unsigned)BH;
        const int grc = min(max(gr, 0), BH - 1), xc = min(max(x, 0), W - 1);
        hoff[i] = (grc * W + xc) * g.ldx + 8 * v;
        hlds[i] = idx < HV ? pix * LDC + 8 * v : -1;
    }

    f32x4 acc[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's two output pixels
    int hbase[2], yrow[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int p = 32 * wave + 16 * ms + fr;
        const int j = p >> WLOG, x = p & (W - 1);
        hbase[ms] = ((j + 1) * WP + x + 1) * LDC + 8 * fq;
        yrow[ms] = (gr0 + j) & (H - 1);
    }

    const int nck = g.Cin / CK;
    for (int ck = 0; ck < nck; ++ck) {
        const int c0 = ck * CK;
        // ---- stage: all global loads first (in flight together), then transform + LDS stores ----
        V hv[HI], wv[WI];
#pragma unroll
        for (int i = 0; i < HI; ++i) hv[i] = *reinterpret_cast<const V*>(X + hoff[i] + c0);
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int idx = min(tid + 256 * i, WV - 1);
            wv[i] = *reinterpret_cast<const V*>(Wp + (int64_t)(idx >> 2) * g.Cin + c0 + 8 * v);
        }
        f32x4 s0, s1, t0, t1;
        if (has_pro) {
            s0 = *reinterpret_cast<const f32x4*>(a.pro_scale + c0 + 8 * v);
            s1 = *reinterpret_cast<const f32x4*>(a.pro_scale + c0 + 8 * v + 4);
            t0 = *reinterpret_cast<const f32x4*>(a.pro_shift + c0 + 8 * v);
            t1 = *reinterpret_cast<const f32x4*>(a.pro_shift + c0 + 8 * v + 4);
        }
        if (ck > 0) __syncthreads();       // previous chunk's MFMAs are done with the LDS tiles
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            V o = zero;
            if (hok[i]) {
                o = hv[i];
                if (has_pro) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (T)act_fwd(to_f(hv[i][j]) * s0[j] + t0[j], a.pro_slope);
                        o[j + 4] = (T)act_fwd(to_f(hv[i][j + 4]) * s1[j] + t1[j], a.pro_slope);
                    }
                }
            }
            if (hlds[i] >= 0) *reinterpret_cast<V*>(halo + hlds[i]) = o;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int idx = tid + 256 * i;
            if (idx < WV) *reinterpret_cast<V*>(wl + (idx >> 2) * LDC + 8 * v) = wv[i];
        }
        __syncthreads();
        // ---- nine taps out of LDS --------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = tap_off(pdy, t), dx = tap_off(pdx, t);
            const int sh = (dy * WP + dx) * LDC;
            V af[2];
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const bool ok = !((dy < 0 && yrow[ms] == 0) || (dy > 0 && yrow[ms] == H - 1));
                const V f = *reinterpret_cast<const V*>(halo + hbase[ms] + sh);
                af[ms] = ok ? f : zero;
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const V wf = *reinterpret_cast<const V*>(wl + ((16 * i + fr) * 9 + t) * LDC + 8 * fq);
                mma32(acc[i][0], wf, af[0]);
                mma32(acc[i][1], wf, af[1]);
            }
        }
    }

    int64_t obase[2];
    bool oval[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int p = 32 * wave + 16 * ms + fr;
        obase[ms] = ((int64_t)(gr0 + (p >> WLOG)) * W + (p & (W - 1))) * g.ldo;
        oval[ms] = true;
    }
    gemm_epilogue<T, NT>(acc, obase, oval, n0, g.N, a, ssum);
}


// ------------------------------------------------------------------------------------------------------
// Persistent variant for layers whose whole weight slab fits in LDS (Cin <= 64 at 32 output channels per
// block): weights are staged ONCE per block, the block then walks a contiguous range of pixel tiles with a
// register-prefetch software pipeline -- while tile i is on the MFMAs, the halo of tile i+1 and the
// residual / raw tensor needed by tile i's epilogue are already in flight -- and the BatchNorm sums are
// kept in registers across tiles and flushed once per block (one shuffle tree, one atomic per channel).
// MODE: the fusion flags at compile time (0 = read from the arguments; 1 = prologue + statistics, 2 = prologue + residual +
// statistics, 3 = activation-backward epilogue, no prologue -- the three launch kinds of the training step; no bias in 1..3).
// (Round 5's mode 6 -- the BatchNorm backward of the layer in front formed in the load path from two tensors -- lost in the step and
//  left in round 6: that fusion lives in bwd3x3f.hip, where ONE kernel forms it for the data and the weight gradient together.)
template <typename T, int WLOG, int CCH, int MODE>      // CCH = Cin / 32
__global__ __launch_bounds__(256, SV_C3P_WAVES) void conv3x3p_kernel(const sv_geom g, const sv_igemm_args_g A, int tiles_per) {
    const sv_igemm_args& a = A.g[blockIdx.y];
    sv_start_signal(a);
    typedef typename V8<T>::type V;
    typedef typename V4<T>::type Q;
    constexpr int NT = 2, BN = 32;
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int CIN = 32 * CCH, LDW = CIN + 16;           // +16 elements: conflict-free ds_read_b128 fragments
    constexpr int VPP = CIN / 8;                             // 8-channel vectors per pixel
    // LDS halo rows: row 0 / the last row are the vertical halo, and when a tile holds several whole images
    // (W = 8: TR = 16 > H = 8) a zero spacer row separates them -- so zero padding is DATA in LDS and the nine
    // taps need no per-lane masking at all.
    constexpr int HH = (TR < W) ? TR : W;                    // image rows per segment (images are square: H == W)
    constexpr int SEG = TR / HH;
    constexpr int LROWS = TR + SEG + 1;
    constexpr int HV = LROWS * WP * VPP;                     // halo vectors
    constexpr int HI = (HV + 255) / 256;
    constexpr int HPIX = (HI * 256 + VPP - 1) / VPP;         // LDS pixels incl. dummy tail (branch-free staging)
    constexpr int WV = BN * 9 * VPP, WI = (WV + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = reinterpret_cast<T*>(smem);                    // [HPIX][LDW]
    T* wl = halo + HPIX * LDW;                               // [BN*9][LDW]
    double* ssum = reinterpret_cast<double*>(wl + BN * 9 * LDW);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    const int nNt = g.N / BN;
    // the channel tiles of one pixel range read the same input: keep them on one XCD (blocks L, L+8, ... share an
    // L2) so that only the first of them goes to HBM (PMC: 1.73x the algorithmic bytes at 64 channels before).
    // Tile order: INTERLEAVED -- at its step k a block takes tile k * NC + xsub, so that the NC blocks of the launch sweep a
    // compact window of NC consecutive tiles together, front to back, instead of each walking a contiguous range of its own:
    // hundreds of separate streams ran HBM at 3.3-5 TB/s where one moving window reaches 5.5-6.6 (tools/probes/mall_probe.hip).
    // Inside the window an XCD owns NC / 8 consecutive tiles: vertically adjacent tiles share their halo rows through its L2.
    const int NC = gridDim.x / nNt;
    int in_i, xsub;
    if (NC % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        in_i = slot % nNt;
        xsub = SV_C3P_INTERLEAVE ? xcd * (NC / 8) + slot / nNt : (slot / nNt) * 8 + xcd;
    } else {
        in_i = blockIdx.x % nNt;
        xsub = blockIdx.x / nNt;
    }
    const int n0 = in_i * BN;
    // step k of this block: tile k * tstep + t_begin, while it is < t_end
    const int tstep = SV_C3P_INTERLEAVE ? NC : 1;
    const int t_begin = SV_C3P_INTERLEAVE ? xsub : xsub * tiles_per;
    const int t_end = SV_C3P_INTERLEAVE ? nT : min(nT, t_begin + tiles_per);
    if (t_begin >= t_end) return;
    const sv_phase& P = g.phase[0];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(a.w) + P.w_off + (int64_t)n0 * 9 * CIN;
    T* __restrict__ O = reinterpret_cast<T*>(a.out);
    const T* __restrict__ R = MODE == 0 || MODE == 2 ? reinterpret_cast<const T*>(a.residual) : nullptr;
    const T* __restrict__ EX = MODE == 0 || MODE == 3 ? reinterpret_cast<const T*>(a.ex) : nullptr;
    const bool hasR = MODE == 0 ? R != nullptr : MODE == 2, hasEX = MODE == 0 ? EX != nullptr : MODE == 3;
    const bool has_stats = MODE == 0 ? a.stats != nullptr : (MODE == 1 || MODE == 2);
    const bool has_pro = MODE == 0 ? a.pro_scale != nullptr : MODE < 3;
    const bool want_sums = has_stats || hasEX;

    if (tid < 2 * BN) ssum[tid] = 0.0;
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;

    // ---- weights: once per block ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int idx = min(tid + 256 * i, WV - 1);           // (duplicates of the last vector are harmless)
        const int row = idx / VPP, vv = idx - row * VPP;
        *reinterpret_cast<V*>(wl + row * LDW + 8 * vv) = *reinterpret_cast<const V*>(Wp + (int64_t)row * CIN + 8 * vv);
    }
    // ---- halo staging slots: everything that does not depend on the tile -----------------------------
    // kind: 0 = always zero (padding column / spacer / dummy), 1 = image row of this tile,
    //       2 = row above the tile, 3 = row below the tile (valid only inside the same image)
    int hrel[HI], hxc[HI], hlds[HI], hc[HI], hkind[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int pix = idx / VPP;
        hc[i] = 8 * (idx - pix * VPP);
        hlds[i] = pix * LDW + hc[i];
        const int lr = pix / WP, xx = pix - lr * WP;
        const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
        int kind = 1, rel = lr - 1 - seg;
        if (off == 0) {
            if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
            else kind = 0;
        }
        if (idx >= HV || xx == 0 || xx == WP - 1) kind = 0;
        hkind[i] = kind;
        hrel[i] = rel;
        hxc[i] = min(max(xx - 1, 0), W - 1);
    }
    // a thread's 8-channel group is the same for all of its slots (256 % VPP == 0): BN scale / shift live in
    // registers for the whole kernel instead of being re-fetched (a dependent L2 round trip) every tile
    static_assert(256 % VPP == 0, "");
    f32x4 ps0 = {1.f, 1.f, 1.f, 1.f}, ps1 = ps0, pt0 = {0.f, 0.f, 0.f, 0.f}, pt1 = pt0;
    if (has_pro && a.fold_stats) {
        // the BatchNorm in front of this layer has not been finalised: every block derives the coefficients from the raw
        // statistics itself (sv_igemm_args::fold_*; the halo area is free until the first tile is stored), the first block
        // of the launch also stores them for the backward pass
        float* fs = reinterpret_cast<float*>(halo);
        sv_bn_fold_block(a, CIN, reinterpret_cast<double*>(halo), fs + 1024, fs + 1024 + CIN, blockIdx.x == 0);
        ps0 = *reinterpret_cast<const f32x4*>(fs + 1024 + hc[0]);
        ps1 = *reinterpret_cast<const f32x4*>(fs + 1024 + hc[0] + 4);
        pt0 = *reinterpret_cast<const f32x4*>(fs + 1024 + CIN + hc[0]);
        pt1 = *reinterpret_cast<const f32x4*>(fs + 1024 + CIN + hc[0] + 4);
        __syncthreads();
    } else if (has_pro) {
        ps0 = *reinterpret_cast<const f32x4*>(a.pro_scale + hc[0]);
        ps1 = *reinterpret_cast<const f32x4*>(a.pro_scale + hc[0] + 4);
        pt0 = *reinterpret_cast<const f32x4*>(a.pro_shift + hc[0]);
        pt1 = *reinterpret_cast<const f32x4*>(a.pro_shift + hc[0] + 4);
    }
    // two register stages: the halo of tile i+2 is requested while tile i is on the MFMAs, so every halo has
    // two full tile periods to arrive (at 2 blocks per CU one tile period does not cover the memory latency)
    struct HStage { V hv[HI]; bool hok[HI]; };
    HStage HA, HB, HC;          // (HC: SV_C3P_DEPTH 3 only)
    auto load_halo = [&](HStage& S, int tile) {
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            S.hok[i] = hkind[i] == 1 || (hkind[i] == 2 && top_ok) || (hkind[i] == 3 && bot_ok);
            const int grc = min(max(gr0 + hrel[i], 0), BH - 1);
            S.hv[i] = *reinterpret_cast<const V*>(X + ((int64_t)grc * W + hxc[i]) * g.ldx + hc[i]);
        }
    };
    auto store_halo = [&](HStage& S, int tile) {
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            V o = S.hv[i];
            if (has_pro) o = bn_act8(S.hv[i], ps0, ps1, pt0, pt1, a.pro_slope);          // LeakyReLU / ReLU for slope in [0,1]
            *reinterpret_cast<V*>(halo + hlds[i]) = S.hok[i] ? o : zero;
        }
    };

    // this lane's two output pixels inside a tile, per-channel epilogue constants
    int hbase[2], prow[2], pcol[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int p = 32 * wave + 16 * ms + fr;
        prow[ms] = p >> WLOG;
        pcol[ms] = p & (W - 1);
        hbase[ms] = ((prow[ms] + 1 + prow[ms] / HH) * WP + pcol[ms] + 1) * LDW + 8 * fq;
    }
    f32x4 bias[NT], esc[NT], esh[NT], emu[NT], ers[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + 16 * i + 4 * fq;
        bias[i] = (MODE == 0 && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (hasEX) {
            esc[i] = *reinterpret_cast<const f32x4*>(a.ex_scale + n);
            esh[i] = *reinterpret_cast<const f32x4*>(a.ex_shift + n);
            emu[i] = *reinterpret_cast<const f32x4*>(a.ex_mean + n);
            ers[i] = *reinterpret_cast<const f32x4*>(a.ex_rstd + n);
        }
    }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;

    load_halo(HA, t_begin);
    if (t_begin + tstep < t_end) load_halo(HB, t_begin + tstep);
    if (SV_C3P_DEPTH == 3 && t_begin + 2 * tstep < t_end) load_halo(HC, t_begin + 2 * tstep);
    store_halo(HA, t_begin);
    __syncthreads();
    // 32 input channels: the block's 18 weight fragments stay in registers (72 of them: with the fusion flags at compile time
    // the variants hold 152-176 without) -- the nine taps read only the pixel fragments from LDS
    constexpr bool WREG = SV_C3P_WREG && CCH == 1 && sizeof(T) == 2 && MODE != 0;
    V wr[WREG ? 9 : 1][NT];
    if (WREG) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < NT; ++i) wr[t][i] = *reinterpret_cast<const V*>(wl + ((16 * i + fr) * 9 + t) * LDW + 8 * fq);
    }
    // one tile of the pipeline; NEXT holds tile+1 (already requested), FREE receives the request for tile+2
    // the epilogue operand (residual / raw tensor) of a tile: requested ONE TILE AHEAD, like the halo two tiles ahead -- loads
    // return in order, so an operand requested inside its own tile made the epilogue wait for everything that tile had
    // requested, the halo of tile + 2 included: one exposed memory round trip per tile (what bounded the kernel at one block per
    // CU, the paired budget of the backward)
    struct EStage { Q v[NT][2]; };
    auto load_eop = [&](EStage& E, int tile) {
        const int gr0 = tile * TR;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int64_t ob = ((int64_t)(gr0 + prow[ms]) * W + pcol[ms]) * g.ldo;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int n = n0 + 16 * i + 4 * fq;
                if (hasR) E.v[i][ms] = *reinterpret_cast<const Q*>(R + ob + n);
                else if (hasEX) E.v[i][ms] = *reinterpret_cast<const Q*>(EX + ob + n);
            }
        }
    };
    EStage EA, EB, EC;
    constexpr int HD = SV_C3P_DEPTH, ED = SV_C3P_DEPTH - 1;      // request distances in tiles: halo, epilogue operand
    if (SV_C3P_EOP_AHEAD && (hasR || hasEX)) {
        load_eop(EA, t_begin);
        if (ED == 2 && t_begin + tstep < t_end) load_eop(EB, t_begin + tstep);
    }
    // FREE receives the halo of tile + HD, ENEXT the epilogue operand of tile + ED
    auto do_tile = [&](int tile, HStage& NEXT, HStage& FREE, EStage& ECUR, EStage& ENEXT) {
        const int gr0 = tile * TR;
        // ---- request the halo HD tiles ahead + the epilogue operands ED tiles ahead; they fly during the MFMAs ----
        const bool more = tile + tstep < t_end;
        if (tile + HD * tstep < t_end) load_halo(FREE, tile + HD * tstep);
        if (SV_C3P_EOP_AHEAD) { if (tile + ED * tstep < t_end && (hasR || hasEX)) load_eop(ENEXT, tile + ED * tstep); }
        else if (hasR || hasEX) load_eop(ECUR, tile);
        Q (&eop)[NT][2] = ECUR.v;
        int64_t obase[2];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) obase[ms] = ((int64_t)(gr0 + prow[ms]) * W + pcol[ms]) * g.ldo;
        // ---- nine taps x CCH channel chunks out of LDS (padding is data: no masks) ---------------------------
        f32x4 acc[NT][2];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int sh = (tap_off(pdy, t) * WP + tap_off(pdx, t)) * LDW;
#pragma unroll
            for (int ck = 0; ck < CCH; ++ck) {
                const V af0 = *reinterpret_cast<const V*>(halo + hbase[0] + sh + 32 * ck);
                const V af1 = *reinterpret_cast<const V*>(halo + hbase[1] + sh + 32 * ck);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const V wf = WREG ? wr[WREG ? t : 0][i] : *reinterpret_cast<const V*>(wl + ((16 * i + fr) * 9 + t) * LDW + 32 * ck + 8 * fq);
                    mma32(acc[i][0], wf, af0);
                    mma32(acc[i][1], wf, af1);
                }
            }
        }
        __syncthreads();                               // all waves are done reading this tile's halo
        if (more) store_halo(NEXT, tile + tstep);      // next tile's halo -> LDS (requested a whole tile ago)
        // ---- epilogue of this tile (operands already in registers) -----------------------------------------
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n0 + 16 * i + 4 * fq;
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                f32x4 vv = acc[i][ms];
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] += bias[i][r];
                }
                if (hasR) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] += to_f(eop[i][ms][r]);
                }
                if (hasEX) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xf = to_f(eop[i][ms][r]);
                        const float gv = vv[r] * act_grad(xf * esc[i][r] + esh[i][r], a.ex_slope);
                        vv[r] = gv;
                        s1[i][r] += gv;
                        s2[i][r] += gv * ((xf - emu[i][r]) * ers[i][r]);
                    }
                } else if (has_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[i][r] += vv[r];
                        s2[i][r] += vv[r] * vv[r];
                    }
                }
                Q o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (T)vv[r];
                *reinterpret_cast<Q*>(O + obase[ms] + n) = o;
            }
        }
        __syncthreads();                               // next halo visible
    };
    if (SV_C3P_DEPTH == 3) {
        for (int tile = t_begin; tile < t_end; tile += 3 * tstep) {
            do_tile(tile, HB, HA, EA, EC);
            if (tile + tstep < t_end) do_tile(tile + tstep, HC, HB, EB, EA);
            if (tile + 2 * tstep < t_end) do_tile(tile + 2 * tstep, HA, HC, EC, EB);
        }
    } else {
        for (int tile = t_begin; tile < t_end; tile += 2 * tstep) {
            do_tile(tile, HB, HA, EA, EB);
            if (tile + tstep < t_end) do_tile(tile + tstep, HA, HB, EB, EA);
        }
    }
    // ---- flush the per-channel sums once per block ---------------------------------------------------------
    if (want_sums) {
        bool allv[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) allv[i] = true;
        flush_channel_sums<NT>(s1, s2, allv, ssum, hasEX ? a.bsums : a.stats, n0, g.N, a.replicas, a.flags);
    }
}


// ------------------------------------------------------------------------------------------------------
// Wide layers (Cin >= 96: WRN-28-10 body, MFMA-bound): a block owns PT consecutive 128-pixel tiles x 16*NT
// output channels with ALL accumulators in registers, and walks the input channels in 32-wide chunks.  The
// [BN][9][32] weight chunk -- by far the largest staging item -- is staged ONCE per chunk and reused by the PT
// pixel tiles (the 128-pixel kernel restaged it for every tile); the halo tiles alternate between two LDS
// buffers and are register-prefetched one step ahead, the next weight chunk a whole chunk ahead.
template <typename T, int NT, int WLOG, int PT>
__global__ __launch_bounds__(256) void conv3x3m_kernel(const sv_geom g, const sv_igemm_args_g A) {
    const sv_igemm_args& a = A.g[blockIdx.y];
    sv_start_signal(a);
    typedef typename V8<T>::type V;
    typedef typename V4<T>::type Q;
    static_assert(PT == 2 || PT == 4, "");
    constexpr int BN = 16 * NT;
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int LDW = CK + 16, VPP = CK / 8;
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;
    constexpr int HV = LROWS * WP * VPP, HI = (HV + 255) / 256, HPIX = (HI * 256 + VPP - 1) / VPP;
    constexpr int WV = BN * 9 * VPP, WI = (WV + 255) / 256, WROWS = (WI * 256 + VPP - 1) / VPP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = reinterpret_cast<T*>(smem);                    // [2][HPIX][LDW]
    T* wl = halo + 2 * HPIX * LDW;                           // [WROWS][LDW]
    double* ssum = reinterpret_cast<double*>(wl + WROWS * LDW);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR, nG = nT / PT;      // tile groups
    const int nNt = g.N / BN;
    const int L = blockIdx.x;
    int in_i, grp;
    if (nG >= 64) {                                   // XCD-affine: channel tiles of one pixel group share an L2
        const int xcd = L & 7, slot = L >> 3;
        in_i = slot % nNt;
        grp = (slot / nNt) * 8 + xcd;
        if (grp >= nG) return;
    } else {
        in_i = L % nNt;
        grp = L / nNt;
    }
    const int n0 = in_i * BN;
    const int t0 = grp * PT;
    const sv_phase& P = g.phase[0];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(a.w) + P.w_off + (int64_t)n0 * 9 * g.Cin;
    T* __restrict__ O = reinterpret_cast<T*>(a.out);
    const T* __restrict__ R = reinterpret_cast<const T*>(a.residual);
    const T* __restrict__ EX = reinterpret_cast<const T*>(a.ex);
    const bool has_pro = a.pro_scale != nullptr;
    const bool want_sums = (a.stats != nullptr) || (EX != nullptr);

    if (tid < 2 * BN) ssum[tid] = 0.0;
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;

    // ---- staging slots (tile / chunk independent parts) ------------------------------------------------
    const int v8 = 8 * (tid & 3);
    int hrel[HI], hxc[HI], hlds[HI], hkind[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int pix = idx / VPP;
        hlds[i] = pix * LDW + v8;
        const int lr = pix / WP, xx = pix - lr * WP;
        const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
        int kind = 1, rel = lr - 1 - seg;
        if (off == 0) {
            if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
            else kind = 0;
        }
        if (idx >= HV || xx == 0 || xx == WP - 1) kind = 0;
        hkind[i] = kind;
        hrel[i] = rel;
        hxc[i] = min(max(xx - 1, 0), W - 1);
    }
    V hv[HI], wv[WI];
    bool hok[HI];
    f32x4 ps0, ps1, pt0, pt1;                                 // BN scale / shift of the halo in flight
    auto load_halo = [&](int tile, int c0) {
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            hok[i] = hkind[i] == 1 || (hkind[i] == 2 && top_ok) || (hkind[i] == 3 && bot_ok);
            const int grc = min(max(gr0 + hrel[i], 0), BH - 1);
            hv[i] = *reinterpret_cast<const V*>(X + ((int64_t)grc * W + hxc[i]) * g.ldx + c0 + v8);
        }
        if (has_pro) {
            ps0 = *reinterpret_cast<const f32x4*>(a.pro_scale + c0 + v8);
            ps1 = *reinterpret_cast<const f32x4*>(a.pro_scale + c0 + v8 + 4);
            pt0 = *reinterpret_cast<const f32x4*>(a.pro_shift + c0 + v8);
            pt1 = *reinterpret_cast<const f32x4*>(a.pro_shift + c0 + v8 + 4);
        }
    };
    auto store_halo = [&](int buf) {
        T* hb = halo + buf * HPIX * LDW;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            V o = hv[i];
            if (has_pro) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u0 = to_f(hv[i][j]) * ps0[j] + pt0[j], u1 = to_f(hv[i][j + 4]) * ps1[j] + pt1[j];
                    o[j] = (T)fmaxf(u0, u0 * a.pro_slope);
                    o[j + 4] = (T)fmaxf(u1, u1 * a.pro_slope);
                }
            }
            *reinterpret_cast<V*>(hb + hlds[i]) = hok[i] ? o : zero;
        }
    };
    auto load_w = [&](int c0) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = min((tid + 256 * i) >> 2, BN * 9 - 1);
            wv[i] = *reinterpret_cast<const V*>(Wp + (int64_t)row * g.Cin + c0 + v8);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WI; ++i) *reinterpret_cast<V*>(wl + ((tid + 256 * i) >> 2) * LDW + v8) = wv[i];
    };

    int hbase[2], prow[2], pcol[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int p = 32 * wave + 16 * ms + fr;
        prow[ms] = p >> WLOG;
        pcol[ms] = p & (W - 1);
        hbase[ms] = ((prow[ms] + 1 + prow[ms] / HH) * WP + pcol[ms] + 1) * LDW + 8 * fq;
    }
    f32x4 acc[PT][NT][2];
#pragma unroll
    for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nck = g.Cin / CK;
    load_w(0);
    load_halo(t0, 0);
    store_w();
    store_halo(0);
    __syncthreads();
    for (int ck = 0; ck < nck; ++ck) {
        const bool more_ck = ck + 1 < nck;
        if (more_ck) load_w((ck + 1) * CK);                 // lands while this chunk's PT tiles are on the MFMAs
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const bool last = q == PT - 1;
            const bool more = !last || more_ck;
            if (more) load_halo(last ? t0 : t0 + q + 1, last ? (ck + 1) * CK : ck * CK);
            const T* hb = halo + (q & 1) * HPIX * LDW;       // PT is even: step parity == q parity
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sh = (tap_off(pdy, t) * WP + tap_off(pdx, t)) * LDW;
                const V af0 = *reinterpret_cast<const V*>(hb + hbase[0] + sh);
                const V af1 = *reinterpret_cast<const V*>(hb + hbase[1] + sh);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const V wf = *reinterpret_cast<const V*>(wl + ((16 * i + fr) * 9 + t) * LDW + 8 * fq);
                    mma32(acc[q][i][0], wf, af0);
                    mma32(acc[q][i][1], wf, af1);
                }
            }
            if (last && more_ck) {
                __syncthreads();                              // every wave is done with this chunk's weights
                store_w();
            }
            if (more) store_halo((q + 1) & 1);                // that buffer was last read one step ago (barrier since)
            __syncthreads();
        }
    }

    // ---- epilogue: PT tiles, BatchNorm sums kept in registers and flushed once ----------------------------------
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int gr0 = (t0 + q) * TR;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n0 + 16 * i + 4 * fq;
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
            f32x4 esc, esh, emu, ers;
            if (EX) {
                esc = *reinterpret_cast<const f32x4*>(a.ex_scale + n);
                esh = *reinterpret_cast<const f32x4*>(a.ex_shift + n);
                emu = *reinterpret_cast<const f32x4*>(a.ex_mean + n);
                ers = *reinterpret_cast<const f32x4*>(a.ex_rstd + n);
            }
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const int64_t ob = ((int64_t)(gr0 + prow[ms]) * W + pcol[ms]) * g.ldo + n;
                f32x4 vv = acc[q][i][ms];
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] += bias[r];
                if (R) {
                    const Q rr = *reinterpret_cast<const Q*>(R + ob);
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] += to_f(rr[r]);
                }
                if (EX) {
                    const Q xe = *reinterpret_cast<const Q*>(EX + ob);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xf = to_f(xe[r]);
                        const float gv = vv[r] * act_grad(xf * esc[r] + esh[r], a.ex_slope);
                        vv[r] = gv;
                        s1[i][r] += gv;
                        s2[i][r] += gv * ((xf - emu[r]) * ers[r]);
                    }
                } else if (a.stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[i][r] += vv[r];
                        s2[i][r] += vv[r] * vv[r];
                    }
                }
                Q o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (T)vv[r];
                *reinterpret_cast<Q*>(O + ob) = o;
            }
        }
    }
    if (want_sums) {
        bool allv[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) allv[i] = true;
        flush_channel_sums<NT>(s1, s2, allv, ssum, EX ? a.bsums : a.stats, n0, g.N, a.replicas, a.flags);
    }
}

template <typename T, int NT, int WLOG, int PT>
int launch_m(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W, BN = 16 * NT, LDW = CK + 16, VPP = CK / 8;
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;
    constexpr int HV = LROWS * (W + 2) * VPP, HI = (HV + 255) / 256, HPIX = (HI * 256 + VPP - 1) / VPP;
    constexpr int WV = BN * 9 * VPP, WI = (WV + 255) / 256, WROWS = (WI * 256 + VPP - 1) / VPP;
    const int nG = g->B * g->Hin / TR / PT;
    const int nNt = g->N / BN;
    const int grid = (nG >= 64 ? ((nG + 7) / 8) * 8 : nG) * nNt;
    const size_t lds = (size_t)(2 * HPIX + WROWS) * LDW * sizeof(T) + 2 * BN * sizeof(double);
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3m_kernel<T, NT, WLOG, PT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3m)");
        optin = true;
    }
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3m_kernel<T, NT, WLOG, PT>), dim3(grid, sv_ngroups(a->groups)), dim3(256), lds, s, *g, sv_expand_groups(*g, *a, (int)sizeof(T)));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(conv3x3m)");
}

template <typename T, int NT, int PT>
int launch_mw(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    switch (g->Win) {
        case 32: return launch_m<T, NT, 5, PT>(g, a, s);
        case 16: return launch_m<T, NT, 4, PT>(g, a, s);
        default: return launch_m<T, NT, 3, PT>(g, a, s);
    }
}

template <typename T, int WLOG, int CCH, int MODE>
int launch_pm(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W, CIN = 32 * CCH, LDW = CIN + 16, VPP = CIN / 8;
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;
    constexpr int HV = LROWS * (W + 2) * VPP, HI = (HV + 255) / 256, HPIX = (HI * 256 + VPP - 1) / VPP;
    const int nT = g->B * g->Hin / TR;
    const int nNt = g->N / 32;
    // persistent blocks (two per CU): fewer / more measured slower; a batched launch shares them among its groups
    const int budget = sv_persistent_blocks();
    const int target = budget / sv_ngroups(a->groups) > 64 ? budget / sv_ngroups(a->groups) : 64;
    int chunks = (target + nNt - 1) / nNt;
    if (chunks > nT) chunks = nT;
    const int tiles_per = (nT + chunks - 1) / chunks;
    chunks = (nT + tiles_per - 1) / tiles_per;
    const size_t lds = (size_t)(HPIX + 32 * 9) * LDW * sizeof(T) + 2 * 32 * sizeof(double);
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3p_kernel<T, WLOG, CCH, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3p)");
        optin = true;
    }
    sv_igemm_args b = *a;          // this kernel folds the BatchNorm finalisation of its prologue (<= 64 channels, <= 64 replicas)
    if (!sv_fold_claim(b.fold_stats && b.fold_replicas <= 64 && (size_t)HPIX * LDW * sizeof(T) >= (size_t)(1024 + 2 * CIN) * 4)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(chunks * nNt, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3p_kernel<T, WLOG, CCH, MODE>), dim3(chunks * nNt, sv_ngroups(a->groups)), dim3(256), lds, s, *g, sv_expand_groups(*g, *a, (int)sizeof(T)), tiles_per);
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(conv3x3p)");
}

// the three launch kinds of the training step take the binaries with their fusion flags at compile time (bf16 only)
template <typename T, int WLOG, int CCH>
int launch_p(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
#if SV_C3P_MODES
    if constexpr (sizeof(T) == 2) {
        if (!a->bias) {
            if (a->pro_scale && a->stats && !a->ex) return a->residual ? launch_pm<T, WLOG, CCH, 2>(g, a, s) : launch_pm<T, WLOG, CCH, 1>(g, a, s);
            if (!a->pro_scale && a->ex && !a->residual && !a->stats) return launch_pm<T, WLOG, CCH, 3>(g, a, s);
        }
    }
#endif
    return launch_pm<T, WLOG, CCH, 0>(g, a, s);
}

template <typename T, int CCH>
int launch_pw(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    switch (g->Win) {
        case 32: return launch_p<T, 5, CCH>(g, a, s);
        case 16: return launch_p<T, 4, CCH>(g, a, s);
        default: return launch_p<T, 3, CCH>(g, a, s);
    }
}

template <typename T, int NT, int WLOG>
int launch(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W, BN = 16 * NT;
    const int nT = g->B * g->Hin / TR;
    const int nNt = g->N / BN;
    const int grid = (nT >= 64 ? ((nT + 7) / 8) * 8 : nT) * nNt;
    const size_t lds = (size_t)((TR + 2) * (W + 2) + BN * 9) * LDC * sizeof(T) + 2 * BN * sizeof(double);
    static bool optin = false;          // > 64 KiB of dynamic LDS needs an opt-in (gfx950 has 160 KiB per CU)
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, NT, WLOG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3)");
        optin = true;
    }
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3_kernel<T, NT, WLOG>), dim3(grid, sv_ngroups(a->groups)), dim3(256), lds, s, *g, sv_expand_groups(*g, *a, (int)sizeof(T)));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(conv3x3)");
}

template <typename T, int NT>
int launch_w(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    switch (g->Win) {
        case 32: return launch<T, NT, 5>(g, a, s);
        case 16: return launch<T, NT, 4>(g, a, s);
        default: return launch<T, NT, 3>(g, a, s);
    }
}

}  // namespace

static bool conv3x3_covers(const sv_geom* g) {
    if (g->nphase != 1 || g->phase[0].ntap != 9 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1) return false;
    if (g->Hq != g->Hin || g->Wq != g->Win || g->Hout != g->Hin || g->Wout != g->Win || g->Hin != g->Win) return false;
    if (g->Win != 8 && g->Win != 16 && g->Win != 32) return false;
    if (g->Cin % CK != 0 || g->ldx != g->Cin || g->N % 32 != 0) return false;
    if (g->phase[0].ooy != 0 || g->phase[0].oox != 0) return false;
    for (int t = 0; t < 9; ++t)
        if (g->phase[0].dy[t] < -1 || g->phase[0].dy[t] > 1 || g->phase[0].dx[t] < -1 || g->phase[0].dx[t] > 1) return false;
    return (g->B * g->Hin) % (128 / g->Win) == 0;
}
// Returns 1 and sets *rc when the geometry is a stride-1 3x3 convolution this kernel covers.
int sv_conv3x3_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (!conv3x3_covers(g)) return 0;
    const int TR = 128 / g->Win;
    const bool no_persist = sv_disabled(SV_K_CONV3X3P);
    if (!no_persist && dtype == SV_BF16 && (g->Cin == 32 || g->Cin == 64)) {
        // whole weight slab resident in LDS: persistent software-pipelined kernel
        *rc = g->Cin == 32 ? launch_pw<bf16, 1>(g, a, s) : launch_pw<bf16, 2>(g, a, s);
        return 1;
    }
    if (!no_persist && dtype == SV_F32 && g->Cin == 32) {
        *rc = launch_pw<float, 1>(g, a, s);
        return 1;
    }
    if (sv_conv3x3w_try(g, dtype, a, s, rc)) return 1;       // wide MFMA-bound layers: conv3x3w.hip
    const bool no_multi = sv_disabled(SV_K_CONV3X3M);
    if (!no_multi && dtype == SV_BF16 && g->Cin >= 96) {
        // MFMA-bound wide layers: multi-tile kernel when the grid still fills the chip
        // (measured at B=512: 160 ch 707 vs 796 us; the 64-channel-tile variant <4,4> lost to the 128-pixel kernel
        //  on 320 / 640 channels -- 447 vs 418 us, 469 vs 356 us -- and is not dispatched)
        const int nT = g->B * g->Hin / TR;
        if (g->N % 80 == 0 && nT % 2 == 0 && (int64_t)(nT / 2) * (g->N / 80) * sv_ngroups(a->groups) >= 256) {
            *rc = launch_mw<bf16, 5, 2>(g, a, s);
            return 1;
        }
    }
    if (dtype == SV_BF16) {
        // 64-channel tiles keep two blocks per CU resident (LDS); wider layers take several tiles
        if (g->N % 64 == 0) *rc = launch_w<bf16, 4>(g, a, s);
        else if (g->N % 80 == 0) *rc = launch_w<bf16, 5>(g, a, s);
        else *rc = launch_w<bf16, 2>(g, a, s);
        return 1;
    }
    *rc = launch_w<float, 2>(g, a, s);     // fp32 parity mode: 32-channel tiles (LDS budget)
    return 1;
}
