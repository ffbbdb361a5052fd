// Tap-fused weight gradient of the "odd" conv-like layers (bf16, gfx950): the ConvTranspose2d(4, 2, 1) decoder stack
// (decoder.py:22-58), the stride-2 3x3 convolutions of the WideResNet (wideresnet.py:29-30) and its 16-channel stem / first
// block (wideresnet.py:13-14) -- everything that is neither a stride-1 3x3 layer with >= 32 channels (wgrad3x3.hip) nor a
// plain GEMM.
//
//   dW[n][torig(ph, t)][c] += sum_{b, q}  dy[b, q * os + oo(ph)][n] * A[b, q * s + d(ph, t)][c],   A = act(x * scale + shift)
//
// The generic kernel (wgrad.hip) gives every TAP its own blocks, so both operands stream once per tap -- 9 to 16 times,
// through L2: measured L2-bound (the 16 -> 32 layer at 4 x 512 images: 201 MB of operands, 1.8 GB of L2 traffic, 189 us).
// Here, as in wgrad3x3.hip, a persistent block owns an NS (n) x CS (c) slab of dW for ALL taps of ALL phases and walks a
// range of 128-position tiles of the per-phase output grid; per tile it stages ONCE
//   * the dy pixels of every phase (the os*TR x os*Wq output region of the tile), and
//   * the BatchNorm-transformed input region the taps touch (zero padding as data),
// with two register stages (operands requested two tiles ahead), and runs phases x taps on MFMA out of LDS: the reduction
// index (pixels) is the MFMA k dimension, both operands are read k-major with the transposing read ds_read_b64_tr_b16 at
// per-lane pixel addresses (a phase / a tap / a stride is nothing but a different LDS pixel address).  With fewer than four
// 16 x 16 sub-blocks in the slab the waves split the pixel chunks among them and publish separate partial slabs.  Partial
// slabs go to the caller's workspace with plain stores and are summed by the slab reduction of wgrad3x3.hip.
#include <stdlib.h>

#include "common.h"

void sv_slab_reduce(const float* ws, int nslabs, int64_t n, float* dw, hipStream_t s);      // wgrad3x3.hip

namespace {


struct hw_params {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    float pro_slope;
    const void* dy;
    float* ws;
    int splits, tiles_per, groups;
    // geometry of a tile
    int wlog, hlog, hhlog, TR, SEG, SR, LW, dymin, dxmin, HP;
    int YW, YP;                               // dy tile: os*Wq columns, os*TR*os*Wq pixels
};

typedef __attribute__((address_space(3))) s16x4 lds_v4;

__device__ __forceinline__ bf16x8 frag2(const bf16* a0, const bf16* a1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a1));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

// NS / CS: slab extent in output / input channels (16 or 32); NPH: 4 = four phases with <= 4 taps each, 1 = one phase with
// <= 16 taps.  Accumulator slot of (phase, tap) = phase * 4 + tap resp. tap: static, so the 16 tiles stay in registers.
template <int NS, int CS, int NPH>
__global__ __launch_bounds__(256, 2) void hwgrad_kernel(const sv_geom g, const sv_wg_g<hw_params> PG) {
    const hw_params& p = PG.g[blockIdx.y];
    constexpr int LDY = NS + (NS == 32 ? 16 : 8), LDX = CS + (CS == 32 ? 16 : 8);     // LDS pixel strides (elements)
    constexpr int VY = NS / 8, VX = CS / 8;
    constexpr int NSUB = (NS / 16) * (CS / 16), KPARTS = 4 / NSUB;
    // 16-byte vectors per thread per register stage: the dy tile (512 pixels with four phases, 128 with one) and the input
    // region (<= 288 pixels for the ConvTranspose layers, <= 612 for a 4x4 stride-2 window)
    constexpr int DYMAX = NPH == 4 ? 2 * VY : (VY + 1) / 2, XMAX = NPH == 4 ? (CS == 32 ? 5 : 3) : (CS == 32 ? 10 : 5);
    typedef bf16x8 V;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Ys = reinterpret_cast<bf16*>(smem);            // [YP][LDY]
    bf16* Xs = Ys + p.YP * LDY;                          // [HP][LDX]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int sub = wave % NSUB, kpart = wave / NSUB;
    const int wi = sub / (CS / 16), wj = sub % (CS / 16);          // this wave's 16 x 16 sub-block (n, c)
    const int Wq = 1 << p.wlog, Hq = 1 << p.hlog, HH = 1 << p.hhlog;
    const int BHq = g.B * Hq, nT = (BHq + p.TR - 1) / p.TR;
    const int nCt = g.Cin / CS, nNt = g.N / NS, nNC = nCt * nNt;
    const int nc = blockIdx.x % nNC, split = blockIdx.x / nNC;
    const int n0 = (nc / nCt) * NS, c0 = (nc % nCt) * CS;
    const int t_begin = split * p.tiles_per, t_end = min(nT, t_begin + p.tiles_per);
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
    const bf16* __restrict__ DY = reinterpret_cast<const bf16*>(p.dy);
    const bool has_pro = p.pro_scale != nullptr;
    float pslope = p.pro_slope;               // pinned in a vector register (conv3x3p_kernel: no re-load from the argument segment)
    asm volatile("v_mov_b32 %0, %0" : "+v"(pslope));
    const int os = g.osy;

    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (has_pro) {                              // a thread's 8-channel group is the same for all of its slots (256 % VX == 0)
        const int cv8 = 8 * (tid % VX);
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + cv8);
        s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + cv8 + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + cv8);
        t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + cv8 + 4);
    }
    // ---- staging slots (tile-invariant parts) ------------------------------------------------------------------------------
    // dy: LDS pixel yp = (yr, yc) of the os*TR x YW region; source = tile base (output row os * R0) + (yr * Wout + yc) * ldo
    const int YVn = p.YP * VY, HVn = p.HP * VX;
    int yoff[DYMAX];            // (every tile is full -- the dispatcher requires it -- so a dy vector is always valid)
#pragma unroll
    for (int i = 0; i < DYMAX; ++i) {
        const int idx = min(tid + 256 * i, YVn - 1);
        const int yp = idx / VY, v = idx - yp * VY;
        const int yr = yp / p.YW, yc = yp - yr * p.YW;
        yoff[i] = (yr * g.Wout + yc) * g.ldo + n0 + 8 * v;
    }
    int hoff[XMAX], hrs[XMAX];  // element offset from the tile base; row within the segment | segment << 8 (or -1: never valid)
#pragma unroll
    for (int i = 0; i < XMAX; ++i) {
        const int idx = tid + 256 * i;
        hoff[i] = 0; hrs[i] = -1;
        if (idx < HVn) {
            const int pix = idx / VX, cv = idx - pix * VX;
            const int lr = pix / p.LW, lc = pix - lr * p.LW;
            const int seg = lr / p.SR, off = lr - seg * p.SR;
            if ((unsigned)(lc + p.dxmin) < (unsigned)g.Win) {
                hrs[i] = off | (seg << 8);
                hoff[i] = ((seg * g.Hin + off) * g.Win + lc) * g.ldx + c0 + 8 * cv;
            }
        }
    }
    struct Stage { V ry[DYMAX], rx[XMAX]; bool xok[XMAX]; };
    Stage SA, SB;
    auto load_tile = [&](Stage& S, int tile) {
        const int R0 = tile * p.TR;
        const int b0 = R0 >> p.hlog, qy0 = R0 & (Hq - 1);
        const int64_t ybase = (int64_t)(os * R0) * g.Wout * g.ldo;
        const int iy0 = g.sy * qy0 + p.dymin;
        const int64_t xbase = ((int64_t)(b0 * g.Hin + iy0) * g.Win + p.dxmin) * g.ldx;
        const int64_t xsafe = ((int64_t)(b0 * g.Hin + g.sy * qy0) * g.Win) * g.ldx + c0;
#pragma unroll
        for (int i = 0; i < DYMAX; ++i) {
            if (256 * i >= YVn) break;
            S.ry[i] = *reinterpret_cast<const V*>(DY + ybase + yoff[i]);
        }
#pragma unroll
        for (int i = 0; i < XMAX; ++i) {
            if (256 * i >= HVn) break;
            const bool ok = hrs[i] >= 0 && (unsigned)(iy0 + (hrs[i] & 255)) < (unsigned)g.Hin;
            S.xok[i] = ok;
            S.rx[i] = *reinterpret_cast<const V*>(X + (ok ? xbase + hoff[i] : xsafe));
        }
    };
    auto store_tile = [&](Stage& S) {
#pragma unroll
        for (int i = 0; i < DYMAX; ++i) {
            const int idx = tid + 256 * i;
            if (idx < YVn) *reinterpret_cast<V*>(Ys + (idx / VY) * LDY + 8 * (idx % VY)) = S.ry[i];
        }
#pragma unroll
        for (int i = 0; i < XMAX; ++i) {
            const int idx = tid + 256 * i;
            if (idx >= HVn) break;
            V o = S.rx[i];
            if (has_pro) o = bn_act8(S.rx[i], s0, s1, t0, t1, pslope);
            *reinterpret_cast<V*>(Xs + (idx / VX) * LDX + 8 * (idx % VX)) = S.xok[i] ? o : zero;
        }
    };

    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma_tile = [&]() {
        for (int kc = kpart; kc < 4; kc += KPARTS) {
            // this lane's pixels of the 32-pixel chunk: pq = 32 kc + 8 fq + (fr >> 2) and pq + 4 (transposing read: four pixel
            // rows x 16 channels per 16-lane group)
            int yb[2], hb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pq = 32 * kc + 8 * fq + (fr >> 2) + 4 * h;
                const int prow = pq >> p.wlog, pcol = pq & (Wq - 1);
                yb[h] = ((os * prow) * p.YW + os * pcol) * LDY + 16 * wi + 4 * (lane & 3);
                const int seg = prow >> p.hhlog, rin = prow & (HH - 1);
                hb[h] = ((seg * p.SR + g.sy * rin) * p.LW + g.sx * pcol) * LDX + 16 * wj + 4 * (lane & 3);
            }
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                const sv_phase& P = g.phase[ph];        // (the dispatcher guarantees nphase == NPH)
                const int yo = (P.ooy * p.YW + P.oox) * LDY;
                const bf16x8 fy = frag2(Ys + yb[0] + yo, Ys + yb[1] + yo);
                constexpr int TMAX = NPH == 4 ? 4 : 16;
#pragma unroll
                for (int t = 0; t < TMAX; ++t) {
                    if (t < P.ntap) {            // (no early exit: the loop must unroll completely -- static accumulator slots)
                        const int sh = ((P.dy[t] - p.dymin) * p.LW + (P.dx[t] - p.dxmin)) * LDX;
                        const bf16x8 fx = frag2(Xs + hb[0] + sh, Xs + hb[1] + sh);
                        const int slot = NPH == 4 ? 4 * ph + t : t;
                        acc[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy, fx, acc[slot], 0, 0, 0);
                    }
                }
            }
        }
    };

    if (t_begin < t_end) {
        load_tile(SA, t_begin);
        if (t_begin + 1 < t_end) load_tile(SB, t_begin + 1);
        auto do_tile = [&](int tile, Stage& CUR) {
            store_tile(CUR);
            __syncthreads();
            if (tile + 2 < t_end) load_tile(CUR, tile + 2);       // in flight during this and the next tile's MFMAs
            mma_tile();
            __syncthreads();                                      // everyone is done reading before the next tile overwrites LDS
        };
        for (int tile = t_begin; tile < t_end; tile += 2) {
            do_tile(tile, SA);
            if (tile + 1 < t_end) do_tile(tile + 1, SB);
        }
    }
    // ---- publish this wave's partial slab: lane holds c = c0 + 16 wj + fr, n = n0 + 16 wi + 4 fq + r ---------------------------
    const int64_t slab = (int64_t)g.N * g.T_orig * g.Cin;
    float* dst = p.ws + (((int64_t)blockIdx.y * p.splits + split) * KPARTS + kpart) * slab;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const sv_phase& P = g.phase[ph];
        constexpr int TMAX = NPH == 4 ? 4 : 16;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < P.ntap) {
                const int to = P.torig[t];
                const int slot = NPH == 4 ? 4 * ph + t : t;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    dst[((int64_t)(n0 + 16 * wi + 4 * fq + r) * g.T_orig + to) * g.Cin + c0 + 16 * wj + fr] = acc[slot][r];
            }
        }
    }
}

int ilog2h(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

template <int NS, int CS, int NPH>
int launch_hw(const sv_geom* g, const hw_params& p, float* dw, size_t lds, hipStream_t s) {
    constexpr int KPARTS = 4 / ((NS / 16) * (CS / 16));
    const int nNC = (g->N / NS) * (g->Cin / CS);
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hwgrad_kernel<NS, CS, NPH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(hwgrad)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((hwgrad_kernel<NS, CS, NPH>), dim3(p.splits * nNC, p.groups), dim3(256), lds, s, *g,
                       sv_expand_wg(*g, p, p.groups, 2));
    sv_prof_end(s);
    sv_slab_reduce(p.ws, p.splits * p.groups * KPARTS, (int64_t)g->N * g->T_orig * g->Cin, dw, s);
    return sv_check_launch("sv_wgrad(hwgrad)");
}

}  // namespace

// Returns 1 and sets *rc when the geometry is covered (bf16, square power-of-two grids, <= 4 taps per phase with four phases
// or <= 16 taps with one, the tile's LDS image within budget) and the caller's workspace holds the partial slabs.
int sv_hwgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                  const void* dy, float* dw, float* ws, int64_t ws_elems, int groups, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_HWGRAD) || dtype != SV_BF16 || !ws) return 0;
    if (g->Hq != g->Wq || g->Hin != g->Win || g->sy != g->sx || g->osy != g->osx) return 0;
    if (g->sy < 1 || g->sy > 2 || g->osy < 1 || g->osy > 2) return 0;
    if (g->nphase != 1 && g->nphase != 4) return 0;
    if (g->ldx != g->Cin || g->ldo != g->N || g->Cin % 16 != 0 || g->N % 16 != 0) return 0;
    hw_params p;
    p.wlog = ilog2h(g->Wq);
    p.hlog = ilog2h(g->Hq);
    if (p.wlog < 2 || p.wlog > 5) return 0;                  // 4 .. 32 columns
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127, ttot = 0;
    for (int ph = 0; ph < g->nphase; ++ph) {
        if (g->phase[ph].ntap > (g->nphase == 4 ? 4 : 16)) return 0;
        for (int t = 0; t < g->phase[ph].ntap; ++t) {
            const int dyv = g->phase[ph].dy[t], dxv = g->phase[ph].dx[t];
            dymin = dyv < dymin ? dyv : dymin; dymax = dyv > dymax ? dyv : dymax;
            dxmin = dxv < dxmin ? dxv : dxmin; dxmax = dxv > dxmax ? dxv : dxmax;
            ++ttot;
        }
    }
    if (ttot < 4 || dymax - dymin > 3 || dxmax - dxmin > 3) return 0;      // (1 x 1 layers stream well through the generic kernel)
    if (ttot != g->T_orig) return 0;            // pruned taps would leave holes in the partial slabs
    const int NS = g->N % 32 == 0 ? 32 : 16, CS = g->Cin % 32 == 0 ? 32 : 16;
    const int Wq = g->Wq, Hq = g->Hq, os = g->osy;
    p.TR = 128 / Wq;
    const int HH = p.TR < Hq ? p.TR : Hq;
    p.hhlog = ilog2h(HH);
    p.SEG = p.TR / HH;
    p.SR = g->sy * (HH - 1) + (dymax - dymin) + 1;
    p.LW = g->sx * (Wq - 1) + (dxmax - dxmin) + 1;
    p.dymin = dymin;
    p.dxmin = dxmin;
    p.HP = p.SEG * p.SR * p.LW;
    p.YW = os * Wq;
    p.YP = os * p.TR * p.YW;
    {
        const int dymax = g->nphase == 4 ? 2 * (NS / 8) : (NS / 8 + 1) / 2;
        const int xmax = g->nphase == 4 ? (CS == 32 ? 5 : 3) : (CS == 32 ? 10 : 5);
        if (p.YP * (NS / 8) > 256 * dymax || p.HP * (CS / 8) > 256 * xmax) return 0;
    }
    if ((g->B * Hq) % p.TR != 0) return 0;
    const size_t lds = ((size_t)p.YP * (NS + (NS == 32 ? 16 : 8)) + (size_t)p.HP * (CS + (CS == 32 ? 16 : 8))) * 2;
    if (lds > 76 * 1024) return 0;
    if ((int64_t)g->B * g->Hin * g->Win * g->ldx >= ((int64_t)1 << 31) || (int64_t)g->B * g->Hout * g->Wout * g->ldo >= ((int64_t)1 << 31))
        return 0;
    const int nT = g->B * Hq / p.TR;
    const int nNC = (g->N / NS) * (g->Cin / CS), kparts = 4 / ((NS / 16) * (CS / 16));
    const int budget = sv_persistent_blocks();
    const int target = budget / groups > 32 ? budget / groups : 32;
    int splits = (target + nNC - 1) / nNC;
    if (splits > nT) splits = nT;
    if (splits < 1) splits = 1;
    p.tiles_per = (nT + splits - 1) / splits;
    splits = (nT + p.tiles_per - 1) / p.tiles_per;
    if (p.tiles_per < 4 && !sv_halo_all()) return 0;             // too little work per block to amortise the slab: generic kernel
    // Measured against the generic kernel at 4 x 512 images (profiles/r02_*): it wins on the thin stride-1 layers, which are
    // L2-bound there (16 -> 32: 189 -> 100 us, last ConvTranspose 155 -> 77 us, stem 132 -> 66 us), and is level or behind on
    // the layers with many slabs or a stride-2 input region (128 -> 64 ConvTranspose 131 -> 139 us, 32 -> 64 stride 2
    // 97 -> 106 us): those stay on the generic kernel (SV_OPT_HALO_ALL = 1 takes the whole range: tests)
    if (!sv_halo_all() && !(nNC <= 2 && g->sy == 1)) return 0;
    const int64_t need = (int64_t)splits * groups * kparts * g->N * g->T_orig * g->Cin;
    if (ws_elems < need) return 0;
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_slope = pro_slope; p.dy = dy; p.ws = ws;
    p.splits = splits; p.groups = groups;
    const bool multi = g->nphase == 4;
    if (NS == 32 && CS == 32) *rc = multi ? launch_hw<32, 32, 4>(g, p, dw, lds, s) : launch_hw<32, 32, 1>(g, p, dw, lds, s);
    else if (NS == 32 && CS == 16) *rc = multi ? launch_hw<32, 16, 4>(g, p, dw, lds, s) : launch_hw<32, 16, 1>(g, p, dw, lds, s);
    else if (NS == 16 && CS == 32) *rc = multi ? launch_hw<16, 32, 4>(g, p, dw, lds, s) : launch_hw<16, 32, 1>(g, p, dw, lds, s);
    else *rc = multi ? launch_hw<16, 16, 4>(g, p, dw, lds, s) : launch_hw<16, 16, 1>(g, p, dw, lds, s);
    return 1;
}
