// LDS-halo gather-GEMM for the "odd" conv-like layers of the SHOT-VAE network -- everything conv-like that is not a
// stride-1 3x3 convolution with >= 32 channels (those have conv3x3*.hip): the ConvTranspose2d(4, 2, 1) decoder stack and
// its data gradients (decoder.py:22-58), the stride-2 3x3 convolutions and 1x1 shortcuts of the WideResNet and their data
// gradients (wideresnet.py:29-30,41-43), the 16-channel stem / first block (wideresnet.py:13-14).  gfx950.
//
// The generic gather-GEMM (igemm.hip) fetches its A operand from global memory once per TAP: a ConvTranspose phase reads
// every input pixel 4 times, the four phases 16 times, a 4x4 stride-2 convolution 16 times -- through L2, with a
// global -> LDS round trip and a barrier per 32..64-deep k step.  Those layers ran at 0.15-1.8 TB/s of algorithmic
// traffic (3-7x their HBM time) and made up a third of the config-2 step.  Here a block owns 128 positions of the
// per-phase output grid (TR = 128 / Wq whole grid rows, possibly several whole images) x 16*NT output channels and
//   * stages the input region those positions touch -- (sy (HH - 1) + dy range + 1) rows x (sx (Wq - 1) + dx range + 1)
//     columns per image segment, zero padding as data -- ONCE per 16/32-channel chunk, the BatchNorm + LeakyReLU/ReLU
//     prologue applied once per element on the way in;
//   * stages the weight chunk of EVERY tap of EVERY phase ([BN][taps][CC]);
//   * runs all phases x taps on MFMA out of LDS (a tap is an LDS address offset), one accumulator set per phase;
//   * ends with the fused epilogue of sv_igemm (bias, residual, next-BatchNorm statistics or activation-backward +
//     BatchNorm-backward sums), the per-channel sums kept in registers over the phases and flushed once.
// With 16 input channels (the stem, the first block, the last decoder gradient) one 32-deep MFMA k step carries TWO taps.
// Two kernels: halop_kernel (persistent blocks, ALL input channels and the weights of every tap LDS-resident, two-stage
// register prefetch of the next tile -- the form that pays: thin layers with Cin <= 64 and at most two channel tiles) and
// halo_kernel (one tile per block, channel-chunked; dispatched only where measured faster than igemm_kernel: the
// ConvTranspose 128 -> 64 forward and thin-OUTPUT 3x3 layers such as the 160 -> 16 data gradient).  sv_halo_try holds the
// rules; SV_OPT_HALO_ALL lets both take everything they can run (tests, A/B runs).
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a drop-in fast path inside it (SV_K_HALO / SV_K_HALOP
// disable).
#include <stdlib.h>

#include "common.h"
#include "epilogue.h"

namespace {

// Blocks per CU of the small-stage persistent variants (measured at 4 x 512 images, tools/thin_ab.sh: three blocks of the
// 2-vector / 16-channel-tile variant: stem 66 -> 55 us, 1x1 16 -> 32 data gradient 54 -> 46 us; four blocks or any of the
// 32-channel-tile variants spill -- 16 -> 32 forward 115 -> 176 / 290 us -- and stay at two)
#ifndef SV_HALOP_EO_AHEAD
#define SV_HALOP_EO_AHEAD 0    // 1: the epilogue operand of a tile is requested one tile ahead (measured mixed here: -4 % .. +6 % per layer, step flat;
                                // conv3x3p, one block per CU in the paired backward, gains 17 % from the same change)
#endif
#ifndef SV_HALOP_MODES
#define SV_HALOP_MODES 1
#endif
#ifndef SV_HALOP_OCC2
#define SV_HALOP_OCC2 3         // 2-vector stage, 16-channel tiles (NT = 1)
#endif
#ifndef SV_HALOP_OCC2B
#define SV_HALOP_OCC2B 2        // ... 32-channel tiles (NT = 2)
#endif
#ifndef SV_HALOP_OCC4
#define SV_HALOP_OCC4 2         // 4-vector stage, NT = 1
#endif
#ifndef SV_HALOP_OCC4B
#define SV_HALOP_OCC4B 2        // ... NT = 2
#endif
#ifndef SV_HALOP_OCC2_M
#define SV_HALOP_OCC2_M 4       // the same with compile-time fusion flags (MODE 1 / 3)
#endif
#ifndef SV_HALOP_OCC2B_M
#define SV_HALOP_OCC2B_M 3
#endif
#ifndef SV_HALOP_OCC4_M
#define SV_HALOP_OCC4_M 3
#endif
#ifndef SV_HALOP_OCC4B_M
#define SV_HALOP_OCC4B_M 2
#endif
#ifndef SV_HALOP_OCC6_M
#define SV_HALOP_OCC6_M 3       // 6-vector stage, NT = 1
#endif
constexpr int HMAXV = 12;       // halo 16-byte vectors per thread (3072 per block)
constexpr int HMAXW = 8;        // weight vectors per thread

struct halo_cfg {
    int wlog, hlog, hhlog;      // log2 of Wq, Hq, HH (HH = grid rows per image segment of a tile = min(TR, Hq))
    int TR, SEG, SR, LW;        // grid rows per tile, image segments per tile, LDS rows per segment, LDS columns
    int dymin, dxmin;
    int HP;                     // LDS halo pixels
    int tslots;                 // tap slots per weight row (phases padded to whole k steps)
    int tap0[SV_MAX_PHASES];    // first tap slot of each phase
    int nks[SV_MAX_PHASES];     // k steps of each phase
    // LDS pixel shift of every tap slot: shift[16 * phase + slot] = (dy - dymin) * LW + (dx - dxmin); a pad slot repeats
    // tap 0 of its phase (its weights are zero).  Lane l of every wave keeps entry l in a register and the k loop reads
    // it with v_readlane: no memory access in the loop.  (Indexing sv_phase::dy[] / dx[] in the loop compiled to FOUR
    // global_load_sbyte from the kernel-argument segment per k step, each followed by s_waitcnt vmcnt(0) -- which also
    // drained the prefetches of the next tiles: the thin layers ran at 1.4 - 2 TB/s with their waves waiting 55 - 63 %
    // of the time, profiles/r03_pmc_odd.txt.)
    int shift[SV_MAX_PHASES * 16];
    // weight source of every tap slot (halo_kernel's staging): wbase[slot] = w_off + tap * Cin of the slot's phase (elements; -1:
    // pad slot), wnst[slot] = ntap * Cin = elements between two output channels in that phase.  Lane-held like `shift`.
    int wbase[SV_MAX_PHASES * 16], wnst[SV_MAX_PHASES * 16];
};

// The k loop of both kernels: all (tap slot, channel chunk) steps of one phase out of LDS, software-pipelined by hand -- the
// fragments of step i + 1 are requested before the MFMAs of step i (two register sets, ping-pong), the tap shift comes from
// the lane-held table.  hb0 / hb1: the lane's two activation fragment bases (elements), wrow0: the lane's weight fragment
// base of the phase (element pointer incl. the 16-row stride term fr * LDW is added here).
template <typename T, int NT, int CC>
__device__ __forceinline__ void halo_phase_mma(f32x4 (&acc)[NT][2], const T* halo, int hb0, int hb1, const T* wph, int fr, int fq,
                                               int LDC, int LDW, int kstride /* elements between k steps of the weights */,
                                               int nks, int nck, int ncklog, int mytab, int tab0) {
    typedef typename V8<T>::type V;
    constexpr int TPK = 32 / CC;
    const int nst = nks << ncklog;
    V a0[2], a1[2], w[2][NT];
    auto load = [&](int st, int buf) __attribute__((always_inline)) {
        const int ks = st >> ncklog, ck = st & (nck - 1);
        int sh = __builtin_amdgcn_readlane(mytab, tab0 + ks * TPK);
        if (TPK == 2) {
            const int shb = __builtin_amdgcn_readlane(mytab, tab0 + ks * TPK + 1);
            sh = (fq >> 1) ? shb : sh;
        }
        const int off = sh * LDC + CC * ck;
        a0[buf] = *reinterpret_cast<const V*>(halo + hb0 + off);
        a1[buf] = *reinterpret_cast<const V*>(halo + hb1 + off);
        const T* wrow = wph + ks * kstride + CC * ck;
#pragma unroll
        for (int i = 0; i < NT; ++i) w[buf][i] = *reinterpret_cast<const V*>(wrow + (16 * i + fr) * LDW);
    };
    auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            mma32(acc[i][0], w[buf][i], a0[buf]);
            mma32(acc[i][1], w[buf][i], a1[buf]);
        }
    };
    load(0, 0);
    int st = 0;
    while (true) {
        if (st + 1 < nst) load(st + 1, 1);
        mma(0);
        if (++st >= nst) break;
        if (st + 1 < nst) load(st + 1, 0);
        mma(1);
        if (++st >= nst) break;
    }
}

template <typename T, int NT, int CC, int NPH>
__global__ __launch_bounds__(256) void halo_kernel(const sv_geom g, const sv_igemm_args_g AG, const halo_cfg c) {
    typedef typename V8<T>::type V;
    typedef typename V4<T>::type Q;
    constexpr int BN = 16 * NT;
    constexpr int VPP = CC / 8;                 // 8-channel vectors per pixel per chunk
    constexpr int LDC = CC + 16;                // LDS pixel stride (elements): 96 B / 64 B rows
    constexpr int TPK = 32 / CC;                // taps per 32-deep k step
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = reinterpret_cast<T*>(smem);                       // [HP][LDC]
    const int LDW = c.tslots * CC + 16;                         // weight row stride (elements)
    T* wl = halo + c.HP * LDC;                                  // [BN][LDW]
    double* ssum = reinterpret_cast<double*>(wl + BN * LDW);    // [2][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int mytab = c.shift[lane];                                  // tap-shift table: entry l in lane l (halo_phase_mma)
    // (the load is waited for HERE, once: a register that is still "in flight" at the head of the tile loop makes the
    //  compiler put s_waitcnt vmcnt(0) in front of its first use in EVERY iteration, draining the tile prefetches)
    asm volatile("v_mov_b32 %0, %0" : "+v"(mytab));
    const int Wq = 1 << c.wlog, Hq = 1 << c.hlog, HH = 1 << c.hhlog;
    const int BHq = g.B * Hq;                                   // rows of the (per-phase) output grid
    const int nT = (BHq + c.TR - 1) / c.TR;
    const int nNt = (g.N + BN - 1) / BN;
    const int L = blockIdx.x;
    int in_i, mt;
    if (nT >= 64) {                                             // XCD-affine: the channel tiles of a pixel tile share an L2
        const int xcd = L & 7, slot = L >> 3;
        in_i = slot % nNt;
        mt = (slot / nNt) * 8 + xcd;
        if (mt >= nT) return;
    } else {
        in_i = L % nNt;
        mt = L / nNt;
    }
    const int n0 = in_i * BN;
    const int R0 = mt * c.TR;                                   // first grid row of the tile (rows run over images)
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(a.w);
    T* __restrict__ O = reinterpret_cast<T*>(a.out);
    const T* __restrict__ R = reinterpret_cast<const T*>(a.residual);
    const T* __restrict__ EX = reinterpret_cast<const T*>(a.ex);
    const bool has_pro = a.pro_scale != nullptr;
    const bool want_sums = (a.stats != nullptr) || (EX != nullptr);
    // (per-launch scalars of the loops, pinned in vector registers: see conv3x3p_kernel -- a uniform value from the argument
    //  segment is otherwise re-loaded where it is used, s_load + s_waitcnt lgkmcnt(0), up to 40 times per tile here)
    float pslope = a.pro_slope, eslope = a.ex_slope;
    int emode = EX ? 2 : (a.stats ? 1 : 0), sparse = a.sparse_out;
    // (halo_kernel: plain locals -- its chunk loop has no such re-loads, and four more vector registers cost it a wave)

    if (tid < 2 * BN) ssum[tid] = 0.0;
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;

    // ---- staging slots (channel-chunk independent).  Halo: vector idx -> (LDS pixel, 8-channel vector); weights: vector idx
    //      -> (output channel, tap slot, 8-channel vector).  Everything a chunk's staging needs is computed HERE, once per
    //      block, and the staging itself is branch-free: a source that does not exist (padding, rows beyond the tensor, pad
    //      slots of a phase) reads a valid address and is replaced by zero, a slot beyond the tile's image is stored into a
    //      per-thread dummy vector behind the sums.  (The straightforward form -- `valid ? load : zero` per slot, the
    //      phase of a weight slot looked up from the kernel arguments inside the chunk loop -- compiled to one basic block
    //      per slot, each ending in s_waitcnt vmcnt(0): the requests of a chunk went out one by one, and the ConvTranspose
    //      128 -> 64 forward ran at 0.7 TB/s with its waves waiting 60 % of the time, profiles/r03_pmc_odd.txt.)
    const int HVn = c.HP * VPP;
    T* const dummy = reinterpret_cast<T*>(ssum + 2 * BN) + tid * 8;
    const int ddst = (int)(dummy - halo);
    constexpr int PFV = 4;                     // halo vectors per thread that are prefetched (every dispatched shape: <= 4)
    // weight vectors per thread: 16 * NT rows x tslots x CC / 8 vectors <= 256 * PFW (one phase: at most 16 tap slots)
    constexpr int PFW = (NPH == 1 && NT == 1) ? 4 : HMAXW;
    auto halo_slot = [&](int i, int& src, int& dst) __attribute__((always_inline)) {      // returns: the slot has a source
        const int idx = tid + 256 * i;
        src = 0;
        dst = ddst;
        if (idx >= HVn) return false;
        const int pix = idx / VPP, cv = idx - pix * VPP;
        const int lr = pix / c.LW, lc = pix - lr * c.LW;
        const int seg = lr / c.SR, off = lr - seg * c.SR;
        const int grow = R0 + (seg << c.hhlog);                 // first grid row of the segment
        const int b = grow >> c.hlog, qy0 = grow & (Hq - 1);
        const int iy = g.sy * qy0 + off + c.dymin, ix = lc + c.dxmin;
        dst = pix * LDC + 8 * cv;
        if (!(grow < BHq && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win)) return false;
        src = ((b * g.Hin + iy) * g.Win + ix) * g.ldx + 8 * cv;
        return true;
    };
    int hsrc[PFV], hdst[PFV];                  // element offset into x (channel chunk 0, clamped) ; LDS element offset
    unsigned hok = 0;                          // bit i: slot i has a source
#pragma unroll
    for (int i = 0; i < PFV; ++i)
        if (halo_slot(i, hsrc[i], hdst[i])) hok |= 1u << i;
    const int WVn = BN * c.tslots * VPP;
    const int mywb = c.wbase[lane], myws = c.wnst[lane];      // (tslots <= 64: slot s lives in lane s)
    int wsrc[PFW], wdst[PFW];                  // element offset into the packed weights (channel chunk 0) ; LDS element offset
    unsigned wok = 0;
    const int wl0 = c.HP * LDC;                // wl - halo
#pragma unroll
    for (int i = 0; i < PFW; ++i) {
        const int idx = tid + 256 * i;
        wsrc[i] = 0;
        wdst[i] = ddst;
        if (idx < WVn) {
            const int v = idx % VPP, r = idx / VPP;
            const int slot = r % c.tslots, n = r / c.tslots;
            wdst[i] = wl0 + n * LDW + slot * CC + 8 * v;
            const int wb = __shfl(mywb, slot), ws = __shfl(myws, slot);
            if (wb >= 0 && n0 + n < g.N) {
                wsrc[i] = wb + (n0 + n) * ws + 8 * v;
                wok |= 1u << i;
            }
        }
    }
    // ---- this lane's two output-grid positions --------------------------------------------------------------------------
    int hbase[2], prow[2], pcol[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int p = 32 * wave + 16 * ms + fr;
        prow[ms] = p >> c.wlog;
        pcol[ms] = p & (Wq - 1);
        const int seg = prow[ms] >> c.hhlog, rin = prow[ms] & (HH - 1);
        hbase[ms] = ((seg * c.SR + g.sy * rin) * c.LW + g.sx * pcol[ms]) * LDC + (CC == 16 ? 8 * (fq & 1) : 8 * fq);
    }

    f32x4 acc[NPH][NT][2];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[ph][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the chunk loop, register-prefetched: the requests of chunk i + 1 (PFV halo vectors + every weight vector + the
    //      BatchNorm coefficients per thread) go out right after chunk i has been published in LDS and fly during its MFMAs
    const int nck = g.Cin / CC;
    const int cv8 = 8 * (tid % VPP);            // a thread's 8-channel group is the same for all of its slots (256 % VPP == 0)
    V hv[PFV], wv[PFW];
    f32x4 ps0, ps1, pt0, pt1;
    // (without a prologue the coefficient requests read the head of the weights -- Cin floats are always there -- and their
    //  values are not used: an unconditional request needs no copy, hence no wait, at the join behind a branch)
    const float* const psc = has_pro ? a.pro_scale : reinterpret_cast<const float*>(Wg);
    const float* const psh = has_pro ? a.pro_shift : reinterpret_cast<const float*>(Wg);
    auto request = [&](int c0) __attribute__((always_inline)) {
        ps0 = *reinterpret_cast<const f32x4*>(psc + c0 + cv8);
        ps1 = *reinterpret_cast<const f32x4*>(psc + c0 + cv8 + 4);
        pt0 = *reinterpret_cast<const f32x4*>(psh + c0 + cv8);
        pt1 = *reinterpret_cast<const f32x4*>(psh + c0 + cv8 + 4);
#pragma unroll
        for (int i = 0; i < PFV; ++i) hv[i] = *reinterpret_cast<const V*>(X + hsrc[i] + c0);
#pragma unroll
        for (int i = 0; i < PFW; ++i) wv[i] = *reinterpret_cast<const V*>(Wg + wsrc[i] + c0);
    };
    auto publish = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PFV; ++i) {
            V o = hv[i];
            if (has_pro) o = bn_act8(hv[i], ps0, ps1, pt0, pt1, pslope);
            *reinterpret_cast<V*>(halo + hdst[i]) = ((hok >> i) & 1u) ? o : zero;
        }
        // (larger halo images -- none of the dispatched shapes, SV_OPT_HALO_ALL only -- stage the rest here, unprefetched)
#pragma unroll
        for (int i = PFV; i < HMAXV; ++i) {
            if (256 * i >= HVn) break;
            int src, dst;
            const bool ok = halo_slot(i, src, dst);
            const V x = *reinterpret_cast<const V*>(X + src + c0);
            V o = x;
            if (has_pro) o = bn_act8(x, ps0, ps1, pt0, pt1, pslope);
            *reinterpret_cast<V*>(halo + dst) = ok ? o : zero;
        }
#pragma unroll
        for (int i = 0; i < PFW; ++i) *reinterpret_cast<V*>(halo + wdst[i]) = ((wok >> i) & 1u) ? wv[i] : zero;
    };
    request(0);
    for (int ck = 0; ck < nck; ++ck) {
        if (ck > 0) __syncthreads();                // the previous chunk's MFMAs are done with the LDS tiles
        publish(ck * CC);
        __syncthreads();
        if (ck + 1 < nck) request((ck + 1) * CC);
        // ---- every phase x k step out of LDS ---------------------------------------------------------------------------------
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            if (ph >= g.nphase) break;
            if (c.nks[ph] > 0)
                halo_phase_mma<T, NT, CC>(acc[ph], halo, hbase[0], hbase[1], wl + c.tap0[ph] * CC + 8 * fq, fr, fq, LDC, LDW,
                                          TPK * CC, c.nks[ph], 1, 0, mytab, 16 * ph);
        }
    }

    // ---- epilogue: every phase's outputs, per-channel sums kept in registers and flushed once ------------------------------
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
    f32x4 bias[NT], esc[NT], esh[NT], emu[NT], ers[NT];
    bool nval[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + 16 * i + 4 * fq;
        nval[i] = n < g.N;
        const int nc = nval[i] ? n : 0;
        bias[i] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (EX) {
            esc[i] = *reinterpret_cast<const f32x4*>(a.ex_scale + nc);
            esh[i] = *reinterpret_cast<const f32x4*>(a.ex_shift + nc);
            emu[i] = *reinterpret_cast<const f32x4*>(a.ex_mean + nc);
            ers[i] = *reinterpret_cast<const f32x4*>(a.ex_rstd + nc);
        }
    }
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph >= g.nphase) break;
        if (sparse && c.nks[ph] == 0) continue;       // tapless phase: exactly zero, left unwritten (sv_bn_branch::sparse)
        const sv_phase& P = g.phase[ph];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            // every operand of the (phase, row) pair is requested before the first is consumed (clamped addresses: rows / channel
            // groups beyond the tensor read a valid element and are not stored) -- one exposed round trip instead of NT
            const int grow = R0 + prow[ms];
            const bool rok = grow < BHq;
            const int growc = rok ? grow : BHq - 1;
            const int b = growc >> c.hlog, qy = growc & (Hq - 1);
            const int64_t ob = ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + pcol[ms] * g.osx + P.oox) * g.ldo;
            Q eo[NT];
            if (R || EX) {
                const T* __restrict__ E = R ? R : EX;
#pragma unroll
                for (int i = 0; i < NT; ++i) eo[i] = *reinterpret_cast<const Q*>(E + ob + (nval[i] ? n0 + 16 * i + 4 * fq : 0));
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (!nval[i] || !rok) continue;
                const int n = n0 + 16 * i + 4 * fq;
                f32x4 vv = acc[ph][i][ms];
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] += bias[i][r];
                if (R) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] += to_f(eo[i][r]);
                } else if (EX) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xf = to_f(eo[i][r]);
                        const float gv = vv[r] * act_grad(xf * esc[i][r] + esh[i][r], eslope);
                        vv[r] = gv;
                        s1[i][r] += gv;
                        s2[i][r] += gv * ((xf - emu[i][r]) * ers[i][r]);
                    }
                }
                if (emode == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[i][r] += vv[r];
                        s2[i][r] += vv[r] * vv[r];
                    }
                }
                Q o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (T)vv[r];
                *reinterpret_cast<Q*>(O + ob + n) = o;
            }
        }
    }
    if (want_sums) flush_channel_sums<NT>(s1, s2, nval, ssum, EX ? a.bsums : a.stats, n0, g.N, a.replicas, a.flags);
}

// ------------------------------------------------------------------------------------------------------------------------
// PERSISTENT variant for the thin layers (Cin <= 64: the stem, the 16 -> 32 block, the 32 -> 64 stride-2 convolution, the last
// ConvTranspose, and the data gradients of all of them): the weights of every tap of every phase and ALL input channels of
// the halo tile are LDS-resident, so a block stages its weights ONCE and walks a contiguous range of tiles with the
// register-prefetch pipeline of conv3x3p (conv3x3.hip): while tile i is on the MFMAs the halo of tile i + 2 is already
// requested (two register stages), the BatchNorm sums stay in registers across tiles and are flushed once per block.
// These layers are HBM-bound (2-4.5 flop/B x C): what matters is that every input byte is fetched once and that the
// requests never stop.
constexpr int PMAXV = 10;       // halo vectors per thread per register stage (upper bound; the kernel is compiled for PV <= PMAXV)

// PV = halo vectors per thread per register stage (2 for the 16-channel layers, 4, or 6).  The small variants fit more blocks
// per CU (OCC = blocks per CU the register budget is held to): the compiler waits for register prefetches with
// s_waitcnt vmcnt(0), so a block never has more than ~one tile of loads in flight -- bytes in flight per CU, i.e. the
// achievable bandwidth of these HBM-bound layers, scale with the number of resident blocks.
// MODE: fusion flags at compile time (0 = from the arguments; 1 = prologue + statistics; 2 = prologue only (the 1x1 shortcuts);
// 3 = activation-backward epilogue, no prologue; no bias / residual in 1..3) -- straight-line epilogue, 40-50 registers fewer (conv3x3p_kernel)
template <typename T, int NT, int CC, int NPH, int PV, int OCC, int MODE>
__global__ __launch_bounds__(256, OCC) void halop_kernel(const sv_geom g, const sv_igemm_args_g AG, const halo_cfg c, int tiles_per) {
    constexpr int PMAXV = PV;
    typedef typename V8<T>::type V;
    typedef typename V4<T>::type Q;
    constexpr int BN = 16 * NT;
    constexpr int TPK = 32 / CC;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    const int Cin = g.Cin, VPP = Cin / 8, NCK = Cin / CC;       // NCK = 1 or 2 (Cin <= 64)
    const int ncklog = NCK > 1 ? 1 : 0;
    const int LDC = Cin + 16;                                   // LDS pixel stride (elements)
    const int LDW = c.tslots * Cin + 16;                        // weight row stride (elements)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = reinterpret_cast<T*>(smem);                       // [HP][LDC]
    T* wl = halo + c.HP * LDC;                                  // [BN][LDW]
    double* ssum = reinterpret_cast<double*>(wl + BN * LDW);    // [2][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int mytab = c.shift[lane];                                  // tap-shift table: entry l in lane l (halo_phase_mma)
    // (the load is waited for HERE, once: a register that is still "in flight" at the head of the tile loop makes the
    //  compiler put s_waitcnt vmcnt(0) in front of its first use in EVERY iteration, draining the tile prefetches)
    asm volatile("v_mov_b32 %0, %0" : "+v"(mytab));
    const int Wq = 1 << c.wlog, Hq = 1 << c.hlog, HH = 1 << c.hhlog;
    const int BHq = g.B * Hq;
    const int nT = (BHq + c.TR - 1) / c.TR;
    const int nNt = (g.N + BN - 1) / BN;
    int in_i, chunk;
    if (nNt > 1 && (gridDim.x / nNt) % 8 == 0) {                // the channel tiles of a pixel range on one XCD
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        in_i = slot % nNt;
        chunk = (slot / nNt) * 8 + xcd;
    } else {
        in_i = blockIdx.x % nNt;
        chunk = blockIdx.x / nNt;
    }
    const int n0 = in_i * BN;
    const int t_begin = chunk * tiles_per, t_end = min(nT, t_begin + tiles_per);
    if (t_begin >= t_end) return;
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(a.w);
    T* __restrict__ O = reinterpret_cast<T*>(a.out);
    const T* __restrict__ R = MODE == 0 ? reinterpret_cast<const T*>(a.residual) : nullptr;
    const T* __restrict__ EX = MODE == 0 || MODE == 3 ? reinterpret_cast<const T*>(a.ex) : nullptr;
    const bool hasR = MODE == 0 ? R != nullptr : false, hasEX = MODE == 0 ? EX != nullptr : MODE == 3;
    const bool has_pro = MODE == 0 ? a.pro_scale != nullptr : (MODE == 1 || MODE == 2);
    const bool want_sums = MODE == 0 ? ((a.stats != nullptr) || (EX != nullptr)) : MODE != 2;
    // (per-launch scalars of the loops, pinned in vector registers: see conv3x3p_kernel -- a uniform value from the argument
    //  segment is otherwise re-loaded where it is used, s_load + s_waitcnt lgkmcnt(0), up to 40 times per tile here)
    float pslope = a.pro_slope, eslope = a.ex_slope;
    int emode = MODE == 0 ? (EX ? 2 : (a.stats ? 1 : 0)) : (MODE == 3 ? 2 : MODE == 1 ? 1 : 0), sparse = a.sparse_out;
    asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(pslope), "+v"(eslope), "+v"(emode), "+v"(sparse));

    if (tid < 2 * BN) ssum[tid] = 0.0;
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;

    // ---- weights of every tap slot of every phase, all input channels: once per block --------------------------------------
    {
        const int WVn = BN * c.tslots * VPP;
        for (int idx = tid; idx < WVn; idx += 256) {
            const int v = idx % VPP, r = idx / VPP;
            const int slot = r % c.tslots, n = r / c.tslots;
            int ph = 0;
#pragma unroll
            for (int q = 1; q < NPH; ++q)
                if (q < g.nphase && slot >= c.tap0[q]) ph = q;
            const int t = slot - c.tap0[ph];
            V w = zero;
            if (t < g.phase[ph].ntap && n0 + n < g.N)
                w = *reinterpret_cast<const V*>(Wg + g.phase[ph].w_off + ((int64_t)(n0 + n) * g.phase[ph].ntap + t) * Cin + 8 * v);
            *reinterpret_cast<V*>(wl + n * LDW + slot * Cin + 8 * v) = w;
        }
    }
    // ---- halo staging slots: everything that does not depend on the tile ----------------------------------------------------
    // slot -> (LDS pixel, 8-channel vector); source = tile base + hoff, where the tile base is the input pixel
    // (image of the tile's first grid row, input row sy * qy0 + dymin, column dxmin) -- possibly outside the tensor, the
    // offsets of valid slots bring it back inside -- and segment s of a multi-image tile is the image s further on
    const int HVn = c.HP * VPP;
    int hoff[PMAXV], hdst[PMAXV], hrs[PMAXV];       // element offset from the tile base; LDS element offset; row within the
                                                    // segment | segment << 8 (or -1: never valid)
#pragma unroll
    for (int i = 0; i < PMAXV; ++i) {
        const int idx = tid + 256 * i;
        hoff[i] = 0; hdst[i] = -1; hrs[i] = -1;
        if (idx < HVn) {
            const int pix = idx / VPP, cv = idx - pix * VPP;
            const int lr = pix / c.LW, lc = pix - lr * c.LW;
            const int seg = lr / c.SR, off = lr - seg * c.SR;
            hdst[i] = pix * LDC + 8 * cv;
            if ((unsigned)(lc + c.dxmin) < (unsigned)g.Win) {
                hrs[i] = off | (seg << 8);
                hoff[i] = ((seg * g.Hin + off) * g.Win + lc) * g.ldx + 8 * cv;
            }
        }
    }
    f32x4 ps0 = {1.f, 1.f, 1.f, 1.f}, ps1 = ps0, pt0 = {0.f, 0.f, 0.f, 0.f}, pt1 = pt0;
    if (has_pro && a.fold_stats) {
        // the BatchNorm in front of this layer is finalised HERE (sv_igemm_args::fold_*, claimed by the launcher): every block
        // derives the coefficients of the <= 64 input channels from the raw statistics -- the halo area is free until the first
        // tile is stored --, the first block of the launch stores them for the backward pass
        const int cv8 = 8 * (tid % VPP);
        float* fs = reinterpret_cast<float*>(halo);
        sv_bn_fold_block(a, Cin, reinterpret_cast<double*>(halo), fs + 1024, fs + 1024 + Cin, blockIdx.x == 0);
        ps0 = *reinterpret_cast<const f32x4*>(fs + 1024 + cv8);
        ps1 = *reinterpret_cast<const f32x4*>(fs + 1024 + cv8 + 4);
        pt0 = *reinterpret_cast<const f32x4*>(fs + 1024 + Cin + cv8);
        pt1 = *reinterpret_cast<const f32x4*>(fs + 1024 + Cin + cv8 + 4);
        __syncthreads();
    } else if (has_pro) {                           // a thread's 8-channel group is the same for all of its slots (256 % VPP == 0)
        const int cv8 = 8 * (tid % VPP);
        ps0 = *reinterpret_cast<const f32x4*>(a.pro_scale + cv8);
        ps1 = *reinterpret_cast<const f32x4*>(a.pro_scale + cv8 + 4);
        pt0 = *reinterpret_cast<const f32x4*>(a.pro_shift + cv8);
        pt1 = *reinterpret_cast<const f32x4*>(a.pro_shift + cv8 + 4);
    }
    struct HStage { V hv[PMAXV]; bool hok[PMAXV]; };
    HStage HA, HB;
    auto load_halo = [&](HStage& S, int tile) {
        const int R0 = tile * c.TR;
        const int b0 = R0 >> c.hlog, qy0 = R0 & (Hq - 1);
        const int iy0 = g.sy * qy0 + c.dymin;                                  // input row of LDS row 0 of a segment
        const int64_t base = ((int64_t)(b0 * g.Hin + iy0) * g.Win + c.dxmin) * g.ldx;
        const int64_t safe = ((int64_t)(b0 * g.Hin + g.sy * qy0) * g.Win) * g.ldx;    // the tile's first input pixel
#pragma unroll
        for (int i = 0; i < PMAXV; ++i) {
            if (256 * i >= HVn) break;
            const bool ok = hrs[i] >= 0 && (unsigned)(iy0 + (hrs[i] & 255)) < (unsigned)g.Hin &&
                            R0 + ((hrs[i] >> 8) << c.hhlog) < BHq;
            S.hok[i] = ok;
            S.hv[i] = *reinterpret_cast<const V*>(X + (ok ? base + hoff[i] : safe));
        }
    };
    auto store_halo = [&](HStage& S) {
#pragma unroll
        for (int i = 0; i < PMAXV; ++i) {
            if (256 * i >= HVn) break;
            V o = S.hv[i];
            if (has_pro) o = bn_act8(S.hv[i], ps0, ps1, pt0, pt1, pslope);
            if (hdst[i] >= 0) *reinterpret_cast<V*>(halo + hdst[i]) = S.hok[i] ? o : zero;
        }
    };

    // ---- this lane's two output-grid positions inside a tile, epilogue constants ----------------------------------------------
    int hbase[2], prow[2], pcol[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int p = 32 * wave + 16 * ms + fr;
        prow[ms] = p >> c.wlog;
        pcol[ms] = p & (Wq - 1);
        const int seg = prow[ms] >> c.hhlog, rin = prow[ms] & (HH - 1);
        hbase[ms] = ((seg * c.SR + g.sy * rin) * c.LW + g.sx * pcol[ms]) * LDC + (CC == 16 ? 8 * (fq & 1) : 8 * fq);
    }
    f32x4 bias[NT], esc[NT], esh[NT], emu[NT], ers[NT];
    bool nval[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + 16 * i + 4 * fq;
        nval[i] = n < g.N;
        const int nc = nval[i] ? n : 0;
        bias[i] = (MODE == 0 && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (hasEX) {
            esc[i] = *reinterpret_cast<const f32x4*>(a.ex_scale + nc);
            esh[i] = *reinterpret_cast<const f32x4*>(a.ex_shift + nc);
            emu[i] = *reinterpret_cast<const f32x4*>(a.ex_mean + nc);
            ers[i] = *reinterpret_cast<const f32x4*>(a.ex_rstd + nc);
        }
    }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;

    load_halo(HA, t_begin);
    if (t_begin + 1 < t_end) load_halo(HB, t_begin + 1);
    store_halo(HA);
    __syncthreads();
    // the epilogue's extra operand (residual OR raw tensor) of every (phase, row, channel group) of a tile is requested ONE TILE
    // AHEAD (clamped addresses: rows / channel groups beyond the tensor read a valid element and are not stored).  Requested
    // inside the epilogue it was an exposed round trip per tile (waves of the thin layers waited 70 % of their cycles,
    // tools/pmc_sq.py); requested at the top of its own tile, behind the halo of tile + 2, it still made the epilogue wait for
    // that halo -- loads return in order (conv3x3p: data gradient 112 -> 93 us with the same change).
    struct EStage { Q eo[NPH][2][NT]; };
    auto load_eo = [&](EStage& E, int tile) {
        const int R0 = tile * c.TR;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int grow = R0 + prow[ms];
            const int growc = grow < BHq ? grow : BHq - 1;
            const int b = growc >> c.hlog, qy = growc & (Hq - 1);
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                const sv_phase& P = g.phase[ph < g.nphase ? ph : 0];
                const int64_t ob = ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + pcol[ms] * g.osx + P.oox) * g.ldo;
                if ((hasR || hasEX) && ph < g.nphase && !(sparse && c.nks[ph] == 0)) {
                    const T* __restrict__ E_ = R ? R : EX;
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        E.eo[ph][ms][i] = *reinterpret_cast<const Q*>(E_ + ob + (nval[i] ? n0 + 16 * i + 4 * fq : 0));
                }
            }
        }
    };
    EStage EA, EB;
    if (SV_HALOP_EO_AHEAD && (hasR || hasEX)) load_eo(EA, t_begin);
    auto do_tile = [&](int tile, HStage& NEXT, HStage& FREE, EStage& ECUR, EStage& ENEXT) {
        const int R0 = tile * c.TR;
        const bool more = tile + 1 < t_end;
        if (tile + 2 < t_end) load_halo(FREE, tile + 2);              // flies during this tile's and the next tile's MFMAs
        if (SV_HALOP_EO_AHEAD) { if (more && (hasR || hasEX)) load_eo(ENEXT, tile + 1); }
        else if (hasR || hasEX) load_eo(ECUR, tile);
        Q (&eo)[NPH][2][NT] = ECUR.eo;
        int64_t obv[NPH][2];
        bool rokv[2];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int grow = R0 + prow[ms];
            rokv[ms] = grow < BHq;
            const int growc = rokv[ms] ? grow : BHq - 1;
            const int b = growc >> c.hlog, qy = growc & (Hq - 1);
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                const sv_phase& P = g.phase[ph < g.nphase ? ph : 0];
                obv[ph][ms] = ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + pcol[ms] * g.osx + P.oox) * g.ldo;
            }
        }
        f32x4 acc[NPH][NT][2];
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[ph][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            if (ph >= g.nphase) break;
            if (c.nks[ph] > 0)
                halo_phase_mma<T, NT, CC>(acc[ph], halo, hbase[0], hbase[1], wl + c.tap0[ph] * Cin + 8 * fq, fr, fq, LDC, LDW,
                                          TPK * Cin, c.nks[ph], NCK, ncklog, mytab, 16 * ph);
        }
        __syncthreads();                               // every wave is done reading this tile's halo
        if (more) store_halo(NEXT);                    // the next tile's halo -> LDS (requested a whole tile ago)
        // ---- epilogue of this tile ------------------------------------------------------------------------------------------
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            if (ph >= g.nphase) break;
            if (sparse && c.nks[ph] == 0) continue;   // tapless phase: left unwritten
            const sv_phase& P = g.phase[ph];
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const bool rok = rokv[ms];
                const int64_t ob = obv[ph][ms];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (!nval[i] || !rok) continue;
                    const int n = n0 + 16 * i + 4 * fq;
                    f32x4 vv = acc[ph][i][ms];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] += bias[i][r];
                    if (hasR) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vv[r] += to_f(eo[ph][ms][i][r]);
                    } else if (hasEX) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xf = to_f(eo[ph][ms][i][r]);
                            const float gv = vv[r] * act_grad(xf * esc[i][r] + esh[i][r], eslope);
                            vv[r] = gv;
                            s1[i][r] += gv;
                            s2[i][r] += gv * ((xf - emu[i][r]) * ers[i][r]);
                        }
                    }
                    if (MODE == 0 ? emode == 1 : MODE == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s1[i][r] += vv[r];
                            s2[i][r] += vv[r] * vv[r];
                        }
                    }
                    Q o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (T)vv[r];
                    *reinterpret_cast<Q*>(O + ob + n) = o;
                }
            }
        }
        __syncthreads();                               // the next halo is visible
    };
    for (int tile = t_begin; tile < t_end; tile += 2) {
        do_tile(tile, HB, HA, EA, EB);
        if (tile + 1 < t_end) do_tile(tile + 1, HA, HB, EB, EA);
    }
    if (want_sums) flush_channel_sums<NT>(s1, s2, nval, ssum, hasEX ? a.bsums : a.stats, n0, g.N, a.replicas, a.flags);
}

template <typename T, int NT, int CC, int NPH, int PV, int OCC, int MODE>
int launch_halop_pv(const sv_geom* g, const sv_igemm_args* a, const halo_cfg& c, size_t lds, hipStream_t s) {
    constexpr int BN = 16 * NT;
    const int nT = (g->B * g->Hq + c.TR - 1) / c.TR;
    const int nNt = (g->N + BN - 1) / BN;
    const int budget = sv_persistent_blocks() * OCC / 2;
    const int target = budget / sv_ngroups(a->groups) > 64 ? budget / sv_ngroups(a->groups) : 64;
    int chunks = (target + nNt - 1) / nNt;
    if (chunks > nT) chunks = nT;
    const int tiles_per = (nT + chunks - 1) / chunks;
    chunks = (nT + tiles_per - 1) / tiles_per;
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&halop_kernel<T, NT, CC, NPH, PV, OCC, MODE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(halop)");
        optin = true;
    }
    sv_igemm_args b = *a;          // the persistent kernel folds the BatchNorm finalisation of its prologue (fold_*)
    if (!sv_fold_claim(b.fold_stats && b.fold_replicas <= 64 && 256 % g->Cin == 0 && (size_t)c.HP * (g->Cin + 16) * sizeof(T) >= (1024 + 2 * 64) * 4))
        b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(chunks * nNt, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((halop_kernel<T, NT, CC, NPH, PV, OCC, MODE>), dim3(chunks * nNt, sv_ngroups(a->groups)), dim3(256), lds, s, *g,
                       sv_expand_groups(*g, *a, (int)sizeof(T)), c, tiles_per);
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(halop)");
}

template <typename T, int NT, int CC, int NPH, int MODE>
int launch_halop_m(const sv_geom* g, const sv_igemm_args* a, const halo_cfg& c, size_t lds, hipStream_t s) {
    // register stage size from the tile's halo; the small stages go with more resident blocks (their LDS image permitting)
    const int hvn = c.HP * (g->Cin / 8);
    if constexpr (sizeof(T) == 2 && NPH == 1) {
        // (with the flags at compile time the variants need 20-35 registers fewer: one more resident block each)
        constexpr int O2 = NT == 1 ? (MODE ? SV_HALOP_OCC2_M : SV_HALOP_OCC2) : (MODE ? SV_HALOP_OCC2B_M : SV_HALOP_OCC2B);
        constexpr int O4 = NT == 1 ? (MODE ? SV_HALOP_OCC4_M : SV_HALOP_OCC4) : (MODE ? SV_HALOP_OCC4B_M : SV_HALOP_OCC4B);
        constexpr int O6 = (NT == 1 && MODE) ? SV_HALOP_OCC6_M : 2;
        if constexpr (O2 > 2)
            if (hvn <= 256 * 2 && lds * O2 <= 150 * 1024) return launch_halop_pv<T, NT, CC, NPH, 2, O2, MODE>(g, a, c, lds, s);
        if constexpr (O4 > 2)
            if (hvn <= 256 * 4 && lds * O4 <= 150 * 1024) return launch_halop_pv<T, NT, CC, NPH, 4, O4, MODE>(g, a, c, lds, s);
        if constexpr (O6 > 2)
            if (lds * O6 <= 150 * 1024) return launch_halop_pv<T, NT, CC, NPH, 6, O6, MODE>(g, a, c, lds, s);
    }
    // (the 10-vector stage: the stride-2 3x3 forward 32 -> 64, whose 128 outputs read a 17 x 33 pixel region; the stride-2 1x1
    //  shortcut of the same shape stays with the gather-GEMM -- it needs a quarter of that region: 40 us against 62)
    if constexpr (NPH == 1 && sizeof(T) == 2)
        if (hvn > 256 * 6) return launch_halop_pv<T, NT, CC, NPH, 10, 2, MODE>(g, a, c, lds, s);
    return launch_halop_pv<T, NT, CC, NPH, 6, 2, MODE>(g, a, c, lds, s);
}

// the two launch kinds the step issues most take the binaries with their fusion flags at compile time (bf16)
template <typename T, int NT, int CC, int NPH>
int launch_halop(const sv_geom* g, const sv_igemm_args* a, const halo_cfg& c, size_t lds, hipStream_t s) {
#if SV_HALOP_MODES
    if constexpr (sizeof(T) == 2) {
        if (!a->bias && !a->residual) {
            if (a->pro_scale && a->stats && !a->ex) return launch_halop_m<T, NT, CC, NPH, 1>(g, a, c, lds, s);
            if (a->pro_scale && !a->stats && !a->ex) return launch_halop_m<T, NT, CC, NPH, 2>(g, a, c, lds, s);
            if (!a->pro_scale && a->ex && !a->stats) return launch_halop_m<T, NT, CC, NPH, 3>(g, a, c, lds, s);
        }
    }
#endif
    return launch_halop_m<T, NT, CC, NPH, 0>(g, a, c, lds, s);
}

template <typename T, int CC, int NPH>
int launch_halop_nt(const sv_geom* g, const sv_igemm_args* a, const halo_cfg& c, int nt, size_t lds, hipStream_t s) {
    return nt == 2 ? launch_halop<T, 2, CC, NPH>(g, a, c, lds, s) : launch_halop<T, 1, CC, NPH>(g, a, c, lds, s);
}

int ilog2x(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

template <typename T, int NT, int CC, int NPH>
int launch_halo(const sv_geom* g, const sv_igemm_args* a, const halo_cfg& c, hipStream_t s) {
    constexpr int BN = 16 * NT;
    const int nT = (g->B * g->Hq + c.TR - 1) / c.TR;
    const int nNt = (g->N + BN - 1) / BN;
    const int grid = (nT >= 64 ? ((nT + 7) / 8) * 8 : nT) * nNt;
    const size_t lds = ((size_t)c.HP * (CC + 16) + (size_t)BN * (c.tslots * CC + 16)) * sizeof(T) + 2 * BN * sizeof(double) +
                       256 * 8 * sizeof(T);       // + the per-thread dummy vectors of the branch-free staging
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&halo_kernel<T, NT, CC, NPH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(halo)");
        optin = true;
    }
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((halo_kernel<T, NT, CC, NPH>), dim3(grid, sv_ngroups(a->groups)), dim3(256), lds, s, *g,
                       sv_expand_groups(*g, *a, (int)sizeof(T)), c);
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(halo)");
}

template <typename T, int CC, int NPH>
int launch_halo_nt(const sv_geom* g, const sv_igemm_args* a, const halo_cfg& c, int nt, hipStream_t s) {
    return nt == 2 ? launch_halo<T, 2, CC, NPH>(g, a, c, s) : launch_halo<T, 1, CC, NPH>(g, a, c, s);
}

}  // namespace

// Returns 1 and sets *rc when the geometry is covered: square power-of-two grids, taps within a small window, the
// tile's LDS image within budget.
int sv_halo_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_HALO)) return 0;
    if (a->residual && a->ex) return 0;          // (the epilogues here take one extra operand: residual OR raw tensor)
    // Measured against the generic gather-GEMM on the WRN-28-2 / decoder shapes at 4 x 512 images (profiles/r02_*): this
    // one-tile-per-block version wins where a phase has few output channels and the staging is small -- the last two
    // ConvTranspose layers forward (207 -> 125 us, 181 -> 153 us) -- and loses elsewhere (two blocks per CU of 60-90 KB LDS
    // and no overlap of staging and MFMA: 32 -> 64 stride-2 forward 112 -> 214 us).  Until it is persistent and
    // software-pipelined like conv3x3p it only takes the former; SV_OPT_HALO_ALL = 1 takes everything it covers (tests).
    if (g->Hq != g->Wq || g->Hin != g->Win || g->sy != g->sx || g->osy != g->osx) return 0;
    if (g->sy < 1 || g->sy > 2) return 0;
    halo_cfg c;
    c.wlog = ilog2x(g->Wq);
    c.hlog = ilog2x(g->Hq);
    if (c.wlog < 1 || c.wlog > 5) return 0;              // 2 .. 32 columns (1 x 1 maps are plain GEMMs: igemm.hip)
    if (g->Cin % 16 != 0 || g->N % 16 != 0 || g->ldx != g->Cin) return 0;
    if (g->Cin % 32 != 0 && g->Cin != 16) return 0;
    if (g->nphase < 1 || g->nphase > SV_MAX_PHASES) return 0;
    int dymin = 127, dymax = -127, dxmin = 127, dxmax = -127, ttot = 0;
    for (int p = 0; p < g->nphase; ++p)
        for (int t = 0; t < g->phase[p].ntap; ++t) {
            const int dy = g->phase[p].dy[t], dx = g->phase[p].dx[t];
            dymin = dy < dymin ? dy : dymin; dymax = dy > dymax ? dy : dymax;
            dxmin = dx < dxmin ? dx : dxmin; dxmax = dx > dxmax ? dx : dxmax;
            ++ttot;
        }
    if (ttot == 0) return 0;
    if (dymax - dymin > 3 || dxmax - dxmin > 3) return 0;
    const int CC = g->Cin == 16 ? 16 : 32, TPK = 32 / CC;
    const int Wq = g->Wq, Hq = g->Hq;
    c.TR = 128 / Wq;
    const int HH = c.TR < Hq ? c.TR : Hq;
    c.hhlog = ilog2x(HH);
    c.SEG = c.TR / HH;
    c.SR = g->sy * (HH - 1) + (dymax - dymin) + 1;
    c.LW = g->sx * (Wq - 1) + (dxmax - dxmin) + 1;
    c.dymin = dymin;
    c.dxmin = dxmin;
    c.HP = c.SEG * c.SR * c.LW;
    int slots = 0;
    for (int p = 0; p < SV_MAX_PHASES; ++p) {
        c.tap0[p] = slots;
        c.nks[p] = 0;
        if (p < g->nphase) {
            c.nks[p] = (g->phase[p].ntap + TPK - 1) / TPK;
            slots += c.nks[p] * TPK;
        }
    }
    c.tslots = slots;
    for (int p = 0; p < SV_MAX_PHASES; ++p)
        for (int sl = 0; sl < 16; ++sl) {
            int v = 0;
            if (p < g->nphase && g->phase[p].ntap > 0) {
                const int t = sl < g->phase[p].ntap ? sl : 0;
                v = (g->phase[p].dy[t] - dymin) * c.LW + (g->phase[p].dx[t] - dxmin);
            }
            c.shift[16 * p + sl] = v;
        }
    for (int sl = 0; sl < SV_MAX_PHASES * 16; ++sl) { c.wbase[sl] = -1; c.wnst[sl] = 0; }
    if (slots > SV_MAX_PHASES * 16) return 0;
    for (int p = 0; p < g->nphase; ++p)          // (weight offsets are 32-bit in the kernels)
        if (g->phase[p].w_off + (int64_t)g->N * g->phase[p].ntap * g->Cin >= ((int64_t)1 << 31)) return 0;
    {
        for (int p = 0; p < g->nphase; ++p)
            for (int t = 0; t < g->phase[p].ntap; ++t) {
                c.wbase[c.tap0[p] + t] = (int)(g->phase[p].w_off + (int64_t)t * g->Cin);
                c.wnst[c.tap0[p] + t] = g->phase[p].ntap * g->Cin;
            }
    }
    const int es = dtype == SV_BF16 ? 2 : 4;
    const int nt = (g->N % 32 == 0) ? 2 : 1, BN = 16 * nt;
    if (c.HP * (CC / 8) > 256 * HMAXV) return 0;
    if (BN * c.tslots * (CC / 8) > 256 * HMAXW) return 0;
    const size_t lds = ((size_t)c.HP * (CC + 16) + (size_t)BN * (c.tslots * CC + 16)) * es + 2 * BN * 8 + 256 * 8 * es;
    if (lds > 100 * 1024) return 0;
    if ((int64_t)g->B * g->Hin * g->Win * g->ldx >= ((int64_t)1 << 31)) return 0;
    const bool multi = g->nphase > 1;
    // thin layers: the persistent variant (all channels + all weights LDS-resident, two blocks per CU)
    if (!sv_disabled(SV_K_HALOP) && g->Cin <= 64) {
        const int nt = multi ? 1 : (g->N % 32 == 0 ? 2 : 1), BN = 16 * nt;     // (four accumulator sets: 16-channel tiles)
        const size_t ldsp = ((size_t)c.HP * (g->Cin + 16) + (size_t)BN * (c.tslots * g->Cin + 16)) * es + 2 * BN * 8;
        const int nT = (g->B * g->Hq + c.TR - 1) / c.TR;
        // (at most two channel tiles: every tile re-stages the input region -- 16 -> 160 as five tiles ran 564 us against 302)
        if (ldsp <= 76 * 1024 && c.HP * (g->Cin / 8) <= 256 * ((multi || dtype != SV_BF16 || ttot < 4) ? 6 : PMAXV) && 256 % (g->Cin / 8) == 0 && (g->N + BN - 1) / BN <= 2 &&
            (sv_halo_all() || nT * sv_ngroups(a->groups) >= 1024)) {       // (a few tiles per block at least; tests: any size)
            if (dtype == SV_BF16) {
                if (CC == 16) *rc = multi ? launch_halop_nt<bf16, 16, 4>(g, a, c, nt, ldsp, s) : launch_halop_nt<bf16, 16, 1>(g, a, c, nt, ldsp, s);
                else *rc = multi ? launch_halop_nt<bf16, 32, 4>(g, a, c, nt, ldsp, s) : launch_halop_nt<bf16, 32, 1>(g, a, c, nt, ldsp, s);
            } else {
                if (CC == 16) *rc = multi ? launch_halop_nt<float, 16, 4>(g, a, c, nt, ldsp, s) : launch_halop_nt<float, 16, 1>(g, a, c, nt, ldsp, s);
                else *rc = multi ? launch_halop_nt<float, 32, 4>(g, a, c, nt, ldsp, s) : launch_halop_nt<float, 32, 1>(g, a, c, nt, ldsp, s);
            }
            return 1;
        }
    }
    // ... and the stride-1 3x3 layers with few output channels and many input channels (the data gradient of the first
    // WRN-28-10 block, 160 -> 16: every dy element is staged once instead of gathered per tap, 417 -> 232 us)
    const bool thin_out = g->nphase == 1 && g->phase[0].ntap == 9 && g->sy == 1 && g->sx == 1 && g->N <= 32 && g->Cin >= 128 && g->Wq >= 8;
    // (Wq >= 4: svhn_VAE's ConvTranspose 128 -> 64 at 4x4 -> 8x8 and the data gradient of its third convolution, 41 -> 30 / 55 -> 40 us)
    if (!sv_halo_all() && !thin_out && !(g->nphase == 4 && g->sy == 1 && g->N <= 64 && g->Cin <= 128 && g->Wq >= 4)) return 0;
    if (dtype == SV_BF16) {
        if (CC == 16) *rc = multi ? launch_halo_nt<bf16, 16, 4>(g, a, c, nt, s) : launch_halo_nt<bf16, 16, 1>(g, a, c, nt, s);
        else *rc = multi ? launch_halo_nt<bf16, 32, 4>(g, a, c, nt, s) : launch_halo_nt<bf16, 32, 1>(g, a, c, nt, s);
    } else {
        if (CC == 16) *rc = multi ? launch_halo_nt<float, 16, 4>(g, a, c, nt, s) : launch_halo_nt<float, 16, 1>(g, a, c, nt, s);
        else *rc = multi ? launch_halo_nt<float, 32, 4>(g, a, c, nt, s) : launch_halo_nt<float, 32, 1>(g, a, c, nt, s);
    }
    return 1;
}
