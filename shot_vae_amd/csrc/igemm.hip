// Fused gather-GEMM ("implicit GEMM") on MFMA for every conv-like forward and data-gradient of the
// SHOT-VAE step: conv3x3 s1/s2, conv1x1 s1/s2, ConvTranspose 4x4 s2 p1 (as 4 sub-pixel phases),
// the 1x1 ConvTranspose (plain GEMM) and all of their dgrads.  NHWC, gfx950.
//
//   out[b, qy*osy+ooy, qx*osx+oox, n] = sum_{t,c} A(b,qy,qx,t,c) * W[n][t][c]
//   A = LeakyReLU(x*scale[c]+shift[c])  (BatchNorm-apply prologue, zero outside the image)
//
// Tile: 64*MS output positions x (16*NT) channels x 32*KV-deep k; 4 waves, wave w owns rows [16 MS w, 16 MS (w+1)).
// MS = 2 (128 rows) by default; MS = 4 (256 rows x 160 / 128 channels, two blocks per CU) for the wide layers.
// Weights are the MFMA A-operand (rows = n) and activations the B-operand (cols = m), so that each
// lane ends up with 4 consecutive channels of one output pixel -> 8/16-byte epilogue accesses.
// Epilogue: +bias, +residual, either per-channel (sum, sumsq) for the next BatchNorm or the
// activation backward + the two BatchNorm-backward sums; wave shuffle -> LDS atomics -> one global
// atomic per channel per block.
#include <stdlib.h>

#include "common.h"
#include "epilogue.h"

#ifndef SV_IG_DMA2
#define SV_IG_DMA2 1            // 128-row LDS-DMA tiles for the small products (0: off, for A/B runs with tools/ab.sh)
#endif
#ifndef SV_IG_MIN_TILES
#define SV_IG_MIN_TILES 512        // blocks a channel-tile width must yield to be taken (two per CU)
#endif

namespace {

constexpr int BK = 32;

// KV = 32-deep k sub-chunks per loop iteration (one barrier per iteration).  KV = 2 for deep-K layers with few tiles
// (the decoder, the stride-2 convs): those are bound by one global->LDS round trip per iteration, so twice the bytes
// in flight and twice the MFMAs per barrier nearly halve their time.
// MS = 16-row sub-tiles per wave: block tile = 64 MS output positions (128 by default; 256 x 160/128 channels for the
// wide layers, whose small tiles were bound by the L2 -> LDS traffic of re-gathering the input per channel tile).
// AL: Cin is a multiple of the k depth of an iteration, so an iteration never straddles a tap: tap and channel offset are
// block-uniform (scalar registers), the gather addresses of a tap are computed once per tap instead of once per iteration
// (the loop of the round-1 loader issued ~285 VALU instructions per 40 MFMAs on the wide layers -- the SQ counters showed
// the kernel VALU-issue-bound: tools/pmc_sq.py).
template <typename T, int NT, int KV, int MS, bool AL>
__global__ __launch_bounds__(256, MS == 4 ? 2 : 1) void igemm_kernel(const sv_geom g, const sv_igemm_args_g A) {
    const sv_igemm_args& a = A.g[blockIdx.y];
    sv_start_signal(a);
    typedef typename V8<T>::type V;
    typedef typename V4<T>::type Q;
    constexpr int BM = 64 * MS;
    constexpr int BN = 16 * NT;
    constexpr int NBV = (BN * 4 + 255) / 256;   // weight vectors per thread per 32-deep sub-chunk
    constexpr int BKK = BK * KV;
    constexpr int LDK = BKK + 16;               // LDS row stride (elements): 96 / 160 B rows keep the b128 fragment reads conflict-free

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);           // [2][BM][LDK]
    T* Bs = As + 2 * BM * LDK;                    // [2][BN][LDK]
    // [2][BN] channel sums; the 256-row tiles (two blocks per CU: <= 80 KB each) put them over the A buffers after the k loop
    double* ssum = MS == 4 ? reinterpret_cast<double*>(smem) : reinterpret_cast<double*>(Bs + 2 * BN * LDK);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HWq = g.Hq * g.Wq;
    const int M = g.B * HWq;
    const int nMt = (M + BM - 1) / BM;
    const int nNt = (g.N + BN - 1) / BN;
    const int inner = nNt * g.nphase;
    // XCD-aware mapping: blocks L and L+8 share an XCD (and its L2); all channel tiles and phases
    // of one m-tile run back to back on the same XCD so the gathered input is fetched once.
    const int L = blockIdx.x;
    int in_i, mt;
    if (nMt >= 64) {
        const int xcd = L & 7, slot = L >> 3;
        in_i = slot % inner;
        mt = (slot / inner) * 8 + xcd;
        if (mt >= nMt) return;
    } else {            // few m-tiles: plain mapping so that every XCD gets work
        in_i = L % inner;
        mt = L / inner;
    }
    const int ph = in_i / nNt;
    const int n0 = (in_i % nNt) * BN;
    const sv_phase& P = g.phase[ph];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const int ntap = P.ntap;
    const int Ktot = ntap * g.Cin;
    const int nk = (Ktot + BKK - 1) / BKK;
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ W = reinterpret_cast<const T*>(a.w) + P.w_off;
    const bool has_pro = a.pro_scale != nullptr;

    // A phase without taps (three of the four output parities of a stride-2 1x1 data gradient) is exactly zero, and so
    // are its contributions to the BatchNorm sums of either epilogue: plain 16-byte zero stores, no reads, no atomics.
    if (ntap == 0 && a.sparse_out) return;          // (the consumer does not read these positions: sv_bn_branch::sparse)
    if (ntap == 0 && !a.bias && !a.residual && g.ldo % 8 == 0) {
        constexpr int VR = BN / 8;
        V z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
        T* __restrict__ O = reinterpret_cast<T*>(a.out);
        for (int idx = tid; idx < BM * VR; idx += 256) {
            const int row = idx / VR, vv = idx - row * VR;
            const int m = mt * BM + row;
            if (m >= M || n0 + 8 * vv >= g.N) continue;
            const int b = m / HWq, r = m - b * HWq, qy = r / g.Wq, qx = r - qy * g.Wq;
            *reinterpret_cast<V*>(O + ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + qx * g.osx + P.oox) * g.ldo +
                                  n0 + 8 * vv) = z;
        }
        return;
    }

    if (MS != 4) {
        if (tid < 2 * BN) ssum[tid] = 0.0;
        if (BN > 128 && tid + 256 < 2 * BN) ssum[tid + 256] = 0.0;
    }

    // ---- loader state -------------------------------------------------------------------------
    const int v = tid & 3;                 // 8-element k vector inside the 32-deep chunk
    const int lrow = tid >> 2;             // 0..63
    int iy0[MS], ix0[MS], pixb[MS];
    bool mval[MS];
#pragma unroll
    for (int i = 0; i < MS; ++i) {
        const int m = mt * BM + lrow + 64 * i;
        mval[i] = m < M;
        const int mm = mval[i] ? m : 0;
        const int b = mm / HWq;
        const int r = mm - b * HWq;
        const int qy = r / g.Wq;
        const int qx = r - qy * g.Wq;
        iy0[i] = qy * g.sy;
        ix0[i] = qx * g.sx;
        pixb[i] = b * g.Hin * g.Win;
    }
    int tap[KV], c[KV];
#pragma unroll
    for (int s = 0; s < KV; ++s) {
        tap[s] = 0;
        c[s] = 8 * v + BK * s;
        while (c[s] >= g.Cin) { c[s] -= g.Cin; ++tap[s]; }
    }

    V ra[KV][MS], rb[KV][NBV];
    bool oka[KV][MS];
    f32x4 ps0[KV], ps1[KV], pt0[KV], pt1[KV];        // BN scale / shift of the chunk in flight (fetched WITH its data)

    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;
    // ---- aligned loader state (AL): per-row gather pointer and validity of the CURRENT tap, weight row pointers ----------
    const T* pa[MS];
    bool okr[MS];
    const T* pw[NBV];
    bool wok[NBV];
    int tapu = 0, cu = 0;                   // block-uniform: tap and first channel of the next iteration to load
    auto set_tap = [&](int t) {
        const int tt = t < SV_MAX_TAPS ? t : SV_MAX_TAPS - 1;
        const int dy = tap_off(pdy, tt), dx = tap_off(pdx, tt);
#pragma unroll
        for (int i = 0; i < MS; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            okr[i] = mval[i] && t < ntap && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
            const int iyc = min(max(iy, 0), g.Hin - 1), ixc = min(max(ix, 0), g.Win - 1);
            pa[i] = X + ((int64_t)(pixb[i] + iyc * g.Win + ixc) * g.ldx + 8 * v);
        }
    };
    if (AL) {
        set_tap(0);
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            const int nb = lrow + 64 * i;
            wok[i] = nb < BN && n0 + nb < g.N;
            pw[i] = W + (int64_t)min(n0 + nb, g.N - 1) * Ktot + 8 * v;
        }
    }
    auto load_aligned = [&](int kc) {
#pragma unroll
        for (int s = 0; s < KV; ++s) {
            const int cs = cu + BK * s;                                   // uniform
            if (has_pro) {
                const float* sc = a.pro_scale + cs + 8 * v;
                const float* sh = a.pro_shift + cs + 8 * v;
                ps0[s] = *reinterpret_cast<const f32x4*>(sc);
                ps1[s] = *reinterpret_cast<const f32x4*>(sc + 4);
                pt0[s] = *reinterpret_cast<const f32x4*>(sh);
                pt1[s] = *reinterpret_cast<const f32x4*>(sh + 4);
            }
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                oka[s][i] = okr[i];
                const V val = *reinterpret_cast<const V*>(pa[i] + cs);
                ra[s][i] = okr[i] ? val : zero;
            }
            const int ks = kc * BKK + BK * s;                             // uniform; < Ktot by construction
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                const V val = *reinterpret_cast<const V*>(pw[i] + ks);
                rb[s][i] = wok[i] ? val : zero;
            }
        }
        cu += BKK;
        if (cu >= g.Cin) {          // (uniform branch) next iteration starts the next tap
            cu = 0;
            ++tapu;
            set_tap(tapu);
        }
    };
    // Branch-free general loader: clamped addresses, unconditional loads (all in flight together), select.
    auto load_general = [&](int kc) {
#pragma unroll
        for (int s = 0; s < KV; ++s) {
            const int tt = tap[s] < SV_MAX_TAPS ? tap[s] : SV_MAX_TAPS - 1;
            const int dy = tap_off(pdy, tt), dx = tap_off(pdx, tt);
            const int cc = min(c[s], g.Cin - 8);
            if (has_pro) {
                ps0[s] = *reinterpret_cast<const f32x4*>(a.pro_scale + cc);
                ps1[s] = *reinterpret_cast<const f32x4*>(a.pro_scale + cc + 4);
                pt0[s] = *reinterpret_cast<const f32x4*>(a.pro_shift + cc);
                pt1[s] = *reinterpret_cast<const f32x4*>(a.pro_shift + cc + 4);
            }
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                const int iy = iy0[i] + dy, ix = ix0[i] + dx;
                const bool ok = mval[i] && tap[s] < ntap && (unsigned)iy < (unsigned)g.Hin &&
                                (unsigned)ix < (unsigned)g.Win;
                oka[s][i] = ok;
                const int iyc = min(max(iy, 0), g.Hin - 1), ixc = min(max(ix, 0), g.Win - 1);
                const V val = *reinterpret_cast<const V*>(X + ((int64_t)(pixb[i] + iyc * g.Win + ixc) * g.ldx + cc));
                ra[s][i] = ok ? val : zero;
            }
            const int k8 = kc * BKK + BK * s + 8 * v;
            const int k8c = min(k8, Ktot - 8);
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                const int nb = lrow + 64 * i;
                const bool ok = nb < BN && n0 + nb < g.N && k8 < Ktot;
                const V val = *reinterpret_cast<const V*>(W + (int64_t)min(n0 + nb, g.N - 1) * Ktot + k8c);
                rb[s][i] = ok ? val : zero;
            }
            // advance (tap, c) to this sub-chunk's position in the next iteration
            c[s] += BKK;
            while (c[s] >= g.Cin) { c[s] -= g.Cin; ++tap[s]; }
        }
    };

    auto load_global = [&](int kc) {
        if (AL) load_aligned(kc); else load_general(kc);
    };

    auto store_lds = [&](int buf) {
        T* Ab = As + buf * BM * LDK;
        T* Bb = Bs + buf * BN * LDK;
#pragma unroll
        for (int s = 0; s < KV; ++s) {
            if (has_pro) {
#pragma unroll
                for (int i = 0; i < MS; ++i) {
                    const V o = bn_act8(ra[s][i], ps0[s], ps1[s], pt0[s], pt1[s], a.pro_slope);     // LeakyReLU / ReLU for slope in [0,1]
                    ra[s][i] = oka[s][i] ? o : zero;
                }
            }
#pragma unroll
            for (int i = 0; i < MS; ++i)
                *reinterpret_cast<V*>(Ab + (lrow + 64 * i) * LDK + BK * s + 8 * v) = ra[s][i];
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                const int nb = lrow + 64 * i;
                if (nb < BN) *reinterpret_cast<V*>(Bb + nb * LDK + BK * s + 8 * v) = rb[s][i];
            }
        }
    };

    f32x4 acc[NT][MS];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    if (nk > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_global(kc + 1);   // in flight while this chunk's MFMAs run
        const T* Ab = As + buf * BM * LDK + (16 * MS * wave + fr) * LDK + 8 * fq;
        const T* Bb = Bs + buf * BN * LDK + fr * LDK + 8 * fq;
#pragma unroll
        for (int s = 0; s < KV; ++s) {
            V af[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) af[j] = *reinterpret_cast<const V*>(Ab + 16 * j * LDK + BK * s);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const V wf = *reinterpret_cast<const V*>(Bb + 16 * i * LDK + BK * s);
#pragma unroll
                for (int j = 0; j < MS; ++j) mma32(acc[i][j], wf, af[j]);
            }
        }
        if (kc + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    if (MS == 4) {          // (the loop ended with a barrier: the A buffers are free)
        for (int i = tid; i < 2 * BN; i += 256) ssum[i] = 0.0;
        __syncthreads();
    }
    int64_t obase[MS];
    bool oval[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        const int m = mt * BM + 16 * MS * wave + 16 * ms + fr;
        oval[ms] = m < M;
        const int mm = oval[ms] ? m : 0;
        const int b = mm / HWq;
        const int r = mm - b * HWq;
        const int qy = r / g.Wq;
        const int qx = r - qy * g.Wq;
        obase[ms] = ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + qx * g.osx + P.oox) * g.ldo;
    }
    // (the k loop ended with a barrier: the operand buffers are free for the epilogue's constants)
    gemm_epilogue<T, NT, MS>(acc, obase, oval, n0, g.N, a, ssum, reinterpret_cast<float*>(smem) + (MS == 4 ? 4 * BN : 0));
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for geometries WITHOUT a load prologue (data gradients, layers that read a plain tensor), bf16, Cin a
// multiple of 32.  The SQ counters showed the register-staged loop issue-bound on its non-MFMA instructions (9 VALU per
// MFMA: address arithmetic, selects, LDS stores of both operands); with nothing to transform, both operands can go
// global -> LDS directly (global_load_lds_dwordx4): one address add per 16 bytes, no staging registers, no ds_write.
//   * LDS rows are 64 bytes (one 32-deep k chunk), lane-linear as the DMA requires; the 16-byte k-quarter of a row sits in
//     slot q ^ ((row >> 2) & 3) -- applied on the DMA SOURCE address -- which makes the ds_read_b128 fragment reads
//     conflict-free (rows r, r+4, r+8, r+12 share a bank group and get distinct quarters);
//   * rows whose tap falls outside the image (and channel rows beyond N) read 16 bytes of a zero page instead;
//   * three stages: the DMAs of chunks k+1 and k+2 are in flight during the MFMAs of chunk k; every vector-memory
//     instruction of the loop is an LDS-DMA, so the counted `s_waitcnt vmcnt` in front of the barrier counts a single kind.
__device__ __attribute__((aligned(256))) const unsigned char sv_zero_page[256] = {0};

typedef __attribute__((address_space(3))) void* ig_lds_ptr;
typedef const __attribute__((address_space(1))) void* ig_glb_ptr;

template <int NT, int MS>
__global__ __launch_bounds__(256, 2) void igemm_dma_kernel(const sv_geom g, const sv_igemm_args_g A) {
    const sv_igemm_args& a = A.g[blockIdx.y];
    sv_start_signal(a);
    typedef bf16 T;
    typedef bf16x8 V;
    constexpr int BM = 64 * MS, BN = 16 * NT;
    constexpr int NBV = (BN + 63) / 64;              // 64-row passes over the weight tile
    constexpr int STAGE = (BM + BN) * 64;            // bytes per stage: [BM][32] + [BN][32] bf16; three stages

    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ssum = reinterpret_cast<double*>(smem);  // [2][BN] over the operand buffers, after the k loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HWq = g.Hq * g.Wq;
    const int M = g.B * HWq;
    const int nMt = (M + BM - 1) / BM;
    const int nNt = (g.N + BN - 1) / BN;
    const int inner = nNt * g.nphase;
    const int L = blockIdx.x;
    int in_i, mt;
    if (nMt >= 64) {
        const int xcd = L & 7, slot = L >> 3;
        in_i = slot % inner;
        mt = (slot / inner) * 8 + xcd;
        if (mt >= nMt) return;
    } else {
        in_i = L % inner;
        mt = L / inner;
    }
    const int ph = in_i / nNt;
    const int n0 = (in_i % nNt) * BN;
    const sv_phase& P = g.phase[ph];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const int ntap = P.ntap;
    const int Ktot = ntap * g.Cin;
    const int nk = Ktot / 32;
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ W = reinterpret_cast<const T*>(a.w) + P.w_off;

    if (ntap == 0 && a.sparse_out) return;
    if (ntap == 0 && !a.bias && !a.residual && g.ldo % 8 == 0) {          // (see igemm_kernel)
        constexpr int VR = BN / 8;
        V z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
        T* __restrict__ O = reinterpret_cast<T*>(a.out);
        for (int idx = tid; idx < BM * VR; idx += 256) {
            const int row = idx / VR, vv = idx - row * VR;
            const int m = mt * BM + row;
            if (m >= M || n0 + 8 * vv >= g.N) continue;
            const int b = m / HWq, r = m - b * HWq, qy = r / g.Wq, qx = r - qy * g.Wq;
            *reinterpret_cast<V*>(O + ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + qx * g.osx + P.oox) * g.ldo +
                                  n0 + 8 * vv) = z;
        }
        return;
    }

    // ---- DMA slots: lane (row lrow + 64 i, k-quarter slot v) --------------------------------------------------------
    const int v = tid & 3, lrow = tid >> 2;
    const int vs = v ^ ((lrow >> 2) & 3);            // the k-quarter this slot holds
    int iy0[MS], ix0[MS], pixb[MS];
    bool mval[MS];
#pragma unroll
    for (int i = 0; i < MS; ++i) {
        const int m = mt * BM + lrow + 64 * i;
        mval[i] = m < M;
        const int mm = mval[i] ? m : 0;
        const int b = mm / HWq;
        const int r = mm - b * HWq;
        const int qy = r / g.Wq;
        const int qx = r - qy * g.Wq;
        iy0[i] = qy * g.sy;
        ix0[i] = qx * g.sx;
        pixb[i] = b * g.Hin * g.Win;
    }
    const T* const zp = reinterpret_cast<const T*>(sv_zero_page);
    const T* pa[MS];            // source of this lane's 16 bytes of the current tap (channel 0), or the zero page
    int astep[MS];              // elements to advance per 32-channel chunk (0 on the zero page)
    const T* pw[NBV];
    int wstep[NBV];
    auto set_tap = [&](int t) {
        const int tt = t < SV_MAX_TAPS ? t : SV_MAX_TAPS - 1;
        const int dy = tap_off(pdy, tt), dx = tap_off(pdx, tt);
#pragma unroll
        for (int i = 0; i < MS; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = mval[i] && t < ntap && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
            pa[i] = ok ? X + ((int64_t)(pixb[i] + iy * g.Win + ix) * g.ldx + 8 * vs) : zp;
            astep[i] = ok ? 32 : 0;
        }
    };
    set_tap(0);
#pragma unroll
    for (int i = 0; i < NBV; ++i) {
        const int nb = lrow + 64 * i;
        const bool ok = nb < BN && n0 + nb < g.N;
        pw[i] = ok ? W + (int64_t)(n0 + nb) * Ktot + 8 * vs : zp;
        wstep[i] = ok ? 32 : 0;
    }
    int cu = 0, tapu = 0;       // block-uniform: first channel / tap of the next chunk to issue
    auto issue = [&](int stage) {
        char* const sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < MS; ++i) {
            __builtin_amdgcn_global_load_lds((ig_glb_ptr)pa[i], (ig_lds_ptr)(sb + (64 * i + 16 * wave) * 64), 16, 0, 0);
            pa[i] += astep[i];
        }
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            if (64 * i + 16 * wave < BN) {           // (wave-uniform: rows beyond the tile have no LDS)
                __builtin_amdgcn_global_load_lds((ig_glb_ptr)pw[i], (ig_lds_ptr)(sb + BM * 64 + (64 * i + 16 * wave) * 64), 16, 0, 0);
                pw[i] += wstep[i];
            }
        }
        cu += 32;
        if (cu >= g.Cin) {      // (uniform) the next chunk starts the next tap
            cu = 0;
            ++tapu;
            set_tap(tapu);
        }
    };

    f32x4 acc[NT][MS];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int qa = fq ^ ((fr >> 2) & 3);             // slot of k-quarter fq in rows fr, fr + 16, ...
    // three stages: chunks k+1 and k+2 are in flight during the MFMAs of chunk k.  The wait in front of the barrier lets
    // the DMAs of chunk k+2 stay outstanding: a wave issues MS + (weight passes it takes part in) of them per chunk, at
    // least DMIN -- vmcnt(DMIN) therefore covers chunk k+1 for every wave (all of them LDS-DMA: one kind).
    constexpr int DMIN = MS + (BN / 64);
    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMIN) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int st = 0;                 // stage of chunk kc
    for (int kc = 0; kc < nk; ++kc) {
        const int st2 = st >= 1 ? st - 1 : 2;       // (st + 2) % 3: the stage chunk kc - 1 has released
        if (kc + 2 < nk) issue(st2);
        const char* const sb = smem + st * STAGE;
        const char* Ab = sb + (16 * MS * (tid >> 6) + fr) * 64 + 16 * qa;
        const char* Bb = sb + BM * 64 + fr * 64 + 16 * qa;
        V af[MS];
#pragma unroll
        for (int j = 0; j < MS; ++j) af[j] = *reinterpret_cast<const V*>(Ab + 16 * j * 64);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const V wf = *reinterpret_cast<const V*>(Bb + 16 * i * 64);
#pragma unroll
            for (int j = 0; j < MS; ++j) mma32(acc[i][j], wf, af[j]);
        }
        if (kc + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMIN) : "memory");     // chunk kc + 1 has landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        st = st == 2 ? 0 : st + 1;
    }

    // ---- epilogue -----------------------------------------------------------------------------------------------
    for (int i = tid; i < 2 * BN; i += 256) ssum[i] = 0.0;
    __syncthreads();
    int64_t obase[MS];
    bool oval[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        const int m = mt * BM + 16 * MS * (tid >> 6) + 16 * ms + fr;
        oval[ms] = m < M;
        const int mm = oval[ms] ? m : 0;
        const int b = mm / HWq;
        const int r = mm - b * HWq;
        const int qy = r / g.Wq;
        const int qx = r - qy * g.Wq;
        obase[ms] = ((int64_t)(b * g.Hout + qy * g.osy + P.ooy) * g.Wout + qx * g.osx + P.oox) * g.ldo;
    }
    gemm_epilogue<T, NT, MS>(acc, obase, oval, n0, g.N, a, ssum, reinterpret_cast<float*>(ssum + 2 * BN));
}

template <int NT, int MS>
int launch_dma(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int BM = 64 * MS, BN = 16 * NT;
    const int M = g->B * g->Hq * g->Wq;
    const int nMt = (M + BM - 1) / BM;
    const int nNt = (g->N + BN - 1) / BN;
    const int grid = (nMt >= 64 ? ((nMt + 7) / 8) * 8 : nMt) * nNt * g->nphase;
    const size_t lds = (size_t)3 * (BM + BN) * 64;
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_dma_kernel<NT, MS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(igemm_dma)");
        optin = true;
    }
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((igemm_dma_kernel<NT, MS>), dim3(grid, sv_ngroups(a->groups)), dim3(256), lds, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(dma)");
}

template <typename T, int NT, int KV, int MS, bool AL>
int launch_al(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int BM = 64 * MS, BN = 16 * NT, LDK = BK * KV + 16;
    const int M = g->B * g->Hq * g->Wq;
    const int nMt = (M + BM - 1) / BM;
    const int nNt = (g->N + BN - 1) / BN;
    const int grid = (nMt >= 64 ? ((nMt + 7) / 8) * 8 : nMt) * nNt * g->nphase;
    const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(T) + (MS == 4 ? 0 : 2 * BN * sizeof(double));
    static bool optin = false;
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, NT, KV, MS, AL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(igemm)");
        optin = true;
    }
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((igemm_kernel<T, NT, KV, MS, AL>), dim3(grid, sv_ngroups(a->groups)), dim3(256), lds, s, *g, sv_expand_groups(*g, *a, (int)sizeof(T)));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm");
}

template <typename T, int NT, int KV, int MS = 2>
int launch_kv(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    if (sizeof(T) == 2 && g->Cin % (BK * KV) == 0 && !sv_disabled(SV_K_IGEMM_ALIGNED)) return launch_al<T, NT, KV, MS, true>(g, a, s);
    return launch_al<T, NT, KV, MS, false>(g, a, s);
}

template <typename T, int NT>
int launch(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    // two sub-chunks per iteration when K is deep (>= 8 chunks in the shallowest phase) -- bf16 only (LDS budget)
    if (sizeof(T) == 2 && NT <= 5 && !sv_disabled(SV_K_IGEMM_KV2)) {
        int kmin = 1 << 30;
        for (int p = 0; p < g->nphase; ++p) kmin = min(kmin, g->phase[p].ntap * g->Cin);
        if (kmin >= 8 * BK && kmin % (2 * BK) == 0) return launch_kv<T, NT, 2>(g, a, s);
    }
    return launch_kv<T, NT, 1>(g, a, s);
}

}  // namespace

extern "C" int sv_igemm(const sv_geom* g, int dtype, const sv_igemm_args* a_in, void* stream) {
    SV_REQUIRE(g && a_in && a_in->x && a_in->w && a_in->out, SV_E_ARG, "sv_igemm: null argument");
    sv_igemm_args a_loc = *a_in;                       // `flags` belongs to the library
    a_loc.flags = sv_det_stats() ? SV_FLAG_DET : 0;
    const sv_igemm_args* a = &a_loc;
    SV_REQUIRE(dtype == SV_F32 || dtype == SV_BF16, SV_E_ARG, "sv_igemm: bad dtype %d", dtype);
    SV_REQUIRE(g->Cin % 16 == 0 && g->N % 16 == 0 && g->ldx % 8 == 0 && g->ldo % 4 == 0, SV_E_SHAPE,
               "sv_igemm: Cin=%d N=%d must be multiples of 16 (ldx=%d ldo=%d)", g->Cin, g->N, g->ldx, g->ldo);
    SV_REQUIRE(g->nphase >= 1 && g->nphase <= SV_MAX_PHASES, SV_E_SHAPE, "sv_igemm: nphase=%d", g->nphase);
    for (int p = 0; p < g->nphase; ++p)
        SV_REQUIRE(g->phase[p].ntap >= 0 && g->phase[p].ntap <= SV_MAX_TAPS, SV_E_SHAPE, "sv_igemm: ntap");
    SV_REQUIRE(!(a->stats && a->ex), SV_E_ARG, "sv_igemm: stats and ex epilogues are exclusive");
    SV_REQUIRE(a->groups >= 0 && a->groups <= SV_MAX_GROUPS, SV_E_ARG, "sv_igemm: groups=%d (at most %d)", a->groups, SV_MAX_GROUPS);
    SV_REQUIRE(!a->ex || (a->ex_scale && a->ex_shift && a->ex_mean && a->ex_rstd && a->bsums), SV_E_ARG,
               "sv_igemm: incomplete act-backward epilogue");
    SV_REQUIRE(!a->pro_scale || a->pro_shift, SV_E_ARG, "sv_igemm: prologue shift missing");
    SV_REQUIRE(!a->pro_scale || (a->pro_slope >= 0.f && a->pro_slope <= 1.f), SV_E_ARG,
               "sv_igemm: activation slope %g outside [0, 1]", (double)a->pro_slope);
    SV_REQUIRE(!(a->stats || a->ex) || (a->replicas >= 1 && (a->replicas & (a->replicas - 1)) == 0), SV_E_ARG,
               "sv_igemm: replicas=%d must be a power of two", a->replicas);
    SV_REQUIRE(a->block_budget == 0 || a->block_budget >= 8, SV_E_ARG, "sv_igemm: block_budget=%d", a->block_budget);
    SV_REQUIRE(!a->sparse_out || (!a->bias && !a->residual), SV_E_ARG, "sv_igemm: sparse_out with a bias / residual (the skipped positions would not be zero)");
    hipStream_t s = (hipStream_t)stream;
    if (a->fold_stats) {
        SV_REQUIRE(a->pro_scale && a->pro_shift && a->fold_gamma && a->fold_beta && a->fold_mean && a->fold_rstd &&
                   a->fold_replicas >= 1 && a->fold_count > 0.f, SV_E_ARG, "sv_igemm: incomplete BatchNorm fold (fold_*)");
        if (sv_in_query()) a_loc.fold_stats = nullptr;            // (a grid query: nothing is launched, the dispatch does not depend on it)
        else sv_fold_begin(g, a, stream);
    }
    struct FoldEnd { ~FoldEnd() { sv_fold_end(); } } fold_end;
    SvBudgetScope budget_scope(a->block_budget);
    {   // stride-1 3x3 convolutions take the LDS-halo kernels (conv3x3*.hip) unless switched off (tests: generic vs special)
        int rc = 0;
        // the last decoder layer's data gradient (4x4 stride-2 convolution 16 -> 64; dconv.hip)
        if (sv_dconv_try(g, dtype, a, s, &rc)) return rc;
        // the thin stride-1 3x3 layers at 32x32 (stem, 16 -> 32 and its data gradient; thconv.hip)
        if (sv_thconv_try(g, dtype, a, s, &rc)) return rc;
        // the 1x1 shortcut forwards: B fragments straight from global memory, no LDS (pconv.hip)
        if (sv_pconv_try(g, dtype, a, s, &rc)) return rc;
        if (!sv_disabled(SV_K_CONV3X3) && sv_conv3x3_try(g, dtype, a, s, &rc)) return rc;
        // ConvTranspose2d(4, 2, 1) 128 -> 64 forward: the weights of a phase register-resident (tconv.hip)
        if (sv_tconvr_try(g, dtype, a, s, &rc)) return rc;
        // the stride-2 3x3 forward convolutions 32 -> 64 / 64 -> 128: register-resident weights, parity-split LDS image (sconv.hip)
        if (sv_sconv_try(g, dtype, a, s, &rc)) return rc;
        // the other conv-like layers with a spatial extent: LDS-halo gather-GEMM over all phases / taps (halo.hip)
        if (sv_halo_try(g, dtype, a, s, &rc)) return rc;
    }
    const int64_t M = (int64_t)g->B * g->Hq * g->Wq;
    // wide layers (N a multiple of 160 / 128) with enough rows: 256-row tiles at two blocks per CU -- the input is gathered
    // and BatchNorm-transformed once per 160 / 128 channels instead of once per 64 / 80 (measured on the WRN-28-10 stride-2
    // 3x3 layers at 4 x 256 images: forward 798 -> 517 us, data gradient 1102 -> 820 us; tools/ig_ablate.sh)
    if (dtype == SV_BF16 && !sv_disabled(SV_K_IGEMM_BIG)) {
        const int64_t mt256 = (M + 255) / 256 * g->nphase * sv_ngroups(a->groups);
        // no load prologue (data gradients): both operands by LDS-DMA
        const bool dma = !a->pro_scale && g->Cin % 32 == 0 && g->ldx % 8 == 0 && !sv_disabled(SV_K_IGEMM_DMA);
        if (dma && g->N % 160 == 0 && mt256 * (g->N / 160) >= sv_wide_min_blocks()) return launch_dma<10, 4>(g, a, s);
#if SV_IG_DMA2
        // Small products (the decoder's ConvTranspose layers and their data gradients): while the 256-row tiles give fewer
        // than two blocks per CU, 128-row tiles of the WIDEST channel extent (128 / 64 / 32) that still yields two blocks per
        // CU -- the kernel needs its second resident block to cover the k loop's latency -- else the narrowest one
        // (measured at 4 x 512 / 2 x 512 images: ConvT 1024->512 forward 43 -> 36 us, 512->256 forward 74 -> 60,
        //  512->256 data gradient 88 -> 69, 256->128 data gradient 68 -> 55, 128->64 data gradient 60 -> 44 us)
        if (dma && g->N % 32 == 0 && mt256 * ((g->N + 127) / 128) < 2 * sv_wide_min_blocks()) {
            const int64_t mt128 = (M + 127) / 128 * g->nphase * sv_ngroups(a->groups);
            const int64_t two = 2 * sv_wide_min_blocks();
            const bool ok8 = g->N % 128 == 0, ok4 = g->N % 64 == 0;
            if (mt128 * (g->N / 32) >= (sv_wide_min_blocks() + 1) / 2) {
                if (ok8 && mt128 * (g->N / 128) >= two) return launch_dma<8, 2>(g, a, s);
                if (ok4 && mt128 * (g->N / 64) >= two) return launch_dma<4, 2>(g, a, s);
                if (mt128 * (g->N / 32) >= sv_wide_min_blocks()) return launch_dma<2, 2>(g, a, s);      // at least one block per CU
                if (ok4 && mt128 * (g->N / 64) >= (sv_wide_min_blocks() + 1) / 2) return launch_dma<4, 2>(g, a, s);
                if (ok8 && mt128 * (g->N / 128) >= (sv_wide_min_blocks() + 1) / 2) return launch_dma<8, 2>(g, a, s);
            }
        }
#endif
        if (dma && g->N % 128 == 0 && mt256 * (g->N / 128) >= (sv_wide_min_blocks() + 1) / 2) return launch_dma<8, 4>(g, a, s);
        if (g->N % 160 == 0 && mt256 * (g->N / 160) >= sv_wide_min_blocks()) return launch_kv<bf16, 10, 1, 4>(g, a, s);
        // (128-channel tiles from half a block per slot: the 4x4 stride-2 data gradient of ConvT 512 -> 256, 128 tiles, 163 -> 142 us)
        // (single-phase launches WITH a load prologue want a whole block per slot: svhn_VAE's third convolution -- 4x4 stride 2, 64 -> 128
        //  at 2 048 images, 128 tiles -- 56 -> 36 us and its Linear 512 -> 2048 34 -> 24 us on the 128-row tiles below)
        if (g->N % 128 == 0 && mt256 * (g->N / 128) >= (g->nphase == 1 ? sv_wide_min_blocks() : (sv_wide_min_blocks() + 1) / 2))
            return launch_kv<bf16, 8, 1, 4>(g, a, s);
    }
    const int64_t mtiles = (M + 127) / 128 * g->nphase;
    // widest channel tile that still yields >= 2 blocks per CU; never below 32 channels unless N is
    const int N = g->N;
    int nt = 1;
    const int cand[4] = {8, 4, 2, 1};
    for (int i = 0; i < 4; ++i) {
        const int c = cand[i];
        if (N % (16 * c) != 0) continue;
        if (dtype == SV_F32 && c > 2) continue;     // LDS budget (<= 64 KiB without opt-in)
        nt = c;
        if (mtiles * (N / (16 * c)) >= SV_IG_MIN_TILES) break;
        if (c <= 2) break;
    }
    if (N % 80 == 0 && N % 64 != 0 && mtiles * (N / 80) >= 256 && dtype == SV_BF16) nt = 5;
    if (dtype == SV_BF16) {
        switch (nt) {
            case 8: return launch<bf16, 8>(g, a, s);
            case 5: return launch<bf16, 5>(g, a, s);
            case 4: return launch<bf16, 4>(g, a, s);
            case 2: return launch<bf16, 2>(g, a, s);
            default: return launch<bf16, 1>(g, a, s);
        }
    }
    switch (nt) {
        case 2: return launch<float, 2>(g, a, s);
        default: return launch<float, 1>(g, a, s);
    }
}
