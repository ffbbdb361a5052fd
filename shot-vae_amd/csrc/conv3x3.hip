// Stride-1 3x3 convolution (forward and data gradient) with an LDS-resident halo tile.  gfx950.
//
// The generic gather-GEMM (igemm.hip) re-gathers the input once per tap, i.e. moves ~9x the
// algorithmic bytes through L1/L2 -- measured L2-bound on the WRN body convs.  Here a block owns a
// tile of 128 output pixels (TR = 128/W whole image rows), stages the (TR+2) x (W+2) input halo ONCE
// per 32-channel chunk -- BatchNorm-apply + LeakyReLU applied once per element on the way in, zero
// padding written as zeros -- together with the [BN][9][32] weight chunk, and then runs all nine taps
// x BN channels on MFMA straight out of LDS (the tap shift is just an LDS address offset).
// Same sv_geom / packed weights / fused epilogue as sv_igemm: it is a drop-in fast path inside it.
#include "common.h"
#include "epilogue.h"

namespace {

constexpr int CK = 32;          // channel chunk = one MFMA k step
constexpr int LDC = CK + 8;     // LDS pixel / weight-row stride (elements): 80 B (bf16) keeps b128 reads conflict-free

template <typename T, int NT, int WLOG>
__global__ __launch_bounds__(256) void conv3x3_kernel(const sv_geom g, const sv_igemm_args a) {
    typedef typename V8<T>::type V;
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int HP = (TR + 2) * WP;           // halo pixels
    constexpr int BN = 16 * NT;
    constexpr int HV = HP * (CK / 8), WV = BN * 9 * (CK / 8);
    constexpr int HI = (HV + 255) / 256, WI = (WV + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* halo = reinterpret_cast<T*>(smem);             // [HP][LDC]
    T* wl = halo + HP * LDC;                          // [BN*9][LDC]
    float* ssum = reinterpret_cast<float*>(wl + BN * 9 * LDC);   // [2][BN]

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

    if (tid < 2 * BN) ssum[tid] = 0.f;

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

template <typename T, int NT, int WLOG>
int launch(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W, BN = 16 * NT;
    const int nT = g->B * g->Hin / TR;
    const int nNt = g->N / BN;
    const int grid = (nT >= 64 ? ((nT + 7) / 8) * 8 : nT) * nNt;
    const size_t lds = (size_t)((TR + 2) * (W + 2) + BN * 9) * LDC * sizeof(T) + 2 * BN * sizeof(float);
    static bool optin = false;          // > 64 KiB of dynamic LDS needs an opt-in (gfx950 has 160 KiB per CU)
    if (lds > 64 * 1024 && !optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, NT, WLOG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3_kernel<T, NT, WLOG>), dim3(grid), dim3(256), lds, s, *g, *a);
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

// Returns 1 and sets *rc when the geometry is a stride-1 3x3 convolution this kernel covers.
int sv_conv3x3_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (g->nphase != 1 || g->phase[0].ntap != 9 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1) return 0;
    if (g->Hq != g->Hin || g->Wq != g->Win || g->Hout != g->Hin || g->Wout != g->Win || g->Hin != g->Win) return 0;
    if (g->Win != 8 && g->Win != 16 && g->Win != 32) return 0;
    if (g->Cin % CK != 0 || g->ldx != g->Cin || g->N % 32 != 0) return 0;
    if (g->phase[0].ooy != 0 || g->phase[0].oox != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (g->phase[0].dy[t] < -1 || g->phase[0].dy[t] > 1 || g->phase[0].dx[t] < -1 || g->phase[0].dx[t] > 1) return 0;
    const int TR = 128 / g->Win;
    if ((g->B * g->Hin) % TR != 0) return 0;
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
