// Weight gradient of the THIN stride-1 3x3 layers at 32x32 (16 input channels: the stem, wideresnet.py:13-14, and the first convolution
// of block 1, 16 -> 32, wideresnet.py:29-30) with the layer's WHOLE gradient in every block.  gfx950.
// dW is tiny here ([32][9][16] = 4 608 floats): what swgrad.hip could not pay for at 73 728 -- one atomic add per element and block
// -- costs 1.2 M adds per launch.  So: a persistent block owns all of dW, stages every band of 16 rows ONCE (the input through the
// BatchNorm + LeakyReLU prologue, with its halo; dy as it is), a wave takes two rows of the band (a row's 32 pixels = the k dimension of
// one v_mfma_f32_16x16x32_bf16 step, both operands read k-major with ds_read_b64_tr_b16 as in wgrad.hip / swgrad.hip), accumulates
// [N][9 taps][16] in 36 / 72 registers over the block's bands, and at the end the eight waves meet in an LDS copy of dW (one wave
// after the other, 16-byte adds) that the block adds to the fp32 gradient.  The layers move 134 / 201 MB against 1-2 GFLOP: HBM-bound; the tap-fused
// LDS-halo kernel (hwgrad.hip) ran them at 2.2-2.3 TB/s.
// Same sv_wgrad contract: a fast path inside it (SV_K_THWGRAD disables); declines the deterministic mode.
#include "common.h"

namespace {

struct thwg_params {
    const void* x;
    const void* dy;
    const float* pro_scale;
    const float* pro_shift;
    float pro_slope;
    float* dw;
};

// NOUT output channels per block of the layer's NTOT (NTOT > NOUT: the first convolution of block 1 at width 10, 16 -> 160: five
// blocks of ONE XCD walk the same bands, each with 32 of the channels -- dy is read once, the small input five times, from L2)
template <int NOUT, int NTOT = NOUT>
struct thwg_cfg {
    static constexpr int CIN = 16, W = 32, NH = NOUT / 16;
    static constexpr int BR = 16, RPW = BR / 8;                    // rows per band (the more bytes a band has in flight the better: the loop
                                                                   // is a chain request -> stage -> barrier -> MFMA per band), rows per wave
    static constexpr int LDX = CIN + 8, LDY = NOUT + 8;            // LDS row strides (elements): 48 / 80 or 48 bytes
    static constexpr int XIMG = (BR + 2) * 34 * LDX * 2, YIMG = BR * W * LDY * 2, IMG = XIMG + YIMG;
    static constexpr int NTH = 512, XVEC = (BR + 2) * W * 2, XV = (XVEC + NTH - 1) / NTH, YVEC = BR * W * (NOUT / 8), YV = YVEC / NTH;
    static constexpr int OFF_RED = 2 * IMG, RED = NOUT * 9 * CIN * 4, LDS = OFF_RED + RED;
    static_assert(YVEC % NTH == 0 && LDS <= 160 * 1024, "staging / LDS budget");
};

// k-major fragment of v_mfma_f32_16x16x32_bf16 from a pixel-major LDS image (see swgrad.hip): a0 = the lane's address of
// (row0 + (i >> 2), col0 + 4 (i & 3)), i = lane & 15; ldb = bytes per LDS row
__device__ __forceinline__ bf16x8 thwg_frag(const char* a0, int ldb) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * ldb));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

template <int NOUT, int NTOT>
__global__ __launch_bounds__(512, 1) void thwgrad_kernel(const sv_geom g, const sv_wg_g<thwg_params> PG, const int nparts) {
    typedef thwg_cfg<NOUT, NTOT> C;
    constexpr int CIN = C::CIN, W = C::W, NH = C::NH, LDX = C::LDX, LDY = C::LDY, IMG = C::IMG, NTH = C::NTH, XV = C::XV, YV = C::YV, BR = C::BR, RPW = C::RPW;
    const thwg_params& p = PG.g[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = row of the band
    const int gq = lane >> 4, li = lane & 15;
    const sv_phase& P = g.phase[0];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
    const bf16* __restrict__ DY = reinterpret_cast<const bf16*>(p.dy);
    constexpr int BPI = W / BR;
    const int nband = g.B * BPI;
    // (see s2wgrad.hip: the nparts blocks of a band slot sit on one XCD)
    const int nslot = gridDim.x / nparts, bid = blockIdx.x;
    const bool xcd_map = (nslot & 7) == 0;
    const int part = xcd_map ? (bid >> 3) % nparts : bid % nparts, n0 = part * NOUT;
    int band = xcd_map ? (bid & 7) + 8 * (bid / (8 * nparts)) : bid / nparts;

    // ---- a band's vectors: x = 10 input rows 8 b - 1 .. 8 b + 8 (64 vectors each, contiguous), dy = 8 rows (contiguous)
    struct VS { bf16x8 x[XV], y[YV]; };
    VS S0;
    auto x_ok = [&](int b, int v) { const int r = v >> 6; return v < C::XVEC && (b > 0 || r > 0) && (b < BPI - 1 || r < BR + 1); };
    auto request = [&](int bd, VS& V) __attribute__((always_inline)) {
        const int im = bd / BPI, b = bd - im * BPI;
        const bf16* const xi = X + ((int64_t)im * W + BR * b - 1) * (W * CIN);
        const bf16* const yi = DY + ((int64_t)im * W + BR * b) * (W * NTOT) + n0;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + NTH * i;
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            V.x[i] = x_ok(b, v) ? *reinterpret_cast<const bf16x8*>(xi + v * 8) : z;
        }
#pragma unroll
        for (int i = 0; i < YV; ++i) {
            const int v = tid + NTH * i;
            V.y[i] = *reinterpret_cast<const bf16x8*>(yi + (v / (NOUT / 8)) * NTOT + 8 * (v % (NOUT / 8)));
        }
    };
    const int step = nslot;
    if (band < nband) request(band, S0);
    const bool has_pro = p.pro_scale != nullptr;
    const float slope = has_pro ? p.pro_slope : 1.f;
    // prologue coefficients of this thread's 8 channels (chunk tid & 1: the same for both of its vectors)
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (has_pro) {
        const int c0 = 8 * (tid & 1);
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0); s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0); t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 4);
    }
    for (int i = tid; i < C::LDS / 16; i += NTH) *reinterpret_cast<f32x4*>(smem + 16 * i) = f32x4{0.f, 0.f, 0.f, 0.f};      // images (halo = padding) and the dW copy
    // staging destinations: x vector i = input row (tid >> 6) + 8 i, pixel (tid & 63) >> 1, half tid & 1; dy vector i = pixel / chunk from tid + 512 i
    const int xdst = (((tid >> 6) * 34 + ((tid & 63) >> 1) + 1) * LDX + 8 * (tid & 1)) * 2;
    auto stage = [&](int buf, int bd, const VS& V) __attribute__((always_inline)) {
        char* const base = smem + buf * IMG;
        const int b = bd % BPI;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + NTH * i;
            if (v < C::XVEC) {
                // (a padding row stays zero: it is not transformed)
                const bf16x8 val = (has_pro && x_ok(b, v)) ? bn_act8(V.x[i], s0, s1, t0, t1, slope) : V.x[i];
                *reinterpret_cast<bf16x8*>(base + xdst + i * (8 * 34 * LDX * 2)) = val;        // (vector i: 8 rows further)
            }
        }
#pragma unroll
        for (int i = 0; i < YV; ++i) {
            const int v = tid + NTH * i, px = v / (NOUT / 8), ck = v % (NOUT / 8);
            *reinterpret_cast<bf16x8*>(base + C::XIMG + (px * LDY + 8 * ck) * 2) = V.y[i];
        }
    };
    // fragment addresses (byte offsets inside an image): dy -- pixels 32 wave + 8 gq + (li >> 2) .., channels 16 nh + 4 (li & 3);
    // x at tap t -- LDS row wave + dy + 1, columns 8 gq + dx + 1 + (li >> 2) ..
    const int yoff = C::XIMG + ((32 * wave + 8 * gq + (li >> 2)) * LDY + 4 * (li & 3)) * 2;       // (+ 8 rows per further row of this wave)
    int xoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) xoff[t] = (((wave + P.dy[t] + 1) * 34 + 8 * gq + P.dx[t] + 1 + (li >> 2)) * LDX + 4 * (li & 3)) * 2;
    f32x4 acc[NH][9];
#pragma unroll
    for (int a_ = 0; a_ < NH; ++a_)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[a_][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (band < nband) stage(0, band, S0);
    __syncthreads();

    {
        int buf = 0;
        for (; band < nband; band += step, buf ^= 1) {
            const int nxt = band + step;
            const bool has_next = nxt < nband;
            if (has_next) request(nxt, S0);
            const char* const IB = smem + buf * IMG;
#pragma unroll
            for (int rw = 0; rw < RPW; ++rw) {          // this wave's rows wave, wave + 8
                bf16x8 af[NH];
#pragma unroll
                for (int a_ = 0; a_ < NH; ++a_) af[a_] = thwg_frag(IB + yoff + (rw * 8 * 32 * LDY + 16 * a_) * 2, LDY * 2);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const bf16x8 bf_ = thwg_frag(IB + xoff[t] + rw * (8 * 34 * LDX * 2), LDX * 2);
#pragma unroll
                    for (int a_ = 0; a_ < NH; ++a_) mma32(acc[a_][t], af[a_], bf_);
                }
            }
            if (has_next) stage(buf ^ 1, nxt, S0);
            __syncthreads();
        }
    }
    // ---- the eight waves meet in the LDS copy of dW, the block adds it to the gradient: acc[a][t][e] = (n = 16 a + 4 gq + e, c = li).
    // One wave after the other adds its registers to the copy with 16-byte reads / writes (layout [a][gq][t][c][e]: a lane's four
    // values are contiguous).  Round 6: the LDS float atomics this replaces (36 / 72 instructions per wave, the four gq groups of an
    // instruction on the same banks) were a third to a half of the launch (stem 64.5 -> 40.4 us, 16 -> 32: 100 -> 50.8 us; k4wgrad.hip:
    // ~70 of ~100 us).  (Per-block slabs + sv_slab_reduce instead of the global atomics below: -3 / -5 us alone, nothing in the step.)
    {
        f32x4* const red4 = reinterpret_cast<f32x4*>(smem + C::OFF_RED);
        const float* const red = reinterpret_cast<const float*>(smem + C::OFF_RED);
        for (int w = 0; w < 8; ++w) {
            if (wave == w) {
#pragma unroll
                for (int a_ = 0; a_ < NH; ++a_)
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        f32x4* const q = red4 + ((a_ * 4 + gq) * 9 + t) * CIN + li;
                        f32x4 v = acc[a_][t];
                        if (w > 0) {
                            const f32x4 o = *q;
                            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                        }
                        *q = v;
                    }
            }
            __syncthreads();
        }
        const int T = g.T_orig;
        for (int i = tid; i < NOUT * 9 * CIN; i += NTH) {
            const int c = i % CIN, t = (i / CIN) % 9, n = i / (9 * CIN);
            const float v = red[((((n >> 4) * 4 + ((n >> 2) & 3)) * 9 + t) * CIN + c) * 4 + (n & 3)];
            atomicAdd(p.dw + ((size_t)(n0 + n) * T + P.torig[t]) * CIN + c, v);
        }
    }
}

template <int NOUT, int NTOT = NOUT>
int launch_thwgrad(const sv_geom* g, const thwg_params& p, int groups, hipStream_t s) {
    typedef thwg_cfg<NOUT, NTOT> C;
    constexpr int nparts = NTOT / NOUT;
    const int nband = g->B * (32 / C::BR);
    int per = sv_persistent_blocks() / 2 / groups / nparts;        // band slots: one block per CU, nparts blocks per slot
    if (per < 1) per = 1;
    if (per > nband) per = nband;
    const int rounds = (nband + per - 1) / per;
    const int slots = (nband + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&thwgrad_kernel<NOUT, NTOT>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(thwgrad)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((thwgrad_kernel<NOUT, NTOT>), dim3(slots * nparts, groups), dim3(C::NTH), C::LDS, s, *g, sv_expand_wg(*g, p, groups, 2), nparts);
    sv_prof_end(s);
    return sv_check_launch("sv_wgrad(thwgrad)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is the weight gradient of a thin stride-1 3x3 layer at 32x32 (16 input channels).
int sv_thwgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                   const void* dy, float* dw, int groups, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_THWGRAD) || dtype != SV_BF16 || sv_deterministic()) return 0;
    if (g->nphase != 1 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 9 || g->T_orig != 9) return 0;
    for (int t = 0; t < 9; ++t)
        if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
    if (g->Cin != 16 || g->ldx != 16 || g->Hin != 32 || g->Win != 32 || g->Hout != 32 || g->Wout != 32 || g->ldo != g->N) return 0;
    thwg_params p;
    p.x = x; p.dy = dy; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_slope = pro_slope; p.dw = dw;
    if (g->N == 32) { *rc = launch_thwgrad<32>(g, p, groups, s); return 1; }
    if (g->N == 16) { *rc = launch_thwgrad<16>(g, p, groups, s); return 1; }
    if (g->N == 160) { *rc = launch_thwgrad<32, 160>(g, p, groups, s); return 1; }
    return 0;
}
