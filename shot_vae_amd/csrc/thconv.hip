// The data gradient of the THIN stride-1 3x3 convolution at 32x32 (block 1's first convolution 16 -> 32, wideresnet.py:29-30: its
// data gradient 32 -> 16 with the activation-backward epilogue) with register-resident weights.  gfx950.
// (Round 5 also carried FORWARD forms of this kernel -- stem 16 -> 16 with bias, 16 -> 32 with the folded BatchNorm finalisation --:
//  as exact as the kernels they replace, but their different rounding of the first two layers moved the bf16 step's posterior term over
//  a 5e-3 gate for 19 us; they left the library in round 6, docs/lab_notes_r05.md.)
// These layers move 134-268 MB against 1-5 GFLOP: they are HBM-bound, and the LDS-halo kernels ran them at 2.4-3.8 TB/s.  sconv.hip's
// scheme at stride 1: unit of work = a band of 8 output rows (10 input rows), a persistent block of eight waves = the band's eight
// rows (a wave: one row of 32 pixels = one MFMA tile, all output channels); the few weights ([32 or 16 (padded to 32 rows)][9 taps x
// CIN]: 9 / 18 A fragments) live in registers; the band is staged once into a
// zero-bordered LDS image of 16-channel planes (k-step = immediate offset, tap = per-lane base; a pixel's two 16-byte halves are
// swapped where (column >> 3) is odd, which makes a ds_read_b128 lane group -- columns 0-3, 12-15, 20-27 of one row -- conflict-
// free for every tap shift); two bands, one barrier per band; epilogue out of the accumulators (the activation-backward form with the
// raw tensor redistributed by v_permlane32_swap), 16-byte stores.
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a fast path inside it (SV_K_THCONV disables).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int CIN, int NOUT>
struct thconv_cfg {
    static constexpr int W = 32, KC = CIN / 16, KS = 9 * KC, CPP = CIN / 8;
    static constexpr int PITCH = 34, PLANE = 10 * PITCH * 32 + 32, TILE = KC * PLANE;
    static constexpr int NTH = 512, VROW = W * CPP, NVEC = 10 * VROW, VPT = (NVEC + NTH - 1) / NTH;
    static constexpr int NG = NOUT / 16;                            // 16-byte stores per lane and pixel (16 channels per lane pair each)
    static constexpr int OFF_WSUM = 2 * TILE;                      // [8 waves][2][32] floats
    static constexpr int OFF_COEF = OFF_WSUM + 8 * 2 * 32 * 4;     // [NOUT] x {scale, shift, rstd, -mean rstd}
    static constexpr int LDS = OFF_COEF + 32 * 16;
    static_assert(NTH % VROW == 0 && LDS <= 64 * 1024, "staging / LDS");
};

template <int CIN, int NOUT>
__global__ __launch_bounds__(512, 1) void thconv_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    typedef thconv_cfg<CIN, NOUT> C;
    constexpr int W = C::W, KC = C::KC, KS = C::KS, CPP = C::CPP, PITCH = C::PITCH, PLANE = C::PLANE, TILE = C::TILE, NTH = C::NTH;
    constexpr int VROW = C::VROW, VPT = C::VPT, NG = C::NG;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = output row of the band
    const int q = lane & 31, h = lane >> 5;
    const sv_phase& P = g.phase[0];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    const bf16* __restrict__ EXP = reinterpret_cast<const bf16*>(a.ex);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    constexpr int BPI = W / 8;                                     // bands per image
    const int nband = g.B * BPI;
    int band = blockIdx.x;

    // ---- a band's vectors: v = tid + 512 i is vector v of the 10 input rows 8 b - 1 .. 8 b + 8 (contiguous)
    // (two register sets: a band's vectors are requested TWO bands ahead of their staging -- one band of this loop lasts ~1.8 us, less
    //  than a loaded HBM round trip)
    bf16x8 xrA[VPT], xrB[VPT];
    auto row_ok = [&](int b, int v) { const int r = v / VROW; return v < C::NVEC && (b > 0 || r > 0) && (b < BPI - 1 || r < 9); };
    auto request = [&](int bd, bf16x8 (&xr)[VPT]) __attribute__((always_inline)) {
        const int im = bd / BPI, b = bd - im * BPI;
        const bf16* const xi = X + ((int64_t)im * W + 8 * b - 1) * (W * CIN);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = tid + NTH * i;
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            xr[i] = row_ok(b, v) ? *reinterpret_cast<const bf16x8*>(xi + v * 8) : z;
        }
    };
    if (band < nband) request(band, xrA);
    // ---- weights: A fragments (row = output channel q -- rows >= NOUT are zero --, k = 16 ks + 8 h ..) of the packed [N][9][CIN]
    bf16x8 wf[KS];
    {
        const bf16* __restrict__ Wp = reinterpret_cast<const bf16*>(a.w) + P.w_off + (q % NOUT) * (9 * CIN) + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 f = *reinterpret_cast<const bf16x8*>(Wp + 16 * ks);
            if (NOUT < 32 && q >= NOUT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (bf16)0.f;
            }
            wf[ks] = f;
        }
    }
    float* const coef = reinterpret_cast<float*>(smem + C::OFF_COEF);
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * TILE / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    if (tid < NOUT) {
        const float rs = a.ex_rstd[tid];
        reinterpret_cast<f32x4*>(coef)[tid] = f32x4{a.ex_scale[tid], a.ex_shift[tid], rs, -a.ex_mean[tid] * rs};     // xhat = x rstd - mean rstd
    }
    // staging: vector i of this thread = input row (tid / VROW) + (NTH / VROW) i, pixel (tid % VROW) / CPP, chunk tid % CPP
    const int sc = tid % CPP;
    int sdst;
    {
        const int r = tid / VROW, xx = (tid % VROW) / CPP + 1;
        sdst = (sc >> 1) * PLANE + (r * PITCH + xx) * 32 + ((((sc & 1) ^ (xx >> 3)) & 1) << 4);
    }
    auto stage = [&](int buf, int bd, const bf16x8 (&xr)[VPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = tid + NTH * i;
            if (v < C::NVEC) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * ((NTH / VROW) * PITCH * 32)) = xr[i];
        }
    };
    // B fragments: output pixel (row wave, column q) at tap (dy, dx) reads LDS row wave + dy + 1, column q + dx + 1, channels 16 kc + 8 h ..
    int rb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int rr = wave + P.dy[t] + 1, xx = q + P.dx[t] + 1;
        rb[t] = (rr * PITCH + xx) * 32 + (((h ^ (xx >> 3)) & 1) << 4);
    }
    const int opix = (wave * W + q) * g.ldo + 8 * h;
    const float ex_slope = a.ex_slope;
    float ps1[8 * NG], ps2[8 * NG];               // this lane's NOUT / 2 channels
#pragma unroll
    for (int e = 0; e < 8 * NG; ++e) ps1[e] = ps2[e] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the weights are here (no counted waits for them inside the loop)
    __syncthreads();
    if (band < nband) stage(0, band, xrA);
    if (band + (int)gridDim.x < nband) request(band + gridDim.x, xrB);
    __syncthreads();

    // the loop body: LDS holds band `band` (image buf), the register set xnear holds band + step (requested one iteration ago), xfar is
    // requested now for band + 2 step
    const int step = gridDim.x;
    auto body = [&](int buf, bf16x8 (&xfar)[VPT], const bf16x8 (&xnear)[VPT]) __attribute__((always_inline)) {
        const int nxt = band + step, nx2 = band + 2 * step;
        const bool has_next = nxt < nband;
        if (nx2 < nband) request(nx2, xfar);
        const int im = band / BPI, b = band - im * BPI;
        const int64_t obase = ((int64_t)im * W + 8 * b) * W * g.ldo;
        u32x4 opr[NG];                        // the raw tensor at this lane's 16-byte store positions
#pragma unroll
        for (int gp = 0; gp < NG; ++gp) opr[gp] = *reinterpret_cast<const u32x4*>(EXP + obase + opix + 16 * gp);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const char* const IB = smem + buf * TILE;
        constexpr int PD = 2, NB = PD + 1;
        bf16x8 bfr[NB];
        auto fetch = [&](int ks) __attribute__((always_inline)) {
            bfr[ks % NB] = *reinterpret_cast<const bf16x8*>(IB + rb[ks / KC] + (ks % KC) * PLANE);
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) fetch(d);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PD < KS) fetch(ks + PD);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], bfr[ks % NB], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: acc[4 gq + e] = channel 8 gq + 4 h + e of pixel q (gq < NG)
#pragma unroll
        for (int gp = 0; gp < NG; ++gp) {
            uint32_t xw[2][2], ow[2][2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                // loaded: lanes 0-31 channels 16 gp + 0 .. 7, lanes 32-63 channels 16 gp + 8 .. 15; wanted: 8 gq + 4 h + e
                const auto rr = __builtin_amdgcn_permlane32_swap(opr[gp][d], opr[gp][2 + d], false, false);
                xw[0][d] = rr[0];
                xw[1][d] = rr[1];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int gq = 2 * gp + k, e0 = 4 * gq + 2 * d, cch = 8 * gq + 4 * h + 2 * d;
                    float g0 = acc[e0], g1 = acc[e0 + 1];
                    {
                        const f32x4 c0 = reinterpret_cast<const f32x4*>(coef)[cch], c1 = reinterpret_cast<const f32x4*>(coef)[cch + 1];
                        const uint32_t w = xw[k][d];
                        const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
                        g0 *= (x0 * c0[0] + c0[1] > 0.f) ? 1.f : ex_slope;
                        g1 *= (x1 * c1[0] + c1[1] > 0.f) ? 1.f : ex_slope;
                        ps1[e0] += g0;
                        ps2[e0] += g0 * (x0 * c0[2] + c0[3]);
                        ps1[e0 + 1] += g1;
                        ps2[e0 + 1] += g1 * (x1 * c1[2] + c1[3]);
                    }
                    typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    const bf16x2 pr = {(bf16)g0, (bf16)g1};
                    ow[k][d] = __builtin_bit_cast(uint32_t, pr);
                }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const auto rr = __builtin_amdgcn_permlane32_swap(ow[0][d], ow[1][d], false, false);
                ow[0][d] = rr[0];
                ow[1][d] = rr[1];
            }
            const u32x4 o = {ow[0][0], ow[0][1], ow[1][0], ow[1][1]};
            *reinterpret_cast<u32x4*>(O + obase + opix + 16 * gp) = o;
        }
        if (has_next) stage(buf ^ 1, nxt, xnear);
        __syncthreads();
    };
    while (band < nband) {
        body(0, xrA, xrB);
        band += step;
        if (band >= nband) break;
        body(1, xrB, xrA);
        band += step;
    }
    // ---- sums: 32 pixel lanes -> lanes 0 / 32, the eight waves through LDS, one double atomic per channel and block
    {
        float* const wsum = reinterpret_cast<float*>(smem + C::OFF_WSUM) + wave * 64;
#pragma unroll
        for (int e = 0; e < 8 * NG; ++e) {
            float v1 = ps1[e], v2 = ps2[e];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                v1 += __shfl_xor(v1, o);
                v2 += __shfl_xor(v2, o);
            }
            if (q == 0) {
                const int n = 8 * (e >> 2) + 4 * h + (e & 3);
                wsum[n] = v1;
                wsum[32 + n] = v2;
            }
        }
        __syncthreads();
        if (tid < 2 * NOUT) {
            const int which = tid / NOUT, n = tid - which * NOUT;
            const float* const ws = reinterpret_cast<const float*>(smem + C::OFF_WSUM) + which * 32 + n;
            float v = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) v += ws[m * 64];
            atomicAdd(a.bsums + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * NOUT + tid, (double)v);
        }
    }
}

template <int CIN, int NOUT>
int launch_thconv(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    typedef thconv_cfg<CIN, NOUT> C;
    const int G = sv_ngroups(a->groups);
    const int nband = g->B * (C::W / 8);
    int per = sv_persistent_blocks() / 2 / G;          // (the budget counts two blocks per CU; this kernel is one)
    if (per < 1) per = 1;
    if (per > nband) per = nband;
    const int rounds = (nband + per - 1) / per;
    const int grid = (nband + rounds - 1) / rounds;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((thconv_kernel<CIN, NOUT>), dim3(grid, G), dim3(C::NTH), C::LDS, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(thconv)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a thin stride-1 3x3 convolution at 32x32 this kernel covers.
int sv_thconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_THCONV) || dtype != SV_BF16) return 0;
    if (!a->ex) return 0;
    if (a->residual || a->sparse_out) return 0;
    if (a->stats || a->pro_scale || a->bias || a->fold_stats) return 0;
    if (a->flags & SV_FLAG_DET) return 0;
    if (g->nphase != 1 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 9 || P.ooy != 0 || P.oox != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
    if (g->Hin != 32 || g->Win != 32 || g->Hout != 32 || g->Wout != 32 || g->Hq != 32 || g->Wq != 32) return 0;
    if (g->ldx != g->Cin || g->ldo % 8 != 0 || (int64_t)g->B * 1024 * g->ldo >= ((int64_t)1 << 31)) return 0;
    if (g->Cin == 32 && g->N == 16) { *rc = launch_thconv<32, 16>(g, a, s); return 1; }
    return 0;
}
