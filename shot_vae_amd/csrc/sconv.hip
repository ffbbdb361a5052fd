// Stride-2 3x3 convolution forward with register-resident weights: the first convolution of WideResNet blocks 2 and 3
// (wideresnet.py:29-30 with stride 2: 32 -> 64 at 32x32 -> 16x16, 64 -> 128 at 16x16 -> 8x8), fused with the BatchNorm + LeakyReLU in
// front of it (wideresnet.py:27-28) as the load prologue and the statistics of the BatchNorm behind it (wideresnet.py:32) as the
// epilogue: the sv_igemm contract.  gfx950.  The forward counterpart of tconv.hip's data-gradient kernels, same building blocks:
//   * unit of work = a BAND of 8 output rows (a whole 8x8 image, or half of a 16x16 one): 17 input rows x 2 W pixels, 34 KB;
//   * a persistent block of eight waves = NOUT / 32 channel tiles x 8 / (NOUT / 32) pixel tiles (32 pixels: 4 rows of 8, or 2 rows
//     of 16); a wave holds its channel tile's A fragments -- [32][9 taps x CIN] -- for the block's lifetime (64 -> 128: 26 of 36 in
//     registers, 10 in a lane-linear LDS slice shared by the two waves of a channel tile; 32 -> 64: all 18 in registers), fetched
//     once through LDS in whole 128-byte lines;
//   * the band is staged once for all waves, BatchNorm + LeakyReLU applied on the way in, into an LDS image split by ROW and COLUMN
//     PARITY (a stride-2 tap reads consecutive pixels of one parity sub-image: the stride-1 access pattern, conflict-free with the
//     same row pitch / half-swap rule as tconv.hip) and into planes of 16 channels (k-step = immediate offset, tap = one per-lane
//     base register); the zero padding of the convolution is the zeroed left column / top row of the odd sub-images; two bands
//     (the next one is requested into registers before the MFMAs of this one and written behind them, one barrier per band);
//   * epilogue out of the accumulators: fp32 partial sums of y and y^2 per lane over the block's bands, 16-byte stores through
//     v_permlane32_swap, the sums to the double accumulators once per block.
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a fast path inside it (SV_K_SCONV disables).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef SV_SCONV_PD
#define SV_SCONV_PD 2           // the B fragments are requested this many k-steps ahead of their MFMA
#endif

template <int CIN, int NOUT, int W>
struct sconv_cfg {
    static constexpr int NT = NOUT / 32, MTW = 8 / NT;             // channel tiles, pixel tiles per band (one of each per wave)
    static constexpr int TROWS = 32 / W;                           // output rows per 32-pixel tile
    static constexpr int KC = CIN / 16, KS = 9 * KC;               // k-steps per tap / in all
    static constexpr int KL = KS > 24 ? KS - 26 : 0, KR = KS - KL; // A fragments in LDS / in registers
    static constexpr int PITCH = W == 8 ? 12 : 24;
    static constexpr int SUB = 9 * PITCH * 32 + (W == 8 ? 0 : 32); // a parity sub-image: 9 rows (row 0 / column 0: the halo)
    static constexpr int PLANE = 4 * SUB + (W == 8 ? 32 : 64), NPL = KC, TILE = NPL * PLANE;
    static constexpr int NTH = 512, NVEC = 17 * 128, VPT = (NVEC + NTH - 1) / NTH;      // 16-byte vectors of a band (17 rows x 2 KB)
    static constexpr int CPP = CIN / 8;                            // vectors per pixel
    static constexpr int OFF_WSUM = 2 * TILE;                      // [8 waves][2][32] floats
    static constexpr int OFF_COEF = OFF_WSUM + 8 * 2 * 32 * 4;     // [CIN] pairs {scale, shift}
    static constexpr int OFF_WLDS = OFF_COEF + CIN * 8;            // [NT][KL][64 lanes][16 B]
    static constexpr int LDS = OFF_WLDS + NT * KL * 1024;
    static_assert(W * MTW * TROWS == 8 * W && NT * MTW == 8, "eight waves: a band = 8 output rows");
    static_assert(2 * W * CPP == 128, "a row of the input = 128 vectors");
    static_assert(LDS <= 160 * 1024 && 8 * 32 * 144 <= LDS, "LDS budget (the weight staging area of the start-up lies over everything)");
    static_assert((NPL - 1) * PLANE + 4 * SUB < 65536, "plane and sub-image offsets are ds_read immediates");
    static_assert((SUB / 16) % 8 == (W == 8 ? 0 : 2) && (PLANE / 16) % 8 == (W == 8 ? 2 : 4), "staging stores: 8 lanes on 8 bank groups");
};

template <int CIN, int NOUT, int W>
__global__ __launch_bounds__(512, 1) void sconv_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    typedef sconv_cfg<CIN, NOUT, W> C;
    constexpr int NT = C::NT, TROWS = C::TROWS, KC = C::KC, KS = C::KS, KL = C::KL, KR = C::KR, PITCH = C::PITCH, SUB = C::SUB;
    constexpr int PLANE = C::PLANE, TILE = C::TILE, NTH = C::NTH, VPT = C::VPT, CPP = C::CPP;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave % NT, mt = wave / NT;                     // this wave's channel tile and pixel tile of the band
    const int q = lane & 31, h = lane >> 5;
    const int ty = q / W, tx = q % W;                              // pixel of the tile
    const sv_phase& P = g.phase[0];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    const int bpi = g.Hout / 8;                                    // bands per image
    const int nband = g.B * bpi;
    int band = blockIdx.x;

    // ---- a band's vectors: v = tid + 512 i is vector v of the 17 input rows 16 b - 1 .. 16 b + 15 (2 KB each, contiguous)
    bf16x8 xr[VPT];
    auto request = [&](int bd) __attribute__((always_inline)) {
        const int im = bd / bpi, b = bd - im * bpi;
        const bf16* const xi = X + ((int64_t)im * g.Hin + 16 * b - 1) * (2 * W * CIN);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = tid + NTH * i;
            const bool ok = v < C::NVEC && (b > 0 || v >= 128);      // (row -1 of the image is padding)
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            xr[i] = ok ? *reinterpret_cast<const bf16x8*>(xi + v * 8) : z;
        }
    };
    if (band < nband) request(band);

    // ---- weights: A fragments of channel tile nt (row = channel 32 nt + q, k = 16 ks + 8 h ..), through LDS in whole lines
    bf16x8 wf[KR], wtail[KL > 0 ? KL : 1];
    char* const wlds = smem + C::OFF_WLDS + nt * (KL * 1024) + lane * 16;
    {
        constexpr int ROWB = 9 * CIN * 2;                          // bytes of a weight row
        constexpr int PASSB = 2 * CIN, KPP = KC, NPASS = 9;        // a pass = one tap of every row (128 / 64 bytes)
        constexpr int VPR = PASSB / 16, RPI = 64 / VPR, NI = 32 / RPI, WPITCH = PASSB + 16;       // (rows 144 / 80 B apart: row = lane is conflict-free)
        const char* const Wb = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(a.w) + P.w_off) + (32 * nt) * ROWB;
        char* const wst = smem + wave * (32 * 144);
        const int vrow = lane / VPR, vcol = lane % VPR;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            bf16x8 tmp[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) tmp[i] = *reinterpret_cast<const bf16x8*>(Wb + (vrow + RPI * i) * ROWB + PASSB * pass + 16 * vcol);
#pragma unroll
            for (int i = 0; i < NI; ++i) *reinterpret_cast<bf16x8*>(wst + (vrow + RPI * i) * WPITCH + 16 * vcol) = tmp[i];
#pragma unroll
            for (int j = 0; j < KPP; ++j) {
                const bf16x8 f = *reinterpret_cast<const bf16x8*>(wst + q * WPITCH + (2 * j + h) * 16);
                const int ks = KPP * pass + j;
                if (ks < KR) wf[ks < KR ? ks : 0] = f;
                else wtail[ks >= KR ? ks - KR : 0] = f;
            }
        }
    }
    const bool has_pro = a.pro_scale != nullptr;
    float* const coef = reinterpret_cast<float*>(smem + C::OFF_COEF);
    const float slope = has_pro ? a.pro_slope : 1.f;
    __syncthreads();                              // every wave is done with the weight staging area (it lies over what follows)
    if (KL > 0 && mt == 0) {
#pragma unroll
        for (int j = 0; j < KL; ++j) *reinterpret_cast<bf16x8*>(wlds + j * 1024) = wtail[j];
    }
    // (BatchNorm finalisation folded into this launch -- sv_igemm_args::fold_*: every block derives the coefficients itself, block 0
    //  of a group stores the four vectors; the scratch lies in the image area, zeroed below)
    if (a.fold_stats) sv_bn_fold_block512(a, CIN, reinterpret_cast<double*>(smem), coef, blockIdx.x == 0);
    else if (has_pro && tid < 2 * CIN) coef[tid] = (tid & 1) ? a.pro_shift[tid >> 1] : a.pro_scale[tid >> 1];
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * TILE / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    // staging: vector i of this thread = input row r = (tid >> 7) + 4 i (same parity for every i), pixel / chunk from tid & 127
    const int sc = tid & (CPP - 1), six = (tid & 127) / CPP;
    int sdst;
    {
        const int r = tid >> 7, rowidx = (r + 1) >> 1, pr = (r & 1) ^ 1, pc = six & 1, colidx = (six >> 1) + 1;
        sdst = (sc >> 1) * PLANE + (2 * pr + pc) * SUB + (rowidx * PITCH + colidx) * 32 + (((sc ^ rowidx) & 1) << 4);
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
        f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
        if (has_pro) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(coef + 16 * sc + 4 * j);        // channels 8 sc + 2 j, + 1
                (j < 2 ? s0 : s1)[2 * (j & 1)] = c[0]; (j < 2 ? t0 : t1)[2 * (j & 1)] = c[1];
                (j < 2 ? s0 : s1)[2 * (j & 1) + 1] = c[2]; (j < 2 ? t0 : t1)[2 * (j & 1) + 1] = c[3];
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = tid + NTH * i;
            // (row -1 of the image stays zero: the padding is not transformed; vector i is 2 sub-image rows below vector i - 1)
            if (v < C::NVEC)
                *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (2 * PITCH * 32)) =
                    has_pro ? bn_act8(xr[i], s0, s1, t0, t1, slope) : xr[i];
        }
    };
    // the padding row: zero vectors pass through the prologue as act(shift) -- put back to zero below (band 0 of an image only)
    // B fragments: output pixel (TROWS mt + ty, tx) at tap t reads input (2 y + dy, 2 x + dx) = sub-image (dy & 1, dx & 1),
    // row y + (dy >= 0), column x + (dx >= 0) (both stored one further: the halo), channels 16 kc + 8 h ..
    int rb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = P.dy[t], dx = P.dx[t];
        const int yy = TROWS * mt + ty + (dy >= 0 ? 1 : 0), xx = tx + (dx >= 0 ? 1 : 0);
        rb[t] = (2 * (dy & 1) + (dx & 1)) * SUB + (yy * PITCH + xx) * 32 + (((h ^ yy) & 1) << 4);
    }
    const int opix = ((TROWS * mt + ty) * g.Wout + tx) * g.ldo + 32 * nt + 8 * h;
    const bool want_stats = a.stats != nullptr;
    float ps1[16], ps2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ps1[e] = ps2[e] = 0.f;

    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the weights are here (no counted waits for them inside the loop)
    __syncthreads();
    // (the padded row of band 0: act(0 * scale + shift) != 0 -- its vectors are forced to zero at the request, and the stage
    //  must not transform them: handled by staging raw zeros for them)
    const bool pad_thread = tid < 128;             // vector 0 of these threads is row -1
    auto stage_band = [&](int buf, int bd) __attribute__((always_inline)) {
        stage(buf);
        if (pad_thread && (bd % bpi) == 0) {
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst) = z;
        }
    };
    if (band < nband) stage_band(0, band);
    __syncthreads();

    f32x16 acc;
    auto body = [&](auto bufc, int bd) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        const int nxt = bd + gridDim.x;
        const bool has_next = nxt < nband;
        if (has_next) request(nxt);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        constexpr int PD = SV_SCONV_PD, NB = PD + 1;
        bf16x8 bfr[NB], afr[NB];
        int rbb[9];                                // (the image offset does not fit the 16-bit immediate beside the plane offset)
#pragma unroll
        for (int t = 0; t < 9; ++t) rbb[t] = rb[t] + BUF * TILE;
        auto fetch = [&](int ks, bf16x8& dst, bf16x8& adst) __attribute__((always_inline)) {
            const int t = ks / KC, kc = ks % KC;
            dst = *reinterpret_cast<const bf16x8*>(smem + rbb[t] + kc * PLANE);
            if (ks >= KR) adst = *reinterpret_cast<const bf16x8*>(wlds + (ks - KR) * 1024);
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) fetch(d, bfr[d % NB], afr[d % NB]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PD < KS) fetch(ks + PD, bfr[(ks + PD) % NB], afr[(ks + PD) % NB]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks < KR ? wf[ks < KR ? ks : 0] : afr[ks % NB], bfr[ks % NB], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: acc[4 gq + e] = channel 32 nt + 8 gq + 4 h + e of pixel q of this wave's tile
        {
            const int im = bd / bpi, b = bd - im * bpi;
            bf16* const ob = O + ((int64_t)im * g.Hout + 8 * b) * g.Wout * g.ldo;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t pk[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const int e0 = 4 * (2 * gp + k) + 2 * d;
                        const float v0 = acc[e0], v1 = acc[e0 + 1];
                        if (want_stats) {
                            ps1[e0] += v0; ps2[e0] += v0 * v0;
                            ps1[e0 + 1] += v1; ps2[e0 + 1] += v1 * v1;
                        }
                        typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 pr = {(bf16)v0, (bf16)v1};
                        pk[k][d] = __builtin_bit_cast(uint32_t, pr);
                    }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto r = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                    pk[0][d] = r[0];
                    pk[1][d] = r[1];
                }
                const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                *reinterpret_cast<u32x4*>(ob + opix + 16 * gp) = o;
            }
        }
        if (has_next) stage_band(BUF ^ 1, nxt);
        __syncthreads();
    };
    {
        const int step = gridDim.x;
        while (band < nband) {
            body(std::integral_constant<int, 0>{}, band);
            band += step;
            if (band >= nband) break;
            body(std::integral_constant<int, 1>{}, band);
            band += step;
        }
    }
    // ---- statistics: 32 pixel lanes -> lanes 0 / 32, the waves of a channel tile through LDS, one double atomic per channel and block
    if (want_stats) {
        float* const wsum = reinterpret_cast<float*>(smem + C::OFF_WSUM) + wave * 64;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
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
            // tid = which * NOUT + channel; the channel's tile nt = channel / 32 is held by waves nt, nt + NT, ...
            const int which = tid / NOUT, n = tid - which * NOUT, cn = n >> 5, cl = n & 31;
            const float* const ws = reinterpret_cast<const float*>(smem + C::OFF_WSUM) + which * 32 + cl;
            float v = 0.f;
#pragma unroll
            for (int m = 0; m < C::MTW; ++m) v += ws[(cn + NT * m) * 64];
            atomicAdd(a.stats + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * NOUT + tid, (double)v);
        }
    }
}

template <int CIN, int NOUT, int W>
int launch_sconv(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    typedef sconv_cfg<CIN, NOUT, W> C;
    const int G = sv_ngroups(a->groups);
    const int nband = g->B * (g->Hout / 8);
    int per = sv_persistent_blocks() / 2 / G;          // (the budget counts two blocks per CU; this kernel is one)
    if (per < 1) per = 1;
    if (per > nband) per = nband;
    const int rounds = (nband + per - 1) / per;
    const int grid = (nband + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_kernel<CIN, NOUT, W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(sconv)");
        optin = true;
    }
    sv_igemm_args b = *a;          // this kernel folds the BatchNorm finalisation of its prologue
    if (!sv_fold_claim(b.fold_stats != nullptr)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((sconv_kernel<CIN, NOUT, W>), dim3(grid, G), dim3(C::NTH), C::LDS, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(sconv)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a stride-2 3x3 forward convolution this kernel covers.
int sv_sconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_SCONV) || dtype != SV_BF16) return 0;
    if (a->bias || a->residual || a->ex || a->sparse_out) return 0;
    if ((a->flags & SV_FLAG_DET) && a->stats) return 0;
    if (g->nphase != 1 || g->sy != 2 || g->sx != 2 || g->osy != 1 || g->osx != 1) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 9 || P.ooy != 0 || P.oox != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
    if (g->Hin != g->Win || g->Hout != g->Wout || g->Hq != g->Hout || g->Wq != g->Wout || g->Hin != 2 * g->Hout) return 0;
    if (g->ldx != g->Cin || g->ldo % 4 != 0 || (int64_t)g->B * g->Hout * g->Wout * g->ldo >= ((int64_t)1 << 31)) return 0;
    if (g->Cin == 64 && g->N == 128 && g->Hout == 8) {
        *rc = launch_sconv<64, 128, 8>(g, a, s);
        return 1;
    }
    if (g->Cin == 32 && g->N == 64 && g->Hout == 16) {
        *rc = launch_sconv<32, 64, 16>(g, a, s);
        return 1;
    }
    return 0;
}
