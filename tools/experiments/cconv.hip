// Stride-1 3x3 convolution 64 -> 64 at 16x16 with register-resident weights: the body of WideResNet block 2 (wideresnet.py:29-35,
// 46-49 -- conv1 / conv2 of every unit, forward and data gradient), fused like every sv_igemm launch: BatchNorm + LeakyReLU load
// prologue, residual add, statistics of the next BatchNorm (forward); activation-backward epilogue with the two BatchNorm-backward
// sums (data gradient).  gfx950.  The scheme of tconv.hip / sconv.hip at stride 1:
//   * a persistent block of eight waves = 2 channel tiles x 4 row groups of the image (a wave: 32 channels x 4 rows of 16 pixels
//     = two 32-pixel tiles that share every A fragment); the wave holds its channel tile's A fragments [32][9 taps x 64] for the
//     block's lifetime -- 22 of 36 in registers, 14 in a lane-linear LDS slice shared by the four waves of the tile -- fetched once,
//     through LDS, in whole lines;
//   * the image (16 x 16 x 64, 32 KB) is staged once for all waves into a zero-bordered LDS image of four 16-channel planes
//     (k-step = immediate offset, tap = one per-lane base register; rows 24 pixels apart + half swap on odd rows: conflict-free
//     ds_read_b128 for every tap), BatchNorm + LeakyReLU applied on the way in; two images, one barrier per image;
//   * per image a wave runs 36 k-steps x 2 tiles = 72 v_mfma_f32_32x32x16_bf16 on 72 + 14 ds_read_b128 -- one fragment read per
//     32-cycle MFMA where the LDS-resident-weight kernel (conv3x3p: 16x16x32 MFMAs, pixel AND weight fragments from LDS) reads one
//     per 16-cycle MFMA and sits on its LDS floor ("where a 64-channel tile's time goes");
//   * epilogue out of the accumulators; the residual / raw-tensor operand is requested at the start of the image interval with the
//     store's own addressing (16 bytes per lane) and handed to the accumulator layout by v_permlane32_swap, as are the stores.
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a fast path inside it (SV_K_CCONV disables).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef SV_CCONV_PD
#define SV_CCONV_PD 2           // the B fragments are requested this many k-steps ahead of their MFMAs
#endif
#ifndef SV_CCONV_KL
#define SV_CCONV_KL 14          // A fragments kept in LDS instead of registers (forward)
#endif
#ifndef SV_CCONV_KL_EX
#define SV_CCONV_KL_EX 22       // ... data gradient (its epilogue holds the raw-tensor operand and four constants per channel)
#endif
#ifndef SV_CCONV_PD_EX
#define SV_CCONV_PD_EX 1
#endif

template <bool EX>
struct cconv_cfg {
    static constexpr int CIN = 64, NOUT = 64, H = 16, W = 16;
    static constexpr int NT = 2, KC = CIN / 16, KS = 9 * KC, KL = EX ? SV_CCONV_KL_EX : SV_CCONV_KL, KR = KS - KL;
    static constexpr int PITCH = 24, PLANE = (H + 2) * PITCH * 32 + 32, NPL = KC, TILE = NPL * PLANE;
    static constexpr int NTH = 512, VPT = H * W * (CIN / 8) / NTH;
    static constexpr int OFF_WSUM = 2 * TILE;                      // [8 waves][2][32] floats
    static constexpr int OFF_COEF = OFF_WSUM + 8 * 2 * 32 * 4;     // [CIN] pairs {scale, shift}  /  (EX) [NOUT] x {scale, shift, rstd, -mean rstd}
    static constexpr int OFF_WLDS = OFF_COEF + NOUT * 16;          // [NT][KL][64 lanes][16 B]
    static constexpr int LDS = OFF_WLDS + NT * KL * 1024;
    static_assert(VPT == 4 && LDS <= 160 * 1024 && 8 * 32 * 144 <= LDS, "staging / LDS budget");
    static_assert((NPL - 1) * PLANE + 2 * PITCH * 32 + (H + 2) * PITCH * 32 < 65536, "plane and tile offsets are ds_read immediates");
    static_assert((PLANE / 16) % 8 == 2, "staging stores: the 8 chunks of a pixel on 8 bank groups");
};

template <bool EX>
__global__ __launch_bounds__(512, 1) void cconv_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    typedef cconv_cfg<EX> C;
    constexpr int CIN = C::CIN, NOUT = C::NOUT, NT = C::NT, KC = C::KC, KS = C::KS, KL = C::KL, KR = C::KR, PITCH = C::PITCH;
    constexpr int PLANE = C::PLANE, TILE = C::TILE, NTH = C::NTH, VPT = C::VPT;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1, tp = wave >> 1;                       // channel tile; rows 4 tp .. 4 tp + 3 of the image
    const int q = lane & 31, h = lane >> 5, r = q >> 4, x = q & 15;
    const sv_phase& P = g.phase[0];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    const bf16* __restrict__ OPD = reinterpret_cast<const bf16*>(EX ? a.ex : a.residual);     // epilogue operand at the output positions
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    const int nimg = g.B;
    int img = blockIdx.x;

    bf16x8 xr[VPT];
    auto request = [&](int im) __attribute__((always_inline)) {
        const bf16* const xi = X + (int64_t)im * (C::H * C::W * CIN);
#pragma unroll
        for (int i = 0; i < VPT; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(xi + tid * 8 + i * (NTH * 8));
    };
    if (img < nimg) request(img);

    // ---- weights: A fragments of channel tile nt (row = channel 32 nt + q, k = 16 ks + 8 h ..), through LDS in whole lines
    bf16x8 wf[KR], wtail[KL];
    char* const wlds = smem + C::OFF_WLDS + nt * (KL * 1024) + lane * 16;
    {
        constexpr int ROWB = 9 * CIN * 2, PASSB = 2 * CIN;          // a pass = one tap of every row (128 bytes)
        const char* const Wb = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(a.w) + P.w_off) + (32 * nt) * ROWB;
        char* const wst = smem + wave * (32 * 144);
        const int vrow = lane >> 3, vcol = lane & 7;
#pragma unroll
        for (int pass = 0; pass < 9; ++pass) {
            bf16x8 tmp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) tmp[i] = *reinterpret_cast<const bf16x8*>(Wb + (vrow + 8 * i) * ROWB + PASSB * pass + 16 * vcol);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8*>(wst + (vrow + 8 * i) * 144 + 16 * vcol) = tmp[i];
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const bf16x8 f = *reinterpret_cast<const bf16x8*>(wst + q * 144 + (2 * j + h) * 16);
                const int ks = KC * pass + j;
                if (ks < KR) wf[ks < KR ? ks : 0] = f;
                else wtail[ks >= KR ? ks - KR : 0] = f;
            }
        }
    }
    const bool has_pro = !EX && a.pro_scale != nullptr;
    float* const coef = reinterpret_cast<float*>(smem + C::OFF_COEF);
    const float slope = has_pro ? a.pro_slope : 1.f;
    __syncthreads();                              // every wave is done with the weight staging area (it lies over what follows)
    if (tp == 0) {
#pragma unroll
        for (int j = 0; j < KL; ++j) *reinterpret_cast<bf16x8*>(wlds + j * 1024) = wtail[j];
    }
    // (BatchNorm finalisation folded into this launch -- sv_igemm_args::fold_*; it is not what the forward form loses in the step:
    //  +0.32 ms with the fold, +0.42 without)
    if (!EX && a.fold_stats) sv_bn_fold_block512(a, CIN, reinterpret_cast<double*>(smem), coef, blockIdx.x == 0);
    else if (has_pro && tid < 2 * CIN) coef[tid] = (tid & 1) ? a.pro_shift[tid >> 1] : a.pro_scale[tid >> 1];
    if (EX && tid < NOUT) {
        const float rs = a.ex_rstd[tid];
        reinterpret_cast<f32x4*>(coef)[tid] = f32x4{a.ex_scale[tid], a.ex_shift[tid], rs, -a.ex_mean[tid] * rs};     // xhat = x rstd - mean rstd
    }
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * TILE / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    // staging: vector i of this thread = pixel (tid >> 3) + 64 i (4 rows further: same row parity), chunk sc = tid & 7
    const int sc = tid & 7;
    int sdst;
    {
        const int p = tid >> 3, yy = (p >> 4) + 1, xx = (p & 15) + 1;
        sdst = (sc >> 1) * PLANE + (yy * PITCH + xx) * 32 + (((sc ^ yy) & 1) << 4);
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
        if (has_pro) {
            f32x4 s0, s1, t0, t1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(coef + 16 * sc + 4 * j);        // channels 8 sc + 2 j, + 1
                (j < 2 ? s0 : s1)[2 * (j & 1)] = c[0]; (j < 2 ? t0 : t1)[2 * (j & 1)] = c[1];
                (j < 2 ? s0 : s1)[2 * (j & 1) + 1] = c[2]; (j < 2 ? t0 : t1)[2 * (j & 1) + 1] = c[3];
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = bn_act8(xr[i], s0, s1, t0, t1, slope);
        } else {
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = xr[i];
        }
    };
    // B fragments: pixel (4 tp + 2 i + r, x) of tile i at tap t, channels 16 kc + 8 h ..  ->  rb[t] + i (2 rows) + kc PLANE
    int rb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = 4 * tp + r + P.dy[t] + 1, xx = x + P.dx[t] + 1;
        rb[t] = (yy * PITCH + xx) * 32 + (((h ^ yy) & 1) << 4);
    }
    const int opix = ((4 * tp + r) * C::W + x) * g.ldo + 32 * nt + 8 * h;
    const int otile = 2 * C::W * g.ldo;
    const int64_t ostride = (int64_t)C::H * C::W * g.ldo;
    const bool want_stats = EX || a.stats != nullptr;
    const bool has_res = !EX && a.residual != nullptr;
    const float ex_slope = EX ? a.ex_slope : 1.f;
    float ps1[16], ps2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ps1[e] = ps2[e] = 0.f;

    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the weights are here (no counted waits for them inside the loop)
    __syncthreads();
    if (img < nimg) stage(0);
    __syncthreads();

    f32x16 acc[2];
    u32x4 opr[2][2];                               // [tile][16-byte half]: residual / raw tensor at this lane's store positions
    auto epilogue = [&](int im) __attribute__((always_inline)) {
        bf16* const oimg = O + (int64_t)im * ostride;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            uint32_t xw[2][2][2], ow[2][2][2];
            if (EX || has_res) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        // loaded: lanes 0-31 channels 16 gp + 0 .. 7, lanes 32-63 channels 16 gp + 8 .. 15; wanted: 8 gq + 4 h + e
                        const auto rr = __builtin_amdgcn_permlane32_swap(opr[i][gp][d], opr[i][gp][2 + d], false, false);
                        xw[i][0][d] = rr[0];
                        xw[i][1][d] = rr[1];
                    }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int e0 = 4 * (2 * gp + k) + 2 * d;
                    f32x4 c0, c1;
                    if (EX) {
                        c0 = reinterpret_cast<const f32x4*>(coef)[32 * nt + 8 * (2 * gp + k) + 4 * h + 2 * d];
                        c1 = reinterpret_cast<const f32x4*>(coef)[32 * nt + 8 * (2 * gp + k) + 4 * h + 2 * d + 1];
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float g0 = acc[i][e0], g1 = acc[i][e0 + 1];
                        if (EX) {
                            const uint32_t w = xw[i][k][d];
                            const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
                            g0 *= (x0 * c0[0] + c0[1] > 0.f) ? 1.f : ex_slope;
                            g1 *= (x1 * c1[0] + c1[1] > 0.f) ? 1.f : ex_slope;
                            ps1[e0] += g0;
                            ps2[e0] += g0 * (x0 * c0[2] + c0[3]);
                            ps1[e0 + 1] += g1;
                            ps2[e0 + 1] += g1 * (x1 * c1[2] + c1[3]);
                        } else {
                            if (has_res) {
                                const uint32_t w = xw[i][k][d];
                                g0 += __builtin_bit_cast(float, w << 16);
                                g1 += __builtin_bit_cast(float, w & 0xffff0000u);
                            }
                            if (want_stats) {
                                ps1[e0] += g0; ps2[e0] += g0 * g0;
                                ps1[e0 + 1] += g1; ps2[e0 + 1] += g1 * g1;
                            }
                        }
                        typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 pr = {(bf16)g0, (bf16)g1};
                        ow[i][k][d] = __builtin_bit_cast(uint32_t, pr);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto rr = __builtin_amdgcn_permlane32_swap(ow[i][0][d], ow[i][1][d], false, false);
                    ow[i][0][d] = rr[0];
                    ow[i][1][d] = rr[1];
                }
                const u32x4 o = {ow[i][0][0], ow[i][0][1], ow[i][1][0], ow[i][1][1]};
                *reinterpret_cast<u32x4*>(oimg + i * otile + opix + 16 * gp) = o;
            }
        }
    };
    auto body = [&](auto bufc, int im) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        const int nxt = im + gridDim.x;
        const bool has_next = nxt < nimg;
        if (has_next) request(nxt);
        if (EX || has_res) {
            const bf16* const eimg = OPD + (int64_t)im * ostride;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) opr[i][gp] = *reinterpret_cast<const u32x4*>(eimg + i * otile + opix + 16 * gp);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        int rbb[9];                                // (the image offset does not fit the 16-bit immediate beside the plane offset)
#pragma unroll
        for (int t = 0; t < 9; ++t) rbb[t] = rb[t] + BUF * TILE;
        constexpr int PD = EX ? SV_CCONV_PD_EX : SV_CCONV_PD, NB = PD + 1;
        bf16x8 bfr[NB][2], afr[NB];
        auto fetch = [&](int ks, bf16x8 (&dst)[2], bf16x8& adst) __attribute__((always_inline)) {
            const int t = ks / KC, kc = ks % KC;
#pragma unroll
            for (int i = 0; i < 2; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(smem + rbb[t] + (kc * PLANE + i * (2 * PITCH * 32)));
            if (ks >= KR) adst = *reinterpret_cast<const bf16x8*>(wlds + (ks - KR) * 1024);
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) fetch(d, bfr[d % NB], afr[d % NB]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + PD < KS) fetch(ks + PD, bfr[(ks + PD) % NB], afr[(ks + PD) % NB]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks < KR ? wf[ks < KR ? ks : 0] : afr[ks % NB], bfr[ks % NB][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        epilogue(im);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) stage(BUF ^ 1);
        __syncthreads();
    };
    {
        const int step = gridDim.x;
        while (img < nimg) {
            body(std::integral_constant<int, 0>{}, img);
            img += step;
            if (img >= nimg) break;
            body(std::integral_constant<int, 1>{}, img);
            img += step;
        }
    }
    // ---- sums: 32 pixel lanes -> lanes 0 / 32, the four waves of a channel tile through LDS, one double atomic per channel and block
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
            const int which = tid / NOUT, n = tid - which * NOUT, cn = n >> 5, cl = n & 31;
            const float* const ws = reinterpret_cast<const float*>(smem + C::OFF_WSUM) + which * 32 + cl;
            float v = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) v += ws[(cn + NT * m) * 64];
            atomicAdd((EX ? a.bsums : a.stats) + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * NOUT + tid, (double)v);
        }
    }
}

template <bool EX>
int launch_cconv(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    typedef cconv_cfg<EX> C;
    const int G = sv_ngroups(a->groups);
    int per = sv_persistent_blocks() / 2 / G;          // (the budget counts two blocks per CU; this kernel is one)
    if (per < 1) per = 1;
    if (per > g->B) per = g->B;
    const int rounds = (g->B + per - 1) / per;
    const int grid = (g->B + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_kernel<EX>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(cconv)");
        optin = true;
    }
    sv_igemm_args b = *a;          // the forward form folds the BatchNorm finalisation of its prologue
    if (!sv_fold_claim(!EX && b.fold_stats != nullptr)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((cconv_kernel<EX>), dim3(grid, G), dim3(C::NTH), C::LDS, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(cconv)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a stride-1 3x3 convolution 64 -> 64 at 16x16 (forward or data gradient) this kernel covers.
int sv_cconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (!sv_enabled(a->ex ? SV_K_CCONV_EX : SV_K_CCONV) || dtype != SV_BF16) return 0;
    if (a->bias || a->x2 || a->sparse_out || (a->residual && a->ex)) return 0;
    if (a->ex && (a->stats || a->pro_scale)) return 0;
    if ((a->flags & SV_FLAG_DET) && (a->stats || a->ex)) return 0;
    if (g->nphase != 1 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 9 || P.ooy != 0 || P.oox != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
    if (g->Hin != 16 || g->Win != 16 || g->Hout != 16 || g->Wout != 16 || g->Hq != 16 || g->Wq != 16) return 0;
    if (g->Cin != 64 || g->N != 64 || g->ldx != 64 || g->ldo % 4 != 0) return 0;
    *rc = a->ex ? launch_cconv<true>(g, a, s) : launch_cconv<false>(g, a, s);
    return 1;
}
