// Data gradient of the last decoder layer: ConvTranspose2d(64, 3, 4, 2, 1) backward = a 4x4 stride-2 convolution 16 (3 padded) -> 64
// at 32x32 -> 16x16 (decoder.py:58), fused with the activation backward of the BatchNorm + ReLU in front of that layer
// (decoder.py:55-56: the raw tensor at the output positions, g * act'(BatchNorm(x)) stored, sum g and sum g * xhat to bsums).  gfx950.
// sconv.hip's scheme (bands of 8 output rows, row / column parity-split LDS image, register-resident weights) with 16 taps of one
// 16-channel k-step each and cconv.hip's activation-backward epilogue; no load prologue (the input is the loss's gradient).  The
// LDS-halo kernel ran this layer (two groups: 100 MB) at 2.2 TB/s: 46 us.
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a fast path inside it (SV_K_TCONVR_EX disables).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// NOUT_ = 64: the decoder layer above, bands of 8 output rows (two channel tiles x four pixel tiles of 2 rows).  NOUT_ = 32 (round 6):
// svhn_VAE's first convolution (3 (16 padded) -> 32, svhn_vae.py:62; EX_ = false: the bias epilogue) and the data gradient of its last
// ConvTranspose2d(32, 3, 4, 2, 1) (svhn_vae.py:131; EX_ = true): one channel tile x eight pixel tiles = a whole 16 x 16 image per pass.
template <int NOUT_>
struct dconv_cfg {
    static constexpr int CIN = 16, NOUT = NOUT_, W = 16, NTAP = 16;
    static constexpr int NCT = NOUT / 32, NPT = 8 / NCT, BR = 2 * NPT;                // channel tiles, pixel tiles, output rows of a band
    static constexpr int PITCH = 24, SUB = (BR + 2) * PITCH * 32 + 64, TILE = 4 * SUB;   // four parity sub-images of (BR + 2) x 18 pixels (32 B each)
    static constexpr int NTH = 512, VROW = 64, ROWS = 2 * BR + 2, NVEC = ROWS * VROW, VPT = (NVEC + NTH - 1) / NTH;
    static constexpr int OFF_WSUM = 2 * TILE;                      // [8 waves][2][32] floats
    static constexpr int OFF_CST = OFF_WSUM + 8 * 2 * 32 * 4;      // [NOUT] x {scale, shift, rstd, -mean rstd} (EX) / {bias, -, -, -}
    static constexpr int LDS = OFF_CST + NOUT * 16;
    static_assert((SUB / 16) % 8 == 4, "staging: 8 lanes = 4 pixels x 2 halves on 8 bank groups");
};

template <int NOUT_, bool EX_>
__global__ __launch_bounds__(512, 1) void dconv_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    typedef dconv_cfg<NOUT_> C;
    constexpr int CIN = C::CIN, NOUT = C::NOUT, W = C::W, NTAP = C::NTAP, PITCH = C::PITCH, SUB = C::SUB, TILE = C::TILE, NTH = C::NTH, VPT = C::VPT;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave % C::NCT, mt = wave / C::NCT;              // channel tile, pixel tile (2 rows x 16) of the band
    const int q = lane & 31, h = lane >> 5, ty = q >> 4, tx = q & 15;
    const sv_phase& P = g.phase[0];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    const bf16* __restrict__ EXP = reinterpret_cast<const bf16*>(a.ex);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    const int bpi = g.Hout / C::BR;                                // bands per image
    const int nband = g.B * bpi;
    int band = blockIdx.x;

    // ---- a band's vectors: v = tid + 512 i is vector v of the 2 BR + 2 input rows 2 BR b - 1 .. 2 BR (b + 1) (1 KB each, contiguous)
    bf16x8 xr[VPT];
    auto request = [&](int bd) __attribute__((always_inline)) {
        const int im = bd / bpi, b = bd - im * bpi;
        const bf16* const xi = X + ((int64_t)im * g.Hin + 2 * C::BR * b - 1) * (2 * W * CIN);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = tid + NTH * i, r = v >> 6;
            const bool ok = v < C::NVEC && (b > 0 || r > 0) && (b < bpi - 1 || r < C::ROWS - 1);     // (rows -1 and 32 of the image are padding)
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            xr[i] = ok ? *reinterpret_cast<const bf16x8*>(xi + v * 8) : z;
        }
    };
    if (band < nband) request(band);
    // ---- weights: A fragments (row = channel 32 nt + q, k = tap t: 16 channels, 8 h ..) of the packed [N][16 taps][16]
    bf16x8 wf[NTAP];
    {
        const bf16* __restrict__ Wp = reinterpret_cast<const bf16*>(a.w) + P.w_off + (32 * nt + q) * (NTAP * CIN) + 8 * h;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(Wp + CIN * t);
    }
    float* const cst = reinterpret_cast<float*>(smem + C::OFF_CST);
    if (tid < NOUT) {
        if (EX_) {
            const float rs = a.ex_rstd[tid];
            reinterpret_cast<f32x4*>(cst)[tid] = f32x4{a.ex_scale[tid], a.ex_shift[tid], rs, -a.ex_mean[tid] * rs};     // xhat = x rstd - mean rstd
        } else {
            reinterpret_cast<f32x4*>(cst)[tid] = f32x4{a.bias[tid], 0.f, 0.f, 0.f};
        }
    }
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * TILE / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    // staging: vector i of this thread = input row r = (tid >> 6) + 8 i (same parity for every i), pixel (tid & 63) >> 1, half tid & 1
    int sdst;
    {
        const int r = tid >> 6, rowidx = (r + 1) >> 1, pr = (r & 1) ^ 1, ix = (tid & 63) >> 1, pc = ix & 1, colidx = (ix >> 1) + 1;
        sdst = (2 * pr + pc) * SUB + (rowidx * PITCH + colidx) * 32 + ((((tid & 1) ^ rowidx) & 1) << 4);
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            if (tid + NTH * i < C::NVEC) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = xr[i];
    };
    // B fragments: output pixel (2 mt + ty, tx) at tap (dy, dx) reads input (2 y + dy, 2 x + dx) = sub-image (dy & 1, dx & 1), row
    // y + ((dy + 2) >> 1), column x + ((dx + 2) >> 1) (stored one further: the halo), channels 8 h ..
    int rb[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int dy = P.dy[t], dx = P.dx[t];
        const int yy = 2 * mt + ty + ((dy + 2) >> 1), xx = tx + ((dx + 2) >> 1);
        rb[t] = (2 * (dy & 1) + (dx & 1)) * SUB + (yy * PITCH + xx) * 32 + (((h ^ yy) & 1) << 4);
    }
    const int opix = ((2 * mt + ty) * g.Wout + tx) * g.ldo + 32 * nt + 8 * h;
    const float ex_slope = a.ex_slope;
    float ps1[16], ps2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ps1[e] = ps2[e] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the weights are here (no counted waits for them inside the loop)
    __syncthreads();
    if (band < nband) stage(0);
    __syncthreads();

    // (one copy of the loop body: the image buffer is a run-time value)
    {
        const int step = gridDim.x;
        int buf = 0;
        for (; band < nband; band += step, buf ^= 1) {
            const int nxt = band + step;
            const bool has_next = nxt < nband;
            if (has_next) request(nxt);
            const int im = band / bpi, b = band - im * bpi;
            const int64_t obase = ((int64_t)im * g.Hout + C::BR * b) * g.Wout * g.ldo;
            u32x4 opr[2] = {};                    // the raw tensor at this lane's two 16-byte store positions
            if (EX_) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) opr[gp] = *reinterpret_cast<const u32x4*>(EXP + obase + opix + 16 * gp);
            }
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const char* const IB = smem + buf * TILE;
            constexpr int PD = 2, NB = PD + 1;
            bf16x8 bfr[NB];
#pragma unroll
            for (int d = 0; d < PD; ++d) bfr[d] = *reinterpret_cast<const bf16x8*>(IB + rb[d]);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                if (t + PD < NTAP) bfr[(t + PD) % NB] = *reinterpret_cast<const bf16x8*>(IB + rb[t + PD]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], bfr[t % NB], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: acc[4 gq + e] = channel 32 nt + 8 gq + 4 h + e of pixel q
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
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
                        const int e0 = 4 * (2 * gp + k) + 2 * d;
                        const f32x4 c0 = reinterpret_cast<const f32x4*>(cst)[32 * nt + 8 * (2 * gp + k) + 4 * h + 2 * d];
                        const f32x4 c1 = reinterpret_cast<const f32x4*>(cst)[32 * nt + 8 * (2 * gp + k) + 4 * h + 2 * d + 1];
                        const uint32_t w = xw[k][d];
                        const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
                        float g0, g1;
                        if (EX_) {
                            g0 = acc[e0] * ((x0 * c0[0] + c0[1] > 0.f) ? 1.f : ex_slope);
                            g1 = acc[e0 + 1] * ((x1 * c1[0] + c1[1] > 0.f) ? 1.f : ex_slope);
                            ps1[e0] += g0;
                            ps2[e0] += g0 * (x0 * c0[2] + c0[3]);
                            ps1[e0 + 1] += g1;
                            ps2[e0 + 1] += g1 * (x1 * c1[2] + c1[3]);
                        } else {
                            g0 = acc[e0] + c0[0];
                            g1 = acc[e0 + 1] + c1[0];
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
            if (has_next) stage(buf ^ 1);
            __syncthreads();
        }
    }
    // ---- sums: 32 pixel lanes -> lanes 0 / 32, the waves of a channel tile through LDS, one double atomic per channel and block
    if (EX_) {
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
            for (int m = 0; m < C::NPT; ++m) v += ws[(cn + C::NCT * m) * 64];
            atomicAdd(a.bsums + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * NOUT + tid, (double)v);
        }
    }
}

}  // namespace

namespace {

template <int NOUT_, bool EX_>
int launch_dconv(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    typedef dconv_cfg<NOUT_> C;
    const int G = sv_ngroups(a->groups);
    const int nband = g->B * (g->Hout / C::BR);
    int per = sv_persistent_blocks() / 2 / G;          // (the budget counts two blocks per CU; this kernel is one)
    if (per < 1) per = 1;
    if (per > nband) per = nband;
    const int rounds = (nband + per - 1) / per;
    const int grid = (nband + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dconv_kernel<NOUT_, EX_>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(dconv)");
        optin = true;
    }
    int gate_rc = SV_OK;
    if (sv_dry_run(grid, a, &gate_rc)) return gate_rc;
    sv_prof_begin(s);
    hipLaunchKernelGGL((dconv_kernel<NOUT_, EX_>), dim3(grid, G), dim3(C::NTH), C::LDS, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(dconv)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a 4x4 stride-2 convolution 16 -> 64 / 16 -> 32 at 32x32 this file covers: the data gradient of
// the last decoder layer (activation-backward epilogue), svhn_VAE's first convolution (bias epilogue) and the data gradient of its last
// ConvTranspose2d.
int sv_dconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_TCONVR_EX) || dtype != SV_BF16) return 0;
    if (a->residual || a->sparse_out || a->stats || a->pro_scale || (a->flags & SV_FLAG_DET)) return 0;
    if (a->ex ? a->bias != nullptr : a->bias == nullptr) return 0;          // one epilogue: activation backward OR bias
    if (g->nphase != 1 || g->sy != 2 || g->sx != 2 || g->osy != 1 || g->osx != 1) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 16 || P.ooy != 0 || P.oox != 0) return 0;
    for (int t = 0; t < 16; ++t)
        if (P.dy[t] < -1 || P.dy[t] > 2 || P.dx[t] < -1 || P.dx[t] > 2) return 0;
    if (g->Cin != 16 || g->ldx != 16 || (g->N != 64 && g->N != 32) || g->Hin != 32 || g->Win != 32 || g->Hout != 16 || g->Wout != 16 || g->Hq != 16 ||
        g->Wq != 16 || g->ldo % 4 != 0 || (int64_t)g->B * g->Hout * g->Wout * g->ldo >= ((int64_t)1 << 31))
        return 0;
    if (g->N == 64) {
        if (!a->ex) return 0;
        *rc = launch_dconv<64, true>(g, a, s);
    } else {
        *rc = a->ex ? launch_dconv<32, true>(g, a, s) : launch_dconv<32, false>(g, a, s);
    }
    return 1;
}
