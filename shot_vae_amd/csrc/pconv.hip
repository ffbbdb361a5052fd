// 1x1 convolution forward (the WideResNet shortcuts, wideresnet.py:41-43: 16 -> 32 at stride 1, 32 -> 64 at stride 2) fused
// with the BatchNorm + LeakyReLU in front of it (wideresnet.py:39-40) as the load prologue.  gfx950.
//
// A pointwise convolution re-uses no input pixel across output positions: the LDS staging of the gather-GEMM kernels buys nothing.
// Here the B fragments of v_mfma_f32_32x32x16_bf16 come STRAIGHT from global memory -- a lane's 8 consecutive channels of its pixel
// are 16 contiguous bytes of the NHWC tensor -- pass through the prologue in registers (the lane's channels never change: their
// scale / shift stay in registers) and meet the weights, which are a handful of A fragments held for the kernel's lifetime.  No
// LDS in the loop, no barrier; a wave streams 32-pixel tiles (the next tile's vectors requested before this tile's MFMAs) and
// writes 16-byte vectors through v_permlane32_swap.  The BatchNorm finalisation of the prologue is folded into the launch
// (sv_igemm_args::fold_*).  Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm (SV_K_PCONV disables).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int CIN, int NOUT>
__global__ __launch_bounds__(256) void pconv_kernel(const sv_geom g, const sv_igemm_args_g AG, int ntiles) {
    constexpr int KS = CIN / 16, NT = NOUT / 32;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    __shared__ __attribute__((aligned(16))) double fold_scratch[512];
    __shared__ float sc_lds[64], sh_lds[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane & 31, h = lane >> 5;
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    // ---- prologue coefficients (channels 16 ks + 8 h + j of this lane), folded finalisation first
    const bool has_pro = a.pro_scale != nullptr;
    if (a.fold_stats) sv_bn_fold_block(a, CIN, fold_scratch, sc_lds, sh_lds, blockIdx.x == 0);
    float cs[KS][8], ct[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * ks + 8 * h + j;
            cs[ks][j] = !has_pro ? 1.f : (a.fold_stats ? sc_lds[c] : a.pro_scale[c]);
            ct[ks][j] = !has_pro ? 0.f : (a.fold_stats ? sh_lds[c] : a.pro_shift[c]);
        }
    const float slope = has_pro ? a.pro_slope : 1.f;
    // ---- weights: A fragments (row = channel 32 nt + q, k = 16 ks + 8 h ..) of the packed [N][CIN] matrix
    bf16x8 wf[NT][KS];
    {
        const bf16* __restrict__ W = reinterpret_cast<const bf16*>(a.w) + g.phase[0].w_off;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[nt][ks] = *reinterpret_cast<const bf16x8*>(W + (32 * nt + q) * CIN + 16 * ks + 8 * h);
    }
    // ---- tiles of 32 output positions, wave-strided
    const int wlog = __builtin_ctz(g.Wq), hwlog = __builtin_ctz(g.Hq * g.Wq);     // (powers of two: checked by the launcher)
    const int wave_id = blockIdx.x * 4 + (tid >> 6), nwaves = gridDim.x * 4;
    auto src = [&](int tile) {
        const int m = 32 * tile + q, b = m >> hwlog, r = m & ((1 << hwlog) - 1), y = r >> wlog, x = r & ((1 << wlog) - 1);
        return X + ((int64_t)(b * g.Hin + y * g.sy) * g.Win + x * g.sx) * CIN + 8 * h;
    };
    bf16x8 xv[KS], xn[KS];
    int tile = wave_id;
    if (tile < ntiles) {
        const bf16* s = src(tile);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xv[ks] = *reinterpret_cast<const bf16x8*>(s + 16 * ks);
    }
    for (; tile < ntiles; tile += nwaves) {
        const int nxt = tile + nwaves;
        if (nxt < ntiles) {
            const bf16* s = src(nxt);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xn[ks] = *reinterpret_cast<const bf16x8*>(s + 16 * ks);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 bfr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = (float)xv[ks][j] * cs[ks][j] + ct[ks][j];
                bfr[j] = has_pro ? (bf16)fmaxf(u, u * slope) : xv[ks][j];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][ks], bfr, acc[nt], 0, 0, 0);
        }
        // acc[nt][4 gq + e] = channel 32 nt + 8 gq + 4 h + e of position q: 16-byte stores through v_permlane32_swap
        bf16* const op = O + (int64_t)(32 * tile + q) * g.ldo + 8 * h;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t pk[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 pr = {(bf16)acc[nt][4 * (2 * gp + k) + 2 * d], (bf16)acc[nt][4 * (2 * gp + k) + 2 * d + 1]};
                        pk[k][d] = __builtin_bit_cast(uint32_t, pr);
                    }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto r = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                    pk[0][d] = r[0];
                    pk[1][d] = r[1];
                }
                const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                *reinterpret_cast<u32x4*>(op + 32 * nt + 16 * gp) = o;
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xv[ks] = xn[ks];
    }
}

template <int CIN, int NOUT>
int launch_pconv(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    const int G = sv_ngroups(a->groups);
    const int64_t M = (int64_t)g->B * g->Hq * g->Wq;
    const int ntiles = (int)(M / 32);
    int grid = 256 * 8 / G;                         // ~8 blocks of 4 waves per CU over the batched launch
    if (grid > (ntiles + 3) / 4) grid = (ntiles + 3) / 4;
    if (grid < 1) grid = 1;
    sv_igemm_args b = *a;                           // folds the BatchNorm finalisation of its prologue (<= 64 channels, <= 64 replicas)
    if (!sv_fold_claim(b.fold_stats && b.fold_replicas <= 64)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((pconv_kernel<CIN, NOUT>), dim3(grid, G), dim3(256), 0, s, *g, sv_expand_groups(*g, *a, 2), ntiles);
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(pconv)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a 1x1 forward convolution this kernel covers.
int sv_pconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_PCONV) || dtype != SV_BF16) return 0;
    if (a->bias || a->residual || a->ex || a->sparse_out || a->stats) return 0;
    if (g->nphase != 1 || g->phase[0].ntap != 1 || g->phase[0].dy[0] != 0 || g->phase[0].dx[0] != 0) return 0;
    if (g->phase[0].ooy != 0 || g->phase[0].oox != 0 || g->osy != 1 || g->osx != 1 || g->sy != g->sx || g->sy < 1 || g->sy > 2) return 0;
    if (g->Hq != g->Hout || g->Wq != g->Wout || g->Hin != g->sy * g->Hout || g->Win != g->sx * g->Wout) return 0;
    if (g->ldx != g->Cin || g->ldo % 8 != 0 || (g->Wq & (g->Wq - 1)) || ((g->Hq * g->Wq) & (g->Hq * g->Wq - 1))) return 0;
    const int64_t M = (int64_t)g->B * g->Hq * g->Wq;
    if (M % 32 != 0 || M * g->ldo >= ((int64_t)1 << 31) || (int64_t)g->B * g->Hin * g->Win * g->Cin >= ((int64_t)1 << 31)) return 0;
    if (g->Cin == 16 && g->N == 32) { *rc = launch_pconv<16, 32>(g, a, s); return 1; }
    if (g->Cin == 32 && g->N == 64) { *rc = launch_pconv<32, 64>(g, a, s); return 1; }
    if (g->Cin == 16 && g->N == 160) { *rc = launch_pconv<16, 160>(g, a, s); return 1; }      // (the first shortcut at width 10: 368 MB)
    // (an activation-backward form for the dense 1x1 data gradients of the stride-2 shortcuts was built and measured: 64 -> 32 level with
    //  the gather-GEMM (33.8 vs 34.0 us), 128 -> 64 at 288 registers twice as slow (51 vs 27 us): not kept)
    // (64 -> 128: 240 registers -- weights 64, coefficients 64, accumulators 64 -- leave one wave per SIMD: 25.0 us against the
    //  gather-GEMM's 22.9; not dispatched)
    return 0;
}
