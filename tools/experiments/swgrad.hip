// Weight gradient of the stride-2 3x3 convolution 64 -> 128 at 16x16 -> 8x8 (wideresnet.py:29-30, first convolution of block 3) with
// the WHOLE gradient of the layer resident in one block's registers.  gfx950.
//
// The generic kernel (wgrad.hip) gives a block one (64 channels-out) x (one tap) x (64 channels-in) tile: dy and the activated input are
// staged once per TAP -- nine times per layer through L2 (~900 MB for a 100 MB layer: 96 us).  Here a persistent block of eight waves
// owns dW[128][9][64] entirely -- wave (nt, ct) accumulates [32 n][9 taps][32 c] = 36 tiles of v_mfma_f32_16x16x32_bf16 = 144
// registers -- and every image is staged ONCE: the input through the BatchNorm + LeakyReLU prologue into a row / column PARITY-split
// LDS image (sconv.hip's layout, pixel-major rows of 64 channels: a stride-2 tap is a row offset inside one sub-image), dy as it
// is; the k dimension of the products is the PIXEL index, so both operands are read back with the transposing ds_read_b64_tr_b16
// (wgrad.hip's fragment recipe: 4 rows x 16 columns per 16-lane group).  Per image and wave 72 MFMAs on 80 transposed reads; two
// images in LDS (the next one in registers during the MFMAs), one barrier per image.  At the end every block adds its 73 728
// partial sums to the fp32 gradient with float atomics (64 contiguous floats per wave instruction).
// MEASURED (4 x 512 images): 92 us against the generic kernel's 69 -- the loop itself takes 29 us (tools/probes/swgrad_ablate.sh), the
// final atomics 61: 256 blocks x 73 728 floats = 18.9 M atomic adds at the ~300 G/s the memory side sustains.  Fewer, larger blocks trade
// that one for one against compute; slabs in a workspace + a reduction pass cost as much.  OFF (SV_OPT_ENABLE_MASK, SV_K_SWGRAD): kept
// as the record of why the whole-gradient-per-block form does not pay at 256 blocks, with its test.
// Same sv_wgrad contract; declines the deterministic mode.
#include <type_traits>

#include "common.h"

#ifndef SV_SWG_DBG
#define SV_SWG_DBG 0             // ablation (tools/probes/swgrad_ablate.sh): 1 no final atomics, 2 no MFMA loop, 4 no staging of the next image
#endif

namespace {

struct swg_params {
    const void* x;
    const void* dy;
    const float* pro_scale;
    const float* pro_shift;
    float pro_slope;
    float* dw;
};

struct swg_cfg {
    static constexpr int CIN = 64, NOUT = 128, W = 8, HIN = 16;
    static constexpr int LDX = CIN + 8, LDY = NOUT + 8;            // LDS row strides (elements): 144 / 272 bytes
    static constexpr int SUBPIX = 9 * 9, SUB = SUBPIX * LDX * 2;   // a parity sub-image: 9 x 9 pixels (row 0 / column 0: the halo)
    static constexpr int XIMG = 4 * SUB, YIMG = W * W * LDY * 2, IMG = XIMG + YIMG;
    static constexpr int NTH = 512, XV = HIN * HIN * (CIN / 8) / NTH, YV = W * W * (NOUT / 8) / NTH;
    static constexpr int OFF_COEF = 2 * IMG, LDS = OFF_COEF + CIN * 8;
    static_assert(XV == 4 && YV == 2 && LDS <= 160 * 1024, "staging / LDS budget");
};

// k-major fragment of v_mfma_f32_16x16x32_bf16 from a pixel-major LDS image: the lane's 8 k indices are the pixels row0 .. row0 + 7
// (consecutive LDS rows of ld elements), its row / column index the channel col0 + (lane & 15).  ds_read_b64_tr_b16: lane 4 q + p of a
// 16-lane group addresses row q, channels 4 p .. 4 p + 3 and receives channel (lane & 15) of the 4 rows.
// `a0` = the lane's address of (row0 + (i >> 2), col0 + 4 (i & 3)), i = lane & 15 -- one register per tap, everything else of a read is
// an immediate offset; ldb = bytes per LDS row.
__device__ __forceinline__ bf16x8 frag_px(const char* a0, int ldb) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * ldb));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

__global__ __launch_bounds__(512, 1) void swgrad_kernel(const sv_geom g, const sv_wg_g<swg_params> PG) {
    typedef swg_cfg C;
    constexpr int CIN = C::CIN, NOUT = C::NOUT, W = C::W, LDX = C::LDX, LDY = C::LDY, SUB = C::SUB, IMG = C::IMG, NTH = C::NTH;
    const swg_params& p = PG.g[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, ct = wave >> 2;                       // 32-channel tiles of dy (n) and of the input (c)
    const int gq = lane >> 4;                                      // k group of the fragments: 8 pixels = one output row
    const sv_phase& P = g.phase[0];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
    const bf16* __restrict__ DY = reinterpret_cast<const bf16*>(p.dy);
    const int nimg = g.B;
    int img = blockIdx.x;

    bf16x8 xr[C::XV], yr[C::YV];
    auto request = [&](int im) __attribute__((always_inline)) {
        const bf16* const xi = X + (int64_t)im * (C::HIN * C::HIN * CIN);
        const bf16* const yi = DY + (int64_t)im * (W * W * NOUT);
#pragma unroll
        for (int i = 0; i < C::XV; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(xi + tid * 8 + i * (NTH * 8));
#pragma unroll
        for (int i = 0; i < C::YV; ++i) yr[i] = *reinterpret_cast<const bf16x8*>(yi + tid * 8 + i * (NTH * 8));
    };
    if (img < nimg) request(img);
    const bool has_pro = p.pro_scale != nullptr;
    float* const coef = reinterpret_cast<float*>(smem + C::OFF_COEF);
    const float slope = has_pro ? p.pro_slope : 1.f;
    if (has_pro && tid < 2 * CIN) coef[tid] = (tid & 1) ? p.pro_shift[tid >> 1] : p.pro_scale[tid >> 1];
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * IMG / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    // staging: x vector i = pixel (tid >> 3) + 64 i of the 16 x 16 image (4 rows further: same parities), chunk tid & 7;
    // dy vector i = pixel (tid >> 4) + 32 i, chunk tid & 15
    const int sc = tid & 7;
    int xdst, ydst;
    {
        const int pxl = tid >> 3, iy = pxl >> 4, ix = pxl & 15;
        xdst = (2 * (iy & 1) + (ix & 1)) * SUB + ((((iy >> 1) + 1) * 9 + (ix >> 1) + 1) * LDX + 8 * sc) * 2;
        ydst = C::XIMG + ((tid >> 4) * LDY + 8 * (tid & 15)) * 2;
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
        char* const base = smem + buf * IMG;
        if (has_pro) {
            f32x4 s0, s1, t0, t1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(coef + 16 * sc + 4 * j);        // channels 8 sc + 2 j, + 1
                (j < 2 ? s0 : s1)[2 * (j & 1)] = c[0]; (j < 2 ? t0 : t1)[2 * (j & 1)] = c[1];
                (j < 2 ? s0 : s1)[2 * (j & 1) + 1] = c[2]; (j < 2 ? t0 : t1)[2 * (j & 1) + 1] = c[3];
            }
#pragma unroll
            for (int i = 0; i < C::XV; ++i) *reinterpret_cast<bf16x8*>(base + xdst + i * (2 * 9 * LDX * 2)) = bn_act8(xr[i], s0, s1, t0, t1, slope);
        } else {
#pragma unroll
            for (int i = 0; i < C::XV; ++i) *reinterpret_cast<bf16x8*>(base + xdst + i * (2 * 9 * LDX * 2)) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < C::YV; ++i) *reinterpret_cast<bf16x8*>(base + ydst + i * (32 * LDY * 2)) = yr[i];
    };
    // B fragments: tap t reads sub-image (dy & 1, dx & 1) at row y + (dy >= 0), column x + (dx >= 0): per tap the LDS pixel row of
    // (output row gq of the chunk, column 0)
    // (byte offsets of the lane's fragment address inside an image: one register per tap, the chunk / channel-half / image are
    //  immediates or one add per image)
    const int li = lane & 15;
    int xoffs[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = P.dy[t], dx = P.dx[t];
        const int row = (2 * (dy & 1) + (dx & 1)) * C::SUBPIX + (gq + (dy >= 0 ? 1 : 0)) * 9 + (dx >= 0 ? 1 : 0);
        xoffs[t] = ((row + (li >> 2)) * LDX + 32 * ct + 4 * (li & 3)) * 2;
    }
    const int yoffs = C::XIMG + ((8 * gq + (li >> 2)) * LDY + 32 * nt + 4 * (li & 3)) * 2;
    f32x4 acc[2][9][2];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) acc[a_][t][b_] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (img < nimg) stage(0);
    __syncthreads();

    // (one copy of the loop body -- the image buffer and the 32-pixel chunk are run-time values: with the body unrolled over both, the
    //  register allocator spilled 120-190 registers next to the 144 accumulators)
    {
        const int step = gridDim.x;
        int buf = 0;
        for (; img < nimg; img += step, buf ^= 1) {
            const int nxt = img + step;
            const bool has_next = nxt < nimg;
            if (has_next && !(SV_SWG_DBG & 4)) request(nxt);
            const char* const IB = smem + buf * IMG;
#pragma unroll 1
            for (int kc = 0; kc < ((SV_SWG_DBG & 2) ? 0 : 2); ++kc) {             // 32 output pixels = rows 4 kc .. 4 kc + 3
                const char* const KB = IB + kc * (4 * 9 * LDX * 2);
                bf16x8 af[2];
#pragma unroll
                for (int a_ = 0; a_ < 2; ++a_) af[a_] = frag_px(IB + yoffs + kc * (32 * LDY * 2) + 16 * a_ * 2, LDY * 2);
                // the fragments of tap t + 1 are requested before the MFMAs of tap t; the scheduling barrier keeps the compiler from
                // hoisting all eighteen of a chunk (72 registers) to its top
                bf16x8 bf_[2][2];
                auto fetch = [&](int t, bf16x8 (&dst)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_) dst[b_] = frag_px(KB + xoffs[t] + 16 * b_ * 2, LDX * 2);
                };
                fetch(0, bf_[0]);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 1 < 9) fetch(t + 1, bf_[(t + 1) & 1]);
#pragma unroll
                    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                        for (int b_ = 0; b_ < 2; ++b_) mma32(acc[a_][t][b_], af[a_], bf_[t & 1][b_]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (has_next && !(SV_SWG_DBG & 4)) stage(buf ^ 1);
            __syncthreads();
        }
    }
    // ---- dW[n][torig][c] += : acc[a][t][b][e] = (n = 32 nt + 16 a + 4 (lane >> 4) + e, c = 32 ct + 16 b + (lane & 15))
    {
        const int T = g.T_orig;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int to = P.torig[t];
#pragma unroll
            for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = 32 * nt + 16 * a_ + 4 * gq + e, c = 32 * ct + 16 * b_ + (lane & 15);
                        if (!(SV_SWG_DBG & 1) || acc[a_][t][b_][e] == 123.25f) atomicAdd(p.dw + ((size_t)n * T + to) * CIN + c, acc[a_][t][b_][e]);
                    }
        }
    }
}

}  // namespace

// Returns 1 and sets *rc when the launch is the weight gradient of the stride-2 3x3 convolution 64 -> 128 at 16x16.
int sv_swgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                  const void* dy, float* dw, int groups, hipStream_t s, int* rc) {
    typedef swg_cfg C;
    if (!sv_enabled(SV_K_SWGRAD) || dtype != SV_BF16 || sv_deterministic()) return 0;      // OFF by default: see the header
    if (g->nphase != 1 || g->sy != 2 || g->sx != 2 || g->osy != 1 || g->osx != 1) return 0;
    const sv_phase& P = g->phase[0];
    if (P.ntap != 9 || g->T_orig != 9) return 0;
    for (int t = 0; t < 9; ++t)
        if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
    if (g->Cin != C::CIN || g->N != C::NOUT || g->Hin != C::HIN || g->Win != C::HIN || g->Hout != C::W || g->Wout != C::W) return 0;
    if (g->ldx != g->Cin || g->ldo != g->N) return 0;
    swg_params p;
    p.x = x; p.dy = dy; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_slope = pro_slope; p.dw = dw;
    int per = sv_persistent_blocks() / 2 / groups;
    if (per < 1) per = 1;
    if (per > g->B) per = g->B;
    const int rounds = (g->B + per - 1) / per;
    const int grid = (g->B + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&swgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) {
            *rc = sv_check_launch("hipFuncSetAttribute(swgrad)");
            return 1;
        }
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL(swgrad_kernel, dim3(grid, groups), dim3(C::NTH), C::LDS, s, *g, sv_expand_wg(*g, p, groups, 2));
    sv_prof_end(s);
    *rc = sv_check_launch("sv_wgrad(swgrad)");
    return 1;
}
