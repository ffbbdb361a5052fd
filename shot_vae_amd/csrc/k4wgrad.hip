// Weight gradient of the THIN 4x4 stride-2 layers between a 16-channel 32x32 tensor and a 32-channel 16x16 tensor: svhn_VAE's first
// convolution, Conv2d(3 (16 padded), 32, 4, 2, 1) (svhn_vae.py:62), and its last layer, ConvTranspose2d(32, 3 (16 padded), 4, 2, 1)
// (svhn_vae.py:131).  gfx950.
// Both are ONE sum: with `small` the 16x16 tensor (the convolution's dy / the transposed convolution's input) and `big` the 32x32 one
// (the convolution's input / the transposed convolution's dy),
//     G[s][ky][kx][b] = sum over images and (i, j) of small[i][j][s] * big[2 i - 1 + ky][2 j - 1 + kx][b],
// and only the place of (s, b) in the gradient differs (convolution: dW[n = s][tap][c = b]; transposed: dW[n = b][tap][c = s]).
// thwgrad.hip's scheme: dW is tiny (8 192 floats), so a persistent block owns ALL of it, stages every image ONCE (the big tensor with a
// zero border, the load prologue of the layer -- BatchNorm-free here: ReLU as scale 1 / shift 0 / slope 0 -- on whichever tensor is the
// layer's input), a wave takes two output rows (32 positions = the k dimension of one v_mfma_f32_16x16x32_bf16 step; both operands
// read k-major with ds_read_b64_tr_b16, the stride-2 walk over the big image is the lanes' own addresses), accumulates [32][16 taps][16]
// in 128 registers over the block's images, and at the end the eight waves meet in an LDS copy of G.  With a workspace the block stores
// that copy as a slab and sv_slab_reduce adds the slabs to the fp32 gradient; without one the block adds it itself (256 blocks x 8 192
// float atomics on the same 8 192 addresses).  The generic gather kernel ran these at 67 / 82 us per launch (2 048 images: 100 MB, 1.2-1.5 TB/s).
// Same sv_wgrad contract: a fast path inside it (SV_K_THWGRAD disables); declines the deterministic mode.
#include "common.h"

void sv_slab_reduce(const float* ws, int nslabs, int64_t n, float* dw, hipStream_t s);      // wgrad3x3.hip

namespace {

struct k4wg_params {
    const void* small;             // [B][16][16][32]
    const void* big;               // [B][32][32][16]
    const float* pro_scale;        // load prologue of the layer's input (may be null)
    const float* pro_shift;
    float pro_slope;
    int pro_on_small;              // 1: the input is the small tensor (transposed convolution), 0: the big one
    int transposed;                // 1: dW[n = b][tap][c = s], 0: dW[n = s][tap][c = b]
    float* dw;
    int8_t ky[16], kx[16];         // slot -> kernel position
    int8_t slot_of[16];            // tap index of the master layout -> slot
    float* slabs;                  // null: the blocks add to dw themselves; else block i of the launch stores its copy of the gradient at
                                   // slabs + 8192 i (master layout) and sv_slab_reduce adds the slabs to dw
};

struct k4wg_cfg {
    static constexpr int SC = 32, BC = 16, NH = SC / 16, NT = 16;
    static constexpr int LDB = BC + 8, LDS_ = SC + 8;              // LDS pixel strides (elements): 48 / 80 bytes
    static constexpr int BIMG = 34 * 34 * LDB * 2, SIMG = 256 * LDS_ * 2, IMG = BIMG + SIMG;
    static constexpr int NTH = 512, BV = 32 * 32 * 2 / NTH, SV = 256 * 4 / NTH;          // 16-byte vectors per thread and image
    static constexpr int RED = SC * NT * BC * 4;                   // the block's copy of G: over the image buffers once the loop is done
    static constexpr int LDS = 2 * IMG;
    static_assert(RED <= LDS && LDS <= 160 * 1024, "LDS budget");
};

// k-major fragment of v_mfma_f32_16x16x32_bf16 from a pixel-major LDS image (thwgrad.hip): a0 = the lane's address of
// (position 8 (lane >> 4) + ((lane & 15) >> 2), channel 4 (lane & 3)); the second half of the k group is 4 positions = `four` bytes further
__device__ __forceinline__ bf16x8 k4wg_frag(const char* a0, int four) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + four));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

__global__ __launch_bounds__(512, 1) void k4wgrad_kernel(const int B, const sv_wg_g<k4wg_params> PG) {
    typedef k4wg_cfg C;
    constexpr int SC = C::SC, BC = C::BC, NH = C::NH, NT = C::NT, LDB = C::LDB, LDS_ = C::LDS_, IMG = C::IMG, NTH = C::NTH, BV = C::BV, SV = C::SV;
    const k4wg_params& p = PG.g[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // output rows 2 wave, 2 wave + 1
    const int gq = lane >> 4, li = lane & 15;
    const bf16* __restrict__ SM = reinterpret_cast<const bf16*>(p.small);
    const bf16* __restrict__ BG = reinterpret_cast<const bf16*>(p.big);
    int im = blockIdx.x;
    const int step = gridDim.x;

    struct VS { bf16x8 b[BV], s[SV]; };
    VS S0;
    auto request = [&](int i_, VS& V) __attribute__((always_inline)) {
        const bf16* const bi = BG + (int64_t)i_ * (32 * 32 * BC);
        const bf16* const si = SM + (int64_t)i_ * (256 * SC);
#pragma unroll
        for (int i = 0; i < BV; ++i) V.b[i] = *reinterpret_cast<const bf16x8*>(bi + (tid + NTH * i) * 8);
#pragma unroll
        for (int i = 0; i < SV; ++i) V.s[i] = *reinterpret_cast<const bf16x8*>(si + (tid + NTH * i) * 8);
    };
    if (im < B) request(im, S0);
    const bool has_pro = p.pro_scale != nullptr;
    const bool pro_small = has_pro && p.pro_on_small, pro_big = has_pro && !p.pro_on_small;
    const float slope = has_pro ? p.pro_slope : 1.f;
    // prologue coefficients of this thread's 8 channels (the same for all of its vectors of the tensor: 512 % 4 == 0, 512 % 2 == 0)
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (has_pro) {
        const int c0 = p.pro_on_small ? 8 * (tid & 3) : 8 * (tid & 1);
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0); s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0); t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 4);
    }
    for (int i = tid; i < C::LDS / 16; i += NTH) *reinterpret_cast<f32x4*>(smem + 16 * i) = f32x4{0.f, 0.f, 0.f, 0.f};      // (the border of the big image = padding)
    // staging destinations: big vector i = row (tid >> 6) + 8 i, pixel (tid & 63) >> 1, half tid & 1; small vector i = position (tid >> 2) + 128 i, chunk tid & 3
    const int bdst = ((((tid >> 6) + 1) * 34 + ((tid & 63) >> 1) + 1) * LDB + 8 * (tid & 1)) * 2;
    const int sdst = C::BIMG + ((tid >> 2) * LDS_ + 8 * (tid & 3)) * 2;
    auto stage = [&](int buf, const VS& V) __attribute__((always_inline)) {
        char* const base = smem + buf * IMG;
#pragma unroll
        for (int i = 0; i < BV; ++i)
            *reinterpret_cast<bf16x8*>(base + bdst + i * (8 * 34 * LDB * 2)) = pro_big ? bn_act8(V.b[i], s0, s1, t0, t1, slope) : V.b[i];
#pragma unroll
        for (int i = 0; i < SV; ++i)
            *reinterpret_cast<bf16x8*>(base + sdst + i * (128 * LDS_ * 2)) = pro_small ? bn_act8(V.s[i], s0, s1, t0, t1, slope) : V.s[i];
    };
    // fragment addresses (byte offsets inside an image): this lane's position of the k group is output row 2 wave + (gq >> 1), column
    // 8 (gq & 1) + (li >> 2) (+ 4 for the second half)
    const int oy = 2 * wave + (gq >> 1), ox = 8 * (gq & 1) + (li >> 2);
    const int soff = C::BIMG + ((16 * oy + ox) * LDS_ + 4 * (li & 3)) * 2;
    int boff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) boff[t] = (((2 * oy + p.ky[t]) * 34 + 2 * ox + p.kx[t]) * LDB + 4 * (li & 3)) * 2;       // (border: input row -1 is LDS row 0)
    f32x4 acc[NH][NT];
#pragma unroll
    for (int a_ = 0; a_ < NH; ++a_)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[a_][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (im < B) stage(0, S0);
    __syncthreads();

    {
        int buf = 0;
        for (; im < B; im += step, buf ^= 1) {
            const int nxt = im + step;
            const bool has_next = nxt < B;
            if (has_next) request(nxt, S0);
            const char* const IB = smem + buf * IMG;
            bf16x8 af[NH];
#pragma unroll
            for (int a_ = 0; a_ < NH; ++a_) af[a_] = k4wg_frag(IB + soff + 32 * a_, 4 * LDS_ * 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 bf_ = k4wg_frag(IB + boff[t], 8 * LDB * 2);        // (4 output columns = 8 input columns further)
#pragma unroll
                for (int a_ = 0; a_ < NH; ++a_) mma32(acc[a_][t], af[a_], bf_);
            }
            if (has_next) stage(buf ^ 1, S0);
            __syncthreads();
        }
    }
    // ---- the eight waves meet in the LDS copy of G: acc[a][t][e] = (s = 16 a + 4 gq + e, b = li).  One wave after the other adds its
    // registers to the copy with 16-byte reads / writes (layout [a][gq][t][b][e]: a lane's four values are contiguous) -- LDS float
    // atomics (128 instructions per wave, four-way bank conflicts between the gq groups) took ~70 us of a ~100 us launch.
    {
        f32x4* const red4 = reinterpret_cast<f32x4*>(smem);
        const float* const red = reinterpret_cast<const float*>(smem);
        for (int w = 0; w < 8; ++w) {
            if (wave == w) {
#pragma unroll
                for (int a_ = 0; a_ < NH; ++a_)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4* const q = red4 + ((a_ * 4 + gq) * NT + t) * BC + li;
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
        // in the ORDER of the master layout (lane-contiguous stores / atomics): element o of dW is (n, tap, c) = (o / (16 Cin), ..)
        float* const slab = p.slabs ? p.slabs + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (SC * NT * BC) : nullptr;
        for (int o = tid; o < SC * NT * BC; o += NTH) {
            int sidx, b, t;
            if (p.transposed) { sidx = o % SC; t = p.slot_of[(o / SC) % NT]; b = o / (SC * NT); }
            else { b = o % BC; t = p.slot_of[(o / BC) % NT]; sidx = o / (BC * NT); }
            const float v = red[((((sidx >> 4) * 4 + ((sidx >> 2) & 3)) * NT + t) * BC + b) * 4 + (sidx & 3)];
            if (slab) slab[o] = v;
            else atomicAdd(p.dw + o, v);
        }
    }
}

}  // namespace

// Returns 1 and sets *rc when the launch is the weight gradient of Conv2d(16, 32, 4, 2, 1) at 32x32 or of ConvTranspose2d(32, 16, 4, 2, 1)
// at 16x16.
int sv_k4wgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                   const void* dy, float* dw, float* ws, int64_t ws_elems, int groups, hipStream_t s, int* rc) {
    typedef k4wg_cfg C;
    if (sv_disabled(SV_K_THWGRAD) || dtype != SV_BF16 || sv_deterministic() || g->T_orig != 16) return 0;
    k4wg_params p;
    p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_slope = pro_slope; p.dw = dw;
    bool seen[16] = {};
    int torig[16];
    if (g->nphase == 1 && g->sy == 2 && g->sx == 2 && g->osy == 1 && g->osx == 1) {
        // the convolution: small = dy, big = x
        if (g->Cin != 16 || g->ldx != 16 || g->N != 32 || g->ldo != 32 || g->Hin != 32 || g->Win != 32 || g->Hout != 16 || g->Wout != 16) return 0;
        const sv_phase& P = g->phase[0];
        if (P.ntap != 16 || P.ooy != 0 || P.oox != 0) return 0;
        for (int t = 0; t < 16; ++t) {
            const int ky = P.dy[t] + 1, kx = P.dx[t] + 1;
            if (ky < 0 || ky > 3 || kx < 0 || kx > 3 || P.torig[t] < 0 || P.torig[t] > 15 || seen[4 * ky + kx]) return 0;
            seen[4 * ky + kx] = true;
            p.ky[t] = (int8_t)ky; p.kx[t] = (int8_t)kx; torig[t] = P.torig[t];
        }
        p.small = dy; p.big = x; p.pro_on_small = 0; p.transposed = 0;
    } else if (g->nphase == 4 && g->sy == 1 && g->sx == 1 && g->osy == 2 && g->osx == 2) {
        // the transposed convolution: small = x, big = dy; phase (py, px), input offset (dy, dx) <-> kernel position (py + 1 - 2 dy, px + 1 - 2 dx)
        if (g->Cin != 32 || g->ldx != 32 || g->N != 16 || g->ldo != 16 || g->Hin != 16 || g->Win != 16 || g->Hout != 32 || g->Wout != 32) return 0;
        int n = 0;
        for (int ph = 0; ph < 4; ++ph) {
            const sv_phase& P = g->phase[ph];
            if (P.ntap != 4 || P.ooy < 0 || P.ooy > 1 || P.oox < 0 || P.oox > 1) return 0;
            for (int t = 0; t < 4; ++t, ++n) {
                const int ky = P.ooy + 1 - 2 * P.dy[t], kx = P.oox + 1 - 2 * P.dx[t];
                if (ky < 0 || ky > 3 || kx < 0 || kx > 3 || P.torig[t] < 0 || P.torig[t] > 15 || seen[4 * ky + kx]) return 0;
                seen[4 * ky + kx] = true;
                p.ky[n] = (int8_t)ky; p.kx[n] = (int8_t)kx; torig[n] = P.torig[t];
            }
        }
        p.small = x; p.big = dy; p.pro_on_small = 1; p.transposed = 1;
    } else {
        return 0;
    }
    for (int t = 0; t < 16; ++t) p.slot_of[torig[t]] = (int8_t)t;
    if (g->B < 1) return 0;
    int per = sv_persistent_blocks() / 2 / groups;                 // one block per CU
    if (per < 1) per = 1;
    if (per > g->B) per = g->B;
    const int rounds = (g->B + per - 1) / per;
    const int grid = (g->B + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k4wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) {
            *rc = sv_check_launch("hipFuncSetAttribute(k4wgrad)");
            return 1;
        }
        optin = true;
    }
    p.slabs = (ws && ws_elems >= (int64_t)grid * groups * 8192) ? ws : nullptr;
    // the groups' operands follow each other (sv_expand_wg works on the fields x / dy of the convolution's view: redo it here)
    sv_wg_g<k4wg_params> PG;
    const int64_t ss = (int64_t)g->B * 256 * 32 * 2, bs = (int64_t)g->B * 1024 * 16 * 2;
    for (int grp = 0; grp < SV_MAX_GROUPS; ++grp) {
        k4wg_params q = p;
        if (grp > 0 && grp < groups) {
            q.small = reinterpret_cast<const char*>(p.small) + grp * ss;
            q.big = reinterpret_cast<const char*>(p.big) + grp * bs;
            if (p.pro_scale) { q.pro_scale = p.pro_scale + grp * g->Cin; q.pro_shift = p.pro_shift + grp * g->Cin; }
        }
        PG.g[grp] = q;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL(k4wgrad_kernel, dim3(grid, groups), dim3(C::NTH), C::LDS, s, g->B, PG);
    sv_prof_end(s);
    if (p.slabs) sv_slab_reduce(p.slabs, grid * groups, 8192, dw, s);
    *rc = sv_check_launch("sv_wgrad(k4wgrad)");
    return 1;
}
