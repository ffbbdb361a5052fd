// Wide stride-1 3x3 convolution (forward and data gradient), bf16, MFMA-bound shapes (Cin >= 96: the WRN-28-10
// body convs 160/320/640 channels, reference shot_vae_model/wideresnet.py:29-35).  gfx950.
//
// conv3x3.hip's tiles (32 pixels x 80 channels per wave) read ~7 LDS fragments per 10 MFMAs: LDS-bound on these
// layers.  This kernel is built like a large-tile GEMM instead:
//   * block = 256 output pixels (whole image rows) x 32*NF output channels, 4 waves, ONE wave per SIMD with the
//     full 512-register file: every wave owns 64 pixels x 32*NF channels = 2 x NF accumulators of
//     v_mfma_f32_32x32x16_bf16 (weights = A operand, pixels = B operand) -> 14 fragment reads per 20 MFMAs;
//   * K loop = (32-channel chunk) x (9 taps).  The input halo of a chunk is staged ONCE (LDS-DMA, raw), gets
//     BatchNorm-apply + LeakyReLU + zero padding in an LDS->LDS pass spread over the MFMA steps of the previous
//     chunk, and serves all nine taps (tap shift = immediate LDS offset);
//   * the [32*NF][32] weight slice of every (chunk, tap) step arrives by LDS-DMA (global_load_lds_dwordx4) three
//     steps ahead into a ring of four buffers, XOR-swizzled on the SOURCE address so the lane-linear LDS image is
//     conflict-free for ds_read_b128;
//   * the first-half (k 0..15) fragments of step k+1 are read into a second register set while step k is on the
//     MFMAs, the second-half fragments at the start of their own step behind the first-half MFMAs; one raw s_barrier
//     per step with counted vmcnt (the DMA queue is never drained inside the loop);
//   * epilogue per 32-channel group through a wave-private LDS transpose: 16-byte coalesced residual / raw-tensor
//     reads and output stores, BatchNorm sums (or activation-backward + BatchNorm-backward sums) in registers.
// Same sv_geom / packed weights / sv_igemm_args contract as the other conv-like kernels.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

__device__ __forceinline__ void glds16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((glb_ptr)gsrc, (lds_ptr)ldst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NF, int WLOG>
struct WCfg {
    static constexpr int BN = 32 * NF;
    static constexpr int W = 1 << WLOG, TR = 256 / W, WP = W + 2;
    static constexpr int HH = TR < W ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;   // images are square
    static constexpr int HPIX = LROWS * WP;                 // halo pixels (incl. padding columns / spacer rows)
    static constexpr int HI = (4 * HPIX + 255) / 256;       // 16-byte slots per thread
    static constexpr int HPIXF = HI * 64;                   // pixels incl. the dummy tail
    static constexpr int LDH = 80;                          // bytes per pixel of the transformed halo (64 + 16 pad:
                                                            // odd 16-byte stride -> conflict-free 32-lane fragments)
    static constexpr int FIN = HPIXF * LDH;
    static constexpr int RAW = HI * 256 * 16;
    static constexpr int WI = (4 * BN + 255) / 256;
    static constexpr int WBUF = WI * 256 * 16;
    static constexpr int OFF_RAW = 2 * FIN, OFF_W = OFF_RAW + RAW, OFF_SSUM = OFF_W + 4 * WBUF,
                         OFF_PS = OFF_SSUM + 2 * BN * 4;
    static constexpr int SCR = 64 * 36 * 4;                 // epilogue transpose scratch per wave
    static_assert(4 * SCR <= 2 * FIN, "epilogue scratch must fit in the halo buffers");
    static_assert(HI <= 6, "transform schedule covers at most 6 slots per thread");
};

template <int NF, int WLOG, bool REV>
__global__ __launch_bounds__(256, 1) void conv3x3w_kernel(const sv_geom g, const sv_igemm_args a) {
    using C = WCfg<NF, WLOG>;
    constexpr int BN = C::BN, W = C::W, TR = C::TR, WP = C::WP, HH = C::HH, SEG = C::SEG, HI = C::HI, WI = C::WI;
    constexpr int LDH = C::LDH, FIN = C::FIN, WBUF = C::WBUF;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const raw = smem + C::OFF_RAW;
    float* const ssum = reinterpret_cast<float*>(smem + C::OFF_SSUM);
    float* const psc = reinterpret_cast<float*>(smem + C::OFF_PS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR, nNt = g.N / BN;
    const int Cin = g.Cin, nck = Cin / 32, KT = 9 * nck;
    float* const psh = psc + Cin;

    // XCD-affine mapping: the 32 CUs of an XCD work on consecutive pixel tiles (shared halo rows and one copy of the
    // weights in that XCD's L2); the channel tiles of a pixel tile are neighbours on the same XCD
    const int per = (nT + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int in_i = slot % nNt, mt = xcd * per + slot / nNt;
    if (mt >= nT) return;
    const int n0 = in_i * BN, gr0 = mt * TR;

    const sv_phase& P = g.phase[0];
    const char* const Xb = reinterpret_cast<const char*>(a.x);
    const char* const Wb = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(a.w) + P.w_off + (int64_t)n0 * 9 * Cin);
    const bool has_pro = a.pro_scale != nullptr;
    const float slope = has_pro ? a.pro_slope : 1.f;

    for (int c = tid; c < Cin; c += 256) {
        psc[c] = has_pro ? a.pro_scale[c] : 1.f;
        psh[c] = has_pro ? a.pro_shift[c] : 0.f;
    }
    for (int c = tid; c < 2 * BN; c += 256) ssum[c] = 0.f;

    // ---- per-thread staging slots: slot s = 256 j + tid is DMA'd AND transformed by this thread ------------------
    uint32_t hsrc[HI];          // byte offset into x (channel chunk 0)
    uint32_t hokm = 0;          // bit j: slot j is a real pixel (else: zero padding / spacer / dummy)
    {
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
#pragma unroll
        for (int j = 0; j < HI; ++j) {
            const int s = 256 * j + tid, pix = s >> 2, q = s & 3;
            const int lr = pix / WP, xx = pix - lr * WP;
            const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
            int kind = 1, rel = lr - 1 - seg;
            if (off == 0) {
                if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
                else kind = 0;
            }
            if (pix >= C::HPIX || xx == 0 || xx == WP - 1) kind = 0;
            if (kind == 1 || (kind == 2 && top_ok) || (kind == 3 && bot_ok)) hokm |= 1u << j;
            const int grc = min(max(gr0 + rel, 0), BH - 1), xc = min(max(xx - 1, 0), W - 1);
            hsrc[j] = (uint32_t)((grc * W + xc) * g.ldx + 8 * q) * 2u;
        }
    }
    const int hdst0 = (tid >> 2) * LDH + 16 * (tid & 3);          // slot j lands at hdst0 + 64 j LDH
    const int qc = 8 * (tid & 3);                                  // this thread's 8-channel group inside a chunk
    // weight slots: row = 64 i + tid/4 (clamped for the dummy tail of the last slot), source k-quarter XOR-swizzled
    const uint32_t wq = (uint32_t)((tid & 3) ^ ((tid >> 4) & 3));
    const uint32_t wsrc0 = (uint32_t)((tid >> 2) * 9 * Cin + 8 * wq) * 2u;
    const uint32_t wsrcL = (uint32_t)(min(64 * (WI - 1) + (tid >> 2), BN - 1) * 9 * Cin + 8 * wq) * 2u;
    const uint32_t wstep = (uint32_t)(64 * 9 * Cin) * 2u;
    auto issue_h = [&](int c) {
#pragma unroll
        for (int j = 0; j < HI; ++j) {
            const uint32_t o = hsrc[j] + (uint32_t)(c * 64);      // one 32-bit offset: SGPR base + VGPR offset form
            glds16(Xb + o, raw + (j * 4 + wave) * 1024);
        }
    };
    auto issue_w = [&](int c, int t, int buf) {
        const uint32_t o = (uint32_t)(t * Cin + c * 32) * 2u;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const uint32_t oo = (i == WI - 1 ? wsrcL : wsrc0 + (uint32_t)i * wstep) + o;
            glds16(Wb + oo, smem + C::OFF_W + buf * WBUF + (i * 4 + wave) * 1024);
        }
    };
    bf16x8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (bf16)0.f;
    auto transform = [&](int c, auto jc, char* fin) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        // BatchNorm scale / shift of this thread's 8 channels of chunk c (LDS-resident: no registers held across steps)
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(psc + 32 * c + qc);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(psc + 32 * c + qc + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(psh + 32 * c + qc);
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(psh + 32 * c + qc + 4);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(raw + 16 * (256 * j + tid));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u0 = (float)v[e] * sc0[e] + sh0[e], u1 = (float)v[e + 4] * sc1[e] + sh1[e];
            o[e] = (bf16)fmaxf(u0, u0 * slope);        // LeakyReLU / ReLU / identity for slope in [0, 1]
            o[e + 4] = (bf16)fmaxf(u1, u1 * slope);
        }
        *reinterpret_cast<bf16x8*>(fin + hdst0 + j * 64 * LDH) = ((hokm >> j) & 1u) ? o : zero;
    };

    // ---- fragment addressing (tap / chunk parts are immediates, the weight ring slot one add per step) ------------
    int bb[2];                  // pixel fragments: byte offset of (pixel - one halo row - one column) + lane half
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int p = 64 * wave + 32 * f + r, prow = p >> WLOG, pcol = p & (W - 1);
        bb[f] = ((prow + prow / HH) * WP + pcol) * LDH + 16 * h;
    }
    const int ab0 = C::OFF_W + 16 * (4 * r + (h ^ ((r >> 2) & 3)));           // k sub-step 0 (channels 0..15)
    const int ab1 = C::OFF_W + 16 * (4 * r + ((2 + h) ^ ((r >> 2) & 3)));     // k sub-step 1 (channels 16..31)

    // register sets: the k-sub-step-0 fragments are double-buffered ACROSS steps (read during the previous step),
    // the k-sub-step-1 fragments are read at the start of their own step, behind the sub-step-0 MFMAs
    bf16x8 A0[2][NF], B0[2][2], A1[NF], B1[2];
    f32x16 acc[2][NF];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][i][e] = 0.f;

    // tap shift + halo buffer of (tap t, chunk parity par) as an immediate
    auto load_k0 = [&](auto setc, auto tc, auto parc, int ring) __attribute__((always_inline)) {
        constexpr int set = decltype(setc)::value, t = decltype(tc)::value, par = decltype(parc)::value;
        constexpr int sh = (REV ? ((2 - t / 3) * WP + (2 - t % 3)) : ((t / 3) * WP + t % 3)) * LDH + par * FIN;
        const int aw = ab0 + ring * WBUF;
#pragma unroll
        for (int i = 0; i < NF; ++i) A0[set][i] = *reinterpret_cast<const bf16x8*>(smem + aw + i * 2048);
#pragma unroll
        for (int f = 0; f < 2; ++f) B0[set][f] = *reinterpret_cast<const bf16x8*>(smem + bb[f] + sh);
    };
    auto load_k1 = [&](auto tc, auto parc, int ring) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, par = decltype(parc)::value;
        constexpr int sh = (REV ? ((2 - t / 3) * WP + (2 - t % 3)) : ((t / 3) * WP + t % 3)) * LDH + par * FIN + 32;
        const int aw = ab1 + ring * WBUF;
#pragma unroll
        for (int i = 0; i < NF; ++i) A1[i] = *reinterpret_cast<const bf16x8*>(smem + aw + i * 2048);
#pragma unroll
        for (int f = 0; f < 2; ++f) B1[f] = *reinterpret_cast<const bf16x8*>(smem + bb[f] + sh);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue ---------------------------------------------------------------------------------------------
    __syncthreads();                                   // psc / psh / ssum visible (no DMA in flight yet)
    issue_h(0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);
    wait_vm<0>();
    transform(0, std::integral_constant<int, 0>{}, smem);
    if (HI > 1) transform(0, std::integral_constant<int, (HI > 1 ? 1 : 0)>{}, smem);
    if (HI > 2) transform(0, std::integral_constant<int, (HI > 2 ? 2 : 0)>{}, smem);
    if (HI > 3) transform(0, std::integral_constant<int, (HI > 3 ? 3 : 0)>{}, smem);
    if (HI > 4) transform(0, std::integral_constant<int, (HI > 4 ? 4 : 0)>{}, smem);
    if (HI > 5) transform(0, std::integral_constant<int, (HI > 5 ? 5 : 0)>{}, smem);
    wait_lds();
    barrier();
    load_k0(I0{}, I0{}, I0{}, 0);
    wait_lds();

    // ---- one (chunk, tap) step: weights of step k live in ring slot k & 3 -------------------------------------------
    auto step = [&](int c, auto tc, auto parc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, par = decltype(parc)::value;
        constexpr int cur = (t + par) & 1, nxt = cur ^ 1;
        constexpr int t1 = (t + 1) % 9, par1 = t == 8 ? par ^ 1 : par;
        const int k = c * 9 + t;
        const bool more_c = c + 1 < nck;
        const bool w_issue = k + 3 < KT;
        // (1) asynchronous copies: weights three steps ahead (into the slot step k-1 released), the next chunk's raw
        //     halo at the chunk's first step
        if (w_issue) issue_w(c + (t + 3) / 9, (t + 3) % 9, (k + 3) & 3);
        if (t == 0 && more_c) issue_h(c + 1);
        // (2) this step's second-half fragments, then the next step's first-half fragments (at the very last step the
        //     latter reads stale but in-bounds LDS and is never used)
        load_k1(tc, parc, k & 3);
        load_k0(std::integral_constant<int, nxt>{}, std::integral_constant<int, t1>{},
                std::integral_constant<int, par1>{}, (k + 1) & 3);
        // (3) BatchNorm-apply + LeakyReLU + padding of the next chunk's halo, spread over steps 3..7 (unconditional:
        //     after the last chunk it rewrites an unused buffer)
        if (t >= 3 && t <= 7) {
            char* fin = smem + (par ^ 1) * FIN;
            const int cn = min(c + 1, nck - 1);
            if (t == 3) {
                transform(cn, I0{}, fin);
                if (HI > 1) transform(cn, std::integral_constant<int, (HI > 1 ? 1 : 0)>{}, fin);
            } else if (t - 2 < HI) {
                transform(cn, std::integral_constant<int, (t - 2 < HI ? (t >= 4 ? t - 2 : 0) : 0)>{}, fin);
            }
        }
        // (4) this step's 4 NF MFMAs
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < NF; ++i)
                acc[f][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[cur][i], B0[cur][f], acc[f][i], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < NF; ++i)
                acc[f][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[i], B1[f], acc[f][i], 0, 0, 0);
        // (5) the weights of step k+2 (first read from LDS during step k+1) must have landed; everything issued
        //     after them may stay in flight: this step's weights, and the raw halo for two more steps
        const bool h_fly = t <= 1 && more_c;
        if (w_issue) {
            if (h_fly) wait_vm<WI + HI>(); else wait_vm<WI>();
        } else {
            if (h_fly) wait_vm<HI>(); else wait_vm<0>();
        }
        wait_lds();
        barrier();
    };
    auto chunk = [&](int c, auto parc) __attribute__((always_inline)) {
        step(c, std::integral_constant<int, 0>{}, parc);
        step(c, std::integral_constant<int, 1>{}, parc);
        step(c, std::integral_constant<int, 2>{}, parc);
        step(c, std::integral_constant<int, 3>{}, parc);
        step(c, std::integral_constant<int, 4>{}, parc);
        step(c, std::integral_constant<int, 5>{}, parc);
        step(c, std::integral_constant<int, 6>{}, parc);
        step(c, std::integral_constant<int, 7>{}, parc);
        step(c, std::integral_constant<int, 8>{}, parc);
    };
    for (int c = 0; c < nck; c += 2) {
        chunk(c, I0{});
        if (c + 1 < nck) chunk(c + 1, I1{});
    }

    // ---- epilogue: per 32-channel group through a wave-private LDS transpose ------------------------------------------
    float* const scr = reinterpret_cast<float*>(smem + wave * C::SCR);      // [64 pixels][36]
    bf16* const O = reinterpret_cast<bf16*>(a.out);
    const bf16* const R = reinterpret_cast<const bf16*>(a.residual);
    const bf16* const EX = reinterpret_cast<const bf16*>(a.ex);
    const bool want_stats = a.stats != nullptr && EX == nullptr;
    const int ipix = lane >> 2, cg = lane & 3;
    const int64_t gp0 = (int64_t)gr0 * W + 64 * wave;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int nl = 32 * i + 8 * cg, n = n0 + nl;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 v = {acc[f][i][4 * gq], acc[f][i][4 * gq + 1], acc[f][i][4 * gq + 2], acc[f][i][4 * gq + 3]};
                *reinterpret_cast<f32x4*>(scr + (32 * f + r) * 36 + 8 * gq + 4 * h) = v;
            }
        float bias[8], esc[8], esh[8], emu[8], ers[8];
        auto load8 = [&](const float* p, float (&d)[8]) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(p + n), hi = *reinterpret_cast<const f32x4*>(p + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[e] = lo[e]; d[e + 4] = hi[e]; }
        };
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = 0.f;
        if (a.bias) load8(a.bias, bias);
        if (EX) {
            load8(a.ex_scale, esc);
            load8(a.ex_shift, esh);
            load8(a.ex_mean, emu);
            load8(a.ex_rstd, ers);
        }
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        bf16x8 eop[4];
        if (R || EX) {
            const bf16* src = R ? R : EX;
#pragma unroll
            for (int it = 0; it < 4; ++it)
                eop[it] = *reinterpret_cast<const bf16x8*>(src + (gp0 + 16 * it + ipix) * g.ldo + n);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pix = 16 * it + ipix;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + pix * 36 + 8 * cg);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + pix * 36 + 8 * cg + 4);
            float vv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vv[e] = v0[e] + bias[e];
                vv[e + 4] = v1[e] + bias[e + 4];
            }
            if (R) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] += (float)eop[it][e];
            }
            if (EX) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xf = (float)eop[it][e];
                    const float gv = vv[e] * act_grad(xf * esc[e] + esh[e], a.ex_slope);
                    vv[e] = gv;
                    s1[e] += gv;
                    s2[e] += gv * ((xf - emu[e]) * ers[e]);
                }
            } else if (want_stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += vv[e];
                    s2[e] += vv[e] * vv[e];
                }
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)vv[e];
            *reinterpret_cast<bf16x8*>(O + (gp0 + pix) * g.ldo + n) = o;
        }
        if (want_stats || EX) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {
                    s1[e] += __shfl_xor(s1[e], o);
                    s2[e] += __shfl_xor(s2[e], o);
                }
            }
            if (ipix == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(&ssum[nl + e], s1[e]);
                    atomicAdd(&ssum[BN + nl + e], s2[e]);
                }
            }
        }
    }
    if (want_stats || EX) {
        __syncthreads();
        float* dst = (EX ? a.bsums : a.stats) + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * g.N;
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, nl = i - which * BN;
            atomicAdd(dst + which * g.N + n0 + nl, ssum[i]);
        }
    }
}

template <int NF, int WLOG, bool REV>
int launch_w3(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    using C = WCfg<NF, WLOG>;
    const int nT = g->B * g->Hin / C::TR, nNt = g->N / C::BN;
    const int grid = 8 * ((nT + 7) / 8) * nNt;
    const size_t lds = (size_t)C::OFF_PS + (size_t)g->Cin * 8;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3w_kernel<NF, WLOG, REV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3w)");
        optin = true;
    }
    if (lds > 160 * 1024) return -1;
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3w_kernel<NF, WLOG, REV>), dim3(grid), dim3(256), lds, s, *g, *a);
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(conv3x3w)");
}

template <int NF, bool REV>
int launch_w2(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    switch (g->Win) {
        case 32: return launch_w3<NF, 5, REV>(g, a, s);
        case 16: return launch_w3<NF, 4, REV>(g, a, s);
        default: return launch_w3<NF, 3, REV>(g, a, s);
    }
}

}  // namespace

// Returns 1 and sets *rc when the geometry is a wide bf16 stride-1 3x3 convolution this kernel covers.
// (The caller, sv_conv3x3_try, has already checked the generic stride-1 3x3 / square-image conditions.)
int sv_conv3x3w_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    static const bool off = getenv("SV_NO_CONV3X3W") != nullptr;
    if (off || dtype != SV_BF16) return 0;
    if (g->Cin < 96 || g->Cin % 32 != 0 || g->ldx % 8 != 0 || g->ldo % 8 != 0) return 0;
    if (g->N % 160 != 0 && g->N % 128 != 0) return 0;
    const int TR = 256 / g->Win;
    if ((g->B * g->Hin) % TR != 0) return 0;
    if ((int64_t)g->B * g->Hin * g->Win * g->ldx * 2 >= ((int64_t)1 << 31)) return 0;
    if ((int64_t)g->N * 9 * g->Cin * 2 >= ((int64_t)1 << 31)) return 0;
    if ((size_t)g->Cin * 8 + 144 * 1024 > 160 * 1024) return 0;
    // tap order: canonical (forward) or reversed (data gradient)
    const sv_phase& P = g->phase[0];
    bool fwd = true, rev = true;
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        fwd = fwd && P.dy[t] == dy && P.dx[t] == dx;
        rev = rev && P.dy[t] == -dy && P.dx[t] == -dx;
    }
    if (!fwd && !rev) return 0;
    if (g->N % 160 == 0) *rc = fwd ? launch_w2<5, false>(g, a, s) : launch_w2<5, true>(g, a, s);
    else *rc = fwd ? launch_w2<4, false>(g, a, s) : launch_w2<4, true>(g, a, s);
    return 1;
}
