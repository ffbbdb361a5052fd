// Stride-1 3x3 convolution (forward / data gradient) of the 160-channel-tile layers (the WRN-28-10 body), bf16, gfx950:
// the ONE-WAVE-PER-SIMD variant of conv3x3w.hip, built by the rules tools/probes/issue_probe.hip measured.
//
// conv3x3w runs two 256-register blocks per CU and hopes that one block's LDS reads / DMA issue / BatchNorm pass run
// under the other block's MFMAs; measured, they do not (a block's step costs the SUM of both waves' matrix time and
// other instructions).  Here a CU runs one block of four waves, each with the whole 512-register file:
//   * block = 256 pixels x 160 channels, wave = 64 pixels x 160 channels = 2 x 5 accumulator tiles of
//     v_mfma_f32_32x32x16_bf16, pinned in the AGPR half by spelling the MFMA in assembly;
//   * every other instruction of the K loop sits in the GAP behind one MFMA (a 32x32x16 MFMA holds the matrix pipe for 32
//     cycles; behind it one wave issues two ds_read_b128 or one + four VALU instructions for free, a vector-memory
//     instruction behind every second one): the 14 fragment reads of the NEXT tap (one register set: a fragment is
//     re-loaded right after its last MFMA of this tap), the BatchNorm + LeakyReLU pass over the next 32-channel chunk's
//     halo as single-instruction micro-steps (global -> registers -> LDS; the registers are re-loaded a whole chunk
//     ahead), the weight DMA.  The placement is a compile-time program (make_xsched) with hand-counted vmcnt waits;
//   * LDS (1 block per CU): two halo stages (21.7 KB) + a NINE-slot weight ring, slot = tap (92 KB): every LDS address
//     is a lane base + an immediate, and three barriers per chunk (after taps 1, 4, 7) are all the synchronisation:
//     the slice of (chunk + 1, tap t) is copied as soon as the barrier after tap t's fragment reads has passed;
//   * the epilogue is conv3x3w's (conv3x3w_epilogue.inc).
// Same fused prologue / epilogue contract as sv_igemm (include/shotvae_hip.h); replaces
// shot_vae_model/wideresnet.py:13-43 (Conv2d 3x3 + BatchNorm2d + LeakyReLU + residual) for the wide layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

#ifndef SV_X3_EPD
#define SV_X3_EPD 5        // residual rows of all five 32-channel groups requested up front (512 registers: 80 are spare)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

// ---- the chunk program ------------------------------------------------------------------------------------------------
// gap g = 20 * tap + m follows MFMA m of the tap (m = 10 ks + 2 i + f).  Fragment reads of the next tap are fixed:
// weights (ks, i) in gap 10 ks + 2 i + 1, pixels (ks, f) in gap 10 ks + 8 + f.  Everything else is an item:
//   1000 + 41 slot + step : one instruction of the BatchNorm pass over halo vector `slot` (step 40 = its LDS store)
//   2000 + 3 s + i        : DMA instruction i of weight slice s: s < 6 -> (next chunk, tap s);  s = 6 -> (next chunk, 6);
//                           s = 7, 8 -> (THIS chunk, tap s), issued in taps 0, 1
//   3000 + slot           : global load of halo vector `slot` of the chunk after next
//   3500 + q              : global load q of the BatchNorm coefficients of the chunk after next
//   4000 + slot / 4500    : the vmcnt wait before the slot's / the coefficients' first use
//   5000                  : point the pixel-fragment bases at the other halo stage (start of tap 8)
constexpr int X_HI = 6, X_HSTEPS = 41, X_NGAP = 180;
constexpr bool x_gap_has_read(int m) { return (m & 1) || m == 8 || m == 18; }
struct XSched {
    int item[X_NGAP][6];
    int vm_slot[X_HI], vm_coef, vm_b1, vm_b4, vm_b7;
    bool ok;
};
constexpr XSched make_xsched() {
    XSched S{};
    for (int i = 0; i < X_NGAP; ++i)
        for (int j = 0; j < 6; ++j) S.item[i][j] = 0;
    int used[X_NGAP] = {};      // 1 = holds a VMEM instruction
    auto put = [&](int gap, int code) {
        for (int j = 0; j < 6; ++j)
            if (S.item[gap][j] == 0) { S.item[gap][j] = code; return true; }
        return false;
    };
    bool ok = true;
    // weight DMA: one slice per tap, its three instructions in the read-free gaps 0, 2, 4
    for (int t = 0; t < 9; ++t) {
        const int s = t == 0 ? 7 : t == 1 ? 8 : t - 2;
        for (int i = 0; i < 3; ++i) { ok = put(20 * t + 2 * i, 2000 + 3 * s + i) && ok; used[20 * t + 2 * i] = 1; }
    }
    ok = put(20 * 8 + 6, 5000) && ok;
    // the BatchNorm pass starts with tap 1 (the coefficients were requested at the end of the previous chunk's pass):
    // two or three steps per gap next to fragment reads, four otherwise; a halo register is re-loaded in the first read-free,
    // VMEM-free gap after its vector's store, the coefficients after the last vector
    int m = 0, next_reload = 0, coef_q = 0;
    const int nsteps = X_HSTEPS * X_HI;
    for (int gp = 20; gp < 160; ++gp) {
        const int mm = gp % 20;
        const bool fre = !x_gap_has_read(mm) && !used[gp];
        if (fre && next_reload < X_HI && m >= X_HSTEPS * (next_reload + 1)) {
            ok = put(gp, 3000 + next_reload) && ok;
            used[gp] = 1;
            ++next_reload;
            continue;
        }
        if (fre && m >= nsteps && next_reload == X_HI && coef_q < 4) {
            ok = put(gp, 3500 + coef_q) && ok;
            used[gp] = 1;
            ++coef_q;
            continue;
        }
        if (used[gp]) continue;
        const int cap = (mm == 9 || mm == 19) ? 2 : x_gap_has_read(mm) ? 3 : 4;     // two reads / one read / none
        for (int c = 0; c < cap && m < nsteps; ++c) {
            if (m == 0) ok = put(gp, 4500) && ok;
            if (m % X_HSTEPS == 0) { ok = put(gp, 4000 + m / X_HSTEPS) && ok; }
            ok = put(gp, 1000 + m) && ok;
            ++m;
            if (m % X_HSTEPS == 0) break;          // a slot's store closes its gap
        }
    }
    S.ok = ok && m == nsteps && next_reload == X_HI && coef_q == 4;
    // ---- vmcnt bookkeeping over the steady-state order
    int order[96] = {}, pos[96] = {};
    int nv = 0;
    for (int gp = 0; gp < X_NGAP; ++gp)
        for (int j = 0; j < 6; ++j) {
            const int c = S.item[gp][j];
            if (c >= 2000 && c < 4000) { order[nv] = c; pos[nv] = gp * 8 + j; ++nv; }
        }
    auto find_item = [&](int code) {
        for (int gp = 0; gp < X_NGAP; ++gp)
            for (int j = 0; j < 6; ++j) if (S.item[gp][j] == code) return gp * 8 + j;
        return -1;
    };
    // VMEM issued after order[idx] until position p of the NEXT iteration (p < 0: until the end of this one + |p| ...)
    auto since = [&](int idx, int p, bool wrap) {
        int cnt = 0;
        if (wrap) {
            cnt = nv - 1 - idx;
            for (int i = 0; i < nv; ++i) if (pos[i] < p) ++cnt;
        } else {
            for (int i = idx + 1; i < nv; ++i) if (pos[i] < p) ++cnt;
        }
        return cnt;
    };
    auto index_of = [&](int code) { for (int i = 0; i < nv; ++i) if (order[i] == code) return i; return -1; };
    for (int sl = 0; sl < X_HI; ++sl) S.vm_slot[sl] = since(index_of(3000 + sl), find_item(4000 + sl), true);
    S.vm_coef = since(index_of(3503), find_item(4500), true);
    // barrier after tap 1: slices 3..5 of this chunk (issued in taps 5..7 of the previous one) must have landed
    S.vm_b1 = since(index_of(2000 + 3 * 5 + 2), 40 * 8, true);
    // barrier after tap 4: slices 6 (previous chunk's tap 8) and 7, 8 (taps 0, 1)
    S.vm_b4 = since(index_of(2000 + 3 * 8 + 2), 100 * 8, false);
    // barrier after tap 7: slices 0..2 of the next chunk (taps 2..4)
    S.vm_b7 = since(index_of(2000 + 3 * 2 + 2), 160 * 8, false);
    return S;
}

template <int WLOG>
struct XCfg {
    static constexpr int NF = 5, BN = 160;
    static constexpr int W = 1 << WLOG, TR = 256 / W, WP = W + 2;
    static constexpr int HH = TR < W ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;   // images are square
    static constexpr int HPIX = LROWS * WP;                 // halo pixels (incl. padding columns / spacer rows)
    static constexpr int HS = 4 * HPIX;                     // 16-byte vectors of one halo stage (64 B per pixel)
    static constexpr int HB = HS * 16 + 1024;               // bytes per halo stage (+ a dummy KB for the unused slots)
    static constexpr int SWS = WLOG == 5 ? 2 : 1;           // pixel swizzle: k-quarter ^= (halo column >> SWS) & 3
    static constexpr int WS = 4 * BN, WI = 3, WBUF = WS * 16;
    static constexpr int OFF_W = 2 * HB, OFF_SSUM = OFF_W + 9 * WBUF, LDS = OFF_SSUM + 2 * BN * 4;
    static constexpr int SCR = 64 * 36 * 4;                 // epilogue transpose scratch per wave
    static_assert((HS + 255) / 256 == X_HI, "six halo vectors per thread");
    static_assert(LDS <= 160 * 1024, "one block per CU");
};

template <int WLOG, bool REV>
__global__ __launch_bounds__(256, 1) void conv3x3x_kernel(const sv_geom g, const sv_igemm_args a) {
    using C = XCfg<WLOG>;
    constexpr int NF = C::NF, BN = C::BN, W = C::W, TR = C::TR, WP = C::WP, HH = C::HH, SEG = C::SEG;
    constexpr int HS = C::HS, HB = C::HB, WBUF = C::WBUF, SWS = C::SWS, HI = X_HI, HSTEPS = X_HSTEPS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ssum = reinterpret_cast<float*>(smem + C::OFF_SSUM);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: DMA destinations stay in SGPRs
    const int r = lane & 31, h = lane >> 5;
    const int H = g.Hin, BH = g.B * H, nT = BH / TR, nNt = g.N / BN;
    const int Cin = g.Cin, nck = Cin / 32;

    // XCD-affine mapping (as conv3x3w): the 32 CUs of an XCD work on consecutive pixel tiles
    const int per = (nT + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int in_i = slot_id % nNt, mt = xcd * per + slot_id / nNt;
    if (mt >= nT) return;
    const int n0 = in_i * BN, gr0 = mt * TR;

    const sv_phase& P = g.phase[0];
    const char* const Xb = reinterpret_cast<const char*>(a.x);
    const char* const Wb = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(a.w) + P.w_off + (int64_t)n0 * 9 * Cin);
    const bool has_pro = a.pro_scale != nullptr;
    const float slope = has_pro ? a.pro_slope : 1.f;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;

#ifdef SV_X3_STAGGER       // experiment: desynchronise the CUs (first-round blocks start SV_X3_STAGGER cycles apart, 4 phases)
    if (blockIdx.x < 256) {
        const uint64_t until = __builtin_amdgcn_s_memtime() + (uint64_t)((blockIdx.x >> 3) & 3) * SV_X3_STAGGER;
        while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
    }
#endif
    for (int c = tid; c < 2 * BN; c += 256) ssum[c] = 0.f;
#ifdef SV_X3_STAMP
    const uint64_t st0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- halo vectors of this thread: vector s = 256 j + tid = (halo pixel s >> 2, logical 8-channel quarter tid & 3),
    //      stored at the swizzled quarter; kind: 0 zero (padding column / spacer / dummy), 1..3 image rows -----------------
    const int lq = tid & 3;
    uint32_t hoff[HI];          // byte offset into x of channel chunk 0
    int hlds[HI];               // byte offset inside a halo stage
    bool hok[HI];
    {
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
#pragma unroll
        for (int j = 0; j < HI; ++j) {
            const int s = 256 * j + tid, pix = min(s, HS - 1) >> 2;
            const int lr = pix / WP, xx = pix - lr * WP;
            const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
            int kind = 1, rel = lr - 1 - seg;
            if (off == 0) {
                if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
                else kind = 0;
            }
            if (xx == 0 || xx == WP - 1) kind = 0;
            if (s >= HS) kind = 0;
            hok[j] = (kind == 1) | ((kind == 2) & top_ok) | ((kind == 3) & bot_ok);
            const int grc = min(max(gr0 + rel, 0), BH - 1), xc = min(max(xx - 1, 0), W - 1);
            hoff[j] = hok[j] ? (uint32_t)((grc * W + xc) * g.ldx + 8 * lq) * 2u : (uint32_t)(gr0 * W * g.ldx + 8 * lq) * 2u;
            hlds[j] = s < HS ? pix * 64 + 16 * (lq ^ ((xx >> SWS) & 3)) : HS * 16 + 16 * (tid & 63);
        }
    }
    // ---- weight DMA: 64 consecutive 16-byte vectors of a [160][32] slice per wave instruction; the k-quarter swizzle
    //      (row >> 2) & 3 is applied to the source address
    uint32_t wsrc[3];
    uint32_t wdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int base = min((i * 4 + wave) * 64, C::WS - 64);
        const int s = base + lane, row = s >> 2, q = (s & 3) ^ ((row >> 2) & 3);
        wsrc[i] = (uint32_t)(row * 9 * Cin + 8 * q) * 2u;
        wdst[i] = lds0 + C::OFF_W + (uint32_t)base * 16u;
    }
    auto dma_w = [&](int c, int t, auto I) {             // instruction i of the slice of (chunk c, tap t) -> ring slot t
        constexpr int i = decltype(I)::value;
        const char* src = Wb + (int64_t)(t * Cin + c * 32) * 2;
        const uint32_t dst = wdst[i] + (uint32_t)t * (uint32_t)WBUF, off = wsrc[i];
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(off), "s"(src) : "memory");
    };
    // ---- halo registers + BatchNorm coefficients of the thread's quarter
    u32x4 rh[HI];
    f32x4 csc[2], csh[2];
    {
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zer = {0.f, 0.f, 0.f, 0.f};
        csc[0] = csc[1] = one;
        csh[0] = csh[1] = zer;
    }
    auto load_h = [&](u32x4* rh, int c, auto J) {
        constexpr int j = decltype(J)::value;
        const char* src = Xb + (int64_t)c * 64;
        const uint32_t off = hoff[j];
        u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(src) : "memory");
        rh[j] = v;
    };
    auto wait_h = [&](u32x4* rh, auto J, auto N) {
        constexpr int j = decltype(J)::value, n = decltype(N)::value;
        u32x4 v = rh[j];
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(n) : "memory");
        rh[j] = v;
    };
    auto load_coef = [&](f32x4* csc, f32x4* csh, int c, auto Q) {                // q: 0,1 = scale lo/hi, 2,3 = shift lo/hi
        constexpr int q = decltype(Q)::value;
        if (!has_pro) return;
        const char* src = reinterpret_cast<const char*>(q < 2 ? a.pro_scale : a.pro_shift) + (int64_t)c * 128;
        const uint32_t off = (uint32_t)(8 * lq + 4 * (q & 1)) * 4u;
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(src) : "memory");
        if (q < 2) csc[q & 1] = v; else csh[q & 1] = v;
    };
    auto wait_coef = [&](f32x4* csc, f32x4* csh, auto N) {
        constexpr int n = decltype(N)::value;
        f32x4 s0 = csc[0], s1 = csc[1], t0 = csh[0], t1 = csh[1];
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s0), "+v"(s1), "+v"(t0), "+v"(t1) : "n"(n) : "memory");
        csc[0] = s0; csc[1] = s1; csh[0] = t0; csh[1] = t1;
    };
    // single-instruction steps of the BatchNorm + LeakyReLU pass (see wgrad3x3.hip): per dword d of vector j
    //   0,1: lo = x << 16, hi = x & 0xffff0000   2,3: u = f * scale + shift   4,5: m = u * slope   6,7: u = max(u, m)
    //   8: od = pack_bf16(u)   9: od = valid ? od : 0;   step 40 stores the vector into the stage `stage_off`
    float xlo, xhi, xmlo, xmhi;
    u32x4 od;
    auto hstep = [&](const u32x4* rh, const f32x4* csc, const f32x4* csh, auto J, auto ST, uint32_t stage_off) {
        constexpr int j = decltype(J)::value, st = decltype(ST)::value, d = st / 10, q = st % 10;
        if constexpr (st == 40) {
            *reinterpret_cast<u32x4*>(smem + stage_off + hlds[j]) = od;
        } else {
            const float sc_lo = csc[d >> 1][2 * (d & 1)], sc_hi = csc[d >> 1][2 * (d & 1) + 1];
            const float sh_lo = csh[d >> 1][2 * (d & 1)], sh_hi = csh[d >> 1][2 * (d & 1) + 1];
            if constexpr (q == 0) xlo = __builtin_bit_cast(float, rh[j][d] << 16);
            if constexpr (q == 1) xhi = __builtin_bit_cast(float, rh[j][d] & 0xffff0000u);
            if constexpr (q == 2) xlo = __builtin_fmaf(xlo, sc_lo, sh_lo);
            if constexpr (q == 3) xhi = __builtin_fmaf(xhi, sc_hi, sh_hi);
            if constexpr (q == 4) xmlo = xlo * slope;
            if constexpr (q == 5) xmhi = xhi * slope;
            if constexpr (q == 6) xlo = fmaxf(xlo, xmlo);
            if constexpr (q == 7) xhi = fmaxf(xhi, xmhi);
            if constexpr (q == 8) {
                typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
                const bf16x2v pk = {(__bf16)xlo, (__bf16)xhi};
                od[d] = __builtin_bit_cast(uint32_t, pk);
            }
            if constexpr (q == 9) od[d] = hok[j] ? od[d] : 0u;
        }
    };

    // ---- fragment addressing (byte addresses in LDS; the tap / slot / k half / channel group are immediates) --------------
    // weights: lane (r, h) reads row 32 i + r, k-quarter (2 ks + h) ^ swizzle(row); two bases keep the immediates < 64 KB
    uint32_t wa[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        wa[ks][0] = lds0 + C::OFF_W + 16 * (4 * r + ((2 * ks + h) ^ ((r >> 2) & 3)));
        wa[ks][1] = wa[ks][0] + 5 * WBUF;
    }
    // pixels: 64 * (pixel - one halo row - one column) + 16 * ((2 ks + h) ^ swizzle of the tap's column)
    uint32_t px[2][3][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int p = 64 * wave + 32 * f + r, prow = p >> WLOG, pc = p & (W - 1);
        const int bb = ((prow + prow / HH) * WP + pc) * 64;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                px[f][tx][ks] = lds0 + (uint32_t)(bb + 16 * ((2 * ks + h) ^ (((pc + tx) >> SWS) & 3)));
    }
    typedef __attribute__((address_space(3))) bf16x8 lds_v8;
    bf16x8 A[2][NF], Bf[2][2];
    auto read_w = [&](int t, int ks, int i) {
        const int off = (t < 5 ? t : t - 5) * WBUF + i * 2048;
        A[ks][i] = *reinterpret_cast<const lds_v8*>((uintptr_t)(wa[ks][t < 5 ? 0 : 1] + (uint32_t)off));
    };
    auto read_p = [&](int t, int ks, int f) {
        const int ty = REV ? 2 - t / 3 : t / 3, tx = REV ? 2 - t % 3 : t % 3;
        Bf[ks][f] = *reinterpret_cast<const lds_v8*>((uintptr_t)(px[f][tx][ks] + (uint32_t)((ty * WP + tx) * 64)));
    };

    f32x16 acc[2][NF];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][i][e] = 0.f;

    static constexpr XSched SCHED = make_xsched();
    static_assert(SCHED.ok, "the chunk program does not fit the gaps");

    // ---- prologue: every request first -- chunk 0's halo + coefficients (into a second register set), the halo registers
    //      + coefficients of chunk 1, slices 0..6 of chunk 0 (7, 8 come with taps 0, 1) -- then chunk 0's BatchNorm pass
    //      into stage 0 while the rest is still in flight, and one full wait: the waits of the loop count the VMEM
    //      instructions of a steady-state chunk, which the first chunk has not issued yet
    __syncthreads();                                   // ssum visible
    const int c1 = min(1, nck - 1);
    {
        u32x4 rh0[HI];
        f32x4 csc0[2] = {csc[0], csc[1]}, csh0[2] = {csh[0], csh[1]};
        static_for<HI>([&](auto J) { load_h(rh0, 0, J); });
        static_for<4>([&](auto Q) { load_coef(csc0, csh0, 0, Q); });
        static_for<7>([&](auto S) { static_for<3>([&](auto I) { dma_w(0, decltype(S)::value, I); }); });
        static_for<HI>([&](auto J) { load_h(rh, c1, J); });
        static_for<4>([&](auto Q) { load_coef(csc, csh, c1, Q); });
        // the first HI + 4 requests are the oldest: 21 DMA + HI + 4 younger ones may stay in flight
        wait_coef(csc0, csh0, std::integral_constant<int, 21 + HI + 4>{});
        static_for<HI>([&](auto J) {
            wait_h(rh0, J, std::integral_constant<int, 21 + HI + 4>{});
            static_for<HSTEPS>([&](auto ST) { hstep(rh0, csc0, csh0, J, ST, 0u); });
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    static_for<HI>([&](auto J) { wait_h(rh, J, std::integral_constant<int, 0>{}); });
    wait_coef(csc, csh, std::integral_constant<int, 0>{});
    __syncthreads();
    static_for<2>([&](auto KS) {
        static_for<NF>([&](auto I) { read_w(0, decltype(KS)::value, decltype(I)::value); });
        static_for<2>([&](auto F) { read_p(0, decltype(KS)::value, decltype(F)::value); });
    });

#ifdef SV_X3_STAMP
    const uint64_t st1 = __builtin_amdgcn_s_memtime();
    uint64_t stb = 0;
#endif
    // ---- the K loop: one 32-channel chunk = nine taps = 180 MFMAs -------------------------------------------------------
    int par = 0;
    for (int c = 0; c < nck; ++c) {
        const int cn = min(c + 1, nck - 1), cnn = min(c + 2, nck - 1);
        const uint32_t other = (uint32_t)((par ^ 1) * HB);
        static_for<9>([&](auto T) {
            constexpr int t = decltype(T)::value, tn = (t + 1) % 9;
            static_for<20>([&](auto M) {
                constexpr int m = decltype(M)::value, ks = m / 10, i = (m % 10) / 2, f = m & 1;
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[f][i]) : "v"(A[ks][i]), "v"(Bf[ks][f]));
                static_for<6>([&](auto JJ) {
                    constexpr int code = SCHED.item[20 * t + m][decltype(JJ)::value];
                    if constexpr (code >= 1000 && code < 2000)
                        hstep(rh, csc, csh, std::integral_constant<int, (code - 1000) / HSTEPS>{},
                              std::integral_constant<int, (code - 1000) % HSTEPS>{}, other);
                    if constexpr (code >= 2000 && code < 3000) {
                        constexpr int s = (code - 2000) / 3;
                        dma_w(s >= 7 ? c : cn, s, std::integral_constant<int, (code - 2000) % 3>{});
                    }
                    if constexpr (code >= 3000 && code < 3500) load_h(rh, cnn, std::integral_constant<int, code - 3000>{});
                    if constexpr (code >= 3500 && code < 4000) load_coef(csc, csh, cnn, std::integral_constant<int, code - 3500>{});
                    if constexpr (code >= 4000 && code < 4500)
                        wait_h(rh, std::integral_constant<int, code - 4000>{}, std::integral_constant<int, SCHED.vm_slot[code - 4000]>{});
                    if constexpr (code == 4500) wait_coef(csc, csh, std::integral_constant<int, SCHED.vm_coef>{});
                    if constexpr (code == 5000) {          // the next tap 0 reads the other halo stage
                        const uint32_t flip = par ? (uint32_t)(-HB) : (uint32_t)HB;
#pragma unroll
                        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                                for (int kk = 0; kk < 2; ++kk) px[ff][tx][kk] += flip;
                    }
                });
                // the next tap's fragments, each right after the last MFMA that reads its registers
                if constexpr (f == 1) read_w(tn, ks, i);
                if constexpr (i == 4) read_p(tn, ks, f);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (t == 1 || t == 4 || t == 7) {
                constexpr int n = t == 1 ? SCHED.vm_b1 : t == 4 ? SCHED.vm_b4 : SCHED.vm_b7;
#ifdef SV_X3_STAMP
                const uint64_t sb0 = __builtin_amdgcn_s_memtime();
#endif
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(n) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#ifdef SV_X3_STAMP
                stb += __builtin_amdgcn_s_memtime() - sb0;
#endif
            }
        });
        par ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();                                   // every wave is past its last fragment read: the LDS is free

#ifdef SV_X3_STAMP
    const uint64_t st2 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int SV_EPD = SV_X3_EPD;
#ifdef SV_X3_STAMP
    uint64_t ste[8] = {};
#define SV_EPI_STAMP(k) ste[k] = __builtin_amdgcn_s_memtime();
#else
#define SV_EPI_STAMP(k)
#endif
#define SV_EPI_NSCR 2
#include "conv3x3w_epilogue.inc"
#undef SV_EPI_NSCR
#undef SV_EPI_STAMP
#ifdef SV_X3_STAMP
    if (tid == 0 && blockIdx.x < 2048) {      // (overwrites the statistics: diagnostic build only)
        float* d = a.stats + 8 * 2 * g.N + 8 * blockIdx.x;      // behind the eight statistics replicas
        d[0] = (float)(st1 - st0); d[1] = (float)(st2 - st1); d[2] = (float)(__builtin_amdgcn_s_memtime() - st2); d[3] = (float)stb;
        d[4] = (float)(ste[0] - st2); d[5] = (float)(ste[1] - ste[0]); d[6] = (float)(ste[5] - ste[1]); d[7] = (float)(ste[6] - ste[5]);
    }
#endif
}

template <int WLOG, bool REV>
int launch_x3(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    using C = XCfg<WLOG>;
    const int nT = g->B * g->Hin / C::TR, nNt = g->N / C::BN;
    const int grid = 8 * ((nT + 7) / 8) * nNt;
    const size_t lds = (size_t)C::LDS;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3x_kernel<WLOG, REV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3x)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3x_kernel<WLOG, REV>), dim3(grid), dim3(256), lds, s, *g, *a);
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(conv3x3x)");
}

template <bool REV>
int launch_x2(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    switch (g->Win) {
        case 32: return launch_x3<5, REV>(g, a, s);
        case 16: return launch_x3<4, REV>(g, a, s);
        default: return launch_x3<3, REV>(g, a, s);
    }
}

}  // namespace

// Returns 1 and sets *rc when the geometry is a wide bf16 stride-1 3x3 convolution with 160-channel tiles.
// (The caller, sv_conv3x3w_try, has already checked the stride-1 3x3 / tap-order / size conditions; fwd = canonical taps.)
int sv_conv3x3x_try(const sv_geom* g, const sv_igemm_args* a, bool fwd, hipStream_t s, int* rc) {
    // experimental (round 1): on par with conv3x3w -- the K loop is ~1.3x the matrix-pipe time, but with one block per CU
    // nothing hides the prologue and the epilogue (tools/x3_stamp.sh); SV_CONV3X3X=1 selects it (read per call: tests)
    const char* on = getenv("SV_CONV3X3X");
    if (!on || on[0] == '0') return 0;
    if (g->N % 160 != 0 || g->Cin % 32 != 0 || g->Cin < 96) return 0;
    *rc = fwd ? launch_x2<false>(g, a, s) : launch_x2<true>(g, a, s);
    return 1;
}
