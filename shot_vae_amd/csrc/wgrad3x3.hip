// Weight gradient of stride-1 3x3 convolutions with an LDS-resident halo tile.  gfx950.
//
//   dW[n][torig(t)][c] += sum_pixels dy[p][n] * A[p + d(t)][c],   A = LeakyReLU(x*scale+shift)
//
// The generic sv_wgrad kernel gives every tap its own block, so dy and x are streamed nine times.
// Here a persistent block owns a 32(n) x 32(c) slab of dW for ALL nine taps (36 MFMA accumulator
// tiles = 144 registers per wave) and walks a range of 128-pixel tiles (whole image rows).  Per tile
// it stages dy [128][32] and the BN-transformed x halo [(TR+2)(W+2)][32] once (the next tile's global
// loads are already in flight while the current one is on the MFMAs); wave w reduces over pixels
// 32w..32w+31 of the tile: both operands are read k-major with ds_read_b64_tr_b16, the tap shift being
// nothing but a different LDS pixel address.  Waves are combined through LDS float atomics, blocks
// through one global float atomic per dW element.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

// An MFMA spelled in inline assembly is invisible to the compiler's hazard recognizer: if the instruction in front of it is a
// VALU write of one of its operands (a spill re-load, a register copy -- whatever the allocator decided), the two wait states
// the hardware requires between them are NOT inserted and the MFMA reads the old register (conv3x3x.hip: how this was found).
// conv3x3x gives every MFMA its own `s_nop 1` (free there); here that costs 13 % (16-cycle MFMAs in a tight gap program:
// 228 -> 254 us at 160 channels), so this file relies on the CHECK instead: tools/asm_mfma_lint.py compiles it to ISA and fails
// on any VALU write of an MFMA operand less than two wait states ahead (tests/test_abi_cpu.py runs it on every change).
#ifndef SV_WG3_NOP
#define SV_WG3_NOP 0
#endif
#if SV_WG3_NOP
#define SV_WG3_PRE "s_nop 1\n\t"
#else
#define SV_WG3_PRE
#endif

namespace {

constexpr int TC32 = 32;
#ifndef SV_WG3_LDH_PAD
#define SV_WG3_LDH_PAD 16
#endif
#ifndef SV_WG3_LDY_PAD
#define SV_WG3_LDY_PAD 16
#endif
// LDS row stride (elements) of both the dy tile and the halo.  96 B: the four pixel rows a 16-lane group of a transposing
// 8-byte read touches (32 B each) land on disjoint banks (80 B wrapped the fourth onto the first: 2-way conflicts)
constexpr int LDH = TC32 + SV_WG3_LDH_PAD;

struct wg3_params {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    float pro_slope;
    const void* dy;
    float* dw;
    float* ws;                    // [splits][N*T_orig*Cin] partial slabs (plain stores) or NULL (atomics into dw)
    int splits, tiles_per;        // 128-pixel tiles: range [split*tiles_per, ...)
    int unit;                     // wide kernel: channel chunks per XCD-affinity unit (0: plain block order)
    int groups;                   // batched launch: blockIdx.y = group; slab index = group * splits + split
};

// 8 consecutive pixels (k = 8g + j) of one (shifted) image row, 16 channels starting at col0: k-major
// fragment through the transposing read.  pix_elem_q = element offset of pixel 8g+q, q = (lane&15)>>2.
__device__ __forceinline__ bf16x8 frag_tr(const bf16* S, int pix_elem_q, int col0, int lane) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const bf16* a0 = S + pix_elem_q + col0 + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * LDH));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

// Block = 32(n) x 32(c) slab of dW, all nine taps.  Wave w owns the 16x16 sub-block (w>>1, w&1) for all
// nine taps (9 accumulator tiles) and reduces over ALL 128 pixels of every staged tile, so no cross-wave
// reduction is needed (LDS float atomics are far too slow for that: 92 us of a 140 us kernel).
template <typename T, int WLOG, int STAGES>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const sv_geom g, const sv_wg_g<wg3_params> PG) {
    const wg3_params& p = PG.g[blockIdx.y];
    typedef typename V8<T>::type V;
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    // LDS halo rows: row 0 / the last row are the vertical halo, and when a tile holds several whole images (W = 8:
    // TR = 16 > H = 8) a zero spacer row separates them -- zero padding is DATA in LDS, the nine taps need no masks
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;
    constexpr int HP = LROWS * WP;
    constexpr int HV = HP * 4, HI = (HV + 255) / 256;     // halo vectors (8 channels each)
    constexpr int YI = 2;                                  // 128 px * 4 vectors / 256 threads

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ys = reinterpret_cast<T*>(smem);          // [128][LDH]
    T* halo = Ys + 128 * LDH;                    // [HP][LDH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;      // this wave's (n, c) 16x16 sub-block
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    const int nCt = g.Cin / 32, nNt = g.N / 32, nNC = nCt * nNt;
    const int L = blockIdx.x;
    int nc, split;
    if (p.splits % 8 == 0) {          // blocks L, L+8 share an XCD: all (n,c) slabs of one pixel range on one L2
        const int xcd = L & 7, slot = L >> 3;
        nc = slot % nNC;
        split = (slot / nNC) * 8 + xcd;
    } else {
        nc = L % nNC;
        split = L / nNC;
    }
    const int n0 = (nc / nCt) * 32, c0 = (nc % nCt) * 32;
    const int t_begin = split * p.tiles_per, t_end = min(nT, t_begin + p.tiles_per);
    const sv_phase& P = g.phase[0];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);
    const bool has_pro = p.pro_scale != nullptr;
    float pslope = p.pro_slope;               // pinned in a vector register (conv3x3p_kernel: no re-load from the argument segment)
    asm volatile("v_mov_b32 %0, %0" : "+v"(pslope));

    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;
    const int v = tid & 3;
    f32x4 s0, s1, t0, t1;
    if (has_pro) {
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * v);
        s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * v + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * v);
        t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * v + 4);
    }
    // halo staging slots.  kind: 0 = always zero (padding column / spacer / dummy), 1 = image row of this tile,
    //                            2 = row above the tile, 3 = row below it (valid only inside the same image)
    int hrel[HI], hxc[HI], hlds[HI], hkind[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int pix = min(idx, HV - 1) >> 2;
        const int lr = pix / WP, xx = pix - lr * WP;
        const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
        int kind = 1, rel = lr - 1 - seg;
        if (off == 0) {
            if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
            else kind = 0;
        }
        if (xx == 0 || xx == WP - 1) kind = 0;
        hkind[i] = kind;
        hrel[i] = rel;
        hxc[i] = min(max(xx - 1, 0), W - 1);
        hlds[i] = idx < HV ? pix * LDH + 8 * v : -1;
    }

    // per-thread staging offsets are tile-invariant: a tile only moves the two (uniform) base pointers.  The halo base
    // is the row ABOVE the tile, so that every offset is an unsigned 32-bit byte count (SGPR base + VGPR offset loads);
    // slots that are zero for this tile read the tile's first pixel instead (always inside the tensor)
    uint32_t hoff[HI], yoff[YI];
#pragma unroll
    for (int i = 0; i < HI; ++i) hoff[i] = (uint32_t)(((hrel[i] + 1) * W + hxc[i]) * g.ldx + c0 + 8 * v) * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < YI; ++i) yoff[i] = (uint32_t)(((tid >> 2) + 64 * i) * g.ldo + n0 + 8 * v) * (uint32_t)sizeof(T);
    const uint32_t hsafe = (uint32_t)(W * g.ldx + c0 + 8 * v) * (uint32_t)sizeof(T);

    // STAGES = 2: two register stages -- the operands of tile i + 2 are requested while tile i is on the MFMAs, so every
    // request has two whole tile periods to arrive (at two blocks per CU one tile period does not cover the HBM latency:
    // the 32-channel layer at 4 x 512 images 119 -> 87 us, 64 channels 91 -> 81 us).  STAGES = 1 (requests one tile ahead)
    // for the 128-channel layer, whose operands are re-read by 16 blocks and come from L2 (74 us; 82 us with two stages)
    struct Stage { V ry[YI], rh[HI]; bool hok[HI]; };
    Stage SA, SB;
    auto load_tile = [&](Stage& S, int tile) {
        const int gr0 = tile * TR;
        const bool top_ok = (gr0 & (H - 1)) != 0, bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        const char* ybase = reinterpret_cast<const char*>(DY + (int64_t)gr0 * W * g.ldo);
        const char* hbase = reinterpret_cast<const char*>(X) + ((int64_t)gr0 - 1) * W * g.ldx * (int64_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < YI; ++i) S.ry[i] = *reinterpret_cast<const V*>(ybase + yoff[i]);   // tile pixel (tid>>2) + 64 i
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            S.hok[i] = hkind[i] == 1 || (hkind[i] == 2 && top_ok) || (hkind[i] == 3 && bot_ok);
            S.rh[i] = *reinterpret_cast<const V*>(hbase + (S.hok[i] ? hoff[i] : hsafe));
        }
    };
    auto store_tile = [&](Stage& S) {
#pragma unroll
        for (int i = 0; i < YI; ++i)
            *reinterpret_cast<V*>(Ys + ((tid >> 2) + 64 * i) * LDH + 8 * v) = S.ry[i];
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            V o = zero;
            if (S.hok[i]) {
                o = S.rh[i];
                if (has_pro) o = bn_act8(S.rh[i], s0, s1, t0, t1, pslope);
            }
            if (hlds[i] >= 0) *reinterpret_cast<V*>(halo + hlds[i]) = o;
        }
    };

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (t_begin < t_end) load_tile(SA, t_begin);
    if (STAGES == 2 && t_begin + 1 < t_end) load_tile(SB, t_begin + 1);
    // one tile of the pipeline: CUR holds this tile (requested two tiles ago), and is re-loaded with tile + 2 once stored
    auto do_tile = [&](int tile, Stage& CUR) {
        store_tile(CUR);
        __syncthreads();
        if (tile + STAGES < t_end) load_tile(CUR, tile + STAGES);   // in flight during this (and the next) tile's MFMAs
        if (sizeof(T) == 2) {
            const bf16* Yb = reinterpret_cast<const bf16*>(Ys);
            const bf16* Hb = reinterpret_cast<const bf16*>(halo);
            // fragments of one 32-pixel chunk: the dy fragment + the nine tap-shifted x fragments, requested together
            // and double-buffered over the chunks (with one MFMA per fragment a read-then-multiply chain exposes the
            // LDS latency nine times per chunk: stamped 3 700 of the 6 100 cycles of a tile iteration)
            struct Frags { bf16x8 fy, fx[9]; };
            Frags FA, FB;
            auto load_frags = [&](Frags& F, int kc) {
                // lane addresses pixel pq = 32*kc + 8*fq + q (q = fr>>2) of the tile (and pq + 4)
                const int pq = 32 * kc + 8 * fq + (fr >> 2);
                const int jrow = pq >> WLOG, xcol = pq & (W - 1);
                F.fy = frag_tr(Yb, pq * LDH, 16 * wi, lane);
                const int hbase = ((jrow + 1 + jrow / HH) * WP + xcol + 1) * LDH;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    F.fx[t] = frag_tr(Hb, hbase + (tap_off(pdy, t) * WP + tap_off(pdx, t)) * LDH, 16 * wj, lane);
            };
            auto mma_frags = [&](const Frags& F) {
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.fy, F.fx[t], acc[t], 0, 0, 0);
            };
            load_frags(FA, 0);
            load_frags(FB, 1);
            mma_frags(FA);
            load_frags(FA, 2);
            mma_frags(FB);
            load_frags(FB, 3);
            mma_frags(FA);
            mma_frags(FB);
            // keep the order written above (the scheduler otherwise re-serialises to two fragments in flight)
            __builtin_amdgcn_sched_group_barrier(0x100, 40, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);
        } else {
            // fp32 (parity mode): v_mfma_f32_16x16x4_f32 step j uses pixel 4*j + fq of each 32-pixel chunk
            const float* Yf = reinterpret_cast<const float*>(Ys);
            const float* Hf = reinterpret_cast<const float*>(halo);
            for (int pj = 0; pj < 32; ++pj) {
                const int pp = 4 * pj + fq;
                const int jrow = pp >> WLOG, xcol = pp & (W - 1);
                const float yv = Yf[pp * LDH + 16 * wi + fr];
                const int hbase = ((jrow + 1 + jrow / HH) * WP + xcol + 1) * LDH + 16 * wj + fr;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float xv = Hf[hbase + (tap_off(pdy, t) * WP + tap_off(pdx, t)) * LDH];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, xv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();          // everyone is done reading before the next tile overwrites LDS
    };
    if (STAGES == 2) {
        for (int tile = t_begin; tile < t_end; tile += 2) {
            do_tile(tile, SA);
            if (tile + 1 < t_end) do_tile(tile + 1, SB);
        }
    } else {
        for (int tile = t_begin; tile < t_end; ++tile) do_tile(tile, SA);
    }

    // ---- publish: D layout = lane holds c = 16*wj + fr, n = 16*wi + 4*fq + r ----------------------------
    const int64_t slab = (int64_t)g.N * g.T_orig * g.Cin;
    float* dst = p.ws ? p.ws + ((int64_t)blockIdx.y * p.splits + split) * slab : p.dw;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int to = P.torig[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* q = dst + ((int64_t)(n0 + 16 * wi + 4 * fq + r) * g.T_orig + to) * g.Cin + c0 + 16 * wj + fr;
            if (p.ws) *q = acc[t][r];
            else atomicAdd(q, acc[t][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Wide layers (N % 160 == 0: the WRN-28-10 body, MFMA-bound).  With 32 x 32 slabs every block re-stages a dy tile
// and a halo tile per 36 MFMAs per wave, and the per-CU load path (L2 -> LDS), not the matrix pipe, sets the pace.
// Here a block owns 160 (n) x 32 (c) x 9 taps with ONE wave per SIMD: wave (wi, wj) owns 80 n x 16 c x 9 taps = 45
// accumulator tiles (180 registers, all in the accumulation half of the register file), so one 128-pixel tile feeds
// 4 x 45 MFMAs per wave from 14 transposing fragment reads per 32 pixels.  Same staging / masking / publishing scheme
// as above (the 32-channel chunks of one pixel range run on one XCD, so the dy re-reads are L2 hits).
typedef __attribute__((address_space(3))) void* wg_lds_ptr;
typedef const __attribute__((address_space(1))) void* wg_glb_ptr;
// ---- gap bookkeeping of the wide kernel's phases (see wgrad3x3w_kernel): gap g = 5 t + a follows MFMA a of tap group t
constexpr bool gap_has_read(int g) { return (g % 5) < 2 || ((g % 5) == 2 && g / 5 < 8); }
constexpr int gap_cap(int g) { return gap_has_read(g) ? 1 : 2; }            // single-instruction steps a gap hides
constexpr int cap_before(int g) { int c = 0; for (int i = 0; i < g; ++i) c += gap_cap(i); return c; }
constexpr int free_before(int g) { int c = 0; for (int i = 0; i < g; ++i) c += gap_has_read(i) ? 0 : 1; return c; }

// The per-tile program of everything that is not an MFMA or a fragment read, packed into the gaps at compile time.
//   items: HSTEPS * HI transform micro-steps (code 1000 + 41 slot + step); the YI = 11 dy DMA instructions (2000 + k);
//          the HI halo loads (3000 + slot); 4 * HI single-dword register copies (5000 + 4 slot + dword).
//   VMEM instructions are spread out -- the probe: one per four MFMAs is free, one per two is not (the CU's 64 B/clk L1
//   path).
//   NO HAND-COUNTED vmcnt: LDS-DMA and register loads do not retire in one common order, so a counted wait for one kind
//   may not rely on younger instructions of the other kind -- and a wait that counts only its own kind drains the
//   other kind's instructions in flight (13 % at 160 channels when the halo waits drained the dy copies).  Instead the
//   halo registers are DOUBLE-BUFFERED: the loads of the tile after next go to a second register set at the top of the
//   tile iteration (phases 0..2), everything -- they and the dy copies -- is awaited by the ONE vmcnt(0) in front of the
//   barrier, 1 000+ cycles after the last request, and phase 3 moves the second set into the first (the transform of the
//   next iteration reads it) while the first four dy copies of the next tile start.
constexpr int WG_HSTEPS = 41, WG_YI = 11;
template <int HI>
struct WSched {
    int item[180][3];       // gap (phase * 45 + g) -> up to 3 item codes, executed in this order (0 = none)
    int last_vmem_a;        // gap of the last VMEM instruction of phases 0..2 (it has 135 - that many MFMAs to land)
    bool ok;
};
template <int HI>
constexpr WSched<HI> make_wsched() {
    WSched<HI> S{};
    for (int i = 0; i < 180; ++i) S.item[i][0] = S.item[i][1] = S.item[i][2] = 0;
    // ---- region A (phases 0..2).  A read gap hides one step, a free gap two steps or ONE VMEM instruction (at least
    //      six gaps after the previous one).  The requests come first -- halo registers of the tile after next, then
    //      D4..D10 of the next tile's dy -- the transform steps fill what is left, from the front
    const int nsteps = WG_HSTEPS * HI;
    int queue[16] = {};
    int nq = 0;
    for (int sl = 0; sl < HI; ++sl) queue[nq++] = 3000 + sl;
    for (int k = 4; k < WG_YI; ++k) queue[nq++] = 2000 + k;
    int m = 0, qh = 0, last_v = -100;
    for (int gi = 0; gi < 135; ++gi) {
        int nt = 0;
        const bool fre = !gap_has_read(gi % 45);
        if (fre && qh < nq && gi - last_v >= 6) {
            S.item[gi][nt++] = queue[qh++];
            last_v = gi;
        } else {
            for (int c = 0; c < (fre ? 2 : 1) && m < nsteps; ++c) S.item[gi][nt++] = 1000 + m++;
        }
    }
    S.ok = m == nsteps && qh == nq;
    S.last_vmem_a = last_v;
    // ---- region B (phase 3): D0..D3 four free gaps apart; the register copies (one dword each) two per free gap / one
    //      per read gap in between
    {
        int f = 0, placed = 0, cp = 0;
        for (int g = 0; g < 45; ++g) {
            const bool fre = !gap_has_read(g);
            if (fre && f % 4 == 0 && placed < 4) {
                S.item[135 + g][0] = 2000 + placed++;
                ++f;
                continue;
            }
            if (fre) ++f;
            for (int c = 0; c < (fre ? 2 : 1) && cp < 4 * HI; ++c) S.item[135 + g][c] = 5000 + cp++;
        }
        S.ok = S.ok && placed == 4 && cp == 4 * HI;
    }
    return S;
}

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}
constexpr int LDY = 160 + SV_WG3_LDY_PAD;      // LDS row stride (elements) of the wide dy tile: 352 B = 96 B mod 256, as LDH
constexpr int YVR = LDY / 8;                   // 16-byte vectors per LDS row of the dy tile (20 of data + padding)

__device__ __forceinline__ bf16x8 frag_tr_ld(const bf16* S, int pix_elem_q, int col0, int lane, int ld) {
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    const bf16* a0 = S + pix_elem_q + col0 + 4 * (lane & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * ld));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.b;
}

template <int WLOG>
__global__ __launch_bounds__(256, 1) void wgrad3x3w_kernel(const sv_geom g, const sv_wg_g<wg3_params> PG) {
    // (a batched launch: the groups are an INNER loop of every block here, not a grid dimension -- a block accumulates its
    //  slab over its tile range of every group and publishes ONE slab: with 15-59 MB slabs a slab per (group, split) is
    //  what the workspace and the reduction cannot afford; measured 745 us for the four groups of the 640-channel layer)
    const wg3_params& p = PG.g[0];
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    // LDS halo rows: row 0 / the last row are the vertical halo, and when a tile holds several whole images (W = 8:
    // TR = 16 > H = 8) a zero spacer row separates them -- zero padding is DATA in LDS, the nine taps need no masks
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;
    constexpr int HP = LROWS * WP;
    // the dy tile goes global -> LDS by DMA, 64 consecutive 16-byte vectors per wave instruction, and an LDS row is YVR
    // vectors (20 of data + the LDY padding, which is fetched as copies of vector 19): 128 * 22 vectors = 44 instructions,
    // wave w issues k*4 + w for k < 11 (any past the end repeat the last one)
    constexpr int YV = 128 * YVR, YI = (YV + 255) / 256;
    constexpr int HV = HP * 4, HI = (HV + 255) / 256;         // halo vectors

    // two LDS stages of {dy tile [128][LDY], halo [HP][LDH], 512 dummy elements for the unused staging slots}
    constexpr int BUF = 128 * LDY + HP * LDH + 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const lds0 = reinterpret_cast<bf16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // scalar copy: DMA destinations stay in SGPRs
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    const int nC = g.Cin / 32, nNt = g.N / 160, nNC = nC * nNt;
    const int L = blockIdx.x;
    int nc, split;
    if (p.unit > 0) {
        // XCD affinity for any split count: an affinity unit = p.unit channel chunks of one (pixel range, n tile); the
        // units are dealt round-robin to the XCDs (blocks L, L+8, ... share an L2), so the chunk blocks that re-read
        // one dy range run on one XCD and only the first of them goes past its L2
        const int xcd = L & 7, slot = L >> 3;
        const int U = (slot / p.unit) * 8 + xcd, cu = slot % p.unit;
        const int upg = nC / p.unit;                  // units per (split, n tile)
        const int grp = U / upg, chunk = (U % upg) * p.unit + cu;
        split = grp / nNt;
        if (split >= p.splits) return;
        nc = (grp % nNt) * nC + chunk;
    } else {
        nc = L % nNC;
        split = L / nNC;
    }
    const int n0 = (nc / nC) * 160, c0 = (nc % nC) * 32;
    const int t_begin = split * p.tiles_per, t_end = min(nT, t_begin + p.tiles_per);
    const sv_phase& P = g.phase[0];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const bf16* X = reinterpret_cast<const bf16*>(p.x);            // (of the current group)
    const bf16* DY = reinterpret_cast<const bf16*>(p.dy);
    const bool has_pro = p.pro_scale != nullptr;

    bf16x8 zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    const int v = tid & 3;
    // without a prologue the transform is the identity (scale 1, shift 0, slope 1: exact in bf16 -> fp32 -> bf16), so the
    // tile loop below is ONE basic block whose instruction order the phases control
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    const float slope = has_pro ? p.pro_slope : 1.f;
    if (has_pro) {
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * v);
        s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * v + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * v);
        t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * v + 4);
    }
    // halo staging slots.  kind: 0 = always zero (padding column / spacer / dummy), 1 = image row of this tile,
    //                            2 = row above the tile, 3 = row below it (valid only inside the same image)
    int hrel[HI], hxc[HI], hlds[HI], hkind[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int pix = min(idx, HV - 1) >> 2;
        const int lr = pix / WP, xx = pix - lr * WP;
        const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
        int kind = 1, rel = lr - 1 - seg;
        if (off == 0) {
            if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
            else kind = 0;
        }
        if (xx == 0 || xx == WP - 1) kind = 0;
        hkind[i] = kind;
        hrel[i] = rel;
        hxc[i] = min(max(xx - 1, 0), W - 1);
        hlds[i] = idx < HV ? 128 * LDY + pix * LDH + 8 * v : 128 * LDY + HP * LDH + 8 * (tid & 63);   // or the dummy
    }
    // per-thread staging offsets are tile-invariant: a tile only moves the two (uniform) base pointers.  The halo base
    // is the row ABOVE the tile, so that every offset is an unsigned 32-bit byte count (SGPR base + VGPR offset loads);
    // slots that are zero for this tile read the tile's first pixel instead (always inside the tensor)
    uint32_t hoff[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) hoff[i] = (uint32_t)(((hrel[i] + 1) * W + hxc[i]) * g.ldx + c0 + 8 * v) * 2u;
    const uint32_t hsafe = (uint32_t)(W * g.ldx + c0 + 8 * v) * 2u;
    uint32_t yoff[YI];
#pragma unroll
    for (int k = 0; k < YI; ++k) {
        const int q = (k * 4 + wave) * 64 + lane, pp = q / YVR, vv = min(q - pp * YVR, 19);
        yoff[k] = (uint32_t)(pp * g.ldo + n0 + 8 * vv) * 2u;
    }

    // Every VMEM instruction of the tile loop is spelled in assembly and waited for by hand (WSched::vm_*): through the
    // builtin the compiler knows that the DMA writes LDS and, unable to tell the two stages apart, waits for vmcnt(0)
    // before the next fragment read; and its own vmcnt bookkeeping cannot see the assembly.
    bf16x8 rh[HI], rn[HI];             // halo vectors of the next tile (being transformed) / the tile after next (in flight)
    bool hok[HI], hokn[HI];
    const char* ybase = nullptr;       // dy rows of the tile whose DMA is in progress
    const char* hbase = nullptr;       // x rows (from one above) of the tile whose halo is being loaded
    bool top_ok = false, bot_ok = false;
    auto halo_bases = [&](int tile) {
        const int gr0 = tile * TR;
        top_ok = (gr0 & (H - 1)) != 0;
        bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        hbase = reinterpret_cast<const char*>(X) + ((int64_t)gr0 - 1) * W * g.ldx * 2;
    };
    auto dy_base = [&](int tile) { ybase = reinterpret_cast<const char*>(DY + (int64_t)tile * TR * W * g.ldo); };
    static_assert(YV % 256 == 0 && YI == WG_YI, "every wave issues whole DMA instructions");
    const uint32_t dma0 = (uint32_t)(uintptr_t)(wg_lds_ptr)lds0 + (uint32_t)wave_s * 1024u;   // + stage, + 4096 k
    auto dma_y = [&](int stage, auto K) {
        constexpr int k = decltype(K)::value;
        const uint32_t base = dma0 + (uint32_t)stage * (uint32_t)(BUF * 2), yo = yoff[k];
        const char* yb = ybase;
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                     :: "s"(base), "n"(k * 4096), "v"(yo), "s"(yb) : "memory", "scc");   // (m0 is reserved: the
                                                                                    // compiler never allocates it)
    };
    auto load_h = [&](bf16x8* dst, bool* ok, auto I) {       // halo vector i of the tile halo_bases() was last called for
        constexpr int i = decltype(I)::value;
        ok[i] = (hkind[i] == 1) | ((hkind[i] == 2) & top_ok) | ((hkind[i] == 3) & bot_ok);      // (no short circuits)
        const uint32_t o = ok[i] ? hoff[i] : hsafe;
        const char* hb = hbase;
        bf16x8 r;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(o), "s"(hb) : "memory");
        dst[i] = r;
    };
    // the full drain that makes a register set usable: to the compiler the registers are (re)defined HERE
    auto drain_h = [&](bf16x8* set) {
        if constexpr (HI == 4) {
            bf16x8 r0 = set[0], r1 = set[1], r2 = set[2], r3 = set[HI - 1];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
            set[0] = r0; set[1] = r1; set[2] = r2; set[HI - 1] = r3;
        } else {
            bf16x8 r0 = set[0], r1 = set[1], r2 = set[2];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2) :: "memory");
            set[0] = r0; set[1] = r1; set[2] = r2;
        }
    };
    // The BN + activation transform of the halo vectors, as single-instruction MICRO-STEPS.  Measured with
    // tools/probes/issue_probe.hip (one wave per SIMD): behind one 16x16x32 MFMA a wave issues two (unpacked) VALU
    // instructions or one LDS read for free; every further VALU costs ~4 cycles, a second LDS read ~10, and
    // v_pk_fma_f32 ~12 (so no packed fp32 here).  Halo vector i = four dwords of two bf16 channels; per dword d:
    //   0,1: lo = x << 16, hi = x & 0xffff0000      2,3: u = f * scale + shift      4,5: m = u * slope
    //   6,7: u = max(u, m)                           8: od = pack_bf16(u)            9: od = valid ? od : 0
    // and step 40 stores the vector into the other stage.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int HSTEPS = 41;
    float xlo, xhi, xmlo, xmhi;
    u32x4 od;
    auto hstep = [&](auto I, auto ST, bf16* wr) {
        constexpr int i = decltype(I)::value, st = decltype(ST)::value, d = st / 10, q = st % 10;
        if constexpr (st == 40) {
            *reinterpret_cast<u32x4*>(wr + hlds[i]) = od;
        } else {
            const float sc_lo = d < 2 ? s0[2 * d] : s1[2 * d - 4], sc_hi = d < 2 ? s0[2 * d + 1] : s1[2 * d - 3];
            const float sh_lo = d < 2 ? t0[2 * d] : t1[2 * d - 4], sh_hi = d < 2 ? t0[2 * d + 1] : t1[2 * d - 3];
            if constexpr (q == 0) xlo = __builtin_bit_cast(float, __builtin_bit_cast(u32x4, rh[i])[d] << 16);
            if constexpr (q == 1) xhi = __builtin_bit_cast(float, __builtin_bit_cast(u32x4, rh[i])[d] & 0xffff0000u);
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
            if constexpr (q == 9) od[d] = hok[i] ? od[d] : 0u;
        }
    };
    auto store_h = [&](bf16* buf, auto I) {
        static_for<HSTEPS>([&](auto ST) { hstep(I, ST, buf); });
    };

    f32x4 acc[5][9];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragments of one 32-pixel k chunk: 5 dy fragments (two sets, used by all nine tap groups of a phase) and 9
    // tap-shifted x fragments (ONE rolling set: tap t's registers are free once its five MFMAs have issued, and receive
    // the next chunk's tap t right away -- eight tap groups before they are needed).  A fragment is two transposing
    // 8-byte reads (k 0..3 and 4..7 of the lane's column), kept as halves so that each read gets a gap of its own.
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 fy[2][5], fx[9];
    auto put_half = [](s16x8& f, s16x4 h, int half) {
        const s16x8 w = __builtin_shufflevector(h, h, 0, 1, 2, 3, -1, -1, -1, -1);
        f = half == 0 ? __builtin_shufflevector(w, f, 0, 1, 2, 3, 12, 13, 14, 15)
                      : __builtin_shufflevector(f, w, 0, 1, 2, 3, 8, 9, 10, 11);
    };
    // lane byte addresses in LDS, per stage: the lane reads pixel l = 8*fq + (fr>>2) (and l + 4) of every 32-pixel chunk;
    // the chunk and the dy fragment number are immediates, the tap shift (layer-dependent order) is folded in here
    const int lpix = 8 * fq + (fr >> 2);
    uint32_t yaddr, xaddr[9];                // of the stage being read; flipped to the other stage once per tile
    {
        const uint32_t l0 = (uint32_t)(uintptr_t)(wg_lds_ptr)lds0;
        const int yl = lpix * LDY + 80 * wi + 4 * (lane & 3);
        const int hl = 128 * LDY + (((lpix >> WLOG) + 1) * WP + (lpix & (W - 1)) + 1) * LDH + 16 * wj + 4 * (lane & 3);
        yaddr = l0 + (uint32_t)yl * 2u;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            xaddr[t] = l0 + (uint32_t)(hl + (tap_off(pdy, t) * WP + tap_off(pdx, t)) * LDH) * 2u;
    }
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    auto load_fy = [&](int set, int kc, int a, int half) {
        put_half(fy[set][a], __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_v4*)(uintptr_t)(yaddr + (uint32_t)((32 * kc + 4 * half) * LDY + 16 * a) * 2u)), half);
    };
    auto load_fx = [&](int kc, int t, int half) {
        // chunk kc starts (32 / W) * kc image rows into the tile; the spacer rows of W = 8 tiles come every HH rows
        const int r0 = (32 / W) * kc, rows = r0 + r0 / HH;
        put_half(fx[t], __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_v4*)(uintptr_t)(xaddr[t] + (uint32_t)((rows * WP + 4 * half) * LDH) * 2u)), half);
    };
    // One phase = the 45 MFMAs of one k chunk.  One wave per SIMD: nobody else fills the matrix pipe while this wave
    // stages, so every other instruction of the tile iteration sits in the GAP behind one MFMA, sized by what the probe
    // says a gap hides, and nothing crosses the sched_barrier that closes a gap.  Gap g = 5 t + a (tap group t, MFMA a):
    //   a = 0, 1: one LDS read each -- tap t-1 of the next chunk (its registers were freed by tap group t-1); in group
    //             0: dy fragment 0                    a = 2 (t < 8): one read of dy fragments 1..4 of the next chunk
    //   the other 19 gaps:  phases 0..2: two micro-steps of the halo transform (and one in each read gap; the 41 * HI
    //                       steps fill the capacity of the three phases from the back);  [end of phase 2: barrier]
    //                       phase 3: reads kc0 of the next tile from the other stage; one global load of the halo of the
    //                       tile after next, or one dy DMA instruction into the stage that just became free
    // wr = the stage phase work writes (fragment reads follow yaddr / xaddr)
    static constexpr WSched<HI> SCHED = make_wsched<HI>();
    static_assert(SCHED.ok, "the tile program does not fit the gaps");
    auto phase = [&](auto PH, bf16* wr, int dma_stage) {
        constexpr int ph = decltype(PH)::value;
        constexpr int nkc = (ph + 1) & 3, set = ph & 1;
        static_for<45>([&](auto GP) {
            constexpr int g = decltype(GP)::value, t = g / 5, a = g % 5;
            // in-place accumulation in the AGPR half, spelled out: left to itself the register allocator rotates the
            // 180 accumulators through copies
            asm volatile(SV_WG3_PRE "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[a][t]) : "v"(fy[set][a]), "v"(fx[t]));
            if constexpr (a < 2) {
                if constexpr (t >= 1) load_fx(nkc, t - 1, a);
                else load_fy(set ^ 1, nkc, 0, a);
            }
            if constexpr (a == 2 && t < 8) load_fy(set ^ 1, nkc, 1 + t / 2, t & 1);
            static_for<3>([&](auto J) {
                constexpr int code = SCHED.item[ph * 45 + g][decltype(J)::value], kind = code / 1000, arg = code % 1000;
                if constexpr (kind == 1)
                    hstep(std::integral_constant<int, arg / HSTEPS>{}, std::integral_constant<int, arg % HSTEPS>{}, wr);
                if constexpr (kind == 2) dma_y(dma_stage, std::integral_constant<int, arg>{});
                if constexpr (kind == 3) load_h(rn, hokn, std::integral_constant<int, arg>{});
                if constexpr (kind == 5) {          // one dword of the second register set moves into the first
                    u32x4 d = __builtin_bit_cast(u32x4, rh[arg / 4]);
                    const uint32_t sdw = __builtin_bit_cast(u32x4, rn[arg / 4])[arg % 4];
                    uint32_t o;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(sdw));
                    d[arg % 4] = o;
                    rh[arg / 4] = __builtin_bit_cast(bf16x8, d);
                    if constexpr (arg % 4 == 0) hok[arg / 4] = hokn[arg / 4];
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        load_fx(nkc, 8, 0);
        load_fx(nkc, 8, 1);
    };
    static_assert(HI >= 3 && HI <= 4 && HSTEPS == WG_HSTEPS, "phase work lists");

    int cur = 0;
    for (int gi = 0; gi < p.groups; ++gi) {
    if (gi > 0) {
        // next group of a batched launch: its tensors and BatchNorm coefficients; the pipeline restarts from stage 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (redundant) requests of the previous group
        __syncthreads();                                        // nobody reads the stages any more
        const wg3_params& pg = PG.g[gi];
        X = reinterpret_cast<const bf16*>(pg.x);
        DY = reinterpret_cast<const bf16*>(pg.dy);
        if (has_pro) {
            s0 = *reinterpret_cast<const f32x4*>(pg.pro_scale + c0 + 8 * v);
            s1 = *reinterpret_cast<const f32x4*>(pg.pro_scale + c0 + 8 * v + 4);
            t0 = *reinterpret_cast<const f32x4*>(pg.pro_shift + c0 + 8 * v);
            t1 = *reinterpret_cast<const f32x4*>(pg.pro_shift + c0 + 8 * v + 4);
        }
        if (cur) {                                              // fragment addresses back to stage 0
            const uint32_t flip = (uint32_t)(-(BUF * 2));
            yaddr += flip;
#pragma unroll
            for (int t = 0; t < 9; ++t) xaddr[t] += flip;
            cur = 0;
        }
    }
    // prologue: tile t_begin into stage 0; then the state the steady loop expects -- the halo of the second tile in rh,
    // D0..D3 of its dy DMA on their way into stage 1 -- and the first fragments
    halo_bases(t_begin);
    dy_base(t_begin);
    static_for<YI>([&](auto K) { dma_y(0, K); });
    static_for<HI>([&](auto I) { load_h(rh, hok, I); });
    drain_h(rh);
    static_for<HI>([&](auto I) { store_h(lds0, I); });
    halo_bases(min(t_begin + 1, t_end - 1));
    dy_base(min(t_begin + 1, t_end - 1));
    static_for<HI>([&](auto I) { load_h(rh, hok, I); });
    static_for<4>([&](auto K) { dma_y(1, K); });
    drain_h(rh);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 5; ++a) { load_fy(0, 0, a, 0); load_fy(0, 0, a, 1); }
#pragma unroll
    for (int t = 0; t < 9; ++t) { load_fx(0, t, 0); load_fx(0, t, 1); }
    for (int tile = t_begin; tile < t_end; ++tile) {
        bf16* cur_stage = lds0 + cur * BUF;
        bf16* other = lds0 + (cur ^ 1) * BUF;
        // this iteration: the MFMAs of `tile` (stage cur); the halo of tile + 1 is transformed into the other stage, where
        // the rest (D4..D10) of its dy DMA lands too; the halo registers are re-loaded for tile + 2 (past the end: a
        // harmless re-load of the last tile), and after the barrier D0..D3 of tile + 2 start into this stage
        halo_bases(min(tile + 2, t_end - 1));
        phase(std::integral_constant<int, 0>{}, other, cur ^ 1);
        phase(std::integral_constant<int, 1>{}, other, cur ^ 1);
        phase(std::integral_constant<int, 2>{}, other, cur ^ 1);
        drain_h(rn);              // this wave's share of the dy DMA has landed, and so has the halo of the tile after next
        __syncthreads();          // the other stage is complete, and nobody reads this one any more (kc3 is in registers)
        {   // from here on fragments come from the other stage
            const uint32_t flip = cur ? (uint32_t)(-(BUF * 2)) : (uint32_t)(BUF * 2);
            yaddr += flip;
#pragma unroll
            for (int t = 0; t < 9; ++t) xaddr[t] += flip;
        }
        dy_base(min(tile + 2, t_end - 1));
        phase(std::integral_constant<int, 3>{}, cur_stage, cur);
        cur ^= 1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last (redundant) DMA, before the LDS goes
                                                                            // away; the last MFMAs, before acc is read

    // ---- publish: D layout = lane holds c = c0 + 16*wj + fr, n = n0 + 80*wi + 16*a + 4*fq + r --------------------
    const int64_t slab = (int64_t)g.N * g.T_orig * g.Cin;
    float* dst = p.ws ? p.ws + (int64_t)split * slab : p.dw;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int to = P.torig[t];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* o = dst + ((int64_t)(n0 + 80 * wi + 16 * a + 4 * fq + r) * g.T_orig + to) * g.Cin + c0 + 16 * wj + fr;
                if (p.ws) *o = acc[a][t][r];
                else atomicAdd(o, acc[a][t][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// Narrow layers (N, Cin <= 128: the WRN-28-2 body), round 4.  wgrad3x3_kernel above takes ~75 us for every width at 4 x 512
// images although the layers differ 4x in bytes: its waves own 16 x 16 sub-blocks of a 32 x 32 slab, i.e. per 32-pixel chunk
// 20 transposing LDS reads feed 9 short MFMAs (16x16x32: the matrix instruction holds the SIMD's issue port for half its
// time), and every (n, c) slab block re-transforms (BatchNorm + LeakyReLU) the x halo of its pixel range -- twice at 64, four
// times at 128 channels.  The SIMD issue port, not the matrix pipe / LDS / HBM, set the pace.  Here:
//   * v_mfma_f32_32x32x16_bf16: a wave owns a whole 32 (n) x 32 (c) tile for all nine taps (144 accumulator registers); per
//     16-pixel step 2 + 18 transposing reads feed nine 32-cycle MFMAs (half the reads and half the issue slots per flop);
//   * a block owns NB = 32 or 64 output channels x 32 input channels: at 64 / 128 channels the halo transform is shared by
//     two n tiles.  Waves: NB = 64: (n half) x (pixel half of the tile); NB = 32: the four pixel quarters.  The partial
//     tiles of the pixel parts meet in LDS once per block (fixed order: plain read-modify-write between barriers);
//   * every LDS image is [pixel][32 channels] with 64-byte rows: the four pixel rows a transposing read touches are 256
//     contiguous bytes -- conflict-free without padding, and the dy tile goes global -> LDS by DMA (lane-linear image);
//   * pipeline: halo registers of tile i + 2 and the dy DMA of tile i + 2 are requested at the top of iteration i, awaited
//     by the ONE vmcnt(0) at the top of iteration i + 1 (a whole tile period later), transformed into the other halo stage
//     and consumed in iteration i + 2; three dy stages, two halo stages, one barrier per tile.
constexpr int LDM = 32;                         // elements per LDS row (64 bytes)
template <int WLOG, int NB>
__global__ __launch_bounds__(256, 2) void wgrad3x3m_kernel(const sv_geom g, const sv_wg_g<wg3_params> PG) {
    const wg3_params& p = PG.g[blockIdx.y];
    constexpr int W = 1 << WLOG, TR = 128 / W, WP = W + 2;
    constexpr int HH = (TR < W) ? TR : W, SEG = TR / HH, LROWS = TR + SEG + 1;
    constexpr int HP = LROWS * WP;
    constexpr int HV = HP * 4, HI = (HV + 255) / 256;     // halo vectors (8 channels each)
    constexpr int NP = NB / 32, PS = 4 / NP;               // n parts, pixel parts of a tile
    constexpr int KS = 128 / PS / 16;                      // 16-pixel MFMA steps per wave and tile
    constexpr int YB = NP * 128 * LDM, HB = ((HP * LDM + 511) / 512) * 512;       // elements per dy / halo stage
    constexpr int YI = NP * 2;                             // dy DMA instructions per wave and tile (1 KiB each)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const lds0 = reinterpret_cast<bf16*>(smem);      // [3][YB] dy stages, then [2][HB] halo stages
    bf16* const halo0 = lds0 + 3 * YB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int np = NP == 2 ? (wave_s >> 1) : 0, pp = NP == 2 ? (wave_s & 1) : wave_s;      // this wave's n part / pixel part
    const int H = g.Hin, BH = g.B * H, nT = BH / TR;
    const int nCt = g.Cin / 32, nNt = g.N / NB, nNC = nCt * nNt;
    const int L = blockIdx.x;
    int nc, split;
    if (p.splits % 8 == 0) {          // blocks L, L+8 share an XCD: all (n,c) slabs of one pixel range on one L2
        const int xcd = L & 7, slot = L >> 3;
        nc = slot % nNC;
        split = (slot / nNC) * 8 + xcd;
    } else {
        nc = L % nNC;
        split = L / nNC;
    }
    const int n0 = (nc / nCt) * NB, c0 = (nc % nCt) * 32;
    const int t_begin = split * p.tiles_per, t_end = min(nT, t_begin + p.tiles_per);
    const sv_phase& P = g.phase[0];
    const uint64_t pdy = pack_taps(P.dy), pdx = pack_taps(P.dx);
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
    const bf16* __restrict__ DY = reinterpret_cast<const bf16*>(p.dy);
    const bool has_pro = p.pro_scale != nullptr;
    float pslope = has_pro ? p.pro_slope : 1.f;
    asm volatile("v_mov_b32 %0, %0" : "+v"(pslope));
    const int v = tid & 3;
    // without a prologue the transform is the identity (scale 1, shift 0, slope 1: exact bf16 -> fp32 -> bf16)
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (has_pro) {
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * v);
        s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * v + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * v);
        t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * v + 4);
    }
    // halo staging slots (as wgrad3x3_kernel).  kind: 0 = always zero, 1 = image row of this tile, 2 = row above the tile,
    //                                                  3 = row below it (valid only inside the same image)
    int hlds[HI], hkind[HI];
    uint32_t hoff[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int idx = tid + 256 * i;
        const int pix = min(idx, HV - 1) >> 2;
        const int lr = pix / WP, xx = pix - lr * WP;
        const int seg = lr / (HH + 1), off = lr - seg * (HH + 1);
        int kind = 1, rel = lr - 1 - seg;
        if (off == 0) {
            if (SEG == 1) { kind = seg == 0 ? 2 : 3; rel = seg == 0 ? -1 : TR; }
            else kind = 0;
        }
        if (xx == 0 || xx == WP - 1) kind = 0;
        hkind[i] = kind;
        const int hxc = min(max(xx - 1, 0), W - 1);
        hoff[i] = (uint32_t)(((rel + 1) * W + hxc) * g.ldx + c0 + 8 * v) * 2u;
        hlds[i] = idx < HV ? pix * LDM + 8 * v : -1;
    }
    const uint32_t hsafe = (uint32_t)(W * g.ldx + c0 + 8 * v) * 2u;
    // dy DMA: instruction k of this wave moves 16 pixels x 32 channels (n part k / 2 ... see below) = 1 KiB, lane-linear
    uint32_t yoff[YI];
#pragma unroll
    for (int k = 0; k < YI; ++k) {
        const int inst = 4 * k + wave;                 // 0 .. 8 NP - 1: n part = inst / 8, 16-pixel block = inst % 8
        const int px = 16 * (inst & 7) + (lane >> 2);
        yoff[k] = (uint32_t)(px * g.ldo + n0 + 32 * (inst >> 3) + 8 * (lane & 3)) * 2u;
    }
    const uint32_t l0 = (uint32_t)(uintptr_t)(wg_lds_ptr)lds0;
    const char* ybase = nullptr;
    const char* hbase = nullptr;
    bool top_ok = false, bot_ok = false;
    auto bases = [&](int tile) {
        const int gr0 = tile * TR;
        top_ok = (gr0 & (H - 1)) != 0;
        bot_ok = ((gr0 + TR) & (H - 1)) != 0;
        hbase = reinterpret_cast<const char*>(X) + ((int64_t)gr0 - 1) * W * g.ldx * 2;
        ybase = reinterpret_cast<const char*>(DY + (int64_t)gr0 * W * g.ldo);
    };
    // every VMEM instruction of the loop is spelled in assembly and awaited by the one vmcnt(0) per iteration: the compiler
    // neither sees the DMA's LDS writes nor counts assembly loads
    auto dma_y1 = [&](int ystage, auto K) {
        constexpr int k = decltype(K)::value;
        const uint32_t base = l0 + (uint32_t)ystage * (uint32_t)(YB * 2) + (uint32_t)wave_s * 1024u, yo = yoff[k];
        const char* yb = ybase;
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                     :: "s"(base), "n"(k * 4096), "v"(yo), "s"(yb) : "memory", "scc");
    };
    auto dma_y = [&](int ystage) { static_for<YI>([&](auto K) { dma_y1(ystage, K); }); };
    bf16x8 rh[HI];
    bool hok[HI];
    auto load_h1 = [&](auto I) {
        constexpr int i = decltype(I)::value;
        hok[i] = (hkind[i] == 1) | ((hkind[i] == 2) & top_ok) | ((hkind[i] == 3) & bot_ok);
        const uint32_t o = hok[i] ? hoff[i] : hsafe;
        const char* hb = hbase;
        bf16x8 r;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(o), "s"(hb) : "memory");
        rh[i] = r;
    };
    auto load_h = [&]() { static_for<HI>([&](auto I) { load_h1(I); }); };
    auto drain = [&]() {          // to the compiler the halo registers are (re)defined HERE
        if constexpr (HI == 4) {
            bf16x8 r0 = rh[0], r1 = rh[1], r2 = rh[2], r3 = rh[3];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
            rh[0] = r0; rh[1] = r1; rh[2] = r2; rh[3] = r3;
        } else {
            static_assert(HI == 3, "halo register sets of 3 or 4 vectors");
            bf16x8 r0 = rh[0], r1 = rh[1], r2 = rh[2];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2) :: "memory");
            rh[0] = r0; rh[1] = r1; rh[2] = r2;
        }
    };
    bf16x8 zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (bf16)0.f;
    auto store_h = [&](int hstage) {
        bf16* hs = halo0 + hstage * HB;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
#ifdef SV_WG3M_NO_XFORM
            bf16x8 o = rh[i];
#else
            bf16x8 o = bn_act8(rh[i], s0, s1, t0, t1, pslope);
#endif
            if (!hok[i]) o = zero;
            if (hlds[i] >= 0) *reinterpret_cast<bf16x8*>(hs + hlds[i]) = o;
        }
    };

    // fragment addresses (bytes).  Lane l of a 32x32x16 operand: row / column l & 31, k = 8 (l >> 5) .. + 7 = two transposing
    // reads of 4 pixels x 16 channels per 16-lane group: lane 4 q + p of a group addresses pixel q, channels 4 p .. 4 p + 3.
    const int kg = lane >> 5, grp = (lane >> 4) & 1, q4 = (lane & 15) >> 2;
    const int lpix = 8 * kg + q4;                           // this lane's pixel inside a 16-pixel step (second read: + 4)
    const int px0 = pp * (128 / PS);                        // first tile pixel of this wave's pixel part
    const int jrow0 = px0 >> WLOG;
    const int lrow = (lpix >> WLOG), lcol = lpix & (W - 1);   // (a step spans two image rows only at W = 8)
    const uint32_t chb = (uint32_t)(16 * grp + 4 * (lane & 3)) * 2u;
    uint32_t yaddr = l0 + (uint32_t)(np * 128 * LDM + (px0 + lpix) * LDM) * 2u + chb;      // + ystage * YB * 2
    uint32_t xaddr[9];                                                                       // + hstage * HB * 2
    {
        const int hrow = jrow0 + lrow;
        const uint32_t hb0 = l0 + (uint32_t)(3 * YB) * 2u + (uint32_t)(((hrow + 1 + hrow / HH) * WP + lcol + 1) * LDM) * 2u + chb;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            xaddr[t] = hb0 + (uint32_t)((tap_off(pdy, t) * WP + tap_off(pdx, t)) * LDM * 2);
    }
    typedef __attribute__((address_space(3))) s16x4 lds_v4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto rd = [&](uint32_t addr) {       // one 16-pixel fragment = two transposing reads (pixels 0..3 and 4..7 of the lane's k group)
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(uintptr_t)addr);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(uintptr_t)(addr + 4 * LDM * 2));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // byte offset of 16-pixel step ks inside this wave's pixel part: whole image rows (W <= 16) or half rows (W = 32)
    auto step_off = [](int ks) {
        const int pxs = 16 * ks, r = pxs >> WLOG, c = pxs & (W - 1);
        return (uint32_t)((r * WP + c) * LDM * 2);
    };

    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (t_begin < t_end) {
        // prologue: tile t_begin complete in stage 0; tile t_begin + 1 requested (halo registers, dy into stage 1)
        bases(t_begin);
        dma_y(0);
        load_h();
        drain();
        store_h(0);
        bases(min(t_begin + 1, t_end - 1));
        dma_y(1);
        load_h();
        __syncthreads();
        int ys = 0, hs = 0;                     // stages of the tile on the MFMAs
        for (int tile = t_begin; tile < t_end; ++tile) {
            // the requests of the previous iteration (halo registers + dy of tile + 1) have had a whole tile period
            drain();
            const int ys1 = ys == 2 ? 0 : ys + 1, ys2 = ys1 == 2 ? 0 : ys1 + 1;
#ifndef SV_WG3M_NO_HST
            store_h(hs ^ 1);                    // halo of tile + 1 (nobody reads that stage: tile - 1 ended at the last barrier)
#endif
            bases(min(tile + 2, t_end - 1));    // (past the end: a harmless re-load of the last tile)
#ifndef SV_WG3M_NO_LOAD
            dma_y(ys2);
            load_h();
#endif
            // ---- the MFMAs of `tile` ------------------------------------------------------------------------------
#ifndef SV_WG3M_NO_MMA
            const uint32_t yb = yaddr + (uint32_t)ys * (uint32_t)(YB * 2), hfl = (uint32_t)hs * (uint32_t)(HB * 2);
            bf16x8 fy = rd(yb), fx[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) fx[t] = rd(xaddr[t] + hfl);
            // rolling fragments: tap t's registers receive the next step's tap t right behind its MFMA -- eight MFMAs before
            // they are needed; the sched_barrier keeps that order (left alone, the scheduler sinks every read down to its use
            // to save registers and each MFMA then waits out a full LDS round trip)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fyn = fy;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy, fx[t], acc[t], 0, 0, 0);
                    if (ks + 1 < KS) {
                        fx[t] = rd(xaddr[t] + hfl + step_off(ks + 1));
                        if (t == 0) fyn = rd(yb + (uint32_t)(16 * (ks + 1) * LDM * 2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                fy = fyn;
            }
#endif
            __syncthreads();
            ys = ys1;
            hs ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (redundant) requests, before the stages are reused

    // ---- the pixel parts of a tile meet in LDS (fixed order), then the block publishes ONE 9 x NB x 32 slab -------------
    // D layout of 32x32: lane holds column c = lane & 31, rows n = 8 (r >> 2) + 4 (lane >> 5) + (r & 3), r = 0..15
    float* const red = reinterpret_cast<float*>(smem);      // [9][NB][32] floats (fits the staging area)
    __syncthreads();
#pragma unroll
    for (int part = 0; part < PS; ++part) {
        if (pp == part) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* q = red + ((t * NB + 32 * np + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 32 + (lane & 31));
                    if (part == 0) *q = acc[t][r];
                    else *q += acc[t][r];
                }
        }
        __syncthreads();
    }
    const int64_t slab = (int64_t)g.N * g.T_orig * g.Cin;
    float* dst = p.ws ? p.ws + ((int64_t)blockIdx.y * p.splits + split) * slab : p.dw;
    for (int row = tid >> 3; row < 9 * NB; row += 32) {       // eight lanes per (tap, n) row of 32 floats
        const int t = row / NB, n = row - t * NB;
        const f32x4 val = *reinterpret_cast<const f32x4*>(red + row * 32 + 4 * (tid & 7));
        float* o = dst + ((int64_t)(n0 + n) * g.T_orig + P.torig[t]) * g.Cin + c0 + 4 * (tid & 7);
        if (p.ws) *reinterpret_cast<f32x4*>(o) = val;
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(o + j, val[j]);
        }
    }
}


// dw[i] += sum_s ws[s][i].  A block owns 256/G float4 columns; its G thread groups each sum every G-th slab, meet in
// LDS, and ONE float atomic per output leaves the block (many small slabs -- 512 x 36 KB for the 32-channel stage --
// used to meet in dw through 32 atomics per output, which cost more than reading the slabs).
template <bool ATOMIC>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* ws, int splits, int64_t n, float* dw, int G) {
    __shared__ f32x4 part[256];
    const int cols = 256 / G, col = threadIdx.x % cols, grp = threadIdx.x / cols;
    const int64_t i = ((int64_t)blockIdx.x * cols + col) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
#pragma unroll 4
        for (int k = grp; k < splits; k += G) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (int64_t)k * n + i);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    if (G > 1) {
        part[threadIdx.x] = s;
        __syncthreads();
        for (int h = G >> 1; h >= 1; h >>= 1) {
            if (grp < h) {
                const f32x4 o = part[threadIdx.x + h * cols];
                s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
                part[threadIdx.x] = s;
            }
            __syncthreads();
        }
    }
    // always atomic: the backward of the other branch of the step may be adding to dw concurrently.  The block's 4 * cols
    // consecutive outputs go out lane-contiguous (through LDS): a wave's atomic instruction then touches two cache
    // lines instead of eight (the float atomics of the 3.7 M-element 640-channel slab cost 40 of the kernel's 52 us)
    if (ATOMIC) {
        __syncthreads();
        if (grp == 0) part[col] = s;
        __syncthreads();
        const float* flat = reinterpret_cast<const float*>(part);
        const int64_t b0 = (int64_t)blockIdx.x * cols * 4;
        for (int j = threadIdx.x; j < 4 * cols; j += 256)
            if (b0 + j < n) atomicAdd(dw + b0 + j, flat[j]);
        return;
    }
    if (grp == 0 && i < n) {
        if (ATOMIC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(dw + i + r, s[r]);
        } else {          // (timing ablation SV_SLAB_PLAIN: wrong when two streams accumulate concurrently)
            f32x4 o = *reinterpret_cast<f32x4*>(dw + i);
            o[0] += s[0]; o[1] += s[1]; o[2] += s[2]; o[3] += s[3];
            *reinterpret_cast<f32x4*>(dw + i) = o;
        }
    }
}

static void launch_slab_reduce(const float* ws, int splits, int64_t n, float* dw, hipStream_t s) {
    int G = 1;                                       // >= 8 slabs per thread group, >= ~256 blocks where possible
    while (G < 32 && G * 2 * 8 <= splits && (n / 4 + 256 / G - 1) / (256 / G) < 256) G *= 2;
    const int cols = 256 / G;
    const unsigned gx = (unsigned)((n / 4 + cols - 1) / cols);
    hipLaunchKernelGGL(slab_reduce_kernel<true>, dim3(gx), dim3(256), 0, s, ws, splits, n, dw, G);
}

template <typename T, int WLOG>
int launch(const sv_geom* g, const wg3_params& p, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W;
    const int nNC = (g->N / 32) * (g->Cin / 32);
    const int grid = p.splits * nNC;
    constexpr int HHn = (TR < W) ? TR : W, LROWSn = TR + TR / HHn + 1;
    const size_t lds = (size_t)(128 + LROWSn * (W + 2)) * LDH * sizeof(T);
    sv_prof_begin(s);
    if (sizeof(T) == 2 && g->N * g->Cin <= 64 * 64)
        hipLaunchKernelGGL((wgrad3x3_kernel<T, WLOG, 2>), dim3(grid, p.groups), dim3(256), lds, s, *g, sv_expand_wg(*g, p, p.groups, (int)sizeof(T)));
    else
        hipLaunchKernelGGL((wgrad3x3_kernel<T, WLOG, 1>), dim3(grid, p.groups), dim3(256), lds, s, *g, sv_expand_wg(*g, p, p.groups, (int)sizeof(T)));
    sv_prof_end(s);               // the event bracket times the main kernel only (comparable with rocprofv3)
    if (p.ws) {
        const int64_t n = (int64_t)g->N * g->T_orig * g->Cin;      // multiple of 4 (Cin % 32 == 0)
        launch_slab_reduce(p.ws, p.splits * p.groups, n, p.dw, s);
    }
    return sv_check_launch("sv_wgrad(3x3)");
}

template <int WLOG, int NB>
int launch_m(const sv_geom* g, const wg3_params& p, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W;
    constexpr int HHn = (TR < W) ? TR : W, LROWSn = TR + TR / HHn + 1, HPn = LROWSn * (W + 2);
    constexpr int YB = (NB / 32) * 128 * LDM, HB = ((HPn * LDM + 511) / 512) * 512;
    constexpr size_t stage_bytes = (size_t)(3 * YB + 2 * HB) * 2, red_bytes = (size_t)9 * NB * 32 * 4;
    constexpr size_t lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
    const int nNC = (g->N / NB) * (g->Cin / 32);
    const int grid = p.splits * nNC;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3m_kernel<WLOG, NB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(wgrad3x3m)");
        optin = true;
    }
    auto PG = sv_expand_wg(*g, p, p.groups, 2);
    sv_prof_begin(s);
    hipLaunchKernelGGL((wgrad3x3m_kernel<WLOG, NB>), dim3(grid, p.groups), dim3(256), lds, s, *g, PG);
    sv_prof_end(s);
    if (p.ws) {
        const int64_t n = (int64_t)g->N * g->T_orig * g->Cin;
        launch_slab_reduce(p.ws, p.splits * p.groups, n, p.dw, s);
    }
    return sv_check_launch("sv_wgrad(3x3 m)");
}

template <int WLOG>
int launch_wide(const sv_geom* g, const wg3_params& p, hipStream_t s) {
    constexpr int W = 1 << WLOG, TR = 128 / W;
    const int nC = g->Cin / 32, nNt = g->N / 160;
    const int grid = p.unit > 0 ? 8 * ((p.splits * nNt * (nC / p.unit) + 7) / 8) * p.unit : p.splits * nNt * nC;
    constexpr int HHc = (TR < W) ? TR : W, LROWSc = TR + TR / HHc + 1;
    const size_t lds = (size_t)(128 * LDY + LROWSc * (W + 2) * LDH + 512) * 2 * 2;      // two stages
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3w_kernel<WLOG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(wgrad3x3w)");
        optin = true;
    }
    sv_prof_begin(s);
    hipLaunchKernelGGL((wgrad3x3w_kernel<WLOG>), dim3(grid), dim3(256), lds, s, *g, sv_expand_wg(*g, p, p.groups, 2));
    sv_prof_end(s);
    if (p.ws) {
        const int64_t n = (int64_t)g->N * g->T_orig * g->Cin;
        launch_slab_reduce(p.ws, p.splits, n, p.dw, s);
    }
    return sv_check_launch("sv_wgrad(3x3 wide)");
}

}  // namespace

void sv_slab_reduce(const float* ws, int nslabs, int64_t n, float* dw, hipStream_t s) { launch_slab_reduce(ws, nslabs, n, dw, s); }

// Returns 1 and sets *rc when the geometry is a stride-1 3x3 convolution this kernel covers.
int sv_wgrad3x3_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift,
                    float pro_slope, const void* dy, float* dw, float* ws, int64_t ws_elems, int groups, hipStream_t s,
                    int* rc) {
    if (g->nphase != 1 || g->phase[0].ntap != 9 || g->sy != 1 || g->sx != 1 || g->osy != 1 || g->osx != 1) return 0;
    if (g->Hq != g->Hin || g->Wq != g->Win || g->Hout != g->Hin || g->Wout != g->Win || g->Hin != g->Win) return 0;
    if (g->Win != 8 && g->Win != 16 && g->Win != 32) return 0;
    if (g->Cin % 32 != 0 || g->N % 32 != 0 || g->ldx != g->Cin || g->ldo != g->N) return 0;
    if (g->phase[0].ooy != 0 || g->phase[0].oox != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (g->phase[0].dy[t] < -1 || g->phase[0].dy[t] > 1 || g->phase[0].dx[t] < -1 || g->phase[0].dx[t] > 1) return 0;
    const int TR = 128 / g->Win;
    if ((g->B * g->Hin) % TR != 0) return 0;
    wg3_params p;
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_slope = pro_slope; p.dy = dy; p.dw = dw;
    p.unit = 0;
    p.groups = groups;
    const int nT = g->B * g->Hin / TR;          // per group
    if (!sv_disabled(SV_K_WGRAD3X3W) && dtype == SV_BF16 && g->N % 160 == 0 && g->Cin >= 96) {
        // wide layers: 160 x 32 slabs, one block (one wave per SIMD) per CU.  Pick the split count and the affinity unit
        // (a divisor of the chunk count) that minimise the modelled time:
        //   compute: rounds-of-32-CUs-per-XCD x (tiles per block + publishing a 160 x 32 x 9 slab, ~4 tiles) x 2.0 us
        //   HBM    : dy is read once per affinity unit of chunks (the unit's blocks share an L2), x ~1.5 times (halo), the
        //            slabs are written and read back; ~5 TB/s.  (Measured, 160-channel layer at B = 512: unit 1 = 1.27 GB
        //            per launch, 5.7 TB/s -- the kernel was HBM-bound; the unit of 5 chunks reads dy once)
        //   + the slab reduction after the kernel (every split adds one slab read at ~4 TB/s).
        // Splits whose slabs do not fit the caller's workspace would have to meet in dw through float atomics
        // (measured: 64 splits of the 320-channel layer = 59 M atomics, 393 us instead of ~300) and are only taken when
        // nothing fits.
        const int nC = g->Cin / 32, nNt = g->N / 160;
        const int64_t slab_elems = (int64_t)g->N * g->T_orig * g->Cin;
        const double rows = (double)g->B * g->Hin * g->Win;
        const double dy_bytes = rows * g->N * 2.0, x_bytes = rows * g->Cin * 2.0, slab_bytes = (double)slab_elems * 4.0;
        int splits = 1, unit = 0;
        double best = -1.0;
        for (int sp = 1; sp <= nT && sp <= 128; ++sp) {
            const int tp = (nT + sp - 1) / sp;
            if ((nT + tp - 1) / tp != sp) continue;               // no empty splits
            const bool fits = sp == 1 || (ws && ws_elems >= (int64_t)sp * slab_elems);
            for (int u = 1; u <= nC && u <= 32; ++u) {
                if (nC % u) continue;
                const int units = sp * nNt * (nC / u);
                const int per_xcd = (units + 7) / 8 * u;
                // (a block walks its tile range of every group: + ~1.5 tiles per group for the pipeline restart)
                const double t_mma = (double)((per_xcd + 31) / 32) * (groups * (tp + 1.5) + 2.5) * 2.0e-6;
                const int nsl = sp;                       // slabs to publish and reduce
                const double t_hbm = (groups * (dy_bytes * (nC / u) + 1.5 * x_bytes * nNt) + (nsl > 1 ? 2.0 * nsl * slab_bytes : 0.0)) / 5.0e12;
                double cost = (t_mma > t_hbm ? t_mma : t_hbm) + (nsl > 1 ? nsl * slab_bytes / 4.0e12 : 0.0) - 1e-9 * u;
                if (!fits) cost += 1.0;
                if (best < 0 || cost < best) { best = cost; splits = sp; unit = u; }
            }
        }
        p.splits = splits;
        p.unit = unit;
        p.tiles_per = (nT + splits - 1) / splits;
        const int64_t needw = (int64_t)splits * g->N * g->T_orig * g->Cin;
        p.ws = (ws && ws_elems >= needw && splits > 1) ? ws : nullptr;
        if (!p.ws && splits > 1 && sv_deterministic()) return 0;      // (the splits would meet in dw through float atomics)
        switch (g->Win) {
            case 32: *rc = launch_wide<5>(g, p, s); break;
            case 16: *rc = launch_wide<4>(g, p, s); break;
            default: *rc = launch_wide<3>(g, p, s); break;
        }
        return 1;
    }
    // bf16: the 32x32x16 kernel with 64- (or 32-) channel n tiles
    const bool use_m = dtype == SV_BF16 && !sv_disabled(SV_K_WGRAD3X3M);
    const int NBm = use_m && g->N % 64 == 0 ? 64 : 32;
    const int nNC = (g->N / NBm) * (g->Cin / 32);
    // ~two persistent blocks per CU; every block should still see a few tiles
    const int budget = sv_persistent_blocks();
    const int target = budget / groups > 32 ? budget / groups : 32;
    int splits = (target + nNC - 1) / nNC;
    if (splits > nT) splits = nT;
    if (splits >= 8) splits = splits / 8 * 8;
    if (splits < 1) splits = 1;
    p.tiles_per = (nT + splits - 1) / splits;
    if (splits < 8) splits = (nT + p.tiles_per - 1) / p.tiles_per;
    p.splits = splits;
    const int64_t need = (int64_t)splits * groups * g->N * g->T_orig * g->Cin;
    p.ws = (ws && ws_elems >= need && splits * groups > 1) ? ws : nullptr;    // no workspace: atomics straight into dw
    if (!p.ws && splits * groups > 1 && sv_deterministic()) return 0;
    if (use_m) {
        switch (g->Win) {
            case 32: *rc = NBm == 64 ? launch_m<5, 64>(g, p, s) : launch_m<5, 32>(g, p, s); break;
            case 16: *rc = NBm == 64 ? launch_m<4, 64>(g, p, s) : launch_m<4, 32>(g, p, s); break;
            default: *rc = NBm == 64 ? launch_m<3, 64>(g, p, s) : launch_m<3, 32>(g, p, s); break;
        }
        return 1;
    }
    switch (g->Win) {
        case 32: *rc = dtype == SV_BF16 ? launch<bf16, 5>(g, p, s) : launch<float, 5>(g, p, s); break;
        case 16: *rc = dtype == SV_BF16 ? launch<bf16, 4>(g, p, s) : launch<float, 4>(g, p, s); break;
        default: *rc = dtype == SV_BF16 ? launch<bf16, 3>(g, p, s) : launch<float, 3>(g, p, s); break;
    }
    return 1;
}

extern "C" int sv_debug_wgrad_tile_program(int halo_vectors, int* items, int* waits) {
    SV_REQUIRE(items && waits, SV_E_ARG, "sv_debug_wgrad_tile_program: null argument");
    SV_REQUIRE(halo_vectors == 3 || halo_vectors == 4, SV_E_ARG, "sv_debug_wgrad_tile_program: halo_vectors=%d", halo_vectors);
    auto copy = [&](const auto& S) {
        for (int i = 0; i < 180; ++i)
            for (int j = 0; j < 3; ++j) items[3 * i + j] = S.item[i][j];
        for (int v = 0; v < 4; ++v) waits[v] = 0;      // no counted waits: one vmcnt(0) in front of the barrier
        waits[4] = S.last_vmem_a;
        return S.ok ? SV_OK : SV_E_SHAPE;
    };
    static constexpr WSched<3> S3 = make_wsched<3>();
    static constexpr WSched<4> S4 = make_wsched<4>();
    return halo_vectors == 3 ? copy(S3) : copy(S4);
}
