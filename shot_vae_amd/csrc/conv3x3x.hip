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
//   * LDS (1 block per CU): two halo stages (21.7 KB) + a six-slot weight ring (slot = K step mod 6; 61 KB) + the epilogue's
//     scratch apart from both (40 KB); every LDS address is a lane base + an immediate, and three barriers per chunk
//     (after taps 1, 4, 7) are all the synchronisation: the slices of steps 3m+6..3m+8 are copied in steps 3m+2..3m+3 and
//     awaited at the barrier after step 3m+4;
//   * PERSISTENT blocks: one per CU, each walks its (pixel tile, channel tile) items; the K loop runs on across items --
//     the first chunk of the next item is staged under the last chunk of this one -- so only the epilogue (conv3x3w's,
//     conv3x3w_epilogue.inc) stands between two items; no per-item prologue.
// Same fused prologue / epilogue contract as sv_igemm (include/shotvae_hip.h); replaces
// shot_vae_model/wideresnet.py:13-43 (Conv2d 3x3 + BatchNorm2d + LeakyReLU + residual) for the wide layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

#ifndef SV_X3_EPD
#define SV_X3_EPD 1
#endif
// An MFMA spelled in inline assembly is invisible to the compiler's hazard recognizer.  The hardware needs two wait states
// between a VALU write of a register and an MFMA that reads it; in front of its own MFMAs the compiler inserts them, in front of
// this assembly it does not -- and the VALU write may be its own: a spill re-load (v_accvgpr_read) or a copy of a fragment
// register that the allocator placed right before the statement.  Found with the compile-time epilogue modes below: at
// 506-512 registers the data-gradient variant re-loaded the first tap's fragments of a new item from their spill slots
// directly in front of MFMAs 0, 2, ... and produced wrong even rows; the run-time-flag binary had the same exposure and was
// right by luck of its allocation.  Every MFMA therefore carries its own `s_nop 1` (measured: not slower -- 320 / 299 us
// against 337 / 308 at 160 channels).
#ifndef SV_X3_NOP
#define SV_X3_NOP 1
#endif
#if SV_X3_NOP
#define SV_X3_PRE "s_nop 1\n\t"
#else
#define SV_X3_PRE
#endif
#ifndef SV_X3_DRAIN
#define SV_X3_DRAIN 1      // 1: the epilogue's stores have left before the K loop resumes
#endif
#ifndef SV_X3_DMAH
#define SV_X3_DMAH 1       // data gradients without a load prologue: the halo by LDS-DMA (make_xsched_dma)
#endif
#ifndef SV_X3_MODES
#define SV_X3_MODES 2      // epilogue fusion flags at compile time: 1 = for the forward launch kinds, 2 = and the data gradient; 0 = run-time flags only
#endif

#ifndef SV_X3_STAMP
#define SV_X3_STAMP 0      // diagnostic build: (s_memrealtime, s_memtime) of every block at start / after the prologue / after every K loop / after every epilogue
#endif
#if SV_X3_STAMP
__device__ unsigned long long sv_x3_stamp_buf[256 * 24 * 2];
extern "C" int sv_x3_stamps_read(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sv_x3_stamp_buf), sizeof(sv_x3_stamp_buf)); }
#define SV_X3_STAMP_AT(ev) do { if (tid == 0 && blockIdx.y == 0 && (ev) < 24) { \
    sv_x3_stamp_buf[(blockIdx.x * 24 + (ev)) * 2] = __builtin_amdgcn_s_memrealtime(); \
    sv_x3_stamp_buf[(blockIdx.x * 24 + (ev)) * 2 + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SV_X3_STAMP_AT(ev) do { } while (0)
#endif

namespace {

__device__ __attribute__((aligned(16))) uint32_t sv_x3_zero16[4];      // source of the halo's padding vectors (DMAH)

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
//   2000 + 3 s + i        : DMA instruction i of weight slice s (see make_xsched: four of this chunk, five of the next)
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
    // weight DMA (six ring slots): slice codes 2000 + 3 s + i, s = 0..8 meaning
    //   s = 0: (this chunk, tap 5) in tap 0;  s = 1, 2: (this, 6), (this, 7) in tap 2;  s = 3: (this, 8) in tap 3;
    //   s = 4: (next, 0) in tap 5;  s = 5, 6: (next, 1), (next, 2) in tap 6;  s = 7, 8: (next, 3), (next, 4) in tap 8
    // -- each inside the three-step window its slot's previous fragments allow, in the read-free gaps 0, 2, 4 (10, 12, 14);
    //    the second half of tap 5 stays free for the halo re-loads that follow the barrier after tap 4 (below)
    {
        const int tap_of[9] = {0, 2, 2, 3, 5, 6, 6, 8, 8}, first_gap[9] = {0, 0, 10, 0, 0, 10, 0, 0, 10};
        for (int sl = 0; sl < 9; ++sl)
            for (int i = 0; i < 3; ++i) {
                const int gp = 20 * tap_of[sl] + first_gap[sl] + 2 * i;
                ok = put(gp, 2000 + 3 * sl + i) && ok;
                used[gp] = 1;
            }
    }
    ok = put(20 * 8 + 0, 5000) && ok;          // before the first fragment read of tap 8 (gap 1)
    // the BatchNorm pass starts with tap 1 (the coefficients were requested at the end of the previous chunk's pass):
    // two or three steps per gap next to fragment reads, four otherwise; a halo register is re-loaded in a read-free,
    // VMEM-free gap after its vector's store, the coefficients after the last vector.
    // Every barrier of the chunk waits with vmcnt(0) (a wait for LDS-DMA counts only the younger LDS-DMA, and there is
    // none: see the bookkeeping below), so it also drains the register loads in flight: those are therefore issued in the
    // windows that FOLLOW a barrier (gaps 41..62 behind tap 1's, 101..144 behind tap 4's) and have 1 000+ cycles to land
    // before the next one; none is issued in tap 8 / tap 0, so that after the barrier behind tap 7 every register of the
    // chunk after next has arrived (the compiler is told so there: nothing is pending across the loop's back edge)
    int m = 0, next_reload = 0, coef_q = 0;
    const int nsteps = X_HSTEPS * X_HI;
    for (int gp = 20; gp < 160; ++gp) {
        const int mm = gp % 20;
        const bool window = (gp > 40 && gp <= 62) || (gp > 100 && gp <= 144);
        const bool fre = !x_gap_has_read(mm) && !used[gp];
        if (fre && window && next_reload < X_HI && m >= X_HSTEPS * (next_reload + 1)) {
            ok = put(gp, 3000 + next_reload) && ok;
            used[gp] = 1;
            ++next_reload;
            continue;
        }
        if (fre && window && m >= nsteps && next_reload == X_HI && coef_q < 4) {
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
    auto index_of = [&](int code) { for (int i = 0; i < nv; ++i) if (order[i] == code) return i; return -1; };
    // A hand-counted wait counts only YOUNGER INSTRUCTIONS OF THE SAME KIND: LDS-DMA and register loads do not retire in
    // one common order (a younger weight copy may leave the counter before an older register load and vice versa), so
    // instructions of the other kind must not be relied upon to keep the counter up.  (The price: such a wait also
    // drains every older instruction of the other kind.)
    auto since_reg = [&](int idx, int p) {                 // register loads after order[idx], wrapping to position p
        int cnt = 0;
        for (int i = idx + 1; i < nv; ++i) if (order[i] >= 3000) ++cnt;
        for (int i = 0; i < nv; ++i) if (pos[i] < p && order[i] >= 3000) ++cnt;
        return cnt;
    };
    auto dma_between = [&](int idx, int p) {               // LDS-DMA instructions after order[idx] and before position p
        int cnt = 0;
        for (int i = idx + 1; i < nv; ++i) if (pos[i] < p && order[i] < 3000) ++cnt;
        return cnt;
    };
    auto last_dma_before = [&](int p) { int l = -1; for (int i = 0; i < nv; ++i) if (pos[i] < p && order[i] < 3000) l = i; return l; };
    for (int sl = 0; sl < X_HI; ++sl) S.vm_slot[sl] = since_reg(index_of(3000 + sl), find_item(4000 + sl));
    S.vm_coef = since_reg(index_of(3503), find_item(4500));
    // barrier after tap 1: (this, 3), (this, 4) from the previous chunk's tap 8 and (this, 5) from tap 0 have landed;
    // after tap 4: (this, 6..8) from taps 2, 3;  after tap 7: (next, 0..2) from taps 5, 6.  Each barrier needs EVERY
    // DMA issued before it (none is issued between the last slice it needs and the barrier), so the counts are zero
    S.vm_b1 = dma_between(last_dma_before(40 * 8), 40 * 8);
    S.vm_b4 = dma_between(last_dma_before(100 * 8), 100 * 8);
    S.vm_b7 = dma_between(last_dma_before(160 * 8), 160 * 8);
    S.ok = S.ok && index_of(2000 + 3 * 0 + 2) <= last_dma_before(40 * 8) && index_of(2000 + 3 * 3 + 2) <= last_dma_before(100 * 8) &&
           index_of(2000 + 3 * 6 + 2) <= last_dma_before(160 * 8) && index_of(2000 + 3 * 5 + 2) <= last_dma_before(160 * 8);
    // no register load after the barrier behind tap 7 (see above)
    for (int i = 0; i < nv; ++i) if (order[i] >= 3000 && pos[i] >= 160 * 8) S.ok = false;
    return S;
}

// The chunk program of the DATA GRADIENT without a load prologue (DMAH): the halo needs no transform, so it goes global -> LDS
// by LDS-DMA like the weights -- no halo registers, no BatchNorm steps, no register loads, no hand-counted waits at all (every
// VMEM instruction of the loop is an LDS-DMA, the three barriers drain them).  Items: the weight slices as in make_xsched;
//   6000 + slot : DMA of halo vector `slot` of the NEXT chunk into the other stage (free since the barrier behind tap 7 of the
//                 previous chunk; read from tap 8 on, behind the barrier after tap 7): slots 0..2 behind the barrier after
//                 tap 1 (awaited 34+ MFMAs later at the barrier after tap 4), slots 3..5 behind the barrier after tap 4
constexpr XSched make_xsched_dma() {
    XSched S{};
    for (int i = 0; i < X_NGAP; ++i)
        for (int j = 0; j < 6; ++j) S.item[i][j] = 0;
    bool ok = true;
    int used[X_NGAP] = {};
    auto put = [&](int gap, int code) {
        if (code >= 2000 && code != 5000) {
            if (used[gap] || x_gap_has_read(gap % 20)) return false;
            used[gap] = 1;
        }
        for (int j = 0; j < 6; ++j)
            if (S.item[gap][j] == 0) { S.item[gap][j] = code; return true; }
        return false;
    };
    {
        const int tap_of[9] = {0, 2, 2, 3, 5, 6, 6, 8, 8}, first_gap[9] = {0, 0, 10, 0, 0, 10, 0, 0, 10};
        for (int sl = 0; sl < 9; ++sl)
            for (int i = 0; i < 3; ++i) ok = put(20 * tap_of[sl] + first_gap[sl] + 2 * i, 2000 + 3 * sl + i) && ok;
    }
    ok = put(20 * 8 + 0, 5000) && ok;
    const int hgap[X_HI] = {46, 56, 66, 106, 110, 112};
    for (int j = 0; j < X_HI; ++j) ok = put(hgap[j], 6000 + j) && ok;
    for (int v = 0; v < X_HI; ++v) S.vm_slot[v] = 0;
    S.vm_coef = 0; S.vm_b1 = 0; S.vm_b4 = 0; S.vm_b7 = 0;      // every barrier drains the queue: all of it is LDS-DMA
    S.ok = ok;
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
    static constexpr int WS = 4 * BN, WI = 3, WBUF = WS * 16, NSLOT = 6;
    static constexpr int SCR = 64 * 36 * 4;                 // epilogue transpose scratch per wave
    static constexpr int OFF_W = 2 * HB, OFF_SCR = OFF_W + NSLOT * WBUF;
    static constexpr int OFF_SSUM = OFF_SCR + 4 * SCR + 5 * BN * 4, LDS = OFF_SSUM + 4 * 2 * BN * 4;     // sums: one copy per wave
    static_assert((HS + 255) / 256 == X_HI, "six halo vectors per thread");
    static_assert(LDS <= 160 * 1024, "one block per CU");
};

// MODE: the epilogue's fusion flags at compile time (conv3x3w_epilogue.inc: 0 = from the arguments, 1 = statistics,
// 2 = residual + statistics, 3 = activation-backward)
// DMAH: the halo by LDS-DMA (data gradients without a load prologue: make_xsched_dma)
template <int WLOG, bool REV, int MODE, bool DMAH>
__global__ __launch_bounds__(256, 1) void conv3x3x_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    using C = XCfg<WLOG>;
    constexpr int NF = C::NF, BN = C::BN, W = C::W, TR = C::TR, WP = C::WP, HH = C::HH, SEG = C::SEG;
    constexpr int HS = C::HS, HB = C::HB, WBUF = C::WBUF, SWS = C::SWS, HI = X_HI, HSTEPS = X_HSTEPS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ssum = reinterpret_cast<float*>(smem + C::OFF_SSUM);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: DMA destinations stay in SGPRs
    const int r = lane & 31, h = lane >> 5;
    SV_X3_STAMP_AT(0);
    const int H = g.Hin, BH = g.B * H, nT = BH / TR, nNt = g.N / BN;
    const int Cin = g.Cin, nck = Cin / 32;

    // ---- work items of this block.  XCD-affine: the 32-odd blocks of an XCD share a contiguous range of pixel tiles (one
    //      copy of the weights and of the halo rows in that XCD's L2); item k = (pixel tile, channel tile) number
    //      slot + k * blocks-per-XCD of the range
    const int per = (nT + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int mt_lo = xcd * per, mt_hi = min(mt_lo + per, nT);
    const int n_items_xcd = max(mt_hi - mt_lo, 0) * nNt;
    const int cnt = slot_id < n_items_xcd ? (n_items_xcd - slot_id + nbx - 1) / nbx : 0;
    if (cnt == 0) return;

    const sv_phase& P = g.phase[0];
    const char* const Xb = reinterpret_cast<const char*>(a.x);
    const char* const Wp = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(a.w) + P.w_off);
    const bool has_pro = a.pro_scale != nullptr;
    const float slope = has_pro ? a.pro_slope : 1.f;
    // (held in registers: the inline assembly of the K loop clobbers memory, so every `a.pro_scale` inside it is a scalar load
    //  from the argument segment followed by s_waitcnt lgkmcnt(0) -- four per chunk, each also draining the fragment reads)
    const char* const pscale_b = reinterpret_cast<const char*>(a.pro_scale);
    const char* const pshift_b = reinterpret_cast<const char*>(a.pro_shift);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;

    // ---- a K-loop position: (item, chunk) and what staging needs to know about it -------------------------------------------
    struct Pos {
        int it, c;                 // item number of this block, 32-channel chunk
        int gr0, n0;               // first image row of the pixel tile, first output channel
        const char* xrow;          // x at (row gr0 - 1, column 0, channel 32 c)
        const char* wsl;           // packed weights of (channel tile, tap 0, channel 32 c)
        bool top_ok, bot_ok;       // the rows above / below the tile belong to the same image
    };
    auto make_pos = [&](int it, int c) {
        Pos p;
        p.it = it; p.c = c;
        const int idx = slot_id + it * nbx;
        p.gr0 = (mt_lo + idx / nNt) * TR;
        p.n0 = (idx % nNt) * BN;
        p.xrow = Xb + ((int64_t)(p.gr0 - 1) * W * g.ldx + 32 * c) * 2;
        p.wsl = Wp + ((int64_t)p.n0 * 9 * Cin + 32 * c) * 2;
        p.top_ok = (p.gr0 & (H - 1)) != 0;
        p.bot_ok = ((p.gr0 + TR) & (H - 1)) != 0;
        return p;
    };
    auto next_pos = [&](const Pos& p) {              // past the last position: itself (a harmless re-staging)
        if (p.c + 1 < nck) {                         // same item: the next 32 channels (no divisions in the chunk loop)
            Pos q = p;
            q.c = p.c + 1;
            q.xrow = p.xrow + 64;
            q.wsl = p.wsl + 64;
            return q;
        }
        if (p.it + 1 < cnt) return make_pos(p.it + 1, 0);
        return p;
    };


    // ---- halo vectors of this thread: vector s = 256 j + tid = (halo pixel s >> 2, logical 8-channel quarter tid & 3),
    //      stored at the swizzled quarter; kind: 0 zero (padding column / spacer / dummy), 1 row of the tile, 2 the row
    //      above, 3 the row below.  Offsets are unsigned byte counts from (row gr0 - 1): tile-invariant ----------------------
    const int lq = tid & 3;
    uint32_t hoff[HI], hoffd[HI];      // hoffd: the vector whose PHYSICAL quarter is tid & 3 (LDS-DMA writes lane-linear)
    int hlds[HI], hkind[HI];
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
        hkind[j] = kind;
        const int xc = min(max(xx - 1, 0), W - 1);
        hoff[j] = (uint32_t)(((rel + 1) * W + xc) * g.ldx + 8 * lq) * 2u;
        hoffd[j] = (uint32_t)(((rel + 1) * W + xc) * g.ldx + 8 * (lq ^ ((xx >> SWS) & 3))) * 2u;
        hlds[j] = s < HS ? pix * 64 + 16 * (lq ^ ((xx >> SWS) & 3)) : HS * 16 + 16 * (tid & 63);
    }
    const uint32_t hsafe = (uint32_t)(W * g.ldx + 8 * lq) * 2u;        // the tile's first pixel: always inside the tensor
    // ---- weight DMA: 64 consecutive 16-byte vectors of a [160][32] slice per wave instruction; the k-quarter swizzle
    //      (row >> 2) & 3 is applied to the source address
    uint32_t wsrc[3], wdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int base = min((i * 4 + wave) * 64, C::WS - 64);
        const int s = base + lane, row = s >> 2, q = (s & 3) ^ ((row >> 2) & 3);
        wsrc[i] = (uint32_t)(row * 9 * Cin + 8 * q) * 2u;
        wdst[i] = lds0 + C::OFF_W + (uint32_t)base * 16u;
    }
    // instruction i of the slice of tap t at position p, whose chunk has ring parity `par`: slot = (t + 3 par) mod 6
    // = t mod 6 + 3 par for t mod 6 < 3, t mod 6 - 3 par otherwise -- an immediate on one of two scalar bases
    auto dma_w = [&](const Pos& p, int par, auto TT, auto I) {
        constexpr int i = decltype(I)::value, t = decltype(TT)::value;
        const char* src = p.wsl + (int64_t)t * Cin * 2;
        const uint32_t base = (t % 6) < 3 ? wdst[i] + (uint32_t)(par * 3 * WBUF) : wdst[i] - (uint32_t)(par * 3 * WBUF);
        const uint32_t off = wsrc[i];
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                     :: "s"(base), "n"((t % 6) * WBUF), "v"(off), "s"(src) : "memory", "scc");
    };
    // (DMAH) halo vector j of position p into the stage at `stage_off`: 64 consecutive vectors per wave instruction; a padding
    // vector (padding column, spacer row, a row of the neighbouring image) copies 16 zero bytes; the tail beyond the stage is masked
    auto dma_h = [&](const Pos& p, uint32_t stage_off, auto J) {
        constexpr int j = decltype(J)::value;
        const bool okj = (hkind[j] == 1) | ((hkind[j] == 2) & p.top_ok) | ((hkind[j] == 3) & p.bot_ok);
        const char* src = okj ? p.xrow + hoffd[j] : reinterpret_cast<const char*>(sv_x3_zero16);
        const uint32_t m0v = lds0 + stage_off + (uint32_t)(256 * j + 64 * wave) * 16u;
        if (256 * j + 255 < HS || 256 * j + tid < HS)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(src) : "memory");
    };
    // ---- halo registers (+ their validity: it travels with the data) and the BatchNorm coefficients of the thread's quarter
    u32x4 rh[HI];
    bool hok[HI];
    f32x4 csc[2], csh[2];
    {
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zer = {0.f, 0.f, 0.f, 0.f};
        csc[0] = csc[1] = one;
        csh[0] = csh[1] = zer;
    }
    auto load_h = [&](u32x4* rh, bool* hok, const Pos& p, auto J) {
        constexpr int j = decltype(J)::value;
        hok[j] = (hkind[j] == 1) | ((hkind[j] == 2) & p.top_ok) | ((hkind[j] == 3) & p.bot_ok);
        const char* src = p.xrow;
        const uint32_t off = hok[j] ? hoff[j] : hsafe;
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
    auto load_coef = [&](f32x4* csc, f32x4* csh, const Pos& p, auto Q) {     // q: 0,1 = scale lo/hi, 2,3 = shift lo/hi
        constexpr int q = decltype(Q)::value;
        if (!has_pro) return;
        const char* src = (q < 2 ? pscale_b : pshift_b) + (int64_t)p.c * 128;
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
    auto hstep = [&](const u32x4* rh, const bool* hok, const f32x4* csc, const f32x4* csh, auto J, auto ST, uint32_t stage_off) {
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

    // ---- fragment addressing (byte addresses in LDS; tap / k half / channel group are immediates) ---------------------------
    // weights: lane (r, h) reads row 32 i + r, k-quarter (2 ks + h) ^ swizzle(row), of ring slot (t + 3 par) mod 6:
    // base [ks][0] serves the taps whose slot is t mod 6 + 3 par (t mod 6 < 3), base [ks][1] those with t mod 6 - 3 par
    uint32_t wa[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        wa[ks][0] = wa[ks][1] = lds0 + C::OFF_W + 16 * (4 * r + ((2 * ks + h) ^ ((r >> 2) & 3)));
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
        const int off = (t % 6) * WBUF + i * 2048;
        A[ks][i] = *reinterpret_cast<const lds_v8*>((uintptr_t)(wa[ks][(t % 6) < 3 ? 0 : 1] + (uint32_t)off));
    };
    auto read_p = [&](int t, int ks, int f) {
        const int ty = REV ? 2 - t / 3 : t / 3, tx = REV ? 2 - t % 3 : t % 3;
        Bf[ks][f] = *reinterpret_cast<const lds_v8*>((uintptr_t)(px[f][tx][ks] + (uint32_t)((ty * WP + tx) * 64)));
    };

    static constexpr XSched SCHED = DMAH ? make_xsched_dma() : make_xsched();
    static_assert(SCHED.ok, "the chunk program does not fit the gaps");

    // ---- prologue (once per block): every request first -- the first position's halo + coefficients (into registers of
    //      their own), the halo registers + coefficients of the second position, the weight slices of steps 0..5 -- then the
    //      first BatchNorm pass into stage 0 while the rest is in flight, and one full wait: the waits of the loop count the
    //      VMEM instructions of a steady-state chunk, which the first chunk has not issued yet
    Pos cur = make_pos(0, 0);
    Pos nxt = next_pos(cur);
    if constexpr (DMAH) {
        static_for<HI>([&](auto J) { dma_h(cur, 0u, J); });
        static_for<6>([&](auto S) { static_for<3>([&](auto I) { dma_w(cur, 0, S, I); }); });
    } else {
        u32x4 rh0[HI];
        bool hok0[HI];
        f32x4 csc0[2] = {csc[0], csc[1]}, csh0[2] = {csh[0], csh[1]};
        static_for<HI>([&](auto J) { load_h(rh0, hok0, cur, J); });
        static_for<4>([&](auto Q) { load_coef(csc0, csh0, cur, Q); });
        static_for<6>([&](auto S) { static_for<3>([&](auto I) { dma_w(cur, 0, S, I); }); });
        static_for<HI>([&](auto J) { load_h(rh, hok, nxt, J); });
        static_for<4>([&](auto Q) { load_coef(csc, csh, nxt, Q); });
        // (a full drain, once per block: counting on the first HI + 4 requests -- HBM -- to return before the 18 younger
        //  weight DMAs -- L2 -- is not safe: LDS-DMA and register loads do not retire in one common order)
        wait_coef(csc0, csh0, std::integral_constant<int, 0>{});
        static_for<HI>([&](auto J) {
            wait_h(rh0, J, std::integral_constant<int, 0>{});
            static_for<HSTEPS>([&](auto ST) { hstep(rh0, hok0, csc0, csh0, J, ST, 0u); });
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr (!DMAH) {
        static_for<HI>([&](auto J) { wait_h(rh, J, std::integral_constant<int, 0>{}); });
        wait_coef(csc, csh, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    static_for<2>([&](auto KS) {
        static_for<NF>([&](auto I) { read_w(0, decltype(KS)::value, decltype(I)::value); });
        static_for<2>([&](auto F) { read_p(0, decltype(KS)::value, decltype(F)::value); });
    });

    SV_X3_STAMP_AT(1);
    // ---- the K loop over all positions of the block: one 32-channel chunk = nine taps = 180 MFMAs -------------------------
    int par = 0;                       // halo stage and ring parity of the current chunk
    for (int item = 0; item < cnt; ++item) {
      // (two plain nested loops: the accumulators are loop-carried through the chunk loop only -- a conditional epilogue
      //  inside one loop makes the register allocator merge two versions of all 160 of them)
      f32x16 acc[2][NF];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[f][i][e] = 0.f;
      for (int cc = 0; cc < nck; ++cc) {
        const Pos nn = next_pos(nxt);
        const uint32_t other = (uint32_t)((par ^ 1) * HB);
        static_for<9>([&](auto T) {
            constexpr int t = decltype(T)::value, tn = (t + 1) % 9;
            static_for<20>([&](auto M) {
                constexpr int m = decltype(M)::value, ks = m / 10, i = (m % 10) / 2, f = m & 1;
                asm volatile(SV_X3_PRE "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[f][i]) : "v"(A[ks][i]), "v"(Bf[ks][f]));
                static_for<6>([&](auto JJ) {
                    constexpr int code = SCHED.item[20 * t + m][decltype(JJ)::value];
                    if constexpr (code >= 1000 && code < 2000)
                        hstep(rh, hok, csc, csh, std::integral_constant<int, (code - 1000) / HSTEPS>{},
                              std::integral_constant<int, (code - 1000) % HSTEPS>{}, other);
                    if constexpr (code >= 2000 && code < 3000) {
                        constexpr int sl = (code - 2000) / 3;
                        constexpr int tap = sl < 4 ? 5 + sl : sl - 4;          // (this, 5..8), (next, 0..4)
                        dma_w(sl < 4 ? cur : nxt, sl < 4 ? par : par ^ 1, std::integral_constant<int, tap>{}, std::integral_constant<int, (code - 2000) % 3>{});
                    }
                    if constexpr (code >= 6000 && code < 6000 + HI) dma_h(nxt, other, std::integral_constant<int, code - 6000>{});
                    if constexpr (code >= 3000 && code < 3500) load_h(rh, hok, nn, std::integral_constant<int, code - 3000>{});
                    if constexpr (code >= 3500 && code < 4000) load_coef(csc, csh, nn, std::integral_constant<int, code - 3500>{});
                    if constexpr (code >= 4000 && code < 4500)
                        wait_h(rh, std::integral_constant<int, code - 4000>{}, std::integral_constant<int, SCHED.vm_slot[code - 4000]>{});
                    if constexpr (code == 4500) wait_coef(csc, csh, std::integral_constant<int, SCHED.vm_coef>{});
                    if constexpr (code == 5000) {          // the next tap 0 belongs to the next chunk: other halo stage, ring + 3
                        const uint32_t flip = par ? (uint32_t)(-HB) : (uint32_t)HB;
#pragma unroll
                        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
                            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                                for (int kk = 0; kk < 2; ++kk) px[ff][tx][kk] += flip;
                        const uint32_t wflip = (uint32_t)(3 * WBUF);
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            wa[kk][0] = par ? wa[kk][0] - wflip : wa[kk][0] + wflip;
                            wa[kk][1] = par ? wa[kk][1] + wflip : wa[kk][1] - wflip;
                        }
                    }
                });
                // the next tap's fragments, each right after the last MFMA that reads its registers
                if constexpr (f == 1) read_w(tn, ks, i);
                if constexpr (i == 4) read_p(tn, ks, f);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (t == 1 || t == 4 || t == 7) {
                constexpr int n = t == 1 ? SCHED.vm_b1 : t == 4 ? SCHED.vm_b4 : SCHED.vm_b7;
#if SV_X3_STAMP
                if (item == 1 && cc == 2) SV_X3_STAMP_AT(12 + 2 * (t / 3));          // (before the barrier's wait)
#endif
                if constexpr (t == 7 && !DMAH) {
                    // The halo / coefficient registers of the chunk after next were requested by assembly the compiler
                    // cannot see through: to it they are defined the moment the load is issued.  Nothing it might do with
                    // them later -- a copy on the loop's back edge, a spill into the AGPR half across the epilogue -- may
                    // happen before the data has arrived.  This wait (vmcnt(0): all of them are older, the chunk program
                    // issues none behind this barrier) is where it is told that they have
                    static_assert(n == 0, "the barrier after tap 7 drains the queue");
                    u32x4 h0 = rh[0], h1 = rh[1], h2 = rh[2], h3 = rh[3], h4 = rh[4], h5 = rh[5];
                    f32x4 c0 = csc[0], c1 = csc[1], c2 = csh[0], c3 = csh[1];
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5),
                                 "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) :: "memory");
                    rh[0] = h0; rh[1] = h1; rh[2] = h2; rh[3] = h3; rh[4] = h4; rh[5] = h5;
                    csc[0] = c0; csc[1] = c1; csh[0] = c2; csh[1] = c3;
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(n) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#if SV_X3_STAMP
                if (item == 1 && cc == 2) SV_X3_STAMP_AT(13 + 2 * (t / 3));          // (behind the barrier)
#endif
            }
        });
        par ^= 1;
        if (cc + 1 < nck) { cur = nxt; nxt = nn; }
      }
      {
            // ---- the item is complete: epilogue (its scratch lies apart from the stages and the ring, which already hold
            //      the next item's first chunk), accumulators back to zero.  The stores must have left before the loop's
            //      hand-counted vmcnt waits resume (stores and loads share the counter but not its order)
            // (the per-wave sums need no clearing: every address is written once per item; the flush of item i and the
            //  writes of item i + 1 are separated by the chunk loop's barriers)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            if (item < 5) SV_X3_STAMP_AT(2 + 2 * item);
            const int n0 = cur.n0, gr0 = cur.gr0;
            {
                constexpr int SV_EPD = SV_X3_EPD;
#define SV_EPI_NSCR 1
#define SV_EPI_BASE C::OFF_SCR
#define SV_EPI_ALIAS 1
#define SV_EPI_MODE MODE
#define SV_EPI_WAVE_SUMS 1
#include "conv3x3w_epilogue.inc"
#undef SV_EPI_MODE
#undef SV_EPI_WAVE_SUMS
#undef SV_EPI_NSCR
#undef SV_EPI_BASE
#undef SV_EPI_ALIAS
            }
            // (SV_X3_DRAIN = 0: outstanding stores only make the loop's counted waits conservative -- a wait for a register load
            //  counts the younger register loads, which retire after it: with the load outstanding the counter is above the count)
            if (SV_X3_DRAIN) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (item < 5) SV_X3_STAMP_AT(3 + 2 * item);
      }
      {   // step to the next item's first chunk (staged during this item's last one)
          const Pos nn2 = next_pos(nxt);
          cur = nxt;
          nxt = nn2;
      }
    }
}

template <int WLOG, bool REV, int MODE, bool DMAH = false>
int launch_x4(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    using C = XCfg<WLOG>;
    const int nT = g->B * g->Hin / C::TR, nNt = g->N / C::BN;
    // persistent: one block per CU (256 = 8 XCDs x 32), fewer when there are fewer items per XCD
    // (a batched launch shares the 256 CUs among its groups)
#ifndef SV_X3_CAP
#define SV_X3_CAP 32       // (diagnostic: fewer blocks per XCD)
#endif
    const int G = sv_ngroups(a->groups), cap = SV_X3_CAP / G > 1 ? SV_X3_CAP / G : 1;
    const int per = (nT + 7) / 8, items_xcd = per * nNt;
    const int grid = 8 * (items_xcd < cap ? items_xcd : cap);
    const size_t lds = (size_t)C::LDS;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3x_kernel<WLOG, REV, MODE, DMAH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(conv3x3x)");
        optin = true;
    }
    SV_LAUNCH_GATE(grid, a);          // (deterministic mode: a replica per block -- the gate checks replicas >= 4 * grid)
    sv_prof_begin(s);
    hipLaunchKernelGGL((conv3x3x_kernel<WLOG, REV, MODE, DMAH>), dim3(grid, G), dim3(256), lds, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(conv3x3x)");
}

// forward launches come with statistics (+ residual), data gradients with the activation-backward epilogue; anything else
// (bias, no statistics, ...) takes the binary that reads the flags at run time
template <int WLOG, bool REV>
int launch_x3(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
#if SV_X3_MODES
    if (!a->bias) {
        if constexpr (REV) {
#if SV_X3_MODES > 1
#if SV_X3_DMAH
            if (a->ex && !a->residual && !a->pro_scale && g->ldx % 8 == 0) return launch_x4<WLOG, REV, 3, true>(g, a, s);
#endif
            if (a->ex && !a->residual) return launch_x4<WLOG, REV, 3>(g, a, s);
#endif
        } else {
            if (a->stats && a->residual && !a->ex) return launch_x4<WLOG, REV, 2>(g, a, s);
            if (a->stats && !a->residual && !a->ex) return launch_x4<WLOG, REV, 1>(g, a, s);
        }
    }
#endif
    return launch_x4<WLOG, REV, 0>(g, a, s);
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
    // Default for 160-channel tiles; sv_set_option(SV_OPT_DISABLE_MASK, SV_K_CONV3X3X) falls back to conv3x3w (the two
    // accumulate in the same order: tests compare them bitwise).
    // History: the first persistent version lost ~5 % of the config-4 training runs to a non-finite loss -- its prologue
    // counted on the oldest (HBM) register loads to retire before 18 younger (L2) LDS-DMA instructions; the two kinds do not
    // share one completion order.  Since then every hand-counted wait of this file counts younger instructions of its own
    // kind only (make_xsched; tests/test_abi_cpu.py checks the program).
    if (sv_disabled(SV_K_CONV3X3X)) return 0;
    if (g->N % 160 != 0 || g->Cin % 32 != 0 || g->Cin < 96) return 0;
    *rc = fwd ? launch_x2<false>(g, a, s) : launch_x2<true>(g, a, s);
    return 1;
}

extern "C" int sv_debug_conv_chunk_program(int* items, int* waits) {
    SV_REQUIRE(items && waits, SV_E_ARG, "sv_debug_conv_chunk_program: null argument");
    static constexpr XSched S = make_xsched();
    for (int i = 0; i < X_NGAP; ++i)
        for (int j = 0; j < 6; ++j) items[6 * i + j] = S.item[i][j];
    for (int v = 0; v < X_HI; ++v) waits[v] = S.vm_slot[v];
    waits[6] = S.vm_coef; waits[7] = S.vm_b1; waits[8] = S.vm_b4; waits[9] = S.vm_b7;
    return S.ok ? SV_OK : SV_E_SHAPE;
}
