// Weight gradient of every conv-like layer as a split-M GEMM on MFMA (gfx950):
//
//   dW[n][torig(t)][c] += sum_m dy[opos(m)][n] * A(m,t,c),   A = LeakyReLU(x*scale+shift) gathered
//
// One block owns a (16*TN channels-out) x (one tap) x (16*TC channels-in) tile of dW and a range of
// output positions m.  The reduction index m is the MFMA k dimension, but both operands are stored
// m-major in HBM (NHWC), so each wave stages ITS OWN 32 rows of dy and A row-major in a wave-private
// LDS region (no block barrier in the main loop) and reads the k-major fragments back with the
// gfx950 transposing read ds_read_b64_tr_b16 (bf16) or plain 4-byte reads (fp32 16x16x4 layout).
// The four waves' accumulators are combined with LDS float atomics, then added to the fp32 gradient
// buffer with one global atomic per element per block -- gradients of all four forwards of a step
// accumulate in place (main_shot_vae.py:324,364 semantics).
#include <stdlib.h>

#include "common.h"

void sv_slab_reduce(const float* ws, int nslabs, int64_t n, float* dw, hipStream_t s);      // wgrad3x3.hip

namespace {

constexpr int RW = 32;   // rows (output positions) per wave per iteration = one MFMA k chunk

struct wg_params {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    float pro_slope;
    const void* dy;
    float* dw;
    int splits, m_per;          // m ranges
    int groups;                 // batched launch: blockIdx.y = group (common.h)
    int wsh, hwsh;              // log2(Wq), log2(Hq*Wq) or -1
    int ntap_total;
    int incr;                   // incremental addressing allowed (SV_K_WGRAD_INCR)
    int64_t slab_stride;        // deterministic mode: m range `split` adds into dw + split * slab_stride (zeroed slabs, one adder
                                // per element), a fixed-order pass sums the slabs afterwards; 0 = every split into dw
};

template <bool FAST>
__device__ __forceinline__ void decode_m(const sv_geom& g, const wg_params& p, int m, int& b, int& qy, int& qx) {
    if (FAST) {
        b = m >> p.hwsh;
        const int r = m & ((1 << p.hwsh) - 1);
        qy = r >> p.wsh;
        qx = r & ((1 << p.wsh) - 1);
    } else {
        const int hw = g.Hq * g.Wq;
        b = m / hw;
        const int r = m - b * hw;
        qy = r / g.Wq;
        qx = r - qy * g.Wq;
    }
}

// k-major MFMA fragment (16 indices starting at col0, 32 consecutive rows of S) ---------------------
template <bool USE_TR>
__device__ __forceinline__ bf16x8 frag_t(const bf16* S, int ld, int col0, int lane) {
    const int gq = lane >> 4, i = lane & 15;
    bf16x8 f;
    if (USE_TR) {
        // block = 4 rows x 16 cols; lane 4q+p of the 16-lane group addresses row q, cols 4p..4p+3 and
        // receives column (lane&15) of the 4 rows.
        const bf16* a0 = S + (8 * gq + (i >> 2)) * ld + col0 + 4 * (i & 3);
        typedef __attribute__((address_space(3))) s16x4 lds_v4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * ld));
        union { s16x4 s[2]; bf16x8 b; } u;
        u.s[0] = lo;
        u.s[1] = hi;
        f = u.b;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = S[(8 * gq + j) * ld + col0 + i];
    }
    return f;
}
template <bool USE_TR>
__device__ __forceinline__ f32x8 frag_t(const float* S, int ld, int col0, int lane) {
    // element j feeds v_mfma_f32_16x16x4_f32 step j, whose k index is lane>>4: row = 4*j + (lane>>4)
    const int gq = lane >> 4, i = lane & 15;
    f32x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = S[(4 * j + gq) * ld + col0 + i];
    return f;
}

template <typename T, int TN, int TC, bool USE_TR, bool FAST, bool INCR>
__global__ __launch_bounds__(256) void wgrad_kernel(const sv_geom g, const sv_wg_g<wg_params> PG) {
    const wg_params& p = PG.g[blockIdx.y];
    typedef typename V8<T>::type V;
    constexpr int BNw = 16 * TN, BCw = 16 * TC;
    constexpr int LDN = BNw + 8, LDC = BCw + 8;           // LDS row strides (elements)
    constexpr int WAVE_ELEMS = RW * (LDN + LDC);
    static_assert(4 * WAVE_ELEMS * sizeof(T) >= BNw * BCw * sizeof(float) || true, "");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* Ys = reinterpret_cast<T*>(smem) + wave * WAVE_ELEMS;   // [RW][LDN]
    T* Xs = Ys + RW * LDN;                                    // [RW][LDC]

    const int M = g.B * g.Hq * g.Wq;
    const int nNt = (g.N + BNw - 1) / BNw;
    const int nCt = (g.Cin + BCw - 1) / BCw;
    const int tiles = nNt * p.ntap_total * nCt;
    // splits is either < 8 (plain mapping) or a multiple of 8: then blocks L and L+8 share an XCD and all
    // (tap, channel-tile) blocks of one m-range run on the same XCD, so dy / x are fetched into one L2.
    const int L = blockIdx.x;
    int tile, split;
    if (p.splits % 8 == 0) {
        const int xcd = L & 7, slot = L >> 3;
        tile = slot % tiles;
        split = (slot / tiles) * 8 + xcd;
    } else {
        tile = L % tiles;
        split = L / tiles;
    }
    const int ct = tile % nCt;
    int tapg = (tile / nCt) % p.ntap_total;
    const int n0 = (tile / (nCt * p.ntap_total)) * BNw;
    const int c0 = ct * BCw;
    int ph = 0;
    while (tapg >= g.phase[ph].ntap) { tapg -= g.phase[ph].ntap; ++ph; }
    const sv_phase& P = g.phase[ph];
    const int dy = P.dy[tapg], dx = P.dx[tapg], torig = P.torig[tapg];
    const int ooy = P.ooy, oox = P.oox;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);
    const bool has_pro = p.pro_scale != nullptr;
    float pslope = p.pro_slope;               // pinned in a vector register (conv3x3p_kernel: no re-load from the argument segment)
    asm volatile("v_mov_b32 %0, %0" : "+v"(pslope));

    const int m_begin = split * p.m_per;
    const int m_end = min(M, m_begin + p.m_per);

    // per-lane load slots: TN passes over the dy rows, TC passes over the x rows
    constexpr int VPRN = 2 * TN, RPN = 64 / VPRN;     // vectors per row, rows per pass
    constexpr int VPRC = 2 * TC, RPC = 64 / VPRC;
    const int vn = lane % VPRN, rn = lane / VPRN;
    const int vc = lane % VPRC, rc = lane / VPRC;
    const bool nvec_ok = n0 + 8 * vn < g.N;
    const bool cvec_ok = c0 + 8 * vc < g.Cin;

    f32x4 s0, s1, t0, t1;
    if (has_pro && cvec_ok) {
        s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * vc);
        s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + c0 + 8 * vc + 4);
        t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * vc);
        t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + c0 + 8 * vc + 4);
    }

    V ry[TN], rx[TC];
    bool okx[TC];
    const int nvo = nvec_ok ? n0 + 8 * vn : 0;
    const int cvo = cvec_ok ? c0 + 8 * vc : 0;
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;
    // FAST (power-of-two grid): every slot keeps (row m, grid row qy, element offset) and ADVANCES them by the 128 rows of an
    // iteration with adds -- the decode + address arithmetic per slot per iteration (34 quarter-rate v_mul_lo_u32 and 17
    // 64-bit multiply-adds in the ISA: ~800 issue cycles against 256 of MFMA) made the kernel VALU-bound (23 VALU per MFMA).
    const int STEP = 4 * RW;
    const int dimg = INCR ? (STEP >> p.hwsh) : 0;                               // whole images per step
    const int dq = INCR ? ((STEP - (dimg << p.hwsh)) >> p.wsh) : 0;             // remaining grid rows per step (< Hq)
    const int rs_y = g.osy * g.Wout * g.ldo, is_y = g.Hout * g.Wout * g.ldo;     // (32-bit element offsets: the incremental
    const int rs_x = g.sy * g.Win * g.ldx, is_x = g.Hin * g.Win * g.ldx;         //  path is taken for tensors < 2^31 elements)
    const int st_y = dimg * is_y + dq * rs_y, wr_y = is_y - g.Hq * rs_y;          // per step / extra on a wrap into the next image
    const int st_x = dimg * is_x + dq * rs_x, wr_x = is_x - g.Hq * rs_x;
    constexpr bool incr = INCR;       // (the host checked: power-of-two grid, Wq <= 128, tensors < 2^31 elements)
    int my[TN], qyy[TN], mx[TC], qyx[TC];
    int offy[TN], offx[TC];
    bool ixok[TC];
    if (incr) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            my[i] = m_begin + RW * wave + rn + RPN * i;
            int b, qy, qx;
            decode_m<FAST>(g, p, my[i], b, qy, qx);
            qyy[i] = qy;
            offy[i] = ((b * g.Hout + qy * g.osy + ooy) * g.Wout + qx * g.osx + oox) * g.ldo + nvo;
        }
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            mx[i] = m_begin + RW * wave + rc + RPC * i;
            int b, qy, qx;
            decode_m<FAST>(g, p, mx[i], b, qy, qx);
            qyx[i] = qy;
            const int ix = qx * g.sx + dx;
            ixok[i] = cvec_ok && (unsigned)ix < (unsigned)g.Win;
            offx[i] = ((b * g.Hin + qy * g.sy + dy) * g.Win + ix) * g.ldx + cvo;      // (may point outside: used only when valid)
        }
    }
    // Branch-free loader: every address is inside the tensor (or replaced by element 0), every load is issued
    // unconditionally (so the 2*TN..2*TC loads of an iteration are in flight together), invalid
    // rows / padding are zeroed by a select afterwards.
    auto load_global = [&](int mbase) {   // mbase = first row of this wave's 32-row slab (consecutive calls: + 128 rows)
        if (incr) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const bool ok = my[i] < m_end && nvec_ok;
                const V val = *reinterpret_cast<const V*>(DY + (ok ? offy[i] : 0));
                ry[i] = ok ? val : zero;
                my[i] += STEP;
                qyy[i] += dq;
                offy[i] += st_y;
                if (qyy[i] >= g.Hq) { qyy[i] -= g.Hq; offy[i] += wr_y; }
            }
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const int iy = qyx[i] * g.sy + dy;
                const bool ok = mx[i] < m_end && ixok[i] && (unsigned)iy < (unsigned)g.Hin;
                const V val = *reinterpret_cast<const V*>(X + (ok ? offx[i] : 0));
                okx[i] = ok;
                rx[i] = ok ? val : zero;
                mx[i] += STEP;
                qyx[i] += dq;
                offx[i] += st_x;
                if (qyx[i] >= g.Hq) { qyx[i] -= g.Hq; offx[i] += wr_x; }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int m = mbase + rn + RPN * i;
            const bool ok = m < m_end && nvec_ok;
            int b, qy, qx;
            decode_m<FAST>(g, p, min(m, m_end - 1), b, qy, qx);
            const int64_t op = (int64_t)(b * g.Hout + qy * g.osy + ooy) * g.Wout + qx * g.osx + oox;
            const V val = *reinterpret_cast<const V*>(DY + op * g.ldo + nvo);
            ry[i] = ok ? val : zero;
        }
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const int m = mbase + rc + RPC * i;
            int b, qy, qx;
            decode_m<FAST>(g, p, min(m, m_end - 1), b, qy, qx);
            const int iy = qy * g.sy + dy, ix = qx * g.sx + dx;
            const bool ok = m < m_end && cvec_ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
            const int iyc = min(max(iy, 0), g.Hin - 1), ixc = min(max(ix, 0), g.Win - 1);
            const V val = *reinterpret_cast<const V*>(X + ((int64_t)(b * g.Hin + iyc) * g.Win + ixc) * g.ldx + cvo);
            okx[i] = ok;
            rx[i] = ok ? val : zero;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < TN; ++i)
            *reinterpret_cast<V*>(Ys + (rn + RPN * i) * LDN + 8 * vn) = ry[i];
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            V o = rx[i];
            if (has_pro && okx[i]) o = bn_act8(rx[i], s0, s1, t0, t1, pslope);
            *reinterpret_cast<V*>(Xs + (rc + RPC * i) * LDC + 8 * vc) = o;
        }
    };

    f32x4 acc[TN][TC];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (m_end <= m_begin) return;      // (block-uniform) nothing to do for this m-range
    const int niter = (m_end - m_begin + 4 * RW - 1) / (4 * RW);
    load_global(m_begin + RW * wave);
    for (int it = 0; it < niter; ++it) {
        store_lds();
        if (it + 1 < niter) load_global(m_begin + (it + 1) * 4 * RW + RW * wave);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        V fy[TN], fx[TC];
#pragma unroll
        for (int i = 0; i < TN; ++i) fy[i] = frag_t<USE_TR>(Ys, LDN, 16 * i, lane);
#pragma unroll
        for (int j = 0; j < TC; ++j) fx[j] = frag_t<USE_TR>(Xs, LDC, 16 * j, lane);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j) mma32(acc[i][j], fy[i], fx[j]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- combine the four waves (plain LDS stores + a summing pass: LDS float atomics are far too slow),
    //      then one global atomic per element -----------------------------------------------------------
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][BNw][BCw]
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[wave * BNw * BCw + (16 * i + 4 * fq + r) * BCw + 16 * j + fr] = acc[i][j][r];
    __syncthreads();
    float* const dwp = p.dw + (int64_t)split * p.slab_stride;
    for (int i = tid; i < BNw * BCw; i += 256) {
        const int n = n0 + i / BCw, c = c0 + i % BCw;
        const float v = red[i] + red[BNw * BCw + i] + red[2 * BNw * BCw + i] + red[3 * BNw * BCw + i];
        if (n < g.N && c < g.Cin)
            atomicAdd(dwp + ((int64_t)n * g.T_orig + torig) * g.Cin + c, v);
    }
}

// Cooperative wide variant (bf16, transposing reads): one block owns a (32*TNW) x (one tap) x (32*TCW) tile of dW -- 160 x
// 160 for the WRN-28-10 odd layers -- and all four waves share each 32-row k chunk of dy and A (wave (wn, wc) multiplies the
// (16 TNW) x (16 TCW) quarter).  The 64 x 64 wave-private tiles above re-read every row of dy and x once per (64 channels
// out) x (tap) x (64 channels in): 9 GB of L2 -> LDS traffic on the 160 -> 320 stride-2 3x3 layer at 4 x 256 images, which
// is what bounded it (1.26 ms); here the same layer moves a third of that.
template <int TNW, int TCW, bool FAST>
__global__ __launch_bounds__(256, 2) void wgradc_kernel(const sv_geom g, const sv_wg_g<wg_params> PG) {
    const wg_params& p = PG.g[blockIdx.y];
    typedef bf16 T;
    typedef bf16x8 V;
    constexpr int BNb = 32 * TNW, BCb = 32 * TCW;
    constexpr int LDN = BNb + 8, LDC = BCb + 8;
    constexpr int VRN = BNb / 8, VRC = BCb / 8;                   // 16-byte vectors per row
    constexpr int NVY = (RW * VRN + 255) / 256, NVX = (RW * VRC + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const Ys = reinterpret_cast<T*>(smem);                     // [2][RW][LDN]
    T* const Xs = Ys + 2 * RW * LDN;                              // [2][RW][LDC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;

    const int M = g.B * g.Hq * g.Wq;
    const int nNt = (g.N + BNb - 1) / BNb;
    const int nCt = (g.Cin + BCb - 1) / BCb;
    const int tiles = nNt * p.ntap_total * nCt;
    const int L = blockIdx.x;
    int tile, split;
    if (p.splits % 8 == 0) {        // blocks L and L + 8 share an XCD: all tiles of one m-range read dy / x through one L2
        const int xcd = L & 7, slot = L >> 3;
        tile = slot % tiles;
        split = (slot / tiles) * 8 + xcd;
    } else {
        tile = L % tiles;
        split = L / tiles;
    }
    const int ct = tile % nCt;
    int tapg = (tile / nCt) % p.ntap_total;
    const int n0 = (tile / (nCt * p.ntap_total)) * BNb;
    const int c0 = ct * BCb;
    int ph = 0;
    while (tapg >= g.phase[ph].ntap) { tapg -= g.phase[ph].ntap; ++ph; }
    const sv_phase& P = g.phase[ph];
    const int dy = P.dy[tapg], dx = P.dx[tapg], torig = P.torig[tapg];
    const int ooy = P.ooy, oox = P.oox;

    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);
    const bool has_pro = p.pro_scale != nullptr;
    float pslope = p.pro_slope;               // pinned in a vector register (conv3x3p_kernel: no re-load from the argument segment)
    asm volatile("v_mov_b32 %0, %0" : "+v"(pslope));
    const int m_begin = split * p.m_per;
    const int m_end = min(M, m_begin + p.m_per);
    if (m_end <= m_begin) return;

    // loader slots: vector tid + 256 i of the [RW][VRN] / [RW][VRC] chunk
    int yrow[NVY], yv[NVY], xrow[NVX], xv[NVX];
    bool yon[NVY], xon[NVX];
#pragma unroll
    for (int i = 0; i < NVY; ++i) {
        const int idx = tid + 256 * i;
        yrow[i] = idx / VRN;
        yv[i] = idx - yrow[i] * VRN;
        yon[i] = idx < RW * VRN && n0 + 8 * yv[i] < g.N;
        if (idx >= RW * VRN) yrow[i] = 0, yv[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < NVX; ++i) {
        const int idx = tid + 256 * i;
        xrow[i] = idx / VRC;
        xv[i] = idx - xrow[i] * VRC;
        xon[i] = idx < RW * VRC && c0 + 8 * xv[i] < g.Cin;
        if (idx >= RW * VRC) xrow[i] = 0, xv[i] = 0;
    }
    // two register stages: the loads of step it + 2 are issued before the MFMAs of step it (a step's 25 MFMAs per wave are
    // ~0.2 us, a gather round trip 1.5-2 us: one stage ahead left the block waiting on every step)
    struct Stage { V ry[NVY], rx[NVX]; bool okx[NVX]; } SA, SB;
    V zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (T)0.f;
    auto load_global = [&](Stage& S, int mbase) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NVY; ++i) {
            const int m = mbase + yrow[i];
            const bool ok = m < m_end && yon[i];
            int b, qy, qx;
            decode_m<FAST>(g, p, min(m, m_end - 1), b, qy, qx);
            const int64_t op = (int64_t)(b * g.Hout + qy * g.osy + ooy) * g.Wout + qx * g.osx + oox;
            const V val = *reinterpret_cast<const V*>(DY + op * g.ldo + (yon[i] ? n0 + 8 * yv[i] : 0));
            S.ry[i] = ok ? val : zero;
        }
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
            const int m = mbase + xrow[i];
            int b, qy, qx;
            decode_m<FAST>(g, p, min(m, m_end - 1), b, qy, qx);
            const int iy = qy * g.sy + dy, ix = qx * g.sx + dx;
            const bool ok = m < m_end && xon[i] && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
            const int iyc = min(max(iy, 0), g.Hin - 1), ixc = min(max(ix, 0), g.Win - 1);
            const V val = *reinterpret_cast<const V*>(X + ((int64_t)(b * g.Hin + iyc) * g.Win + ixc) * g.ldx +
                                                      (xon[i] ? c0 + 8 * xv[i] : 0));
            S.okx[i] = ok;
            S.rx[i] = ok ? val : zero;
        }
    };
    auto store_lds = [&](const Stage& S, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NVY; ++i)
            if (tid + 256 * i < RW * VRN) *reinterpret_cast<V*>(Ys + (buf * RW + yrow[i]) * LDN + 8 * yv[i]) = S.ry[i];
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
            V o = S.rx[i];
            if (has_pro && S.okx[i]) {
                const int cc = c0 + 8 * xv[i];
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.pro_scale + cc), s1 = *reinterpret_cast<const f32x4*>(p.pro_scale + cc + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.pro_shift + cc), t1 = *reinterpret_cast<const f32x4*>(p.pro_shift + cc + 4);
                o = bn_act8(S.rx[i], s0, s1, t0, t1, pslope);
            }
            if (tid + 256 * i < RW * VRC) *reinterpret_cast<V*>(Xs + (buf * RW + xrow[i]) * LDC + 8 * xv[i]) = o;
        }
    };

    f32x4 acc[TNW][TCW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TCW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int niter = (m_end - m_begin + RW - 1) / RW;
    // step it: [loads of it + 2 -> the stage step it's data came from] [MFMAs on LDS buffer it & 1] [stage of it + 1 -> LDS]
    auto step = [&](int it, Stage& Snext, Stage& Sfar) __attribute__((always_inline)) {
        const int buf = it & 1;
        if (it + 2 < niter) load_global(Sfar, m_begin + (it + 2) * RW);
        V fy[TNW], fx[TCW];
#pragma unroll
        for (int i = 0; i < TNW; ++i) fy[i] = frag_t<true>(Ys + buf * RW * LDN, LDN, 16 * (TNW * wn + i), lane);
#pragma unroll
        for (int j = 0; j < TCW; ++j) fx[j] = frag_t<true>(Xs + buf * RW * LDC, LDC, 16 * (TCW * wc + j), lane);
#pragma unroll
        for (int i = 0; i < TNW; ++i)
#pragma unroll
            for (int j = 0; j < TCW; ++j) mma32(acc[i][j], fy[i], fx[j]);
        if (it + 1 < niter) store_lds(Snext, buf ^ 1);
        __syncthreads();
    };
    load_global(SA, m_begin);
    store_lds(SA, 0);
    if (niter > 1) load_global(SB, m_begin + RW);
    __syncthreads();
    for (int it = 0; it < niter; it += 2) {
        step(it, SB, SA);                         // step it + 1 is in SB; SA (step it, already in LDS) takes step it + 2
        if (it + 1 < niter) step(it + 1, SA, SB);
    }
    // every wave owns its quarter of the tile: one global atomic per element per block
    const int fr = lane & 15, fq = lane >> 4;
    float* const dwp = p.dw + (int64_t)split * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TCW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * (TNW * wn + i) + 4 * fq + r, c = c0 + 16 * (TCW * wc + j) + fr;
                if (n < g.N && c < g.Cin) atomicAdd(dwp + ((int64_t)n * g.T_orig + torig) * g.Cin + c, acc[i][j][r]);
            }
}

// deterministic mode: `splits` zeroed slabs of dw's size in the caller's workspace (p.dw -> the slabs); det_slabs_end adds them
// to dw in a fixed order
static int det_slabs_cap(const sv_geom* g, float* ws, int64_t ws_elems) {
    const int64_t n = (int64_t)g->N * g->T_orig * g->Cin;
    if (!ws || n % 4 != 0) return 0;
    const int64_t cap = ws_elems / n;
    return cap >= 2 ? (int)(cap > 64 ? 64 : cap) : 0;
}
static bool det_slabs_begin(const sv_geom* g, wg_params& p, float* ws, hipStream_t s) {
    const int64_t n = (int64_t)g->N * g->T_orig * g->Cin;
    if (hipMemsetAsync(ws, 0, (size_t)p.splits * n * sizeof(float), s) != hipSuccess) return false;
    p.slab_stride = n;
    p.dw = ws;
    return true;
}

template <int TNW, int TCW>
int launch_c(const sv_geom* g, wg_params p, int64_t M, hipStream_t s, float* det_ws = nullptr, int det_cap = 0) {
    constexpr int BNb = 32 * TNW, BCb = 32 * TCW;
    const int nNt = (g->N + BNb - 1) / BNb, nCt = (g->Cin + BCb - 1) / BCb;
    const int tiles = nNt * p.ntap_total * nCt;
    // two blocks per CU over the whole (batched) launch, at least 1024 rows per block
    int64_t want = (512 / p.groups + tiles - 1) / tiles;
    const int64_t maxs = (M + 1023) / 1024;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    int splits = (int)want;
    if (splits >= 8 || (maxs >= 8 && tiles * p.groups >= 32)) {
        // a multiple of 8 (XCD-affine mapping): of the three from ~one block per slot upwards, the one that leaves the
        // smallest idle tail on the 512 block slots
        const int k0 = splits >= 8 ? splits / 8 * 8 : 8;
        int best = k0;
        double beste = 0.;
        for (int k = k0; k <= k0 + 16 && (k == k0 || k <= maxs); k += 8) {
            const int64_t blocks = (int64_t)k * tiles * p.groups;
            if (k > k0 && blocks > 2048) break;
            const double e = (double)blocks / (double)((blocks + 511) / 512 * 512);
            if (e > beste + 0.02) beste = e, best = k;
        }
        splits = best;
    }
    if (det_ws && splits > det_cap) splits = det_cap >= 8 ? det_cap / 8 * 8 : det_cap;
    int64_t m_per = (M + splits - 1) / splits;
    m_per = (m_per + RW - 1) / RW * RW;
    if (splits < 8) splits = (int)((M + m_per - 1) / m_per);
    p.splits = splits;
    p.m_per = (int)m_per;
    float* const dw_final = p.dw;
    if (det_ws && !det_slabs_begin(g, p, det_ws, s)) return sv_check_launch("sv_wgrad(wide): slab clear");
    const size_t lds = (size_t)2 * RW * (BNb + 8 + BCb + 8) * sizeof(bf16);
    sv_prof_begin(s);
    if (p.hwsh >= 0)
        hipLaunchKernelGGL((wgradc_kernel<TNW, TCW, true>), dim3(splits * tiles, p.groups), dim3(256), lds, s, *g, sv_expand_wg(*g, p, p.groups, 2));
    else
        hipLaunchKernelGGL((wgradc_kernel<TNW, TCW, false>), dim3(splits * tiles, p.groups), dim3(256), lds, s, *g, sv_expand_wg(*g, p, p.groups, 2));
    sv_prof_end(s);
    if (det_ws) sv_slab_reduce(det_ws, p.splits, (int64_t)g->N * g->T_orig * g->Cin, dw_final, s);
    return sv_check_launch("sv_wgrad(wide)");
}

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

template <typename T, int TN, int TC, bool USE_TR, bool FAST, bool INCR>
int launch2(const sv_geom* g, const wg_params& p, hipStream_t s) {
    constexpr int BNw = 16 * TN, BCw = 16 * TC;
    const int nNt = (g->N + BNw - 1) / BNw, nCt = (g->Cin + BCw - 1) / BCw;
    const int tiles = nNt * p.ntap_total * nCt;
    const int grid = p.splits * tiles;
    size_t lds = (size_t)4 * RW * (BNw + 8 + BCw + 8) * sizeof(T);
    const size_t red = (size_t)4 * BNw * BCw * sizeof(float);
    if (red > lds) lds = red;
    sv_prof_begin(s);
    hipLaunchKernelGGL((wgrad_kernel<T, TN, TC, USE_TR, FAST, INCR>), dim3(grid, p.groups), dim3(256), lds, s, *g, sv_expand_wg(*g, p, p.groups, (int)sizeof(T)));
    sv_prof_end(s);
    return sv_check_launch("sv_wgrad");
}

template <typename T, int TN, int TC, bool USE_TR>
int launch(const sv_geom* g, const wg_params& p, hipStream_t s) {
    if (p.hwsh >= 0) {
        // (measured: a win where a step covers whole images -- decoder layers, 8x8 grids: -10 % -- a loss on the 16x16 grid
        //  of the 32 -> 64 stride-2 layer, 98 -> 112 us)
        const bool incr = sizeof(T) == 2 && p.incr && (1 << p.hwsh) <= 4 * RW &&
                          (int64_t)g->B * g->Hout * g->Wout * g->ldo < ((int64_t)1 << 31) &&
                          (int64_t)g->B * g->Hin * g->Win * g->ldx < ((int64_t)1 << 31);
        if (incr) return launch2<T, TN, TC, USE_TR, true, sizeof(T) == 2>(g, p, s);
        return launch2<T, TN, TC, USE_TR, true, false>(g, p, s);
    }
    return launch2<T, TN, TC, USE_TR, false, false>(g, p, s);
}

template <typename T, bool USE_TR>
int dispatch(const sv_geom* g, const wg_params& p, int tn, int tc, hipStream_t s) {
    if (tn == 4 && tc == 4) return launch<T, 4, 4, USE_TR>(g, p, s);
    if (tn == 4 && tc == 2) return launch<T, 4, 2, USE_TR>(g, p, s);
    if (tn == 4 && tc == 1) return launch<T, 4, 1, USE_TR>(g, p, s);
    if (tn == 2 && tc == 4) return launch<T, 2, 4, USE_TR>(g, p, s);
    if (tn == 2 && tc == 2) return launch<T, 2, 2, USE_TR>(g, p, s);
    if (tn == 2 && tc == 1) return launch<T, 2, 1, USE_TR>(g, p, s);
    if (tn == 1 && tc == 4) return launch<T, 1, 4, USE_TR>(g, p, s);
    if (tn == 1 && tc == 2) return launch<T, 1, 2, USE_TR>(g, p, s);
    return launch<T, 1, 1, USE_TR>(g, p, s);
}

}  // namespace

static_assert(sizeof(sv_wgrad_args) == 80 && sizeof(sv_igemm_args) == 224 && sizeof(sv_param_job) == 112 && sizeof(sv_bwd3x3_args) == 224, "ABI 8 struct layout (tests/test_abi_cpu.py)");

extern "C" int sv_wgrad_ex(const sv_geom* g, int dtype, const sv_wgrad_args* a, void* stream) {
    SV_REQUIRE(g && a, SV_E_ARG, "sv_wgrad_ex: null argument");
    SV_REQUIRE(a->block_budget == 0 || a->block_budget >= 8, SV_E_ARG, "sv_wgrad_ex: block_budget=%d", a->block_budget);
    SvBudgetScope budget_scope(a->block_budget);
    return sv_wgrad(g, dtype, a->x, a->pro_scale, a->pro_shift, a->pro_slope, a->dy, a->dw, a->splits, a->use_tr, a->ws,
                    a->ws_elems, a->groups, stream);
}

extern "C" int sv_wgrad(const sv_geom* g, int dtype, const void* x, const float* pro_scale,
                        const float* pro_shift, float pro_slope, const void* dy, float* dw, int splits,
                        int use_tr, float* ws, int64_t ws_elems, int groups, void* stream) {
    SV_REQUIRE(g && x && dy && dw, SV_E_ARG, "sv_wgrad: null argument");
    SV_REQUIRE(dtype == SV_F32 || dtype == SV_BF16, SV_E_ARG, "sv_wgrad: bad dtype %d", dtype);
    SV_REQUIRE(g->Cin % 16 == 0 && g->N % 16 == 0 && g->ldx % 8 == 0 && g->ldo % 8 == 0, SV_E_SHAPE,
               "sv_wgrad: Cin=%d N=%d must be multiples of 16", g->Cin, g->N);
    SV_REQUIRE(!pro_scale || pro_shift, SV_E_ARG, "sv_wgrad: prologue shift missing");
    SV_REQUIRE(groups >= 0 && groups <= SV_MAX_GROUPS, SV_E_ARG, "sv_wgrad: groups=%d (at most %d)", groups, SV_MAX_GROUPS);
    SV_REQUIRE(!pro_scale || (pro_slope >= 0.f && pro_slope <= 1.f), SV_E_ARG,
               "sv_wgrad: activation slope %g outside [0, 1]", (double)pro_slope);
    if (sv_deterministic() && sv_ngroups(groups) > 1) {
        // fixed summation order: the groups of a batched launch one after the other (stream order), each a launch whose
        // blocks add ONCE per weight (partial slabs + ordered reduction, or a single M range of the generic kernel)
        const int es = dtype == SV_BF16 ? 2 : 4;
        const int64_t xs = (int64_t)g->B * g->Hin * g->Win * g->ldx * es, ys = (int64_t)g->B * g->Hout * g->Wout * g->ldo * es;
        for (int grp = 0; grp < groups; ++grp) {
            const int rc = sv_wgrad(g, dtype, reinterpret_cast<const char*>(x) + grp * xs, pro_scale ? pro_scale + grp * g->Cin : nullptr,
                                    pro_shift ? pro_shift + grp * g->Cin : nullptr, pro_slope,
                                    reinterpret_cast<const char*>(dy) + grp * ys, dw, splits, use_tr, ws, ws_elems, 1, stream);
            if (rc != SV_OK) return rc;
        }
        return SV_OK;
    }
    // deterministic mode, generic kernels: one adder per weight -- the m ranges write slabs of their own in the caller's
    // workspace and a fixed-order pass adds them to dw (without a workspace: a single m range)
    const int det_cap = sv_deterministic() ? det_slabs_cap(g, ws, ws_elems) : 0;
    if (sv_deterministic()) splits = det_cap ? 0 : 1;
    if (dtype == SV_F32 || use_tr) {   // stride-1 3x3: LDS-halo kernels (wgrad3x3.hip) unless switched off
        int rc = 0;
        if (!sv_disabled(SV_K_WGRAD3X3) && sv_wgrad3x3_try(g, dtype, x, pro_scale, pro_shift, pro_slope, dy, dw, ws, ws_elems, sv_ngroups(groups), (hipStream_t)stream, &rc))
            return rc;
        // the thin 3x3 layers at 32x32 (16 input channels): the whole gradient in every block (thwgrad.hip)
        if (use_tr && sv_thwgrad_try(g, dtype, x, pro_scale, pro_shift, pro_slope, dy, dw, sv_ngroups(groups), (hipStream_t)stream, &rc))
            return rc;
        // svhn_VAE's thin 4x4 stride-2 layers (first convolution, last transposed convolution): the whole gradient in every block (k4wgrad.hip)
        if (use_tr && sv_k4wgrad_try(g, dtype, x, pro_scale, pro_shift, pro_slope, dy, dw, ws, ws_elems, sv_ngroups(groups), (hipStream_t)stream, &rc))
            return rc;
        // the stride-2 3x3 layer 32 -> 64: the whole gradient in every block, bands staged once (s2wgrad.hip)
        if (use_tr && sv_s2wgrad_try(g, dtype, x, pro_scale, pro_shift, pro_slope, dy, dw, sv_ngroups(groups), (hipStream_t)stream, &rc))
            return rc;
        // the other layers with a spatial extent: tap-fused LDS-halo weight gradient (hwgrad.hip)
        if (use_tr && sv_hwgrad_try(g, dtype, x, pro_scale, pro_shift, pro_slope, dy, dw, ws, ws_elems, sv_ngroups(groups), (hipStream_t)stream, &rc))
            return rc;
    }
    wg_params p;
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_slope = pro_slope;
    p.dy = dy; p.dw = dw;
    p.groups = sv_ngroups(groups);
    p.wsh = ilog2_exact(g->Wq);
    p.hwsh = ilog2_exact(g->Hq * g->Wq);
    if (p.wsh < 0 || p.hwsh < 0) p.wsh = p.hwsh = -1;
    p.ntap_total = 0;
    p.incr = !sv_disabled(SV_K_WGRAD_INCR);
    p.slab_stride = 0;
    for (int i = 0; i < g->nphase; ++i) p.ntap_total += g->phase[i].ntap;
    if (p.ntap_total == 0) return SV_OK;
    const int64_t M = (int64_t)g->B * g->Hq * g->Wq;
    if (dtype == SV_BF16 && use_tr && splits <= 0 && !sv_disabled(SV_K_WGRAD_WIDE) && g->N % 160 == 0 && g->Cin % 160 == 0 &&
        M * p.groups >= (int64_t)256 * sv_wide_min_blocks() && (!sv_deterministic() || det_cap))
        return launch_c<5, 5>(g, p, M, (hipStream_t)stream, det_cap ? ws : nullptr, det_cap);
    // tile: the widest of {64,32,16} that divides; fp32 is capped at 32 (LDS budget)
    auto pick = [&](int n) { int t = (n % 64 == 0) ? 4 : (n % 32 == 0 ? 2 : 1); if (n >= 64 && t == 1) t = (n % 32 == 0) ? 2 : 1; return t; };
    int tn = pick(g->N), tc = pick(g->Cin);
    if (g->N > 64 && g->N % 64 != 0) tn = 4;       // ragged last tile is masked in-kernel
    if (g->Cin > 64 && g->Cin % 64 != 0) tc = 4;
    if (dtype == SV_F32) { if (tn > 2) tn = 2; if (tc > 2) tc = 2; }
    const int tiles = ((g->N + 16 * tn - 1) / (16 * tn)) * p.ntap_total * ((g->Cin + 16 * tc - 1) / (16 * tc));
    if (splits <= 0) {
        // aim at ~4 blocks per CU, at least 512 rows per block, at most one split per 128 rows
        int64_t want = (1024 / p.groups + tiles - 1) / tiles;        // (the groups of a batched launch share the chip)
        int64_t maxs = (M + 511) / 512;
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        splits = (int)want;
    }
    if (det_cap && splits > det_cap) splits = det_cap;
    if (splits >= 8) splits = splits / 8 * 8;     // multiple of 8: XCD-affine block mapping
    int64_t m_per = (M + splits - 1) / splits;
    m_per = (m_per + 127) / 128 * 128;
    if (splits < 8) splits = (int)((M + m_per - 1) / m_per);
    p.splits = splits;
    p.m_per = (int)m_per;
    hipStream_t s = (hipStream_t)stream;
    const bool slabs = det_cap && splits > 1;
    if (slabs && !det_slabs_begin(g, p, ws, s)) return sv_check_launch("sv_wgrad: slab clear");
    int rc;
    if (dtype == SV_BF16) rc = use_tr ? dispatch<bf16, true>(g, p, tn, tc, s) : dispatch<bf16, false>(g, p, tn, tc, s);
    else rc = dispatch<float, false>(g, p, tn, tc, s);
    if (rc == SV_OK && slabs) {
        sv_slab_reduce(ws, splits, (int64_t)g->N * g->T_orig * g->Cin, dw, s);
        rc = sv_check_launch("sv_wgrad: slab reduce");
    }
    return rc;
}
