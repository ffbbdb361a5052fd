// Shared epilogue of the conv-like MFMA kernels (igemm.hip, conv3x3.hip).
//
// Accumulator layout (weights = MFMA A operand, activations = B operand): lane l holds, for output
// pixel (l & 15) of m-subtile ms, the 4 consecutive channels 16*nt + 4*(l >> 4) + r.
// Fuses: +bias, +residual, then either the per-channel (sum, sum of squares) of the NEXT BatchNorm or
// the activation backward + the two BatchNorm-backward reductions (sum g, sum g*xhat); 8/16-byte
// stores; wave shuffle -> LDS atomics -> one global atomic per channel per block into the accumulator replica
// (blockIdx % replicas).  Both atomic stages add fp32 partial sums into DOUBLES (sv_acc_t): the order in which the waves / blocks
// arrive varies from run to run, an fp64 sum of a few hundred fp32 values does not depend on it (exact additions).
#pragma once
#include "common.h"

// sum over the 16 lanes of a DPP row (= the 16 pixels of an accumulator sub-tile), result in every lane: four rotate-adds on
// the VALU (row_ror:8,4,2,1) instead of four ds_bpermute round trips per value -- the shuffles were ~10 % of the generic
// kernels on the wide layers (tools/ig_ablate.sh)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// Flush of the per-channel sums a block kept in registers (s1 / s2: lane l holds the partial sums of channels
// 16*i + 4*(l >> 4) + r over ITS pixels) into the accumulator replica -- shared by the persistent kernels (conv3x3p,
// conv3x3m, halo, halop).  Default: 16-lane shuffle reduction, LDS float atomics across the waves, one global atomic per
// channel and block into replica blockIdx.x % R.  Deterministic (SV_FLAG_DET): no LDS stage, every wave adds its own sums to
// replica (4 * blockIdx.x + wave) % R -- with R >= 4 * gridDim.x there is ONE adder per address and the consumers sum the
// replicas in index order, so the result does not depend on timing.
template <int NT>
__device__ __forceinline__ void flush_channel_sums(float (&s1)[NT][4], float (&s2)[NT][4], const bool (&nval)[NT],
                                                   double* ssum /* LDS [2][16*NT], zeroed */, double* gsum /* stats | bsums */,
                                                   int n0, int N, int replicas, int flags) {
    constexpr int BN = 16 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[i][r] = row16_sum(s1[i][r]);
            s2[i][r] = row16_sum(s2[i][r]);
        }
    if (flags & SV_FLAG_DET) {
        double* dst = gsum + (size_t)((blockIdx.x * 4 + wave) & (replicas - 1)) * 2 * N;
        if (fr == 0) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (nval[i]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        atomicAdd(dst + n0 + 16 * i + 4 * fq + r, (double)s1[i][r]);
                        atomicAdd(dst + N + n0 + 16 * i + 4 * fq + r, (double)s2[i][r]);
                    }
                }
        }
        return;
    }
    if (fr == 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (nval[i]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(&ssum[16 * i + 4 * fq + r], (double)s1[i][r]);
                    atomicAdd(&ssum[BN + 16 * i + 4 * fq + r], (double)s2[i][r]);
                }
            }
    }
    __syncthreads();
    double* dst = gsum + (size_t)(blockIdx.x & (replicas - 1)) * 2 * N;
    if (tid < 2 * BN) {
        const int which = tid / BN, nl = tid - which * BN;
        if (n0 + nl < N) atomicAdd(dst + which * N + n0 + nl, ssum[tid]);
    }
}

template <typename T, int NT, int MS = 2>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[NT][MS], const int64_t (&obase)[MS],
                                              const bool (&oval)[MS], int n0, int N, const sv_igemm_args& a,
                                              double* ssum /* LDS [2][16*NT], zeroed, visible */,
                                              float* cst = nullptr /* LDS [5][16*NT] scratch or NULL */) {
    typedef typename V4<T>::type Q;
    constexpr int BN = 16 * NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    T* __restrict__ O = reinterpret_cast<T*>(a.out);
    const T* __restrict__ R = reinterpret_cast<const T*>(a.residual);
    const T* __restrict__ EX = reinterpret_cast<const T*>(a.ex);
    const bool want_sums = (a.stats != nullptr) || (EX != nullptr);
    // per-channel constants of the block's channels: one cooperative copy into LDS (when the caller has scratch) instead
    // of a dependent global round trip per 16-channel group
    const bool lds_c = cst != nullptr && (a.bias || EX);
    if (lds_c) {
        for (int c = tid; c < BN; c += 256) {
            const int n = min(n0 + c, N - 1);
            cst[c] = a.bias ? a.bias[n] : 0.f;
            if (EX) {
                cst[BN + c] = a.ex_scale[n];
                cst[2 * BN + c] = a.ex_shift[n];
                cst[3 * BN + c] = a.ex_mean[n];
                cst[4 * BN + c] = a.ex_rstd[n];
            }
        }
        __syncthreads();
    }
    // The residual / raw-tensor operands of all MS rows of a channel group are requested together, one group AHEAD of the
    // arithmetic (every row address is inside the tensor: invalid rows were clamped to row 0 by the caller).  Issued one
    // by one behind an `if (valid)` each of them was an exposed round trip -- 40 per block on the 256 x 160 tiles, most
    // of the data-gradient kernels' time (WRN-28-10 stride-2 data gradient 711 -> 567 us with the batching alone).
    Q rr[2][MS], xe[2][MS];
    auto fetch = [&](int i, int buf) __attribute__((always_inline)) {
        const int n = min(n0 + 16 * i + 4 * fq, N - 4);
        if (R) {
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) rr[buf][ms] = *reinterpret_cast<const Q*>(R + obase[ms] + n);
        }
        if (EX) {
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) xe[buf][ms] = *reinterpret_cast<const Q*>(EX + obase[ms] + n);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int nl = 16 * i + 4 * fq;      // local channel of this lane's 4-vector
        const int n = n0 + nl;
        const bool nval = n < N;
        if (i + 1 < NT) fetch(i + 1, (i + 1) & 1);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (nval) {
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            f32x4 esc, esh, emu, ers;
            if (lds_c) {
                bias = *reinterpret_cast<const f32x4*>(cst + nl);
                if (EX) {
                    esc = *reinterpret_cast<const f32x4*>(cst + BN + nl);
                    esh = *reinterpret_cast<const f32x4*>(cst + 2 * BN + nl);
                    emu = *reinterpret_cast<const f32x4*>(cst + 3 * BN + nl);
                    ers = *reinterpret_cast<const f32x4*>(cst + 4 * BN + nl);
                }
            } else {
                if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
                if (EX) {
                    esc = *reinterpret_cast<const f32x4*>(a.ex_scale + n);
                    esh = *reinterpret_cast<const f32x4*>(a.ex_shift + n);
                    emu = *reinterpret_cast<const f32x4*>(a.ex_mean + n);
                    ers = *reinterpret_cast<const f32x4*>(a.ex_rstd + n);
                }
            }
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                f32x4 vv = acc[i][ms];
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[r] += bias[r];
                if (R) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] += to_f(rr[i & 1][ms][r]);
                }
                const float live = oval[ms] ? 1.f : 0.f;             // rows beyond M contribute nothing to the sums
                if (EX) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xf = to_f(xe[i & 1][ms][r]);
                        const float u = xf * esc[r] + esh[r];
                        const float gv = vv[r] * act_grad(u, a.ex_slope) * live;
                        vv[r] = gv;
                        s1[r] += gv;
                        s2[r] += gv * ((xf - emu[r]) * ers[r]);
                    }
                } else if (a.stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = vv[r] * live;
                        s1[r] += t;
                        s2[r] += t * t;
                    }
                }
                Q o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (T)vv[r];
                if (oval[ms]) *reinterpret_cast<Q*>(O + obase[ms] + n) = o;
            }
        }
        if (want_sums) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = row16_sum(s1[r]);
                s2[r] = row16_sum(s2[r]);
            }
            if (fr == 0 && nval) {
                if (a.flags & SV_FLAG_DET) {      // one adder per address: this wave's replica (see flush_channel_sums)
                    double* dw_ = (EX ? a.bsums : a.stats) + (size_t)((blockIdx.x * 4 + (tid >> 6)) & (a.replicas - 1)) * 2 * N;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        atomicAdd(dw_ + n + r, (double)s1[r]);
                        atomicAdd(dw_ + N + n + r, (double)s2[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        atomicAdd(&ssum[nl + r], (double)s1[r]);
                        atomicAdd(&ssum[BN + nl + r], (double)s2[r]);
                    }
                }
            }
        }
    }
    if (want_sums && !(a.flags & SV_FLAG_DET)) {
        __syncthreads();
        // replica chosen by block index: keeps the number of adders per address low (contended float
        // atomics on a handful of addresses were 2/3 of the kernel time before)
        double* dst = (EX ? a.bsums : a.stats) + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * N;
        for (int i = tid; i < 2 * BN; i += 256) {
            const int which = i / BN, nl = i - which * BN;
            if (n0 + nl < N) atomicAdd(dst + which * N + n0 + nl, ssum[i]);
        }
    }
}

// sv_bn_bwd_affine inside a 512-thread block (sv_bwd3x3_args::fold_*, ABI 8): scale_g / scale_x / shift [C] of a BatchNorm's backward from
// its raw sums [R][2C] into `out` (three rows of C floats; LDS), sv_bn_bwd_affine's arithmetic; `scratch` = 2 x 512 doubles of LDS.
// Every block of a launch derives the same values (fixed summation order); `add` = this block also adds dgamma / dbeta.
template <int C>
__device__ __forceinline__ void sv_bn_bwd_affine_block512(const double* __restrict__ bsums, int R, float inv_count, const float* gamma,
                                                          const float* mean, const float* rstd, float* dgamma, float* dbeta, bool add,
                                                          double* scratch, float* out) {
    static_assert(512 % C == 0, "");
    constexpr int PARTS = 512 / C;
    const int tid = threadIdx.x, c = tid % C, pt = tid / C;
    double s1 = 0.0, s2 = 0.0;
    for (int r = pt; r < R; r += PARTS) {
        s1 += bsums[(size_t)r * 2 * C + c];
        s2 += bsums[(size_t)r * 2 * C + C + c];
    }
    scratch[tid] = s1;
    scratch[512 + tid] = s2;
    __syncthreads();
    if (tid < C) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) { t1 += scratch[q * C + tid]; t2 += scratch[512 + q * C + tid]; }
        const float rs = rstd[tid], A = gamma[tid] * rs, m1 = (float)(t1 * (double)inv_count), m2 = (float)(t2 * (double)inv_count);
        const float bx = -A * m2 * rs;
        out[tid] = A;
        out[C + tid] = bx;
        out[2 * C + tid] = -A * m1 - bx * mean[tid];
        if (add) {
            if (dbeta) atomicAdd(dbeta + tid, (float)t1);
            if (dgamma) atomicAdd(dgamma + tid, (float)t2);
        }
    }
    __syncthreads();
}
