// Small fused kernels of the SHOT-VAE step (everything that is not a conv-like GEMM).  gfx950.
#include "common.h"

namespace {

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int id = SV_F32; };
template <> struct DT<bf16> { static constexpr int id = SV_BF16; };

// block-wide sum of `v` (blockDim multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// ---------------------------------------------------------------------------------------- deterministic mode
// Two-pass reductions: the blocks of the first pass own a slot each in the library's scratch ring (sv_det_scratch), the second
// pass adds the slots in index order -- one add per output per call, whatever order the blocks ran in.
//   out[o * n + i] += sum_p part[(o * P + p) * n + i],  p = 0 .. P-1 in order
template <typename O>
__global__ __launch_bounds__(256) void det_collect_kernel(const float* part, int P, int n, O* out) {
    const int i = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
    if (i >= n) return;
    const float* src = part + (int64_t)o * P * n + i;
    O t = 0;
#pragma unroll 8
    for (int p = 0; p < P; ++p) t += (O)src[(int64_t)p * n];
    atomicAdd(out + (int64_t)o * n + i, t);
}
// Accumulator replicas [R][n] -> [ceil(R / 256)][n]: deterministic mode sizes the BatchNorm accumulators at a replica per producer
// wave (thousands), and every block of sv_bn_bwd_apply summing all of them was 68 us per launch.  Thread (row group of 8, 4
// columns) adds its rows in order, the 32 row groups meet in LDS in order.
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void replica_fold_kernel(const double* src, int R, int n, double* dst) {
    __shared__ f64x4 part[32][8];
    src += (int64_t)blockIdx.z * R * n;                  // batched launch: group z, [G][R][n] -> [G][gridDim.y][n]
    dst += (int64_t)blockIdx.z * gridDim.y * n;
    const int v = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int col = blockIdx.x * 32 + 4 * v;
    const int r0 = blockIdx.y * 256 + rg * 8;
    f64x4 t = {0.0, 0.0, 0.0, 0.0};
    if (col < n) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (r0 + k < R) {
                const f64x4 q = *reinterpret_cast<const f64x4*>(src + (int64_t)(r0 + k) * n + col);
                t[0] += q[0]; t[1] += q[1]; t[2] += q[2]; t[3] += q[3];
            }
    }
    part[rg][v] = t;
    __syncthreads();
    if (rg == 0 && col < n) {
        f64x4 a = part[0][v];
        for (int q = 1; q < 32; ++q) { const f64x4 b = part[q][v]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
        *reinterpret_cast<f64x4*>(dst + (int64_t)blockIdx.y * n + col) = a;
    }
}

// ---------------------------------------------------------------------------------------- BatchNorm
// 256 threads = 8 channels x 32 lanes; the lanes of a channel split the accumulator replicas (the loads
// are independent and in flight together instead of one dependent chain per channel)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* stats, int R, int C, float count,
                                                          const float* gamma, const float* beta, float eps,
                                                          float momentum, float* rm, float* rv, float* scale,
                                                          float* shift, float* mean, float* rstd) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int r0 = threadIdx.x & 31;
    {   // batched launch: blockIdx.y = group, its [R][2C] statistics and [C] outputs follow the previous group's
        const size_t grp = blockIdx.y;
        stats += grp * R * 2 * C;
        scale += grp * C; shift += grp * C; mean += grp * C; rstd += grp * C;
    }
    double s1 = 0.0, s2 = 0.0;                        // (doubles: sv_acc_t -- exact sums of the producers' fp32 partial sums)
    const int cc = c < C ? c : 0;
    const float ga = gamma[cc], be = beta[cc];        // requested with the statistics (one round trip, not two)
    if (c < C)
        for (int r = r0; r < R; r += 32) {
            s1 += stats[(size_t)r * 2 * C + c];
            s2 += stats[(size_t)r * 2 * C + C + c];
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (c >= C || r0 != 0) return;
    float mu, var, rs;
    sv_bn_moments(s1, s2, count, eps, mu, var, rs);
    const float sc = ga * rs;
    scale[c] = sc;
    shift[c] = be - mu * sc;
    mean[c] = mu;
    rstd[c] = rs;
    if (rm) {
        rm[c] = (1.f - momentum) * rm[c] + momentum * mu;
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        rv[c] = (1.f - momentum) * rv[c] + momentum * unb;
    }
}

// Deferred running-statistics update of ALL BatchNorms of one forward (block = one BN).  Used when the
// forwards of a step are issued on several streams: every forward keeps its batch (mean, rstd) and the four
// momentum updates are applied afterwards in the reference's order (1)(2)(3)(4), race free.
// table[bn] = {offset of the BN's (scale,shift,mean,rstd) block in bnbuf, running_mean offset, running_var offset, C}
__global__ __launch_bounds__(256) void bn_running_update_kernel(const int32_t* table, const float* counts,
                                                                const float* bnbuf, float* bufs, float eps,
                                                                float momentum, int align, int groups, int order) {
    const int bn = blockIdx.x;
    const int boff = table[4 * bn], rmo = table[4 * bn + 1], rvo = table[4 * bn + 2], C = table[4 * bn + 3];
    const int ca = (groups * C + align - 1) / align * align;       // each of the four arrays is [groups][C], padded
    const float n = counts[bn];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float rm = bufs[rmo + c], rv = bufs[rvo + c];
        for (int k = 0; k < groups; ++k) {                           // the updates in the reference's forward order
            const int gi = (order >> (4 * k)) & 15;                    // (nibble k = the group of the k-th forward)
            const float mu = bnbuf[boff + 2 * ca + gi * C + c], rs = bnbuf[boff + 3 * ca + gi * C + c];
            float var = 1.f / (rs * rs) - eps;
            var = var > 0.f ? var : 0.f;
            const float unb = n > 1.f ? var * n / (n - 1.f) : var;
            rm = (1.f - momentum) * rm + momentum * mu;
            rv = (1.f - momentum) * rv + momentum * unb;
        }
        bufs[rmo + c] = rm;
        bufs[rvo + c] = rv;
    }
}

__global__ void bn_eval_affine_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

struct bnb_params {
    int64_t M;
    int C, ld, nbranch;
    const void* x;
    const float* mean;
    const float* rstd;
    float inv_count;
    sv_bn_branch br[2];
    const void* residual;
    void* dx;
};

struct bnb_params_g {
    bnb_params g[SV_MAX_GROUPS];
    int det_groups;       // deterministic mode: > 0 = block (0, 0) alone adds dgamma / dbeta, the groups in index order
};
// tensors [G][M][ld], mean / rstd [G][C], bsums [G][R][2C]
static bnb_params_g bnb_expand(const bnb_params& p, int groups, int es) {
    bnb_params_g A;
    for (int64_t grp = 0; grp < SV_MAX_GROUPS; ++grp) {
        bnb_params r = p;
        if (grp > 0 && grp < groups) {
            const int64_t ts = p.M * p.ld * es;
            r.x = reinterpret_cast<const char*>(p.x) + grp * ts;
            r.dx = reinterpret_cast<char*>(p.dx) + grp * ts;
            if (p.residual) r.residual = reinterpret_cast<const char*>(p.residual) + grp * ts;
            r.mean = p.mean + grp * p.C;
            r.rstd = p.rstd + grp * p.C;
            for (int k = 0; k < p.nbranch; ++k) {
                r.br[k].g = reinterpret_cast<const char*>(p.br[k].g) + grp * (p.br[k].sparse < 0 ? ts / 4 : ts);
                r.br[k].bsums = p.br[k].bsums + grp * p.br[k].replicas * 2 * p.C;
            }
        }
        A.g[grp] = r;
    }
    A.det_groups = 0;
    return A;
}

#ifndef SV_BNB_UNROLL2
#define SV_BNB_UNROLL2 0        // 1: two vectors per trip of sv_bn_bwd_apply's loop (twice the bytes in flight; measured no gain: 43.1 vs 43.4 us, tools/probes/small_ab.sh)
#endif
// REG: (threads of the grid) % (C/8) == 0, so a thread always meets the same 8 channels and keeps their
// coefficients [gamma*rstd, mean(g), mean(g*xhat)] (+ mean, rstd) in registers; otherwise they sit in LDS.
template <typename T, bool REG>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bnb_params_g PG) {
    typedef typename V8<T>::type V;
    const bnb_params& p = PG.g[blockIdx.y];        // batched launch: blockIdx.y = group (the host expanded the pointers)
    extern __shared__ __attribute__((aligned(16))) float coef[];
    const int cv = p.C / 8;                       // vectors per row
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    // per-channel coefficients once per block into LDS.  The replica sums are spread over all 256 threads
    // (G = 256/C thread groups per channel, each summing every G-th replica, all loads independent): a
    // serial 32-replica loop per channel put a ~15 us latency floor under every launch of this kernel.
    {
        float* cmean = coef;
        float* crstd = coef + p.C;
        double* part = reinterpret_cast<double*>(coef + (2 + 3 * p.nbranch) * p.C);         // [nbranch][2][G][Cg]
        const int nthr = blockDim.x;                 // 256, or the largest multiple of C/8 below it (see the launcher)
        const int Cg = p.C < nthr ? p.C : nthr;
        const int G = nthr / Cg;
        const int grp = threadIdx.x / Cg, cl = threadIdx.x - grp * Cg;
        for (int cb0 = 0; cb0 < p.C; cb0 += Cg) {
            const int c = cb0 + cl;
            if (grp < G && c < p.C)
                for (int k = 0; k < p.nbranch; ++k) {
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
                    for (int r = grp; r < p.br[k].replicas; r += G) {
                        s1 += p.br[k].bsums[(size_t)r * 2 * p.C + c];
                        s2 += p.br[k].bsums[(size_t)r * 2 * p.C + p.C + c];
                    }
                    part[((k * 2 + 0) * G + grp) * Cg + cl] = s1;
                    part[((k * 2 + 1) * G + grp) * Cg + cl] = s2;
                }
            __syncthreads();
            if (grp == 0 && c < p.C) {
                const float rs = p.rstd[c];
                cmean[c] = p.mean[c];
                crstd[c] = rs;
                for (int k = 0; k < p.nbranch; ++k) {
                    double s1 = 0.0, s2 = 0.0;
                    for (int q = 0; q < G; ++q) {
                        s1 += part[((k * 2 + 0) * G + q) * Cg + cl];
                        s2 += part[((k * 2 + 1) * G + q) * Cg + cl];
                    }
                    float* cb = coef + (2 + 3 * k) * p.C;
                    cb[c] = p.br[k].gamma[c] * rs;
                    cb[p.C + c] = (float)(s1 * (double)p.inv_count);
                    cb[2 * p.C + c] = (float)(s2 * (double)p.inv_count);
                    if (blockIdx.x == 0 && !PG.det_groups) {      // atomics: another stream's backward may add to the same slots
                        if (p.br[k].dbeta) atomicAdd(p.br[k].dbeta + c, (float)s1);
                        if (p.br[k].dgamma) atomicAdd(p.br[k].dgamma + c, (float)s2);
                    } else if (blockIdx.x == 0 && blockIdx.y == 0) {      // one adder per launch, groups and replicas in index order
                        double a1 = 0.0, a2 = 0.0;
                        for (int q = 0; q < PG.det_groups; ++q) {
                            const sv_bn_branch& bq = PG.g[q].br[k];
                            double t1 = 0.0, t2 = 0.0;
                            for (int r = 0; r < bq.replicas; ++r) {
                                t1 += bq.bsums[(size_t)r * 2 * p.C + c];
                                t2 += bq.bsums[(size_t)r * 2 * p.C + p.C + c];
                            }
                            a1 += t1;
                            a2 += t2;
                        }
                        if (p.br[k].dbeta) atomicAdd(p.br[k].dbeta + c, (float)a1);
                        if (p.br[k].dgamma) atomicAdd(p.br[k].dgamma + c, (float)a2);
                    }
                }
            }
            __syncthreads();
        }
    }
    // ... and, when a thread always meets the same 8 channels, from there into registers
    float cm[8], cr[8], ca[2][8], c1[2][8], c2[2][8];
    if (REG) {
        const int c0 = (int)(gtid % cv) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cm[j] = coef[c0 + j];
            cr[j] = coef[p.C + c0 + j];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                ca[k][j] = c1[k][j] = c2[k][j] = 0.f;
                if (k < p.nbranch) {
                    const float* cb = coef + (2 + 3 * k) * p.C;
                    ca[k][j] = cb[c0 + j];
                    c1[k][j] = cb[p.C + c0 + j];
                    c2[k][j] = cb[2 * p.C + c0 + j];
                }
            }
        }
    }
    const int64_t total = p.M * cv;
    const T* X = reinterpret_cast<const T*>(p.x);
    const T* R = reinterpret_cast<const T*>(p.residual);
    T* DX = reinterpret_cast<T*>(p.dx);
    // no 64-bit division in the loop: a thread's channel group is fixed (REG) and its row advances by a constant
    int64_t m = gtid / cv;
    int c = (int)(gtid - m * cv) * 8;
    const int64_t m_step = gsz / cv;
    const int c_step = (int)(gsz - m_step * cv) * 8;          // 0 in the REG case
    // the operands of one vector (loads only) ...
    auto fetch = [&](int64_t mm, int cc, V& xv, V (&gv)[2], V& rv) __attribute__((always_inline)) {
        const int64_t off = mm * p.ld + cc;
        // streamed once: non-temporal loads (5.0 -> 5.16 TB/s; a non-temporal store of dx costs its consumer as much)
        xv = __builtin_nontemporal_load(reinterpret_cast<const V*>(X + off));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= p.nbranch) break;
            // a sparse branch (stride-2 data gradient written with sv_igemm_args::sparse_out) exists at even (row, column) only
            // (sparse < 0: the same positions stored COMPACTLY, [M / 4][ld]: the output of a dense 1x1 product over the stride-2 grid)
            const int spr = p.br[k].sparse, sp = spr < 0 ? -spr : spr;
            if (sp > 0 && ((((int)(mm >> (sp - 1))) | (int)mm) & 1)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[k][j] = (T)0.f;
            } else if (spr < 0) {
                const int64_t mc = ((mm >> (sp - 1)) >> 1 << (sp - 2)) + ((mm & (((int64_t)1 << (sp - 1)) - 1)) >> 1);
                gv[k] = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const T*>(p.br[k].g) + mc * p.ld + cc));
            } else {
                gv[k] = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const T*>(p.br[k].g) + off));
            }
        }
        if (R) rv = __builtin_nontemporal_load(reinterpret_cast<const V*>(R + off));
    };
    // ... and its arithmetic + store
    auto apply = [&](int64_t mm, int cc, const V& xv, const V (&gv)[2], const V& rv) __attribute__((always_inline)) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float mu, rs;
            if (REG) { mu = cm[j]; rs = cr[j]; } else { mu = coef[cc + j]; rs = coef[p.C + cc + j]; }
            const float xh = (to_f(xv[j]) - mu) * rs;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < p.nbranch) {
                    float a_, m1, m2;
                    if (REG) { a_ = ca[k][j]; m1 = c1[k][j]; m2 = c2[k][j]; }
                    else {
                        const float* cb = coef + (2 + 3 * k) * p.C;
                        a_ = cb[cc + j]; m1 = cb[p.C + cc + j]; m2 = cb[2 * p.C + cc + j];
                    }
                    acc += a_ * (to_f(gv[k][j]) - m1 - xh * m2);
                }
            }
            if (R) acc += to_f(rv[j]);
            o[j] = acc;
        }
        V ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (T)o[j];
        *reinterpret_cast<V*>(DX + mm * p.ld + cc) = ov;
    };
    int64_t i = gtid;
    if (REG && SV_BNB_UNROLL2) {
        // two vectors per trip, the second one's loads issued before the first one's arithmetic: twice the bytes in flight per
        // thread (the kernel runs one resident wave of blocks, 5 waves per SIMD: ~60 KB per CU in flight with one vector per trip,
        // at the edge of what the HBM latency needs)
        for (; i + gsz < total; i += 2 * gsz) {
            V xa, ga[2], ra, xb, gb[2], rb_;
            fetch(m, c, xa, ga, ra);
            fetch(m + m_step, c, xb, gb, rb_);
            apply(m, c, xa, ga, ra);
            apply(m + m_step, c, xb, gb, rb_);
            m += 2 * m_step;
        }
    }
    for (; i < total; i += gsz) {
        V xv, gv[2], rv;
        fetch(m, c, xv, gv, rv);
        apply(m, c, xv, gv, rv);
        m += m_step;
        c += c_step;
        if (c >= p.C) { c -= p.C; ++m; }
    }
}

// Affine coefficients of a BatchNorm backward (sv_bn_bwd_affine) for the two-tensor dy operand of the fused backward that consumes
// it (sv_bwd3x3_args::dy2): block = (64 channels, group); thread (channel tid % 64, part tid / 64) sums its share of the replicas,
// the four parts meet in LDS in index order.
__global__ __launch_bounds__(256) void bn_bwd_affine_kernel(const double* bsums, int R, int C, float inv_count, const float* gamma,
                                                            const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                                            float* scale_g, float* scale_x, float* shift) {
    __shared__ double part[2][256];
    const int grp = blockIdx.y, tid = threadIdx.x, c = blockIdx.x * 64 + (tid & 63), pt = tid >> 6;
    const double* b = bsums + (size_t)grp * R * 2 * C;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = pt; r < R; r += 4) {
            s1 += b[(size_t)r * 2 * C + c];
            s2 += b[(size_t)r * 2 * C + C + c];
        }
    part[0][tid] = s1;
    part[1][tid] = s2;
    __syncthreads();
    if (tid < 64 && c < C) {
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < 4; ++q) { t1 += part[0][q * 64 + tid]; t2 += part[1][q * 64 + tid]; }
        const size_t o = (size_t)grp * C + c;
        const float rs = rstd[o], A = gamma[c] * rs, m1 = (float)(t1 * (double)inv_count), m2 = (float)(t2 * (double)inv_count);
        const float bx = -A * m2 * rs;
        scale_g[o] = A;
        scale_x[o] = bx;
        shift[o] = -A * m1 - bx * mean[o];
        if (dbeta) atomicAdd(dbeta + c, (float)t1);
        if (dgamma) atomicAdd(dgamma + c, (float)t2);
    }
}

// out = act(x * scale[c] + shift[c]) (BatchNorm-apply + LeakyReLU / ReLU as a pass of its own): for the weight-heavy decoder
// layers the consumer GEMM re-applied this prologue once per output-channel tile and was VALU-bound by it
// (9 VALU instructions per MFMA); materialised once, the GEMM takes the prologue-free LDS-DMA loader.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* x, const float* scale, const float* shift, float slope, int64_t M,
                                                     int C, T* out) {
    typedef typename V8<T>::type V;
    const int grp = blockIdx.y;
    x += (int64_t)grp * M * C;
    out += (int64_t)grp * M * C;
    scale += (int64_t)grp * C;
    shift += (int64_t)grp * C;
    const int64_t nv = M * C / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 8) % C);
        const V v = *reinterpret_cast<const V*>(x + i * 8);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(shift + c), t1 = *reinterpret_cast<const f32x4*>(shift + c + 4);
        *reinterpret_cast<V*>(out + i * 8) = bn_act8(v, s0, s1, t0, t1, slope);
    }
}

template <typename T>
__global__ void colsum_kernel(const T* y, int64_t M, int N, int ld, float* out, int ostride = 0) {
    // thread (n, slice): blockDim = (N<=64 ? N : 64, 256/that)
    out += (int64_t)blockIdx.x * ostride;             // (deterministic mode: a slot per row range)
    const int n = blockIdx.y * blockDim.x + threadIdx.x;
    const int64_t rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    float s = 0.f;
    if (n < N)
        for (int64_t m = m0 + threadIdx.y; m < m1; m += blockDim.y) s += to_f(y[m * ld + n]);
    __shared__ float red[256];
    red[threadIdx.y * blockDim.x + threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && n < N) {
        float t = 0.f;
        for (int k = 0; k < (int)blockDim.y; ++k) t += red[k * blockDim.x + threadIdx.x];
        atomicAdd(out + n, t);
    }
}

// 16-byte loads: thread (row slot, 8-channel group); per-block LDS reduction, one global atomic per channel per block
// (the element-wise kernel above read 2 bytes per lane: 67 us for the 67 MB bias gradient of the last ConvTranspose)
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* y, int64_t M, int N, int ld, float* out) {
    typedef typename V8<T>::type V;
    extern __shared__ float csum[];                      // [N]
    const int cv = N / 8;
    for (int i = threadIdx.x; i < N; i += 256) csum[i] = 0.f;
    __syncthreads();
    const int rpp = 256 / cv;                            // rows per pass of the block
    const int v = threadIdx.x % cv, r = threadIdx.x / cv;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (r < rpp)
        for (int64_t m = (int64_t)blockIdx.x * rpp + r; m < M; m += (int64_t)gridDim.x * rpp) {
            const V q = __builtin_nontemporal_load(reinterpret_cast<const V*>(y + m * ld + 8 * v));
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += to_f(q[j]);
        }
    if ((cv & (cv - 1)) == 0 && cv <= 32) {              // lanes l, l + cv, ... of a wave share a channel group: butterfly first
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = cv; o < 64; o <<= 1) s[j] += __shfl_xor(s[j], o);
        if ((threadIdx.x & 63) < cv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(&csum[8 * v + j], s[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&csum[8 * v + j], s[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) atomicAdd(out + i, csum[i]);
}

// ---------------------------------------------------------------------------------------- pool
template <typename T>
__global__ void pool_fwd_kernel(const T* x, const float* scale, const float* shift, float slope, int B,
                                int HW, int C, int ld, float* feat, int Bg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    const int gc = (b / Bg) * C + c;                  // batched: image b belongs to group b / Bg (coefficients [G][C])
    const float sc = scale[gc], sh = shift[gc];
    const T* px = x + (int64_t)b * HW * ld + c;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += act_fwd(to_f(px[(int64_t)p * ld]) * sc + sh, slope);
    feat[idx] = s / (float)HW;
}

template <typename T>
__global__ void pool_bwd_kernel(const T* x, const float* scale, const float* shift, float slope,
                                const float* mean, const float* rstd, const float* dfeat, int B, int HW,
                                int C, int ld, T* g, double* bsums, int Bg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    const int grp = b / Bg, gc = grp * C + c;
    bsums += (size_t)grp * 2 * C;
    const float sc = scale[gc], sh = shift[gc], mu = mean[gc], rs = rstd[gc];
    const float d = dfeat[idx] / (float)HW;
    const T* px = x + (int64_t)b * HW * ld + c;
    T* pg = g + (int64_t)b * HW * ld + c;
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < HW; ++p) {
        const float xf = to_f(px[(int64_t)p * ld]);
        const float gv = d * act_grad(xf * sc + sh, slope);
        pg[(int64_t)p * ld] = (T)gv;
        s1 += gv;
        s2 += gv * ((xf - mu) * rs);
    }
    atomicAdd(bsums + c, (double)s1);
    atomicAdd(bsums + C + c, (double)s2);
}

// 16-byte versions of the two pooling kernels: thread (image, pixel part, 8-channel group); the element-wise kernels read
// 2 bytes per lane (pool_bwd 54 us for 67 MB at WRN-28-2, 86 us at WRN-28-10) and put B-way contention on the sums.
constexpr int POOL_PARTS = 4;
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd8_kernel(const T* x, const float* scale, const float* shift, float slope, int B,
                                                        int HW, int C, int ld, float* feat, int Bg, int P) {
    typedef typename V8<T>::type V;
    extern __shared__ float psum[];                   // [ipb][P][C]
    const int cv = C / 8, per = cv * P, ipb = 256 / per > 0 ? 256 / per : 1;
    const int img = threadIdx.x / per, rem = threadIdx.x - img * per, part = rem / cv, v = rem - part * cv;
    const int b = blockIdx.x * ipb + img;
    const bool on = img < ipb && b < B;
    if (on) {
        const int gc = (b / Bg) * C + 8 * v;
        float sc[8], sh[8], sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = scale[gc + j], sh[j] = shift[gc + j], sum[j] = 0.f;
        const T* px = x + (int64_t)b * HW * ld + 8 * v;
        for (int p = part; p < HW; p += P) {
            const V q = *reinterpret_cast<const V*>(px + (int64_t)p * ld);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] += act_fwd(to_f(q[j]) * sc[j] + sh[j], slope);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) psum[(img * P + part) * C + 8 * v + j] = sum[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ipb * C; i += 256) {
        const int im = i / C, c = i - im * C;
        if (blockIdx.x * ipb + im >= B) continue;
        float t = 0.f;
        for (int q = 0; q < P; ++q) t += psum[(im * P + q) * C + c];
        feat[(int64_t)(blockIdx.x * ipb + im) * C + c] = t / (float)HW;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pool_bwd8_kernel(const T* x, const float* scale, const float* shift, float slope,
                                                        const float* mean, const float* rstd, const float* dfeat, int B, int HW,
                                                        int C, int ld, T* g, double* bsums, int Bg, int P) {
    typedef typename V8<T>::type V;
    extern __shared__ double psum_d[];                // [2][C] (doubles: the threads of the block meet here in any order)
    double* psum = psum_d;
    for (int i = threadIdx.x; i < 2 * C; i += 256) psum[i] = 0.0;
    __syncthreads();
    const int cv = C / 8, per = cv * P, ipb = 256 / per > 0 ? 256 / per : 1;     // (ipb images of one group: Bg % ipb == 0)
    const int img = threadIdx.x / per, rem = threadIdx.x - img * per, part = rem / cv, v = rem - part * cv;
    const int b = blockIdx.x * ipb + img;
    const int grp = (blockIdx.x * ipb) / Bg;
    if (img < ipb && b < B) {
        const int gc = grp * C + 8 * v;
        float sc[8], sh[8], mu[8], rs[8], d[8], s1[8], s2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = scale[gc + j], sh[j] = shift[gc + j], mu[j] = mean[gc + j], rs[j] = rstd[gc + j];
            d[j] = dfeat[(int64_t)b * C + 8 * v + j] / (float)HW;
            s1[j] = s2[j] = 0.f;
        }
        const T* px = x + (int64_t)b * HW * ld + 8 * v;
        T* pg = g + (int64_t)b * HW * ld + 8 * v;
        for (int p = part; p < HW; p += P) {
            const V q = *reinterpret_cast<const V*>(px + (int64_t)p * ld);
            V o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xf = to_f(q[j]);
                const float gv = d[j] * act_grad(xf * sc[j] + sh[j], slope);
                o[j] = (T)gv;
                s1[j] += gv;
                s2[j] += gv * ((xf - mu[j]) * rs[j]);
            }
            *reinterpret_cast<V*>(pg + (int64_t)p * ld) = o;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            atomicAdd(&psum[8 * v + j], (double)s1[j]);
            atomicAdd(&psum[C + 8 * v + j], (double)s2[j]);
        }
    }
    __syncthreads();
    double* dst = bsums + (size_t)grp * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(dst + i, psum[i]);
}

// SV_OPT_DETERMINISTIC: gridDim.y blocks per group share the group's images; every thread sums its (image lane, 8-channel
// group) in a fixed order, the image lanes meet in LDS slots and are added in index order, one add per channel leaves the block
// into the block's own slot [group][blockIdx.y][2C] (the launcher collects the slots in order).
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_det_kernel(const T* x, const float* scale, const float* shift, float slope,
                                                           const float* mean, const float* rstd, const float* dfeat, int Bg, int HW,
                                                           int C, int ld, T* g, float* bsums) {
    typedef typename V8<T>::type V;
    extern __shared__ float psum[];                   // [lanes][2C]
    const int cv = C / 8, lanes = 256 / cv;
    const int v = threadIdx.x % cv, im = threadIdx.x / cv;
    const int grp = blockIdx.x;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (im < lanes) {
        const int gc = grp * C + 8 * v;
        float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = scale[gc + j], sh[j] = shift[gc + j], mu[j] = mean[gc + j], rs[j] = rstd[gc + j];
        for (int bi = im + lanes * blockIdx.y; bi < Bg; bi += lanes * gridDim.y) {
            const int64_t b = (int64_t)grp * Bg + bi;
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = dfeat[b * C + 8 * v + j] / (float)HW;
            const T* px = x + b * HW * ld + 8 * v;
            T* pg = g + b * HW * ld + 8 * v;
            for (int p = 0; p < HW; ++p) {
                const V q = *reinterpret_cast<const V*>(px + (int64_t)p * ld);
                V o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = to_f(q[j]);
                    const float gv = d[j] * act_grad(xf * sc[j] + sh[j], slope);
                    o[j] = (T)gv;
                    s1[j] += gv;
                    s2[j] += gv * ((xf - mu[j]) * rs[j]);
                }
                *reinterpret_cast<V*>(pg + (int64_t)p * ld) = o;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            psum[im * 2 * C + 8 * v + j] = s1[j];
            psum[im * 2 * C + C + 8 * v + j] = s2[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float t = 0.f;
        for (int q = 0; q < lanes; ++q) t += psum[q * 2 * C + i];
        atomicAdd(bsums + ((size_t)grp * gridDim.y + blockIdx.y) * 2 * C + i, t);
    }
}

// ---------------------------------------------------------------------------------------- heads
constexpr int HS = 4;   // samples per block

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* feat, int B, int C, const float* W,
                                                       const float* bias, int ldc, int K, float* mu,
                                                       float* ls, float* la) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int NH = 2 * ldc + K;
    float* f = hs;               // [HS][C]
    float* o = hs + HS * C;      // [HS][NH]
    const int b0 = blockIdx.x * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < HS * C; i += 256) {
        const int s = i / C;
        f[i] = (b0 + s < B) ? feat[(int64_t)(b0 + s) * C + (i - s * C)] : 0.f;
    }
    __syncthreads();
    // one output per thread: its weight row streams in as independent float4 loads (C % 4 == 0), the
    // features are LDS broadcasts
    for (int n = tid; n < NH; n += 256) {
        float acc[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) acc[s] = 0.f;
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + (int64_t)n * C);
#pragma unroll 8
        for (int k4 = 0; k4 < C / 4; ++k4) {
            const f32x4 w = wr[k4];
#pragma unroll
            for (int s = 0; s < HS; ++s) {
                const f32x4 fv = *reinterpret_cast<const f32x4*>(f + s * C + 4 * k4);
                acc[s] += w[0] * fv[0] + w[1] * fv[1] + w[2] * fv[2] + w[3] * fv[3];
            }
        }
        const float bb = bias[n];
#pragma unroll
        for (int s = 0; s < HS; ++s) o[s * NH + n] = acc[s] + bb;
    }
    __syncthreads();
    for (int i = tid; i < HS * 2 * ldc; i += 256) {
        const int s = i / (2 * ldc), j = i - s * 2 * ldc;
        if (b0 + s >= B) continue;
        if (j < ldc) mu[(int64_t)(b0 + s) * ldc + j] = o[s * NH + j];
        else ls[(int64_t)(b0 + s) * ldc + (j - ldc)] = o[s * NH + j];
    }
    // log-softmax over the last K outputs: one wave per sample, two samples per wave
    for (int s = wave; s < HS; s += 4) {
        if (b0 + s >= B) continue;
        const float* z = o + s * NH + 2 * ldc;
        float mx = -INFINITY;
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, z[k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float se = 0.f;
        for (int k = lane; k < K; k += 64) se += expf(z[k] - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        for (int k = lane; k < K; k += 64) la[(int64_t)(b0 + s) * K + k] = z[k] - lse;
    }
}

// dout[b][n] = upstream gradient w.r.t. the pre-log-softmax head outputs; dfeat = dout * W
__global__ __launch_bounds__(256) void head_bwd_data_kernel(int B, int C, const float* W, int ldc, int K,
                                                            const float* la, const float* dmu,
                                                            const float* dls, const float* dla,
                                                            float* dfeat, float* dout) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int NH = 2 * ldc + K;
    float* d = hs;   // [HS][NH]
    const int b0 = blockIdx.x * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < HS * 2 * ldc; i += 256) {
        const int s = i / (2 * ldc), j = i - s * 2 * ldc;
        float v = 0.f;
        if (b0 + s < B) v = j < ldc ? dmu[(int64_t)(b0 + s) * ldc + j] : dls[(int64_t)(b0 + s) * ldc + j - ldc];
        d[s * NH + j] = v;
    }
    for (int s = wave; s < HS; s += 4) {
        float* dz = d + s * NH + 2 * ldc;
        if (b0 + s >= B) {
            for (int k = lane; k < K; k += 64) dz[k] = 0.f;
            continue;
        }
        const float* g = dla + (int64_t)(b0 + s) * K;
        const float* l = la + (int64_t)(b0 + s) * K;
        float sg = 0.f;
        for (int k = lane; k < K; k += 64) sg += g[k];
        sg = wave_sum(sg);
        for (int k = lane; k < K; k += 64) dz[k] = g[k] - expf(l[k]) * sg;
    }
    __syncthreads();
    if (blockIdx.y == 0) {
        for (int i = tid; i < HS * NH; i += 256) {
            const int s = i / NH;
            if (b0 + s < B) dout[(int64_t)(b0 + s) * NH + (i - s * NH)] = d[i];
        }
    }
    // blockIdx.y = 256-channel slice of the features (wide encoders: 640 channels, K = 100 -> three times the blocks).
    // With fewer than 256 channels the thread groups (tid / C) split the NH outputs among them and meet in LDS: 128
    // channels left half of the block idle in a 266-step dependent loop (85 us at 2048 samples)
    const int Cb = C < 256 ? C : 256, parts = 256 / Cb;
    if (parts > 1 && gridDim.y == 1 && 256 % Cb == 0) {
        float* red = d + HS * NH;                                   // [parts][HS][Cb]
        const int c = tid % Cb, part = tid / Cb;
        float acc[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) acc[s] = 0.f;
#pragma unroll 8
        for (int n = part; n < NH; n += parts) {
            const float w = W[(int64_t)n * C + c];
#pragma unroll
            for (int s = 0; s < HS; ++s) acc[s] += w * d[s * NH + n];
        }
#pragma unroll
        for (int s = 0; s < HS; ++s) red[(part * HS + s) * Cb + c] = acc[s];
        __syncthreads();
        for (int i = tid; i < HS * Cb; i += 256) {
            const int s = i / Cb, cc = i - s * Cb;
            float t = 0.f;
            for (int q = 0; q < parts; ++q) t += red[(q * HS + s) * Cb + cc];
            if (b0 + s < B) dfeat[(int64_t)(b0 + s) * C + cc] = t;
        }
        return;
    }
    for (int c = blockIdx.y * 256 + tid; c < C; c += 256 * gridDim.y) {
        float acc[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) acc[s] = 0.f;
#pragma unroll 8
        for (int n = 0; n < NH; ++n) {
            const float w = W[(int64_t)n * C + c];
#pragma unroll
            for (int s = 0; s < HS; ++s) acc[s] += w * d[s * NH + n];
        }
#pragma unroll
        for (int s = 0; s < HS; ++s)
            if (b0 + s < B) dfeat[(int64_t)(b0 + s) * C + c] = acc[s];
    }
}

// dW[n][c] += sum_b dout[b][n]*feat[b][c];  dbias[n] += sum_b dout[b][n]: a [NH x B] x [B x C] product in fp32 on the matrix
// cores (v_mfma_f32_32x32x2_f32: full fp32 multiply-add).  One WAVE per (32 outputs, 32 channels, batch slice): both operands
// are loaded from global memory straight into the MFMA layout -- lane (m, k) = (lane % 32, lane / 32) holds dout[b + k][n0 + m]
// and feat[b + k][c0 + m], 128 contiguous bytes per half-wave -- so there is no LDS stage and no re-read: every wave reads its
// slice of the two matrices once (the one-output-per-block version re-read feat 266 times: 33 us; this one 4 us).
// grid = (ceil(NH / 32) * ceil(C / 32), slices of 128 samples); the slices meet through float atomics (dW / dbias accumulate
// anyway).  Deterministic mode: the slices are launched one after the other (one adder per address at any time).
typedef float f32x16s __attribute__((ext_vector_type(16)));
constexpr int HWS = 128;          // samples per wave
__global__ __launch_bounds__(64) void head_bwd_weight_kernel(const float* feat, const float* dout, int B,
                                                             int C, int NH, float* dW, float* dbias, int slice0) {
    const int nct = (C + 31) / 32;
    const int n0 = (blockIdx.x / nct) * 32, c0 = (blockIdx.x % nct) * 32;
    const int lane = threadIdx.x, m = lane & 31, k = lane >> 5;
    // a wave owns HWS = 128 consecutive samples: ALL of its 2 x 64 loads are requested before the first MFMA (128 registers:
    // the memory latency is paid once per wave, not once per unrolled group of steps)
    const int b0 = (blockIdx.y + slice0) * HWS, b1 = min(B, b0 + HWS);
    const bool nok = n0 + m < NH, cok = c0 + m < C;
    const float* pd = dout + (nok ? n0 + m : 0);
    const float* pf = feat + (cok ? c0 + m : 0);
    float dv[HWS / 2], fv[HWS / 2];
#pragma unroll
    for (int u = 0; u < HWS / 2; ++u) {
        const int bb = min(b0 + 2 * u + k, B - 1);          // (clamped: rows beyond the slice are zeroed below)
        dv[u] = pd[(int64_t)bb * NH];
        fv[u] = pf[(int64_t)bb * C];
    }
    f32x16s acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float sb = 0.f;
#pragma unroll
    for (int u = 0; u < HWS / 2; ++u) {
        const bool bok = b0 + 2 * u + k < b1;
        const float d_ = nok && bok ? dv[u] : 0.f, f_ = cok && bok ? fv[u] : 0.f;
        sb += d_;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(d_, f_, acc, 0, 0, 0);
    }
    // D layout of the 32x32 tile: lane (r, h) = (lane % 32, lane / 32) holds column r, rows 8 g + 4 h + e in acc[4 g + e]
    if (cok) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + 8 * g + 4 * k + e;
                if (n < NH) atomicAdd(dW + (int64_t)n * C + c0 + m, acc[4 * g + e]);
            }
    }
    if (c0 == 0) {                                      // bias: the two sample halves of a row meet by shuffle
        sb += __shfl_xor(sb, 32);
        if (k == 0 && nok) atomicAdd(dbias + n0 + m, sb);
    }
}

// ---------------------------------------------------------------------------------------- sampler
template <typename T>
__global__ __launch_bounds__(128) void sample_fwd_kernel(const float* mu, const float* ls, const float* la,
                                                         const float* eps, const float* u,
                                                         const int64_t* label, const int64_t* label_mix,
                                                         float lam, const float* lam_dev, int mode,
                                                         float temperature, int B,
                                                         int ldc, int K, int Lpad, T* latent, float* csoft) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (lam_dev) lam = lam_dev[0];
    __shared__ float red[2];
    __shared__ float bc[2];
    T* out = latent + (int64_t)b * Lpad;
    for (int j = tid; j < ldc; j += 128) {
        const int64_t i = (int64_t)b * ldc + j;
        out[j] = (T)(mu[i] + expf(ls[i]) * eps[i]);
    }
    for (int j = ldc + K + tid; j < Lpad; j += 128) out[j] = (T)0.f;
    if (mode == 0) {
        const float EPS = 1e-12f;
        // K <= 128 handled one class per thread (K up to 1024 via the strided loops)
        float mx = -INFINITY;
        for (int k = tid; k < K; k += 128) {
            const float gum = -logf(-logf(u[(int64_t)b * K + k] + EPS) + EPS);
            mx = fmaxf(mx, (la[(int64_t)b * K + k] + gum) / temperature);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(red[0], red[1]);
        float se = 0.f;
        for (int k = tid; k < K; k += 128) {
            const float gum = -logf(-logf(u[(int64_t)b * K + k] + EPS) + EPS);
            se += expf((la[(int64_t)b * K + k] + gum) / temperature - mx);
        }
        se = wave_sum(se);
        if ((tid & 63) == 0) bc[tid >> 6] = se;
        __syncthreads();
        se = bc[0] + bc[1];
        for (int k = tid; k < K; k += 128) {
            const float gum = -logf(-logf(u[(int64_t)b * K + k] + EPS) + EPS);
            const float c = expf((la[(int64_t)b * K + k] + gum) / temperature - mx) / se;
            out[ldc + k] = (T)c;
            csoft[(int64_t)b * K + k] = c;
        }
    } else {
        const int la_ = (int)label[b];
        const int lb_ = mode == 2 ? (int)label_mix[b] : -1;
        for (int k = tid; k < K; k += 128) {
            float c = (k == la_) ? 1.f : 0.f;
            if (mode == 2) c = lam * c + (1.f - lam) * ((k == lb_) ? 1.f : 0.f);
            out[ldc + k] = (T)c;
            csoft[(int64_t)b * K + k] = c;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(128) void sample_bwd_kernel(const T* dlatent, const float* ls, const float* eps,
                                                         const float* csoft, int mode, float temperature,
                                                         int B, int ldc, int K, int Lpad, float* dmu,
                                                         float* dls, float* dla) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ float bc[2];
    const T* d = dlatent + (int64_t)b * Lpad;
    for (int j = tid; j < ldc; j += 128) {
        const int64_t i = (int64_t)b * ldc + j;
        const float dz = to_f(d[j]);
        dmu[i] += dz;
        dls[i] += dz * eps[i] * expf(ls[i]);
    }
    if (mode == 0) {
        float s = 0.f;
        for (int k = tid; k < K; k += 128) s += csoft[(int64_t)b * K + k] * to_f(d[ldc + k]);
        s = wave_sum(s);
        if ((tid & 63) == 0) bc[tid >> 6] = s;
        __syncthreads();
        s = bc[0] + bc[1];
        for (int k = tid; k < K; k += 128) {
            const float c = csoft[(int64_t)b * K + k];
            dla[(int64_t)b * K + k] += c * (to_f(d[ldc + k]) - s) / temperature;
        }
    }
}

// ---------------------------------------------------------------------------------------- losses
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void elbo_fwd_dev(const float* x, const float* xr, int64_t n_img,
                                                       const float* mu, const float* ls, const float* la,
                                                       int B, int ldc, int K, int bce, float x_sigma,
                                                       float log_prior, float* out3, int ostride) {
    __shared__ float red[4];
    out3 += (int64_t)blockIdx.x * ostride;         // (deterministic mode: a slot per block, det_collect adds them in order)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.f;
    const int64_t n4 = n_img / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(xr);
    for (int64_t i = t0; i < n4; i += stride) {
        const f32x4 a = x4[i], r = r4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (bce) s += fmaxf(r[j], 0.f) - r[j] * a[j] + log1pf(expf(-fabsf(r[j])));
            else { const float d = sigmoidf_(r[j]) - a[j]; s += d * d; }
        }
    }
    for (int64_t i = n4 * 4 + t0; i < n_img; i += stride) {
        const float a = x[i], r = xr[i];
        if (bce) s += fmaxf(r, 0.f) - r * a + log1pf(expf(-fabsf(r)));
        else { const float d = sigmoidf_(r) - a; s += d * d; }
    }
    float kc = 0.f, kd = 0.f;
    for (int64_t i = t0; i < (int64_t)B * ldc; i += stride) {
        const float m = mu[i], l2 = 2.f * ls[i];
        kc += m * m + expf(l2) - l2 - 1.f;
    }
    for (int64_t i = t0; i < (int64_t)B * K; i += stride) {
        const float l = la[i];
        kd += expf(l) * (l - log_prior);
    }
    const float rs = bce ? 1.f / (float)B : 1.f / (2.f * (float)B * x_sigma * x_sigma);
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out3, s * rs);
    kc = block_sum(kc, red);
    if (threadIdx.x == 0 && kc != 0.f) atomicAdd(out3 + 1, 0.5f * kc / (float)B);
    kd = block_sum(kd, red);
    if (threadIdx.x == 0 && kd != 0.f) atomicAdd(out3 + 2, kd / (float)B);
}
__global__ __launch_bounds__(256) void elbo_fwd_kernel(const float* x, const float* xr, int64_t n_img,
                                                       const float* mu, const float* ls, const float* la,
                                                       int B, int ldc, int K, int bce, float x_sigma,
                                                       float log_prior, float* out3, int ostride) { elbo_fwd_dev(x, xr, n_img, mu, ls, la, B, ldc, K, bce, x_sigma, log_prior, out3, ostride); }

__device__ __forceinline__ void elbo_bwd_dev(const float* x, const float* xr, int64_t n_img,
                                                       const float* mu, const float* ls, const float* la,
                                                       int B, int ldc, int K, int bce, float x_sigma,
                                                       float log_prior, const float* gout3, float* dxr,
                                                       float* dmu, float* dls, float* dla) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float g0 = gout3[0], g1 = gout3[1], g2 = gout3[2];
    const float rs = g0 * (bce ? 1.f / (float)B : 1.f / ((float)B * x_sigma * x_sigma));
    for (int64_t i = t0; i < n_img; i += stride) {
        const float sg = sigmoidf_(xr[i]);
        const float d = sg - x[i];
        dxr[i] = bce ? rs * d : rs * d * sg * (1.f - sg);
    }
    const float c1 = g1 / (float)B, c2 = g2 / (float)B;
    for (int64_t i = t0; i < (int64_t)B * ldc; i += stride) {
        dmu[i] = c1 * mu[i];
        dls[i] = c1 * (expf(2.f * ls[i]) - 1.f);
    }
    for (int64_t i = t0; i < (int64_t)B * K; i += stride) {
        const float l = la[i];
        dla[i] = c2 * expf(l) * (l - log_prior + 1.f);
    }
}
__global__ __launch_bounds__(256) void elbo_bwd_kernel(const float* x, const float* xr, int64_t n_img,
                                                       const float* mu, const float* ls, const float* la,
                                                       int B, int ldc, int K, int bce, float x_sigma,
                                                       float log_prior, const float* gout3, float* dxr,
                                                       float* dmu, float* dls, float* dla) { elbo_bwd_dev(x, xr, n_img, mu, ls, la, B, ldc, K, bce, x_sigma, log_prior, gout3, dxr, dmu, dls, dla); }

__device__ __forceinline__ void cls_fwd_dev(const float* pred, const float* label, const float* w,
                                                      int B, int K, float* out, int ostride) {
    __shared__ float red[4];
    out += (int64_t)blockIdx.x * ostride;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * K; i += (int64_t)gridDim.x * 256)
        s += pred[i] * label[i] * (w ? w[i / K] : 1.f);
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, -s / (float)B);
}
__global__ __launch_bounds__(256) void cls_fwd_kernel(const float* pred, const float* label, const float* w,
                                                      int B, int K, float* out, int ostride) { cls_fwd_dev(pred, label, w, B, K, out, ostride); }
__device__ __forceinline__ void cls_bwd_dev(const float* label, const float* w, int B, int K, const float* gout, float* dp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)B * K) dp[i] = -gout[0] * label[i] * (w ? w[i / K] : 1.f) / (float)B;
}
__global__ __launch_bounds__(256) void cls_bwd_kernel(const float* label, const float* w, int B, int K, const float* gout, float* dp) { cls_bwd_dev(label, w, B, K, gout, dp); }

// top-k accuracy counts (main_shot_vae.py:441-447): rank of the true class among the K scores of a row = how many
// classes score higher (ties: the lower index first, as a stable descending sort); hits[0] += rank < 1, hits[1] += rank < k
__global__ __launch_bounds__(256) void topk_hits_kernel(const float* score, const int64_t* label, int B, int K, int k,
                                                        float* hits) {
    __shared__ float red[4];
    float h1 = 0.f, hk = 0.f;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        const float* row = score + (int64_t)b * K;
        const int y = (int)label[b];
        const float sy = row[y];
        int rank = 0;
        for (int c = 0; c < K; ++c) rank += (row[c] > sy || (row[c] == sy && c < y)) ? 1 : 0;
        h1 += rank < 1 ? 1.f : 0.f;
        hk += rank < k ? 1.f : 0.f;
    }
    h1 = block_sum(h1, red);
    __syncthreads();
    hk = block_sum(hk, red);
    if (threadIdx.x == 0) { atomicAdd(hits, h1); atomicAdd(hits + 1, hk); }
}

__device__ __forceinline__ void post_fwd_dev(const float* mu, const float* ls, const float* mt,
                                                       const float* st, int B, int D, float* out, int ostride) {
    __shared__ float red[4];
    out += (int64_t)blockIdx.x * ostride;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * D; i += (int64_t)gridDim.x * 256) {
        const float a = mu[i] - mt[i], b = expf(ls[i]) - st[i];
        s += a * a + b * b;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s / (float)B);
}
__global__ __launch_bounds__(256) void post_fwd_kernel(const float* mu, const float* ls, const float* mt,
                                                       const float* st, int B, int D, float* out, int ostride) { post_fwd_dev(mu, ls, mt, st, B, D, out, ostride); }
__device__ __forceinline__ void post_bwd_dev(const float* mu, const float* ls, const float* mt, const float* st, int B,
                                int D, const float* gout, float* dmu, float* dls) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * D) return;
    const float g = gout[0] * 2.f / (float)B;
    const float e = expf(ls[i]);
    dmu[i] = g * (mu[i] - mt[i]);
    dls[i] = g * (e - st[i]) * e;
}
__global__ __launch_bounds__(256) void post_bwd_kernel(const float* mu, const float* ls, const float* mt, const float* st, int B,
                                int D, const float* gout, float* dmu, float* dls) { post_bwd_dev(mu, ls, mt, st, B, D, gout, dmu, dls); }

// ---- the loss stage of a step in three launches (sv_shot_loss_step2): at ~5 us per launch the 12 reduction / gradient kernels
//      of the stage cost more in launches than in work.  blockIdx.y selects the part; every part is the kernel above, unchanged.
struct shot_elbo_part { const float* x; const float* xr; const float* mu; const float* ls; const float* la; int B; float* out3;
                        const float* gout3; float* dxr; float* dmu; float* dls; float* dla; };
struct shot_post_part { const float* la; const float* label; const float* mu; const float* ls; const float* mt; const float* st; int B;
                        float* out_cls; float* out_post; const float* g_cls; const float* g_post; float* dla; float* dmu; float* dls; };
struct shot_parts { shot_elbo_part e[2]; shot_post_part q[2]; int64_t n_per_img; int D, K, bce; float x_sigma, log_prior; };

__global__ __launch_bounds__(256) void shot_elbo_fwd2_kernel(const shot_parts P) {
    const shot_elbo_part& e = P.e[blockIdx.y];
    elbo_fwd_dev(e.x, e.xr, P.n_per_img * e.B, e.mu, e.ls, e.la, e.B, P.D, P.K, P.bce, P.x_sigma, P.log_prior, e.out3, 0);
}
__global__ __launch_bounds__(256) void shot_post_fwd4_kernel(const shot_parts P) {
    const shot_post_part& q = P.q[blockIdx.y >> 1];
    if (blockIdx.y & 1) post_fwd_dev(q.mu, q.ls, q.mt, q.st, q.B, P.D, q.out_post, 0);
    else cls_fwd_dev(q.la, q.label, nullptr, q.B, P.K, q.out_cls, 0);
}
__global__ __launch_bounds__(256) void shot_bwd6_kernel(const shot_parts P) {
    const int y = blockIdx.y;
    if (y < 2) {
        const shot_elbo_part& e = P.e[y];
        elbo_bwd_dev(e.x, e.xr, P.n_per_img * e.B, e.mu, e.ls, e.la, e.B, P.D, P.K, P.bce, P.x_sigma, P.log_prior, e.gout3, e.dxr, e.dmu, e.dls, e.dla);
    } else {
        const shot_post_part& q = P.q[(y - 2) >> 1];
        if (y & 1) post_bwd_dev(q.mu, q.ls, q.mt, q.st, q.B, P.D, q.g_post, q.dmu, q.dls);
        else cls_bwd_dev(q.label, nullptr, q.B, P.K, q.g_cls, q.dla);
    }
}

// random permutations from uniform keys: perm[r] = i where r = rank of key i among the n keys of its batch (ties: the lower
// index first) -- blockIdx.y = batch; n is a minibatch size (<= a few thousand), so the O(n^2) counting pass in LDS beats a
// device-wide radix sort (torch.rand(n).argsort(): ~5 launches) by an order of magnitude
__global__ __launch_bounds__(256) void rank_perm_kernel(const float* keys, int n, int64_t* perm) {
    extern __shared__ float kk[];
    const float* k = keys + (size_t)blockIdx.y * n;
    int64_t* p = perm + (size_t)blockIdx.y * n;
    for (int i = threadIdx.x; i < n; i += 256) kk[i] = k[i];
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = kk[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (kk[j] < v || (kk[j] == v && j < i)) ? 1 : 0;
        p[r] = i;
    }
}

// ---------------------------------------------------------------------------------------- fused loss stage of the step
// Targets of the mixed forwards in ONE launch: label_smoothing / mixup_vae_data (lib/utils/mixup.py:22-25,36-39) applied to
// the OUTPUTS of forwards (1) and (3) -- gather-lerps of mu, exp(log_sigma), exp(log_alpha) and of the one-hot labels
// (onehot(label[perm]) = onehot(label)[perm], and ClsCriterion is linear in its label argument, so the two label terms of
// main_shot_vae.py:316-318 are one soft label lam * onehot(y) + (1 - lam) * onehot(y[perm])).
__global__ __launch_bounds__(256) void shot_targets_kernel(const float* mu_l, const float* ls_l, const float* mu_u, const float* ls_u,
                                                           const float* la_u, const int64_t* label_l, const int64_t* perm_l,
                                                           const int64_t* perm_u, float lam_l, const float* lam_l_dev, float lam_u,
                                                           const float* lam_u_dev, int Bl, int Bu, int D, int K, float* sm_mu,
                                                           float* sm_sigma, float* lab_mix, float* mx_mu, float* mx_sigma,
                                                           float* mx_alpha) {
    if (lam_l_dev) lam_l = lam_l_dev[0];
    if (lam_u_dev) lam_u = lam_u_dev[0];
    const int b = blockIdx.x;          // grid = max(Bl, Bu): the two loaders' batches may differ (the last labelled batch of an epoch)
    if (b < Bl) {
        const int64_t pl = perm_l[b];
        for (int j = threadIdx.x; j < D; j += blockDim.x) {
            const int64_t i = (int64_t)b * D + j;
            sm_mu[i] = lam_l * mu_l[i] + (1.f - lam_l) * mu_l[pl * D + j];
            sm_sigma[i] = lam_l * expf(ls_l[i]) + (1.f - lam_l) * expf(ls_l[pl * D + j]);
        }
        const int ya = (int)label_l[b], yb = (int)label_l[pl];
        for (int k = threadIdx.x; k < K; k += blockDim.x)
            lab_mix[(int64_t)b * K + k] = lam_l * (k == ya ? 1.f : 0.f) + (1.f - lam_l) * (k == yb ? 1.f : 0.f);
    }
    if (b < Bu) {
        const int64_t pu = perm_u[b];
        for (int j = threadIdx.x; j < D; j += blockDim.x) {
            const int64_t i = (int64_t)b * D + j;
            mx_mu[i] = lam_u * mu_u[i] + (1.f - lam_u) * mu_u[pu * D + j];
            mx_sigma[i] = lam_u * expf(ls_u[i]) + (1.f - lam_u) * expf(ls_u[pu * D + j]);
        }
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            const int64_t i = (int64_t)b * K + k;
            mx_alpha[i] = lam_u * expf(la_u[i]) + (1.f - lam_u) * expf(la_u[pu * K + k]);
        }
    }
}

// terms[0..9] = recon_l, KLc_l, KLd_l, recon_u, KLc_u, KLd_u, disc_post_l, cont_post_l, disc_post_u, cont_post_u ->
// terms[10] = loss_supervised, terms[11] = loss_unsupervised (main_shot_vae.py:293-296,316-323,343-346,358-363) and
// coef[i] = d(loss that contains term i) / d(term i)   (|.| has gradient sign(.), 0 at 0 like torch.abs)
__global__ void shot_compose_kernel(float* terms, sv_shot_schedule s, float* coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    const float cp = s.kl_beta_c * s.pwm;
    const float elbo_l = terms[0] + s.kl_beta_c * fabsf(terms[1] - s.cmi) + s.kl_beta_d * fabsf(terms[2] - s.dmi) + cp * terms[7];
    const float elbo_u = terms[3] + s.kl_beta_c * fabsf(terms[4] - s.cmi) + s.kl_beta_d * fabsf(terms[5] - s.dmi) + cp * terms[9];
    terms[10] = s.ew * elbo_l + terms[6];
    terms[11] = s.ew * elbo_u + s.ucw * terms[8];
    coef[0] = s.ew;
    coef[1] = s.ew * s.kl_beta_c * sgn(terms[1] - s.cmi);
    coef[2] = s.ew * s.kl_beta_d * sgn(terms[2] - s.dmi);
    coef[3] = s.ew;
    coef[4] = s.ew * s.kl_beta_c * sgn(terms[4] - s.cmi);
    coef[5] = s.ew * s.kl_beta_d * sgn(terms[5] - s.dmi);
    coef[6] = 1.f;
    coef[7] = s.ew * cp;
    coef[8] = s.ucw;
    coef[9] = s.ew * cp;
}

// upstream gradients of the two losses times the per-term coefficients -> the `gout` operands of the backward kernels
__global__ void shot_scale_kernel(const float* coef, const float* g_sup, const float* g_unsup, float* gvec) {
    const int i = threadIdx.x;
    if (i >= 10) return;
    const bool sup = i < 3 || i == 6 || i == 7;
    const float g = sup ? (g_sup ? g_sup[0] : 0.f) : (g_unsup ? g_unsup[0] : 0.f);
    gvec[i] = coef[i] * g;
}

// ---------------------------------------------------------------------------------------- mixup
__global__ void mix_lerp_kernel(const float* a, const int64_t* index, float lam, const float* lam_dev, int B,
                                int64_t row, int exp_space, float* out) {
    if (lam_dev) lam = lam_dev[0];
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= row) return;
    const int64_t src = index[b];
    float p = a[(int64_t)b * row + j], q = a[src * row + j];
    if (exp_space) { p = expf(p); q = expf(q); }
    out[(int64_t)b * row + j] = lam * p + (1.f - lam) * q;
}

// pairwise Gaussian KL(N_i || N_j), second-smallest per row (torch.topk(k=2, largest=False)[:,1])
__global__ __launch_bounds__(256) void optimal_match_kernel(const float* mu, const float* ls, int B, int D,
                                                            int64_t* index) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* mi = sm;           // [D]
    float* vi = sm + D;       // [D] sigma_i^2
    float* kl = sm + 2 * D;   // [B]
    const int i = blockIdx.x, tid = threadIdx.x;
    float lsum_i = 0.f;
    for (int d = tid; d < D; d += 256) {
        mi[d] = mu[(int64_t)i * D + d];
        vi[d] = expf(2.f * ls[(int64_t)i * D + d]);
    }
    __syncthreads();
    for (int d = 0; d < D; ++d) lsum_i += ls[(int64_t)i * D + d];
    for (int j = tid; j < B; j += 256) {
        float acc = 0.f, lsum_j = 0.f;
        for (int d = 0; d < D; ++d) {
            const float l = ls[(int64_t)j * D + d];
            const float iv = expf(-2.f * l);
            const float dm = mi[d] - mu[(int64_t)j * D + d];
            acc += 0.5f * (vi[d] + dm * dm) * iv;
            lsum_j += l;
        }
        kl[j] = (lsum_j - lsum_i) + acc - 0.5f * (float)D;
    }
    __syncthreads();
    if (tid < 64) {
        // wave 0: two argmin passes
        int skip = -1, best = -1;
        for (int pass = 0; pass < 2; ++pass) {
            float bv = INFINITY;
            int bi = 0x7fffffff;
            for (int j = tid; j < B; j += 64)
                if (j != skip && (kl[j] < bv || (kl[j] == bv && j < bi))) { bv = kl[j]; bi = j; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            skip = bi;
            best = bi;
        }
        if (tid == 0) index[i] = best;
    }
}

// ---------------------------------------------------------------------------------------- smooth-ELBO conv-VAE (config 5)
// Heads of svhn_VAE / mnist_VAE (smooth_vae_model/svhn_vae.py:137-208) behind the fused head GEMM: o[b] = [mean | logvar |
// logits | pad] -> alpha = softmax(logits), z = mean + exp(logvar / 2) * eps (training) or mean, the Gumbel-softmax sample
// gs = softmax((log(alpha + EPS) + g) / T), g = -log(-log(u + EPS) + EPS) (training) or one-hot(argmax alpha), the
// discrete code c = one-hot(label) for labelled data, else gs; latent = [z | c | 0] in the compute dtype (the decoder's
// first GEMM reads it) and in fp32 (API).  One block per sample.
constexpr float SM_EPS = 1e-12f;
template <typename T>
__global__ __launch_bounds__(64) void smooth_latent_fwd_kernel(const T* o, int ldo, const float* eps, const float* u,
                                                                const int64_t* label, float temperature, int training, int Dc,
                                                                int Dd, int Lpad, float* mean, float* logvar, float* alpha,
                                                                float* gs, T* latent, float* latent32) {
    const int b = blockIdx.x, l = threadIdx.x;
    const T* ob = o + (int64_t)b * ldo;
    T* lat = latent + (int64_t)b * Lpad;
    float* l32 = latent32 + (int64_t)b * (Dc + Dd);
    for (int j = l; j < Dc; j += 64) {
        const float m = to_f(ob[j]), lv = to_f(ob[Dc + j]);
        mean[(int64_t)b * Dc + j] = m;
        logvar[(int64_t)b * Dc + j] = lv;
        const float z = training ? m + expf(0.5f * lv) * eps[(int64_t)b * Dc + j] : m;
        lat[j] = (T)z;
        l32[j] = z;
    }
    // softmax over the Dd logits (Dd <= 64 per pass; strided loops cover more)
    float mx = -INFINITY;
    for (int k = l; k < Dd; k += 64) mx = fmaxf(mx, to_f(ob[2 * Dc + k]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float se = 0.f;
    for (int k = l; k < Dd; k += 64) se += expf(to_f(ob[2 * Dc + k]) - mx);
    se = wave_sum(se);
    // second pass: alpha and the Gumbel-softmax sample
    float ymx = -INFINITY;
    int amax = 0x7fffffff;
    float abest = -INFINITY;
    for (int k = l; k < Dd; k += 64) {
        const float a = expf(to_f(ob[2 * Dc + k]) - mx) / se;
        alpha[(int64_t)b * Dd + k] = a;
        if (a > abest) { abest = a; amax = k; }
        if (training) {
            const float g = -logf(-logf(u[(int64_t)b * Dd + k] + SM_EPS) + SM_EPS);
            ymx = fmaxf(ymx, (logf(a + SM_EPS) + g) / temperature);
        }
    }
    if (training) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymx = fmaxf(ymx, __shfl_xor(ymx, off));
        float ys = 0.f;
        for (int k = l; k < Dd; k += 64) {
            const float a = alpha[(int64_t)b * Dd + k];
            const float g = -logf(-logf(u[(int64_t)b * Dd + k] + SM_EPS) + SM_EPS);
            ys += expf((logf(a + SM_EPS) + g) / temperature - ymx);
        }
        ys = wave_sum(ys);
        for (int k = l; k < Dd; k += 64) {
            const float a = alpha[(int64_t)b * Dd + k];
            const float g = -logf(-logf(u[(int64_t)b * Dd + k] + SM_EPS) + SM_EPS);
            gs[(int64_t)b * Dd + k] = expf((logf(a + SM_EPS) + g) / temperature - ymx) / ys;
        }
    } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {       // argmax (first maximum)
            const float ov = __shfl_xor(abest, off);
            const int oi = __shfl_xor(amax, off);
            if (ov > abest || (ov == abest && oi < amax)) { abest = ov; amax = oi; }
        }
        for (int k = l; k < Dd; k += 64) gs[(int64_t)b * Dd + k] = k == amax ? 1.f : 0.f;
    }
    const int y = label ? (int)label[b] : -1;
    for (int k = l; k < Dd; k += 64) {
        const float c = label ? (k == y ? 1.f : 0.f) : gs[(int64_t)b * Dd + k];
        lat[Dc + k] = (T)c;
        l32[Dc + k] = c;
    }
    for (int k = Dc + Dd + l; k < Lpad; k += 64) lat[k] = (T)0.f;
}

// backward of the above: d_o = gradient w.r.t. the head GEMM's output row.  dlat: gradient of the (compute-dtype) latent
// from the decoder, dmean / dlogvar / dalpha: gradients from the loss (may be NULL), sample_path = 1 when c was the
// Gumbel-softmax sample (unlabelled data in training mode)
template <typename T>
__global__ __launch_bounds__(64) void smooth_latent_bwd_kernel(const T* dlat, int Lpad, const float* dmean, const float* dlogvar,
                                                                const float* dalpha, const float* logvar, const float* eps,
                                                                const float* alpha, const float* gs, float temperature,
                                                                int training, int sample_path, int Dc, int Dd, T* d_o, int ldo) {
    const int b = blockIdx.x, l = threadIdx.x;
    const T* dl = dlat + (int64_t)b * Lpad;
    T* out = d_o + (int64_t)b * ldo;
    for (int j = l; j < Dc; j += 64) {
        const int64_t i = (int64_t)b * Dc + j;
        const float dz = to_f(dl[j]);
        float dm = dz + (dmean ? dmean[i] : 0.f);
        float dv = dlogvar ? dlogvar[i] : 0.f;
        if (training) dv += dz * eps[i] * 0.5f * expf(0.5f * logvar[i]);
        out[j] = (T)dm;
        out[Dc + j] = (T)dv;
    }
    // d alpha: from the loss and, on the sample path, through the Gumbel-softmax
    float sdc = 0.f;
    if (sample_path && training)
        for (int k = l; k < Dd; k += 64) sdc += to_f(dl[Dc + k]) * gs[(int64_t)b * Dd + k];
    sdc = wave_sum(sdc);
    float sda = 0.f;
    for (int k = l; k < Dd; k += 64) {
        const int64_t i = (int64_t)b * Dd + k;
        const float a = alpha[i];
        float da = dalpha ? dalpha[i] : 0.f;
        if (sample_path && training) da += gs[i] * (to_f(dl[Dc + k]) - sdc) / temperature / (a + SM_EPS);
        sda += da * a;
    }
    sda = wave_sum(sda);
    for (int k = l; k < Dd; k += 64) {
        const int64_t i = (int64_t)b * Dd + k;
        const float a = alpha[i];
        float da = dalpha ? dalpha[i] : 0.f;
        if (sample_path && training) da += gs[i] * (to_f(dl[Dc + k]) - sdc) / temperature / (a + SM_EPS);
        out[2 * Dc + k] = (T)(a * (da - sda));
    }
    for (int k = 2 * Dc + Dd + l; k < ldo; k += 64) out[k] = (T)0.f;
}

// decoder output f [B][HW][ld] (compute dtype) -> reconstruction tanh(f[..., :C]) as NCHW fp32 (svhn_vae.py:118-120), and
// its backward d_f = d_rec * (1 - rec^2) in NHWC (pad channels zero)
template <typename T>
__global__ void tanh_to_nchw_kernel(const T* f, int B, int Cc, int HW, int ld, float* out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned)B * Cc * HW) return;
    const unsigned p = i % HW, bc = i / HW, b = bc / Cc, c = bc - b * Cc;
    out[i] = tanhf(to_f(f[((size_t)b * HW + p) * ld + c]));
}
template <typename T>
__global__ void tanh_to_nchw_bwd_kernel(const float* d_out, const float* out, int B, int Cc, int HW, int ld, T* d_f) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;          // over B*HW*ld
    if (i >= (unsigned)B * HW * ld) return;
    const unsigned c = i % ld, bp = i / ld, b = bp / HW, p = bp - b * HW;
    float v = 0.f;
    if (c < (unsigned)Cc) {
        const size_t j = ((size_t)b * Cc + c) * HW + p;
        v = d_out[j] * (1.f - out[j] * out[j]);
    }
    d_f[i] = (T)v;
}

// the same with one thread per PIXEL (Cc <= 8 image channels, ld a multiple of 8): coalesced 4-byte reads along each channel
// plane, the pixel's row written as 16-byte vectors (the element-wise form above: 62 us for 2 048 images, a division and a
// 2-byte store per element of the padded row)
template <typename T>
__global__ __launch_bounds__(256) void tanh_to_nchw_bwd_px_kernel(const float* d_out, const float* out, int B, int Cc, int HW, int ld, T* d_f) {
    typedef typename V8<T>::type V;
    const unsigned i = blockIdx.x * 256 + threadIdx.x;                  // over B * HW
    if (i >= (unsigned)B * HW) return;
    const unsigned b = i / HW, p = i - b * HW;
    V v;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x = 0.f;
        if (c < Cc) {
            const size_t j = ((size_t)b * Cc + c) * HW + p;
            const float o = out[j];
            x = d_out[j] * (1.f - o * o);
        }
        v[c] = (T)x;
    }
    V z;
#pragma unroll
    for (int c = 0; c < 8; ++c) z[c] = (T)0.f;
    V* row = reinterpret_cast<V*>(d_f + (size_t)i * ld);
    row[0] = v;
    for (int k = 1; k < ld / 8; ++k) row[k] = z;
}

// out[b][y][x][:] = in[b][2y][2x][:] -- the even positions of an NHWC tensor, compactly (the operand of a stride-2 1x1 layer's
// data gradient run as a dense product over the stride-2 grid); one 16-byte vector per thread
template <typename T>
__global__ __launch_bounds__(256) void gather_even_kernel(const T* in, int64_t nvec, int Hq, int Wq, int cv, T* out) {
    typedef typename V8<T>::type V;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int v = (int)(i % cv);
    const int64_t px = i / cv, x = px % Wq, by = px / Wq, y = by % Hq, b = by / Hq;
    const int64_t src = ((b * (2 * Hq) + 2 * y) * (2 * Wq) + 2 * x) * cv + v;
    reinterpret_cast<V*>(out)[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(in) + src);
}

// Trainer._loss_function (main_smooth_ELBO_svhn.py:228-310) raw terms: t[0] = num_pixels * MSE = sum (rec - x)^2 / B,
// t[1] = KL_c (:312-335), t[2] = sum alpha log(alpha + EPS) / B (KL_d = log D + t[2], :368-388), t[3] = BCE(alpha, one-hot)
// (mean over B * D; 0 without labels).  t must be zeroed by the caller.
__global__ __launch_bounds__(256) void smooth_elbo_fwd_kernel(const float* data, const float* rec, int64_t n, const float* mean,
                                                              const float* logvar, const float* alpha, const int64_t* label,
                                                              int B, int Dc, int Dd, float* t, int ostride) {
    __shared__ float red[4];
    t += (int64_t)blockIdx.x * ostride;
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    {   // 16-byte loads (n is a multiple of 4: whole images), the tail element-wise
        const int64_t n4 = n / 4;
        const f32x4* r4 = reinterpret_cast<const f32x4*>(rec);
        const f32x4* d4 = reinterpret_cast<const f32x4*>(data);
        for (int64_t i = t0; i < n4; i += stride) {
            const f32x4 a = r4[i], b = d4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = a[j] - b[j]; s += d * d; }
        }
        for (int64_t i = 4 * n4 + t0; i < n; i += stride) { const float d = rec[i] - data[i]; s += d * d; }
    }
    float kc = 0.f;
    for (int64_t i = t0; i < (int64_t)B * Dc; i += stride) { const float m = mean[i], lv = logvar[i]; kc += -0.5f * (1.f + lv - m * m - expf(lv)); }
    float kd = 0.f, bc = 0.f;
    for (int64_t i = t0; i < (int64_t)B * Dd; i += stride) {
        const float a = alpha[i];
        kd += a * logf(a + SM_EPS);
        if (label) {
            const bool y = (int)label[i / Dd] == (int)(i % Dd);
            bc += -(y ? fmaxf(logf(a), -100.f) : fmaxf(log1pf(-a), -100.f));        // F.binary_cross_entropy clamps the logs at -100
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(t, s / (float)B);
    kc = block_sum(kc, red);
    if (threadIdx.x == 0) atomicAdd(t + 1, kc / (float)B);
    kd = block_sum(kd, red);
    if (threadIdx.x == 0) atomicAdd(t + 2, kd / (float)B);
    bc = block_sum(bc, red);
    if (threadIdx.x == 0 && label) atomicAdd(t + 3, bc / ((float)B * (float)Dd));
}

// loss = t0 + gamma_c |C_c - KL_c| + gamma_d |C_d - KL_d| + alpha_cls * t3 with the linearly growing capacities
// C = min((max - min) * steps / iters + min, max) (C_d also capped at log D); steps from the host or from a device counter
// (hipGraph replay).  Writes t[4] = loss, t[5..8] = the four weighted parts (reconstruction, continuous capacity, discrete
// capacity, classification) and coef[i] = d loss / d t[i].
struct smooth_caps { float cmin, cmax, citers, cgamma, dmin, dmax, diters, dgamma, alpha_cls, steps; };
__global__ void smooth_compose_kernel(float* t, smooth_caps c, const float* steps_dev, int Dd, int has_label, float* coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float steps = steps_dev ? steps_dev[0] : c.steps;
    const float cc = fminf((c.cmax - c.cmin) * steps / c.citers + c.cmin, c.cmax);
    const float logD = logf((float)Dd);
    const float cd = fminf(fminf((c.dmax - c.dmin) * steps / c.diters + c.dmin, c.dmax), logD);
    const float klc = t[1], kld = logD + t[2];
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    t[5] = t[0];
    t[6] = c.cgamma * fabsf(cc - klc);
    t[7] = c.dgamma * fabsf(cd - kld);
    t[8] = has_label ? c.alpha_cls * t[3] : 0.f;
    t[4] = t[5] + t[6] + t[7] + t[8];
    coef[0] = 1.f;
    coef[1] = c.cgamma * sgn(klc - cc);
    coef[2] = c.dgamma * sgn(kld - cd);
    coef[3] = has_label ? c.alpha_cls : 0.f;
}

__global__ __launch_bounds__(256) void smooth_elbo_bwd_kernel(const float* data, const float* rec, int64_t n, const float* mean,
                                                              const float* logvar, const float* alpha, const int64_t* label,
                                                              int B, int Dc, int Dd, const float* coef, const float* gout,
                                                              float* d_rec, float* d_mean, float* d_logvar, float* d_alpha) {
    const int64_t stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float g = gout[0], ib = 1.f / (float)B;
    const float c0 = g * coef[0] * 2.f * ib, c1 = g * coef[1] * ib, c2 = g * coef[2] * ib, c3 = g * coef[3] * ib / (float)Dd;
    for (int64_t i = t0; i < n; i += stride) d_rec[i] = c0 * (rec[i] - data[i]);
    for (int64_t i = t0; i < (int64_t)B * Dc; i += stride) {
        d_mean[i] = c1 * mean[i];
        d_logvar[i] = c1 * (-0.5f) * (1.f - expf(logvar[i]));
    }
    for (int64_t i = t0; i < (int64_t)B * Dd; i += stride) {
        const float a = alpha[i];
        float d = c2 * (logf(a + SM_EPS) + a / (a + SM_EPS));
        if (label) {
            const float y = (int)label[i / Dd] == (int)(i % Dd) ? 1.f : 0.f;
            d += c3 * (a - y) / fmaxf((1.f - a) * a, 1e-12f);          // torch's binary_cross_entropy backward
        }
        d_alpha[i] = d;
    }
}

// ---------------------------------------------------------------------------------------- Adam
// torch.optim.Adam (no weight decay, no amsgrad; main_smooth_ELBO_svhn.py:428) on a flat parameter buffer: one launch
// instead of one per tensor.  step = the 1-based count of this update, from the host or from a device counter (hipGraph).
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                                                   float b2, float eps, float step, const float* step_dev, float gscale) {
    const float t = step_dev ? step_dev[0] : step;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, inv_sq_bc2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sq_bc2 + eps);
    }
}

// ---------------------------------------------------------------------------------------- SGD
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* v, int64_t n, float lr,
                                                  float momentum, float wd, float gscale, int first) {
    const int64_t n4 = n / 4;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = p4[i], gg = g4[i], vv = v4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = gg[j] * gscale + wd * pp[j];
            vv[j] = first ? d : momentum * vv[j] + d;
            pp[j] -= lr * vv[j];
        }
        p4[i] = pp;
        v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float d = g[i] * gscale + wd * p[i];
            v[i] = first ? d : momentum * v[i] + d;
            p[i] -= lr * v[i];
        }
}

// ---------------------------------------------------------------------------------------- layout
// one thread per pixel: C strided channel reads (coalesced across the pixels of a wave), the Cpad outputs of the pixel
// as 16-byte stores (one 2-byte store per thread ran at 1.5 TB/s)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* in, int B, int C, int HW, int Cpad, T* out) {
    typedef typename V8<T>::type V;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over B*HW
    if (i >= (int64_t)B * HW) return;
    const int64_t b = i / HW, p = i - b * HW;
    const float* src = in + b * C * HW + p;
    T* dst = out + i * Cpad;
    if (Cpad % 8 == 0) {
        for (int c0 = 0; c0 < Cpad; c0 += 8) {
            V v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (T)(c0 + j < C ? src[(int64_t)(c0 + j) * HW] : 0.f);
            *reinterpret_cast<V*>(dst + c0) = v;
        }
    } else {
        for (int c = 0; c < Cpad; ++c) dst[c] = (T)(c < C ? src[(int64_t)c * HW] : 0.f);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* in, int B, int C, int HW, int ld, float* out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;           // over B*C*HW (< 2^32)
    if (i >= (unsigned)B * C * HW) return;
    const unsigned p = i % HW;
    const unsigned bc = i / HW;
    const unsigned b = bc / C, c = bc - b * C;
    out[i] = to_f(in[((size_t)b * HW + p) * ld + c]);
}

// K21: gather + reflect pad + flip + crop + ToTensor in one pass (one thread per output element; the uint8 source of a
// batch is 3 KB per image and L2-resident after the first touch)
__device__ __forceinline__ int reflect_idx(int t, int n) { return t < 0 ? -t : (t >= n ? 2 * n - 2 - t : t); }
template <typename T>
__global__ void augment_kernel(const uint8_t* data, const int64_t* index, const int32_t* params, int B, int H, int W,
                               int C, int pad, int cpad, T* out_nhwc, float* out_nchw) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned CC = cpad ? cpad : C;
    if (i >= (unsigned)B * H * W * CC) return;
    unsigned b, y, x, c;
    if (cpad) {                      // NHWC: channel fastest
        c = i % CC;
        const unsigned p = i / CC;
        x = p % W;
        y = (p / W) % H;
        b = p / (W * H);
    } else {                         // NCHW: x fastest
        x = i % W;
        y = (i / W) % H;
        c = (i / (W * H)) % C;
        b = i / (W * H * C);
    }
    float v = 0.f;
    if (c < (unsigned)C) {
        int sy = y, sx = x;
        if (params) {
            const int oy = params[3 * b], ox = params[3 * b + 1], flip = params[3 * b + 2];
            const int px = ox + (int)x;                              // column in the (flipped) padded image
            sy = reflect_idx(oy + (int)y - pad, H);
            sx = reflect_idx((flip ? (W + 2 * pad - 1 - px) : px) - pad, W);
        }
        v = (float)data[(((size_t)index[b] * H + sy) * W + sx) * C + c] / 255.0f;
    }
    if (cpad) out_nhwc[i] = (T)v;
    else out_nchw[i] = v;
}

struct repack_params {
    int N, T_orig, C, transpose, nphase;
    int n_real, c_real;              // source extents (beyond them: zero padding)
    int64_t sn, st, sc;              // source strides (elements) of (n, tap, c): master layout = (T_orig * C, C, 1)
    int ntap[SV_MAX_PHASES];
    int64_t w_off[SV_MAX_PHASES], size[SV_MAX_PHASES];
    int8_t torig[SV_MAX_PHASES][SV_MAX_TAPS];
};
// every (layer, direction, phase) of the network in ONE launch: block b serves the job with block0 <= b < next block0;
// a job has ceil(size / 1024) blocks.  Transposed packs (data gradient: dst[c][t][n] from master[n][t][c]) go through a
// 32 x 32 LDS tile so that both the fp32 reads and the stores are coalesced (element-wise the reads were 4 bytes at a
// stride of T * C floats: 0.4 ms for the 36.5 M parameters of WRN-28-10).
template <typename T>
__global__ __launch_bounds__(256) void repack_batch_kernel(const float* master, const sv_repack_job* jobs, int njobs, T* dst) {
    __shared__ float tile[32][33];
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                       // last job whose first block is <= b (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const sv_repack_job& J = jobs[lo];
    const int bj = b - J.block0;
    const bool al4 = (J.dst_off & 3) == 0 && (J.master_off & 3) == 0;          // 16-byte accesses
    if (J.transpose && J.N % 32 == 0 && J.C % 32 == 0 && al4) {      // size / 1024 = ntap * (N / 32) * (C / 32) tiles exactly
        const int nNt = J.N / 32;
        const int nt_i = bj % nNt, t = (bj / nNt) % J.ntap, ct_i = bj / (nNt * J.ntap);
        if (ct_i >= J.C / 32) return;
        const int v = threadIdx.x & 7, r = threadIdx.x >> 3;          // 16-byte reads: 8 lanes per row of 32 channels
        const f32x4 q = *reinterpret_cast<const f32x4*>(master + J.master_off +
                                                        ((int64_t)(nt_i * 32 + r) * J.T_orig + J.torig[t]) * J.C + ct_i * 32 + 4 * v);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[r][4 * v + j] = q[j];
        __syncthreads();
        typename V4<T>::type o;                                        // 4 consecutive n of channel r
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (T)tile[4 * v + j][r];
        *reinterpret_cast<typename V4<T>::type*>(dst + J.dst_off + ((int64_t)(ct_i * 32 + r) * J.ntap + t) * J.N + nt_i * 32 + 4 * v) = o;
        return;
    }
    const unsigned cp = J.transpose ? J.N : J.C, size = (unsigned)J.size;       // (a pack has < 2^31 elements)
    if (!J.transpose && J.C % 4 == 0 && al4) {     // 4 consecutive channels per thread
        const unsigned i = (unsigned)bj * 1024u + 4u * threadIdx.x;
        if (i >= size) return;
        const unsigned c1 = i % cp, qd = i / cp;
        const unsigned t = qd % (unsigned)J.ntap, n1 = qd / (unsigned)J.ntap;
        const f32x4 q = *reinterpret_cast<const f32x4*>(master + J.master_off + ((int64_t)n1 * J.T_orig + J.torig[t]) * J.C + c1);
        typename V4<T>::type o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (T)q[j];
        *reinterpret_cast<typename V4<T>::type*>(dst + J.dst_off + i) = o;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = (unsigned)bj * 1024u + 256u * k + threadIdx.x;
        if (i >= size) continue;
        const unsigned c1 = i % cp, q = i / cp;
        const unsigned t = q % (unsigned)J.ntap, n1 = q / (unsigned)J.ntap;
        const unsigned n = J.transpose ? c1 : n1, c = J.transpose ? n1 : c1;
        dst[J.dst_off + i] = (T)master[J.master_off + ((int64_t)n * J.T_orig + J.torig[t]) * J.C + c];
    }
}
// sv_param_gather / sv_param_scatter_add: the job tables of the smooth-ELBO models (parameters in torch's own layouts)
__device__ __forceinline__ const sv_param_job& find_job(const sv_param_job* jobs, int njobs, int b) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                       // last job whose first block is <= b (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    return jobs[lo];
}
__device__ __forceinline__ int64_t job_elem(const sv_param_job& J, int n, int t, int c) {
    const int n_hi = n / J.n_lo_count, n_lo = n - n_hi * J.n_lo_count;
    return n_hi * J.sn_hi + n_lo * J.sn_lo + (int64_t)J.torig[t] * J.st + c * J.sc;
}
template <typename T>
__global__ __launch_bounds__(256) void param_gather_kernel(const sv_param_job* jobs, int njobs, T* dst) {
    const sv_param_job& J = find_job(jobs, njobs, blockIdx.x);
    const int bj = blockIdx.x - J.block0;
    const int cp = J.transpose ? J.N : J.C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = (int64_t)bj * 1024 + 256 * k + threadIdx.x;
        if (i >= J.size) continue;
        const int c1 = (int)(i % cp);
        const int64_t q = i / cp;
        const int t = (int)(q % J.ntap), n1 = (int)(q / J.ntap);
        const int n = J.transpose ? c1 : n1, c = J.transpose ? n1 : c1;
        const int64_t di = J.dst_ld ? (int64_t)n1 * J.dst_ld + (int64_t)t * cp + c1 : i;
        dst[J.dst_off + di] = (n < J.n_real && c < J.c_real) ? (T)J.ptr[job_elem(J, n, t, c)] : (T)0.f;
    }
}
__global__ __launch_bounds__(256) void param_scatter_add_kernel(const sv_param_job* jobs, int njobs, const float* src) {
    const sv_param_job& J = find_job(jobs, njobs, blockIdx.x);
    const int bj = blockIdx.x - J.block0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = (int64_t)bj * 1024 + 256 * k + threadIdx.x;          // over the REAL extents: [n_real][ntap][c_real]
        if (i >= J.size) continue;
        const int c = (int)(i % J.c_real);
        const int64_t q = i / J.c_real;
        const int t = (int)(q % J.ntap), n = (int)(q / J.ntap);
        // (transpose = 1 marks a source of DOUBLES at float offset dst_off: the per-channel sums a data gradient's epilogue left --
        //  sv_acc_t -- which are the bias gradient of the layer in front)
        const int64_t si = ((int64_t)n * J.ntap + t) * J.C + c;
        J.ptr[job_elem(J, n, t, c)] += J.transpose ? (float)reinterpret_cast<const double*>(src + J.dst_off)[si] : src[J.dst_off + si];
    }
}

template <typename T>
__global__ void repack_kernel(const float* master, const repack_params p, T* dst) {
    // dst rows n' (= N or C when transposed), cols [ntap][c'] per phase
    const int ph = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.size[ph]) return;
    const int cp = p.transpose ? p.N : p.C;   // inner (c') extent
    const int nt = p.ntap[ph];
    const int c1 = (int)(i % cp);
    const int t = (int)((i / cp) % nt);
    const int n1 = (int)(i / ((int64_t)cp * nt));
    const int n = p.transpose ? c1 : n1, c = p.transpose ? n1 : c1;
    int to = 0;
#pragma unroll
    for (int k = 0; k < SV_MAX_TAPS; ++k)
        if (k == t) to = p.torig[ph][k];
    dst[p.w_off[ph] + i] = (n < p.n_real && c < p.c_real) ? (T)master[n * p.sn + to * p.st + c * p.sc] : (T)0.f;
}

inline int nblocks(int64_t n, int bs, int cap = 2048) {
    int64_t b = (n + bs - 1) / bs;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

#define DISPATCH_T(dtype, ...)                          \
    do {                                                \
        if ((dtype) == SV_BF16) { typedef bf16 T; __VA_ARGS__; } \
        else if ((dtype) == SV_F32) { typedef float T; __VA_ARGS__; } \
        else { sv_set_error("bad dtype %d", (int)(dtype)); return SV_E_ARG; } \
    } while (0)

extern "C" {

int sv_bn_finalize(const double* stats, int replicas, int C, float count, const float* gamma, const float* beta, float eps,
                   float momentum, float* rm, float* rv, float* scale, float* shift, float* mean,
                   float* rstd, int groups, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(stats && gamma && beta && scale && shift && mean && rstd && C > 0 && replicas >= 1, SV_E_ARG,
               "sv_bn_finalize: bad args");
    groups = sv_ngroups(groups);
    SV_REQUIRE(groups == 1 || !rm, SV_E_ARG, "sv_bn_finalize: a batched launch cannot update the running statistics in "
               "place (the groups' momentum updates are ordered): pass NULL and call sv_bn_running_update");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8, groups), dim3(256), 0, (hipStream_t)stream, stats, replicas, C,
                       count, gamma, beta, eps, momentum, rm, rv, scale, shift, mean, rstd);
    return sv_check_launch("sv_bn_finalize");
}

int sv_bn_running_update_ex(const int32_t* table, const float* counts, int nbn, const float* bnbuf, float* bufs,
                            float eps, float momentum, int align, int groups, const int32_t* order, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(table && counts && bnbuf && bufs && nbn > 0 && align > 0, SV_E_ARG, "sv_bn_running_update: bad args");
    groups = sv_ngroups(groups);
    SV_REQUIRE(groups <= 8, SV_E_ARG, "sv_bn_running_update: groups=%d (at most 8)", groups);
    int packed = 0, seen = 0;
    for (int k = 0; k < groups; ++k) {
        const int gi = order ? order[k] : k;
        SV_REQUIRE(gi >= 0 && gi < groups && !(seen & (1 << gi)), SV_E_ARG, "sv_bn_running_update: order is not a permutation");
        seen |= 1 << gi;
        packed |= gi << (4 * k);
    }
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(nbn), dim3(256), 0, (hipStream_t)stream, table, counts, bnbuf,
                       bufs, eps, momentum, align, groups, packed);
    return sv_check_launch("sv_bn_running_update");
}

int sv_bn_running_update(const int32_t* table, const float* counts, int nbn, const float* bnbuf, float* bufs,
                         float eps, float momentum, int align, int groups, void* stream) {
    return sv_bn_running_update_ex(table, counts, nbn, bnbuf, bufs, eps, momentum, align, groups, nullptr, stream);
}

int sv_bn_eval_affine(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                      float* scale, float* shift, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(gamma && beta && rm && rv && scale && shift && C > 0, SV_E_ARG, "sv_bn_eval_affine: null");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma,
                       beta, rm, rv, eps, scale, shift);
    return sv_check_launch("sv_bn_eval_affine");
}

// deterministic mode: P zeroed slots of `outer * n` floats for the blocks of a first pass / the ordered second pass over them
static float* det_slots(int P, int n, int outer, hipStream_t s) {
    float* w = sv_det_scratch((size_t)P * n * outer);
    if (w && hipMemsetAsync(w, 0, (size_t)P * n * outer * sizeof(float), s) != hipSuccess) {
        sv_set_error("deterministic mode: clearing the scratch slots failed");
        return nullptr;
    }
    return w;
}
static void det_collect(const float* part, int P, int n, int outer, float* out, hipStream_t s) {
    hipLaunchKernelGGL((det_collect_kernel<float>), dim3((n + 255) / 256, outer), dim3(256), 0, s, part, P, n, out);
}
static void det_collect(const float* part, int P, int n, int outer, double* out, hipStream_t s) {
    hipLaunchKernelGGL((det_collect_kernel<double>), dim3((n + 255) / 256, outer), dim3(256), 0, s, part, P, n, out);
}

int sv_bn_bwd_apply(int dtype, int64_t M, int C, int ld, const void* x, const float* mean, const float* rstd,
                    float count, const sv_bn_branch* br, int nbranch, const void* residual, void* dx,
                    int groups, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(x && mean && rstd && br && dx && nbranch >= 1 && nbranch <= 2, SV_E_ARG, "sv_bn_bwd_apply: bad args");
    SV_REQUIRE(C % 8 == 0 && ld % 8 == 0, SV_E_SHAPE, "sv_bn_bwd_apply: C=%d ld=%d must be multiples of 8", C, ld);
    bnb_params p;
    p.M = M; p.C = C; p.ld = ld; p.nbranch = nbranch; p.x = x; p.mean = mean; p.rstd = rstd;
    p.inv_count = 1.f / count; p.residual = residual; p.dx = dx;
    for (int k = 0; k < nbranch; ++k) {
        p.br[k] = br[k];
        SV_REQUIRE(br[k].g && br[k].bsums && br[k].gamma && br[k].replicas >= 1, SV_E_ARG,
                   "sv_bn_bwd_apply: branch %d incomplete", k);
        {
            const int sa = br[k].sparse < 0 ? -br[k].sparse : br[k].sparse;
            SV_REQUIRE(sa <= 16 && (sa == 0 || M % ((int64_t)1 << (2 * (sa - 1))) == 0) && (br[k].sparse >= 0 || sa >= 2),
                       SV_E_ARG, "sv_bn_bwd_apply: branch %d sparse=%d does not match M", k, br[k].sparse);
        }
    }
    groups = sv_ngroups(groups);
    SV_REQUIRE(groups <= SV_MAX_GROUPS, SV_E_ARG, "sv_bn_bwd_apply: groups=%d (at most %d)", groups, SV_MAX_GROUPS);
    // every block first sums the accumulator replicas into its coefficients (a few us of latency): the groups of a batched
    // launch share the block budget, so that this prologue stays amortised over as many rows per block
    // One resident wave of blocks over the whole (batched) launch: the register-coefficient variant holds 93 registers = 5 blocks
    // of 256 threads per CU (the 2 048 blocks of round 2 were 1.6 waves: the second one ran 60 % full), the LDS-coefficient
    // variant 8.  The occupancy is asked once per variant.
    static int occ_reg = 0, occ_lds = 0;
    if (!occ_reg) {
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, bn_bwd_apply_kernel<bf16, true>, 256, 16 * 1024) != hipSuccess || o < 1) o = 5;
        occ_reg = o;
        o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, bn_bwd_apply_kernel<bf16, false>, 256, 16 * 1024) != hipSuccess || o < 1) o = 8;
        occ_lds = o;
        (void)hipGetLastError();
    }
    const int cv_ = C / 8;
    const bool reg_ = (256 % cv_ == 0 ? 256 : (cv_ <= 256 ? 256 / cv_ * cv_ : 0)) >= 192;
    const int resident = 256 * (reg_ ? occ_reg : occ_lds);
    const int grid = nblocks(M * (C / 8), 256, resident / groups > 64 ? resident / groups : 64);
    const size_t lds = ((size_t)(2 + 3 * nbranch) * C + (size_t)nbranch * 2 * 256 * 2) * sizeof(float);      // (the partial sums are doubles)
    SV_REQUIRE(lds <= 64 * 1024, SV_E_SHAPE, "sv_bn_bwd_apply: C=%d too large", C);
    const int cv = C / 8;
    // every thread keeps one 8-channel group (coefficients in registers) when the block size is a multiple of C/8: 256
    // threads for the power-of-two widths, 240 for the 160 / 320 / 640-channel tensors of WRN-28-10 (C/8 = 20, 40, 80) -- those
    // took the LDS-coefficient path at 3.5 TB/s
    const int nthr = 256 % cv == 0 ? 256 : (cv <= 256 ? 256 / cv * cv : 0);
    const bool reg = nthr >= 192 && (int64_t)grid * nthr >= cv;
    if (sv_det_stats()) {
        // dgamma / dbeta receive ONE add per launch: block (0, 0) walks the groups in index order (mode 1 only).  The accumulators
        // of these modes hold a replica per producer wave: a pre-pass folds them 256 : 1 (in order) so that the blocks of the apply
        // kernel do not each walk thousands of rows
        for (int k = 0; k < nbranch; ++k) {
            const int R = p.br[k].replicas;
            if (R <= 64) continue;
            const int R2 = (R + 255) / 256;
            double* w = reinterpret_cast<double*>(sv_det_scratch((size_t)2 * groups * R2 * 2 * C));
            if (!w) return SV_E_HIP;
            hipLaunchKernelGGL(replica_fold_kernel, dim3((2 * C + 31) / 32, R2, groups), dim3(256), 0, (hipStream_t)stream,
                               p.br[k].bsums, R, 2 * C, w);
            p.br[k].bsums = w;
            p.br[k].replicas = R2;
        }
        bnb_params_g A;
        DISPATCH_T(dtype, A = bnb_expand(p, groups, (int)sizeof(T)));
        A.det_groups = sv_deterministic() ? groups : 0;
        if (reg) DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(grid, groups), dim3(nthr), lds, (hipStream_t)stream, A));
        else DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(grid, groups), dim3(256), lds, (hipStream_t)stream, A));
        return sv_check_launch("sv_bn_bwd_apply");
    }
    if (reg) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(grid, groups), dim3(nthr), lds, (hipStream_t)stream, bnb_expand(p, groups, (int)sizeof(T))));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(grid, groups), dim3(256), lds, (hipStream_t)stream, bnb_expand(p, groups, (int)sizeof(T))));
    }
    return sv_check_launch("sv_bn_bwd_apply");
}

int sv_bn_bwd_affine(const double* bsums, int replicas, int C, float count, const float* gamma, const float* mean, const float* rstd,
                     float* dgamma, float* dbeta, float* scale_g, float* scale_x, float* shift, int groups, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(bsums && gamma && mean && rstd && scale_g && scale_x && shift && replicas >= 1 && count > 0.f && C >= 1, SV_E_ARG,
               "sv_bn_bwd_affine: bad argument");
    groups = sv_ngroups(groups);
    SV_REQUIRE(groups <= SV_MAX_GROUPS, SV_E_ARG, "sv_bn_bwd_affine: groups=%d (at most %d)", groups, SV_MAX_GROUPS);
    hipLaunchKernelGGL(bn_bwd_affine_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, (hipStream_t)stream, bsums, replicas, C,
                       1.f / count, gamma, mean, rstd, dgamma, dbeta, scale_g, scale_x, shift);
    return sv_check_launch("sv_bn_bwd_affine");
}

int sv_bn_act(int dtype, const void* x, const float* scale, const float* shift, float slope, int64_t M, int C, void* out,
              int groups, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(x && scale && shift && out && M > 0 && C > 0, SV_E_ARG, "sv_bn_act: bad argument");
    SV_REQUIRE(C % 8 == 0, SV_E_SHAPE, "sv_bn_act: C=%d must be a multiple of 8", C);
    SV_REQUIRE(slope >= 0.f && slope <= 1.f, SV_E_ARG, "sv_bn_act: activation slope %g outside [0, 1]", (double)slope);
    groups = sv_ngroups(groups);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_act_kernel<T>), dim3(nblocks(M * C / 8, 256, 1024), groups), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)x, scale, shift, slope, M, C, (T*)out));
    return sv_check_launch("sv_bn_act");
}

int sv_colsum(int dtype, const void* y, int64_t M, int N, int ld, float* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(y && out && N > 0, SV_E_ARG, "sv_colsum: null");
    if (sv_deterministic()) {       // row ranges in a fixed order inside a block into the block's slot; the slots in order
        const int bx = N < 64 ? N : 64;
        SV_REQUIRE(256 % bx == 0, SV_E_SHAPE, "sv_colsum: N=%d", N);
        const int P = nblocks(M, 1024, 256);
        float* w = P > 1 ? det_slots(P, N, 1, (hipStream_t)stream) : nullptr;
        if (P > 1 && !w) return SV_E_HIP;
        DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(P, (N + bx - 1) / bx), dim3(bx, 256 / bx), 0, (hipStream_t)stream,
                                             (const T*)y, M, N, ld, P > 1 ? w : out, P > 1 ? N : 0));
        if (P > 1) det_collect(w, P, N, 1, out, (hipStream_t)stream);
        return sv_check_launch("sv_colsum");
    }
    if (N % 8 == 0 && ld % 8 == 0 && N <= 2048 && M >= 4096) {
        const int rpp = 256 / (N / 8);
        DISPATCH_T(dtype, hipLaunchKernelGGL((colsum8_kernel<T>), dim3(nblocks(M, rpp * 8, 512)), dim3(256), N * sizeof(float),
                                             (hipStream_t)stream, (const T*)y, M, N, ld, out));
        return sv_check_launch("sv_colsum");
    }
    const int bx = N < 64 ? N : 64;
    SV_REQUIRE(256 % bx == 0, SV_E_SHAPE, "sv_colsum: N=%d", N);
    dim3 block(bx, 256 / bx);
    // few rows (the bias gradients of the linear layers of the smooth-ELBO trainers: M = the batch): 128 rows per block instead
    // of 1 024 -- two blocks walking 256 rows per thread took 44 us for a 2 048 x 256 tensor
    dim3 grid(nblocks(M, M >= 65536 ? 1024 : 128, 512), (N + bx - 1) / bx);
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, block, 0, (hipStream_t)stream, (const T*)y, M, N, ld, out));
    return sv_check_launch("sv_colsum");
}

int sv_pool_fwd(int dtype, const void* x, const float* scale, const float* shift, float slope, int B, int HW,
                int C, int ld, float* feat, int groups, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(x && scale && shift && feat, SV_E_ARG, "sv_pool_fwd: null");
    SV_REQUIRE(B % sv_ngroups(groups) == 0, SV_E_ARG, "sv_pool_fwd: B=%d is not a multiple of groups=%d", B, groups);
    SV_REQUIRE(slope >= 0.f && slope <= 1.f, SV_E_ARG, "sv_pool_fwd: activation slope %g outside [0, 1]", (double)slope);
    {
        const int Bg = B / sv_ngroups(groups), cv = C / 8;
        int P = POOL_PARTS;
        while (P > 1 && cv * P > 256) P >>= 1;
        const int ipb = 256 / (cv * P) > 0 ? 256 / (cv * P) : 1;
        if (C % 8 == 0 && ld % 8 == 0 && cv <= 256 && Bg % ipb == 0 && HW >= P) {
            const size_t lds = (size_t)ipb * P * C * sizeof(float);
            DISPATCH_T(dtype, hipLaunchKernelGGL((pool_fwd8_kernel<T>), dim3((B + ipb - 1) / ipb), dim3(256), lds, (hipStream_t)stream,
                                                 (const T*)x, scale, shift, slope, B, HW, C, ld, feat, Bg, P));
            return sv_check_launch("sv_pool_fwd");
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((pool_fwd_kernel<T>), dim3((B * C + 255) / 256), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, scale, shift, slope, B, HW, C, ld, feat,
                                         B / sv_ngroups(groups)));
    return sv_check_launch("sv_pool_fwd");
}

int sv_pool_bwd(int dtype, const void* x, const float* scale, const float* shift, float slope, const float* mean,
                const float* rstd, const float* dfeat, int B, int HW, int C, int ld, void* g, double* bsums,
                int groups, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(x && scale && shift && mean && rstd && dfeat && g && bsums, SV_E_ARG, "sv_pool_bwd: null");
    SV_REQUIRE(B % sv_ngroups(groups) == 0, SV_E_ARG, "sv_pool_bwd: B=%d is not a multiple of groups=%d", B, groups);
    if (sv_det_stats() && C % 8 == 0 && ld % 8 == 0 && C / 8 <= 256) {
        const int cv = C / 8, lanes = 256 / cv, G = sv_ngroups(groups), Bg = B / G;
        int P = Bg / lanes;                    // blocks per group: at least one image per image lane
        if (P > 64) P = 64;
        if (P < 1) P = 1;
        float* w = det_slots(P, 2 * C, G, (hipStream_t)stream);       // (fp32 slots, collected in index order into the double accumulator)
        if (!w) return SV_E_HIP;
        DISPATCH_T(dtype, hipLaunchKernelGGL((pool_bwd_det_kernel<T>), dim3(G, P), dim3(256), (size_t)lanes * 2 * C * sizeof(float),
                                             (hipStream_t)stream, (const T*)x, scale, shift, slope, mean, rstd, dfeat, Bg,
                                             HW, C, ld, (T*)g, w));
        det_collect(w, P, 2 * C, G, bsums, (hipStream_t)stream);
        return sv_check_launch("sv_pool_bwd");
    }
    {
        const int Bg = B / sv_ngroups(groups), cv = C / 8;
        int P = POOL_PARTS;
        while (P > 1 && cv * P > 256) P >>= 1;
        const int ipb = 256 / (cv * P) > 0 ? 256 / (cv * P) : 1;
        if (C % 8 == 0 && ld % 8 == 0 && cv <= 256 && Bg % ipb == 0 && HW >= P) {
            DISPATCH_T(dtype, hipLaunchKernelGGL((pool_bwd8_kernel<T>), dim3((B + ipb - 1) / ipb), dim3(256), 2 * C * sizeof(double),
                                                 (hipStream_t)stream, (const T*)x, scale, shift, slope, mean, rstd, dfeat, B, HW, C, ld,
                                                 (T*)g, bsums, Bg, P));
            return sv_check_launch("sv_pool_bwd");
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((pool_bwd_kernel<T>), dim3((B * C + 255) / 256), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, scale, shift, slope, mean, rstd, dfeat,
                                         B, HW, C, ld, (T*)g, bsums, B / sv_ngroups(groups)));
    return sv_check_launch("sv_pool_bwd");
}

int sv_head_fwd(const float* feat, int B, int C, const float* W, const float* bias, int ldc, int K, float* mu,
                float* ls, float* la, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(feat && W && bias && mu && ls && la, SV_E_ARG, "sv_head_fwd: null");
    const size_t lds = (size_t)HS * (C + 2 * ldc + K) * sizeof(float);
    SV_REQUIRE(lds <= 64 * 1024, SV_E_SHAPE, "sv_head_fwd: C=%d too large", C);
    hipLaunchKernelGGL(head_fwd_kernel, dim3((B + HS - 1) / HS), dim3(256), lds, (hipStream_t)stream, feat, B, C, W,
                       bias, ldc, K, mu, ls, la);
    return sv_check_launch("sv_head_fwd");
}

int sv_head_bwd(const float* feat, int B, int C, const float* W, int ldc, int K, const float* la,
                   const float* dmu, const float* dls, const float* dla, float* dfeat, float* dW, float* dbias,
                   float* dout_ws, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(feat && W && la && dmu && dls && dla && dfeat && dW && dbias && dout_ws, SV_E_ARG, "sv_head_bwd: null");
    const int NH = 2 * ldc + K;
    const size_t lds = ((size_t)HS * NH + (size_t)HS * 256) * sizeof(float);       // gradients + the thread groups' partial sums
    hipLaunchKernelGGL(head_bwd_data_kernel, dim3((B + HS - 1) / HS, (C + 255) / 256), dim3(256), lds, (hipStream_t)stream, B, C, W,
                       ldc, K, la, dmu, dls, dla, dfeat, dout_ws);
    {
        const int tiles = ((NH + 31) / 32) * ((C + 31) / 32);
        const int slices = (B + HWS - 1) / HWS;
        if (sv_deterministic()) {
            for (int sl = 0; sl < slices; ++sl)
                hipLaunchKernelGGL(head_bwd_weight_kernel, dim3(tiles, 1), dim3(64), 0, (hipStream_t)stream, feat, dout_ws, B, C,
                                   NH, dW, dbias, sl);
        } else {
            hipLaunchKernelGGL(head_bwd_weight_kernel, dim3(tiles, slices), dim3(64), 0, (hipStream_t)stream, feat, dout_ws, B, C,
                               NH, dW, dbias, 0);
        }
    }
    return sv_check_launch("sv_head_bwd");
}

int sv_sample_fwd(int dtype, const float* mu, const float* ls, const float* la, const float* eps, const float* u,
                  const int64_t* label, const int64_t* label_mix, float lam, const float* lam_dev, int mode,
                  float temperature, int B,
                  int ldc, int K, int Lpad, void* latent, float* csoft, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(mu && ls && la && eps && latent && csoft, SV_E_ARG, "sv_sample_fwd: null");
    SV_REQUIRE((mode == 0 && u) || (mode == 1 && label) || (mode == 2 && label && label_mix), SV_E_ARG,
               "sv_sample_fwd: mode %d inputs missing", mode);
    SV_REQUIRE(Lpad >= ldc + K, SV_E_SHAPE, "sv_sample_fwd: Lpad");
    DISPATCH_T(dtype, hipLaunchKernelGGL((sample_fwd_kernel<T>), dim3(B), dim3(128), 0, (hipStream_t)stream, mu, ls,
                                         la, eps, u, label, label_mix, lam, lam_dev, mode, temperature, B, ldc, K, Lpad,
                                         (T*)latent, csoft));
    return sv_check_launch("sv_sample_fwd");
}

int sv_sample_bwd(int dtype, const void* dlatent, const float* ls, const float* eps, const float* csoft, int mode,
                  float temperature, int B, int ldc, int K, int Lpad, float* dmu, float* dls, float* dla,
                  void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(dlatent && ls && eps && csoft && dmu && dls && dla, SV_E_ARG, "sv_sample_bwd: null");
    DISPATCH_T(dtype, hipLaunchKernelGGL((sample_bwd_kernel<T>), dim3(B), dim3(128), 0, (hipStream_t)stream,
                                         (const T*)dlatent, ls, eps, csoft, mode, temperature, B, ldc, K, Lpad, dmu,
                                         dls, dla));
    return sv_check_launch("sv_sample_bwd");
}

static float log_prior_f32(int K) {
    // the reference builds log(float32(1/K)) in float32 (lib/criterion.py:29-30)
    const float p = (float)(1.0 / (double)K);
    return logf(p);
}

int sv_elbo_fwd(const float* x, const float* x_rec, int64_t n_per_img, const float* mu, const float* ls,
                const float* la, int B, int ldc, int K, int bce, float x_sigma, float* out3, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(x && x_rec && mu && ls && la && out3, SV_E_ARG, "sv_elbo_fwd: null");
    const int64_t n = n_per_img * B;
    // (three float atomics per block on the same three addresses: 256 blocks, not 1 024 -- the tail of serialised atomics
    //  was most of this kernel's 26 us)
    const int P = nblocks(n / 4, 1024, 256);
    float* w = nullptr;
    if (sv_deterministic() && P > 1 && !(w = det_slots(P, 3, 1, (hipStream_t)stream))) return SV_E_HIP;
    hipLaunchKernelGGL(elbo_fwd_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, x, x_rec,
                       n, mu, ls, la, B, ldc, K, bce, x_sigma, log_prior_f32(K), w ? w : out3, w ? 3 : 0);
    if (w) det_collect(w, P, 3, 1, out3, (hipStream_t)stream);
    return sv_check_launch("sv_elbo_fwd");
}

int sv_elbo_bwd(const float* x, const float* x_rec, int64_t n_per_img, const float* mu, const float* ls,
                const float* la, int B, int ldc, int K, int bce, float x_sigma, const float* gout3, float* dx_rec,
                float* dmu, float* dls, float* dla, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(x && x_rec && mu && ls && la && gout3 && dx_rec && dmu && dls && dla, SV_E_ARG, "sv_elbo_bwd: null");
    const int64_t n = n_per_img * B;
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3(nblocks(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, x, x_rec, n,
                       mu, ls, la, B, ldc, K, bce, x_sigma, log_prior_f32(K), gout3, dx_rec, dmu, dls, dla);
    return sv_check_launch("sv_elbo_bwd");
}

int sv_cls_fwd(const float* predict, const float* label, const float* weight, int B, int K, float* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(predict && label && out, SV_E_ARG, "sv_cls_fwd: null");
    const int P = nblocks((int64_t)B * K, 256, 64);
    float* w = nullptr;
    if (sv_deterministic() && P > 1 && !(w = det_slots(P, 1, 1, (hipStream_t)stream))) return SV_E_HIP;
    hipLaunchKernelGGL(cls_fwd_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, predict, label, weight, B, K, w ? w : out, w ? 1 : 0);
    if (w) det_collect(w, P, 1, 1, out, (hipStream_t)stream);
    return sv_check_launch("sv_cls_fwd");
}
int sv_cls_bwd(const float* label, const float* weight, int B, int K, const float* gout, float* dpredict, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(label && gout && dpredict, SV_E_ARG, "sv_cls_bwd: null");
    hipLaunchKernelGGL(cls_bwd_kernel, dim3(((int64_t)B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, label,
                       weight, B, K, gout, dpredict);
    return sv_check_launch("sv_cls_bwd");
}
int sv_topk_hits(const float* score, const int64_t* label, int B, int K, int k, float* hits, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(score && label && hits && B >= 0 && K >= 1 && k >= 1, SV_E_ARG, "sv_topk_hits: bad argument");
    if (B == 0) return SV_OK;
    hipLaunchKernelGGL(topk_hits_kernel, dim3(nblocks(B, 256, 64)), dim3(256), 0, (hipStream_t)stream, score, label, B, K,
                       k, hits);
    return sv_check_launch("sv_topk_hits");
}
int sv_post_fwd(const float* mu, const float* ls, const float* mu_t, const float* sigma_t, int B, int D, float* out,
                void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(mu && ls && mu_t && sigma_t && out, SV_E_ARG, "sv_post_fwd: null");
    const int P = nblocks((int64_t)B * D, 256, 64);
    float* w = nullptr;
    if (sv_deterministic() && P > 1 && !(w = det_slots(P, 1, 1, (hipStream_t)stream))) return SV_E_HIP;
    hipLaunchKernelGGL(post_fwd_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, mu, ls, mu_t, sigma_t, B, D, w ? w : out, w ? 1 : 0);
    if (w) det_collect(w, P, 1, 1, out, (hipStream_t)stream);
    return sv_check_launch("sv_post_fwd");
}
int sv_post_bwd(const float* mu, const float* ls, const float* mu_t, const float* sigma_t, int B, int D,
                const float* gout, float* dmu, float* dls, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(mu && ls && mu_t && sigma_t && gout && dmu && dls, SV_E_ARG, "sv_post_bwd: null");
    hipLaunchKernelGGL(post_bwd_kernel, dim3(((int64_t)B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, mu, ls,
                       mu_t, sigma_t, B, D, gout, dmu, dls);
    return sv_check_launch("sv_post_bwd");
}

int sv_rank_permutation(const float* keys, int n, int batches, int64_t* perm, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(keys && perm && n > 0 && batches > 0, SV_E_ARG, "sv_rank_permutation: bad argument");
    SV_REQUIRE(n <= 16384, SV_E_SHAPE, "sv_rank_permutation: n=%d (at most 16384 keys per permutation)", n);
    hipLaunchKernelGGL(rank_perm_kernel, dim3((n + 255) / 256, batches), dim3(256), n * sizeof(float), (hipStream_t)stream, keys, n, perm);
    return sv_check_launch("sv_rank_permutation");
}

int sv_shot_targets(const float* mu_l, const float* ls_l, const float* mu_u, const float* ls_u, const float* la_u,
                    const int64_t* label_l, const int64_t* perm_l, const int64_t* perm_u, float lam_l, const float* lam_l_dev,
                    float lam_u, const float* lam_u_dev, int B, int D, int K, float* sm_mu, float* sm_sigma, float* lab_mix,
                    float* mx_mu, float* mx_sigma, float* mx_alpha, void* stream) {
    return sv_shot_targets2(mu_l, ls_l, mu_u, ls_u, la_u, label_l, perm_l, perm_u, lam_l, lam_l_dev, lam_u, lam_u_dev, B, B, D, K, sm_mu,
                            sm_sigma, lab_mix, mx_mu, mx_sigma, mx_alpha, stream);
}

int sv_shot_targets2(const float* mu_l, const float* ls_l, const float* mu_u, const float* ls_u, const float* la_u,
                     const int64_t* label_l, const int64_t* perm_l, const int64_t* perm_u, float lam_l, const float* lam_l_dev,
                     float lam_u, const float* lam_u_dev, int Bl, int Bu, int D, int K, float* sm_mu, float* sm_sigma, float* lab_mix,
                     float* mx_mu, float* mx_sigma, float* mx_alpha, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(mu_l && ls_l && mu_u && ls_u && la_u && label_l && perm_l && perm_u && sm_mu && sm_sigma && lab_mix && mx_mu &&
               mx_sigma && mx_alpha && Bl > 0 && Bu > 0 && D > 0 && K > 0, SV_E_ARG, "sv_shot_targets: bad argument");
    hipLaunchKernelGGL(shot_targets_kernel, dim3(Bl > Bu ? Bl : Bu), dim3(128), 0, (hipStream_t)stream, mu_l, ls_l, mu_u, ls_u, la_u, label_l,
                       perm_l, perm_u, lam_l, lam_l_dev, lam_u, lam_u_dev, Bl, Bu, D, K, sm_mu, sm_sigma, lab_mix, mx_mu, mx_sigma,
                       mx_alpha);
    return sv_check_launch("sv_shot_targets");
}

int sv_shot_compose(float* terms, const sv_shot_schedule* sch, float* coef, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(terms && sch && coef, SV_E_ARG, "sv_shot_compose: null");
    hipLaunchKernelGGL(shot_compose_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, terms, *sch, coef);
    return sv_check_launch("sv_shot_compose");
}

int sv_shot_scale(const float* coef, const float* g_sup, const float* g_unsup, float* gvec, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(coef && gvec, SV_E_ARG, "sv_shot_scale: null");
    hipLaunchKernelGGL(shot_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, coef, g_sup, g_unsup, gvec);
    return sv_check_launch("sv_shot_scale");
}

int sv_shot_loss_step(const sv_shot_loss_args* a, void* stream) {
    SV_REQUIRE(a && a->rec && a->mu && a->ls && a->la && a->d_rec && a->d_mu && a->d_ls && a->d_la && a->B > 0 && a->D > 0 && a->K > 0,
               SV_E_ARG, "sv_shot_loss_step: bad argument");
    // the four groups back to back, B rows each, in the order (1)(3)(2)(4) -> the per-group form
    sv_shot_loss_args2 b{};
    const int64_t B = a->B, n = a->n_per_img;
    for (int g = 0; g < 4; ++g) {
        b.mu[g] = a->mu + g * B * a->D;  b.ls[g] = a->ls + g * B * a->D;  b.la[g] = a->la + g * B * a->K;
        b.d_mu[g] = a->d_mu + g * B * a->D;  b.d_ls[g] = a->d_ls + g * B * a->D;  b.d_la[g] = a->d_la + g * B * a->K;
    }
    for (int g = 0; g < 2; ++g) { b.rec[g] = a->rec + g * B * n;  b.d_rec[g] = a->d_rec + g * B * n; }
    b.image_l = a->image_l; b.image_u = a->image_u; b.label_l = a->label_l; b.perm_l = a->perm_l; b.perm_u = a->perm_u;
    b.lam_l = a->lam_l; b.lam_l_dev = a->lam_l_dev; b.lam_u = a->lam_u; b.lam_u_dev = a->lam_u_dev;
    b.Bl = b.Bu = a->B; b.D = a->D; b.K = a->K; b.bce = a->bce; b.n_per_img = n; b.x_sigma = a->x_sigma; b.sch = a->sch;
    b.terms = a->terms; b.coef = a->coef; b.tgt = a->tgt;
    return sv_shot_loss_step2(&b, stream);
}

int sv_shot_loss_step2(const sv_shot_loss_args2* a, void* stream) {
    SV_REQUIRE(a && a->image_l && a->image_u && a->label_l && a->perm_l && a->perm_u && a->terms && a->coef && a->tgt && a->Bl > 0 &&
               a->Bu > 0 && a->D > 0 && a->K > 0, SV_E_ARG, "sv_shot_loss_step2: bad argument");
    for (int g = 0; g < 4; ++g)
        SV_REQUIRE(a->mu[g] && a->ls[g] && a->la[g] && a->d_mu[g] && a->d_ls[g] && a->d_la[g], SV_E_ARG, "sv_shot_loss_step2: group %d: null", g);
    SV_REQUIRE(a->rec[0] && a->rec[1] && a->d_rec[0] && a->d_rec[1], SV_E_ARG, "sv_shot_loss_step2: reconstruction: null");
    const int Bl = a->Bl, Bu = a->Bu, D = a->D, K = a->K;
    const int64_t n = a->n_per_img;
    // targets: sm_mu | sm_sigma [Bl][D], mx_mu | mx_sigma [Bu][D], lab_mix [Bl][K], mx_alpha [Bu][K]
    float* sm_mu = a->tgt;
    float* sm_sigma = sm_mu + (int64_t)Bl * D;
    float* mx_mu = sm_sigma + (int64_t)Bl * D;
    float* mx_sigma = mx_mu + (int64_t)Bu * D;
    float* lab_mix = mx_sigma + (int64_t)Bu * D;
    float* mx_alpha = lab_mix + (int64_t)Bl * K;
    int rc;
#define SV_TRY(call) do { rc = (call); if (rc != SV_OK) return rc; } while (0)
    if (!sv_deterministic()) {
        // three launches for the twelve reduction / gradient kernels of the stage (+ targets and composition)
        SvProfScope prof_scope(stream);
        shot_parts P;
        P.n_per_img = n; P.D = D; P.K = K; P.bce = a->bce; P.x_sigma = a->x_sigma; P.log_prior = log_prior_f32(K);
        const float* img[2] = {a->image_l, a->image_u};
        const int Bs[2] = {Bl, Bu};
        for (int i = 0; i < 2; ++i) {
            shot_elbo_part& e = P.e[i];
            e.x = img[i]; e.xr = a->rec[i]; e.mu = a->mu[i]; e.ls = a->ls[i]; e.la = a->la[i]; e.B = Bs[i]; e.out3 = a->terms + 3 * i;
            e.gout3 = a->coef + 3 * i; e.dxr = a->d_rec[i]; e.dmu = a->d_mu[i]; e.dls = a->d_ls[i]; e.dla = a->d_la[i];
            shot_post_part& q = P.q[i];
            q.la = a->la[2 + i]; q.label = i ? mx_alpha : lab_mix; q.mu = a->mu[2 + i]; q.ls = a->ls[2 + i];
            q.mt = i ? mx_mu : sm_mu; q.st = i ? mx_sigma : sm_sigma; q.B = Bs[i];
            q.out_cls = a->terms + 6 + 2 * i; q.out_post = a->terms + 7 + 2 * i; q.g_cls = a->coef + 6 + 2 * i; q.g_post = a->coef + 7 + 2 * i;
            q.dla = a->d_la[2 + i]; q.dmu = a->d_mu[2 + i]; q.dls = a->d_ls[2 + i];
        }
        const int Bmax = Bl > Bu ? Bl : Bu;
        hipStream_t s_ = (hipStream_t)stream;
        hipLaunchKernelGGL(shot_elbo_fwd2_kernel, dim3(nblocks(n * Bmax / 4, 1024, 256), 2), dim3(256), 0, s_, P);
        SV_TRY(sv_shot_targets2(a->mu[0], a->ls[0], a->mu[1], a->ls[1], a->la[1], a->label_l, a->perm_l, a->perm_u,
                                a->lam_l, a->lam_l_dev, a->lam_u, a->lam_u_dev, Bl, Bu, D, K, sm_mu, sm_sigma, lab_mix, mx_mu, mx_sigma, mx_alpha, stream));
        hipLaunchKernelGGL(shot_post_fwd4_kernel, dim3(nblocks((int64_t)Bmax * D, 256, 64), 4), dim3(256), 0, s_, P);
        SV_TRY(sv_shot_compose(a->terms, &a->sch, a->coef, stream));
        int gx = nblocks(n * Bmax, 256, 2048);                 // (the cls / posterior parts index their elements directly)
        const int need = (int)(((int64_t)Bmax * (D > K ? D : K) + 255) / 256);
        if (gx < need) gx = need;
        hipLaunchKernelGGL(shot_bwd6_kernel, dim3(gx, 6), dim3(256), 0, s_, P);
        return sv_check_launch("sv_shot_loss_step2");
    }
    // forward: ELBO terms of (1), (3); the targets of (2), (4); their posterior terms; the composition
    SV_TRY(sv_elbo_fwd(a->image_l, a->rec[0], n, a->mu[0], a->ls[0], a->la[0], Bl, D, K, a->bce, a->x_sigma, a->terms, stream));
    SV_TRY(sv_elbo_fwd(a->image_u, a->rec[1], n, a->mu[1], a->ls[1], a->la[1], Bu, D, K, a->bce, a->x_sigma, a->terms + 3, stream));
    SV_TRY(sv_shot_targets2(a->mu[0], a->ls[0], a->mu[1], a->ls[1], a->la[1], a->label_l, a->perm_l, a->perm_u,
                            a->lam_l, a->lam_l_dev, a->lam_u, a->lam_u_dev, Bl, Bu, D, K, sm_mu, sm_sigma, lab_mix, mx_mu, mx_sigma, mx_alpha, stream));
    SV_TRY(sv_cls_fwd(a->la[2], lab_mix, nullptr, Bl, K, a->terms + 6, stream));
    SV_TRY(sv_post_fwd(a->mu[2], a->ls[2], sm_mu, sm_sigma, Bl, D, a->terms + 7, stream));
    SV_TRY(sv_cls_fwd(a->la[3], mx_alpha, nullptr, Bu, K, a->terms + 8, stream));
    SV_TRY(sv_post_fwd(a->mu[3], a->ls[3], mx_mu, mx_sigma, Bu, D, a->terms + 9, stream));
    SV_TRY(sv_shot_compose(a->terms, &a->sch, a->coef, stream));
    // backward with upstream gradients 1: the coefficients are the `gout` operands; every slice is written once
    SV_TRY(sv_elbo_bwd(a->image_l, a->rec[0], n, a->mu[0], a->ls[0], a->la[0], Bl, D, K, a->bce, a->x_sigma, a->coef,
                       a->d_rec[0], a->d_mu[0], a->d_ls[0], a->d_la[0], stream));
    SV_TRY(sv_elbo_bwd(a->image_u, a->rec[1], n, a->mu[1], a->ls[1], a->la[1], Bu, D, K, a->bce, a->x_sigma, a->coef + 3,
                       a->d_rec[1], a->d_mu[1], a->d_ls[1], a->d_la[1], stream));
    SV_TRY(sv_cls_bwd(lab_mix, nullptr, Bl, K, a->coef + 6, a->d_la[2], stream));
    SV_TRY(sv_post_bwd(a->mu[2], a->ls[2], sm_mu, sm_sigma, Bl, D, a->coef + 7, a->d_mu[2], a->d_ls[2], stream));
    SV_TRY(sv_cls_bwd(mx_alpha, nullptr, Bu, K, a->coef + 8, a->d_la[3], stream));
    SV_TRY(sv_post_bwd(a->mu[3], a->ls[3], mx_mu, mx_sigma, Bu, D, a->coef + 9, a->d_mu[3], a->d_ls[3], stream));
#undef SV_TRY
    return SV_OK;
}

int sv_mix_lerp(const float* a, const int64_t* index, float lam, const float* lam_dev, int B, int64_t row,
                int exp_space, float* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(a && index && out, SV_E_ARG, "sv_mix_lerp: null");
    if (B == 0 || row == 0) return SV_OK;
    hipLaunchKernelGGL(mix_lerp_kernel, dim3((unsigned)((row + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, a,
                       index, lam, lam_dev, B, row, exp_space, out);
    return sv_check_launch("sv_mix_lerp");
}

int sv_optimal_match(const float* mu, const float* ls, int B, int D, int64_t* index, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(mu && ls && index && B >= 2, SV_E_ARG, "sv_optimal_match: bad args");
    const size_t lds = (size_t)(2 * D + B) * sizeof(float);
    SV_REQUIRE(lds <= 64 * 1024, SV_E_SHAPE, "sv_optimal_match: B=%d D=%d too large", B, D);
    hipLaunchKernelGGL(optimal_match_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, mu, ls, B, D, index);
    return sv_check_launch("sv_optimal_match");
}

int sv_smooth_latent_fwd(int dtype, const void* o, int ldo, const float* eps, const float* u, const int64_t* label,
                         float temperature, int training, int B, int Dc, int Dd, int Lpad, float* mean, float* logvar,
                         float* alpha, float* gs, void* latent, float* latent32, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(o && mean && logvar && alpha && gs && latent && latent32 && B > 0 && Dc > 0 && Dd > 0, SV_E_ARG, "sv_smooth_latent_fwd: bad argument");
    SV_REQUIRE(!training || (eps && u), SV_E_ARG, "sv_smooth_latent_fwd: training mode needs the noise");
    SV_REQUIRE(ldo >= 2 * Dc + Dd && Lpad >= Dc + Dd && temperature > 0.f, SV_E_SHAPE, "sv_smooth_latent_fwd: ldo=%d Lpad=%d", ldo, Lpad);
    DISPATCH_T(dtype, hipLaunchKernelGGL((smooth_latent_fwd_kernel<T>), dim3(B), dim3(64), 0, (hipStream_t)stream, (const T*)o, ldo, eps, u,
                                         label, temperature, training, Dc, Dd, Lpad, mean, logvar, alpha, gs, (T*)latent, latent32));
    return sv_check_launch("sv_smooth_latent_fwd");
}

int sv_smooth_latent_bwd(int dtype, const void* dlat, int Lpad, const float* dmean, const float* dlogvar, const float* dalpha,
                         const float* logvar, const float* eps, const float* alpha, const float* gs, float temperature,
                         int training, int sample_path, int B, int Dc, int Dd, void* d_o, int ldo, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(dlat && logvar && alpha && gs && d_o && B > 0, SV_E_ARG, "sv_smooth_latent_bwd: bad argument");
    SV_REQUIRE(!training || eps, SV_E_ARG, "sv_smooth_latent_bwd: training mode needs eps");
    DISPATCH_T(dtype, hipLaunchKernelGGL((smooth_latent_bwd_kernel<T>), dim3(B), dim3(64), 0, (hipStream_t)stream, (const T*)dlat, Lpad,
                                         dmean, dlogvar, dalpha, logvar, eps, alpha, gs, temperature, training, sample_path, Dc, Dd,
                                         (T*)d_o, ldo));
    return sv_check_launch("sv_smooth_latent_bwd");
}

int sv_tanh_to_nchw(int dtype, const void* f, int B, int C, int H, int W, int ld, float* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(f && out, SV_E_ARG, "sv_tanh_to_nchw: null");
    const int64_t n = (int64_t)B * C * H * W;
    SV_REQUIRE(n < ((int64_t)1 << 32), SV_E_SHAPE, "sv_tanh_to_nchw: tensor too large");
    DISPATCH_T(dtype, hipLaunchKernelGGL((tanh_to_nchw_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)f, B, C, H * W, ld, out));
    return sv_check_launch("sv_tanh_to_nchw");
}

int sv_tanh_to_nchw_bwd(int dtype, const float* d_out, const float* out, int B, int C, int H, int W, int ld, void* d_f, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(d_out && out && d_f, SV_E_ARG, "sv_tanh_to_nchw_bwd: null");
    const int64_t n = (int64_t)B * H * W * ld;
    SV_REQUIRE(n < ((int64_t)1 << 32), SV_E_SHAPE, "sv_tanh_to_nchw_bwd: tensor too large");
    if (C <= 8 && ld % 8 == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((tanh_to_nchw_bwd_px_kernel<T>), dim3((unsigned)(((int64_t)B * H * W + 255) / 256)), dim3(256), 0,
                                             (hipStream_t)stream, d_out, out, B, C, H * W, ld, (T*)d_f));
        return sv_check_launch("sv_tanh_to_nchw_bwd");
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((tanh_to_nchw_bwd_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                         d_out, out, B, C, H * W, ld, (T*)d_f));
    return sv_check_launch("sv_tanh_to_nchw_bwd");
}

int sv_smooth_elbo_fwd(const float* data, const float* rec, int64_t n_per_img, const float* mean, const float* logvar,
                       const float* alpha, const int64_t* label, int B, int Dc, int Dd, const sv_smooth_schedule* sch,
                       const float* steps_dev, float* terms, float* coef, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(data && rec && mean && logvar && alpha && sch && terms && coef && B > 0, SV_E_ARG, "sv_smooth_elbo_fwd: bad argument");
    const int64_t n = n_per_img * B;
    {
        // (four float atomics per block on the same four addresses: 256 blocks, not 1 024 -- see sv_elbo_fwd)
        const int P = nblocks(n / 4, 1024, 256);
        float* w = nullptr;
        if (sv_deterministic() && P > 1 && !(w = det_slots(P, 4, 1, (hipStream_t)stream))) return SV_E_HIP;
        hipLaunchKernelGGL(smooth_elbo_fwd_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream,
                           data, rec, n, mean, logvar, alpha, label, B, Dc, Dd, w ? w : terms, w ? 4 : 0);
        if (w) det_collect(w, P, 4, 1, terms, (hipStream_t)stream);
    }
    smooth_caps c;
    c.cmin = sch->cont_min; c.cmax = sch->cont_max; c.citers = sch->cont_iters; c.cgamma = sch->cont_gamma;
    c.dmin = sch->disc_min; c.dmax = sch->disc_max; c.diters = sch->disc_iters; c.dgamma = sch->disc_gamma;
    c.alpha_cls = sch->alpha_cls; c.steps = sch->steps;
    hipLaunchKernelGGL(smooth_compose_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, terms, c, steps_dev, Dd, label ? 1 : 0, coef);
    return sv_check_launch("sv_smooth_elbo_fwd");
}

int sv_smooth_elbo_bwd(const float* data, const float* rec, int64_t n_per_img, const float* mean, const float* logvar,
                       const float* alpha, const int64_t* label, int B, int Dc, int Dd, const float* coef, const float* gout,
                       float* d_rec, float* d_mean, float* d_logvar, float* d_alpha, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(data && rec && mean && logvar && alpha && coef && gout && d_rec && d_mean && d_logvar && d_alpha, SV_E_ARG,
               "sv_smooth_elbo_bwd: null");
    const int64_t n = n_per_img * B;
    hipLaunchKernelGGL(smooth_elbo_bwd_kernel, dim3(nblocks(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, data, rec, n, mean,
                       logvar, alpha, label, B, Dc, Dd, coef, gout, d_rec, d_mean, d_logvar, d_alpha);
    return sv_check_launch("sv_smooth_elbo_bwd");
}

int sv_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float step,
            const float* step_dev, float grad_scale, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(p && g && m && v && n >= 0, SV_E_ARG, "sv_adam: null");
    SV_REQUIRE(step_dev || step >= 1.f, SV_E_ARG, "sv_adam: step=%g (1-based)", (double)step);
    if (n == 0) return SV_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, step, step_dev, grad_scale);
    return sv_check_launch("sv_adam");
}

int sv_sgd(float* p, const float* g, float* v, int64_t n, float lr, float momentum, float weight_decay,
           float grad_scale, int first_step, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(p && g && v && n >= 0, SV_E_ARG, "sv_sgd: null");
    if (n == 0) return SV_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(nblocks(n / 4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, v, n, lr,
                       momentum, weight_decay, grad_scale, first_step);
    return sv_check_launch("sv_sgd");
}

int sv_nchw_to_nhwc(int dtype, const float* in, int B, int C, int H, int W, int Cpad, void* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(in && out && Cpad >= C, SV_E_ARG, "sv_nchw_to_nhwc: bad args");
    const int64_t n = (int64_t)B * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                         (hipStream_t)stream, in, B, C, H * W, Cpad, (T*)out));
    return sv_check_launch("sv_nchw_to_nhwc");
}
int sv_gather_even(int dtype, const void* in, int B, int H, int W, int C, void* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, SV_E_ARG, "sv_gather_even: bad argument (B=%d H=%d W=%d)", B, H, W);
    SV_REQUIRE(C % 8 == 0, SV_E_SHAPE, "sv_gather_even: C=%d must be a multiple of 8", C);
    const int64_t nvec = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((gather_even_kernel<T>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)in, nvec, H / 2, W / 2, C / 8, (T*)out));
    return sv_check_launch("sv_gather_even");
}
int sv_augment(int dtype, const uint8_t* data, const int64_t* index, const int32_t* params, int B, int H, int W, int C,
               int pad, int nhwc_cpad, void* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(data && index && out, SV_E_ARG, "sv_augment: null");
    SV_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && pad >= 0 && pad < H && pad < W && (nhwc_cpad == 0 || nhwc_cpad >= C),
               SV_E_ARG, "sv_augment: bad shape (B=%d H=%d W=%d C=%d pad=%d cpad=%d)", B, H, W, C, pad, nhwc_cpad);
    const int64_t n = (int64_t)B * H * W * (nhwc_cpad ? nhwc_cpad : C);
    SV_REQUIRE(n < ((int64_t)1 << 32), SV_E_ARG, "sv_augment: batch too large");
    if (nhwc_cpad) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((augment_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                             (hipStream_t)stream, data, index, params, B, H, W, C, pad, nhwc_cpad,
                                             (T*)out, (float*)nullptr));
    } else {
        hipLaunchKernelGGL((augment_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           data, index, params, B, H, W, C, pad, 0, (float*)nullptr, (float*)out);
    }
    return sv_check_launch("sv_augment");
}

int sv_nhwc_to_nchw(int dtype, const void* in, int B, int C, int H, int W, int ld, float* out, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(in && out && ld >= C, SV_E_ARG, "sv_nhwc_to_nchw: bad args");
    const int64_t n = (int64_t)B * H * W * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)in, B, C, H * W, ld, out));
    return sv_check_launch("sv_nhwc_to_nchw");
}

int sv_repack_batch(int dtype, const float* master_base, const sv_repack_job* jobs, int njobs, int total_blocks,
                    void* dst_base, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(master_base && jobs && dst_base && njobs >= 0 && total_blocks >= 0, SV_E_ARG, "sv_repack_batch: bad argument");
    if (njobs == 0 || total_blocks == 0) return SV_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((repack_batch_kernel<T>), dim3((unsigned)total_blocks), dim3(256), 0,
                                         (hipStream_t)stream, master_base, jobs, njobs, (T*)dst_base));
    return sv_check_launch("sv_repack_batch");
}
int sv_param_gather(int dtype, const sv_param_job* jobs, int njobs, int total_blocks, void* dst_base, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(jobs && dst_base && njobs >= 0 && total_blocks >= 0, SV_E_ARG, "sv_param_gather: bad argument");
    if (njobs == 0 || total_blocks == 0) return SV_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((param_gather_kernel<T>), dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, njobs,
                                         (T*)dst_base));
    return sv_check_launch("sv_param_gather");
}
int sv_param_scatter_add(const sv_param_job* jobs, int njobs, int total_blocks, const float* src_base, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(jobs && src_base && njobs >= 0 && total_blocks >= 0, SV_E_ARG, "sv_param_scatter_add: bad argument");
    if (njobs == 0 || total_blocks == 0) return SV_OK;
    hipLaunchKernelGGL(param_scatter_add_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, njobs, src_base);
    return sv_check_launch("sv_param_scatter_add");
}
int sv_repack(int dtype, const float* master, int N, int T_orig, int C, int transpose, const sv_geom* g, void* dst,
              void* stream) {
    return sv_repack_strided(dtype, master, N, C, (int64_t)T_orig * C, C, 1, N, T_orig, C, transpose, g, dst, stream);
}

int sv_repack_strided(int dtype, const float* src, int n_real, int c_real, int64_t sn, int64_t st, int64_t sc, int N, int T_orig,
                      int C, int transpose, const sv_geom* g, void* dst, void* stream) {
    SvProfScope prof_scope(stream);
    SV_REQUIRE(src && g && dst, SV_E_ARG, "sv_repack: null");
    SV_REQUIRE(n_real >= 1 && n_real <= N && c_real >= 1 && c_real <= C, SV_E_ARG, "sv_repack: source extents %d x %d of %d x %d", n_real, c_real, N, C);
    repack_params p;
    p.N = N; p.T_orig = T_orig; p.C = C; p.transpose = transpose; p.nphase = g->nphase;
    p.n_real = n_real; p.c_real = c_real; p.sn = sn; p.st = st; p.sc = sc;
    int64_t mx = 0;
    for (int i = 0; i < SV_MAX_PHASES; ++i) {
        p.ntap[i] = 0; p.w_off[i] = 0; p.size[i] = 0;
        for (int t = 0; t < SV_MAX_TAPS; ++t) p.torig[i][t] = 0;
    }
    for (int i = 0; i < g->nphase; ++i) {
        p.ntap[i] = g->phase[i].ntap;
        p.w_off[i] = g->phase[i].w_off;
        p.size[i] = (int64_t)N * C * g->phase[i].ntap;
        for (int t = 0; t < SV_MAX_TAPS; ++t) p.torig[i][t] = g->phase[i].torig[t];
        if (p.size[i] > mx) mx = p.size[i];
    }
    if (mx == 0) return SV_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((repack_kernel<T>), dim3((unsigned)((mx + 255) / 256), g->nphase), dim3(256),
                                         0, (hipStream_t)stream, src, p, (T*)dst));
    return sv_check_launch("sv_repack");
}

}  // extern "C"
