// Shared device helpers for libshotvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "shotvae_hip.h"

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct V8;
template <> struct V8<float> { typedef f32x8 type; };
template <> struct V8<bf16> { typedef bf16x8 type; };
template <typename T> struct V4;
template <> struct V4<float> { typedef f32x4 type; };
template <> struct V4<bf16> { typedef bf16x4 type; };

// D(16x16) += A(16xK) * B(Kx16) for one 32-deep k chunk.  Lane l supplies A[row l&15][k=8(l>>4)+j]
// and B[k=8(l>>4)+j][col l&15], j=0..7.  bf16: one v_mfma_f32_16x16x32_bf16.  fp32: eight
// v_mfma_f32_16x16x4_f32 (exact fp32), step j consuming element j of every lane group -- the k order
// is a permutation of the same 32 products, so both dtypes share one fragment layout.
__device__ __forceinline__ void mma32(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
}

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }

// LeakyReLU / ReLU; every entry point rejects slopes outside [0, 1], where max(u, slope*u) is the activation
__device__ __forceinline__ float act_fwd(float u, float slope) { return fmaxf(u, u * slope); }
__device__ __forceinline__ float act_grad(float u, float slope) { return u > 0.f ? 1.f : slope; }

// BatchNorm-apply + LeakyReLU / ReLU of 8 consecutive channels (the load prologue of every conv-like kernel):
// o[j] = act(x[j] * s[j] + t[j]).  (A version on packed fp32 pairs -- v_pk_fma_f32 / v_pk_mul_f32, 28 instead of ~40 VALU
// instructions per vector -- measured SLOWER on gfx950: conv3x3p forward 90.4 vs 88.4 us, the generic stride-2 forward 532
// vs 517 us; the packed fp32 instructions issue at half rate.)
__device__ __forceinline__ bf16x8 bn_act8(const bf16x8& x, const f32x4& s0, const f32x4& s1, const f32x4& t0, const f32x4& t1,
                                          float slope) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float u0 = (float)x[j] * s0[j] + t0[j], u1 = (float)x[j + 4] * s1[j] + t1[j];
        o[j] = (bf16)fmaxf(u0, u0 * slope);
        o[j + 4] = (bf16)fmaxf(u1, u1 * slope);
    }
    return o;
}
__device__ __forceinline__ f32x8 bn_act8(const f32x8& x, const f32x4& s0, const f32x4& s1, const f32x4& t0, const f32x4& t1,
                                         float slope) {
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float u0 = x[j] * s0[j] + t0[j], u1 = x[j + 4] * s1[j] + t1[j];
        o[j] = fmaxf(u0, u0 * slope);
        o[j + 4] = fmaxf(u1, u1 * slope);
    }
    return o;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// packed per-tap offsets: 4 bits per tap, biased by 8
__device__ __forceinline__ uint64_t pack_taps(const int8_t* d) {
    uint64_t p = 0;
#pragma unroll
    for (int t = 0; t < SV_MAX_TAPS; ++t) p |= (uint64_t)((d[t] + 8) & 15) << (4 * t);
    return p;
}
__device__ __forceinline__ int tap_off(uint64_t p, int t) { return (int)((p >> (4 * t)) & 15) - 8; }

// Batched launch (sv_igemm_args::groups): blockIdx.y = group; a group is an independent instance of the layer whose
// tensors / coefficient vectors / accumulators follow those of the previous group.  The HOST expands the caller's
// arguments into one argument block per group (all of them travel as the kernel parameter); a kernel starts with
// `const sv_igemm_args& a = A.g[blockIdx.y];` -- a reference into kernel-argument memory, so every field is still a scalar
// load at its point of use (a modified COPY of the block would pin ~40 SGPRs for the kernel's lifetime: measured +40 % on
// the persistent 3x3 kernel) -- and runs on (g, a) exactly as for a single instance.
constexpr int SV_MAX_GROUPS = 4;          // the four forwards of a SHOT-VAE step
struct sv_igemm_args_g { sv_igemm_args g[SV_MAX_GROUPS]; };
__host__ __device__ __forceinline__ int sv_ngroups(int groups) { return groups > 0 ? groups : 1; }
inline sv_igemm_args_g sv_expand_groups(const sv_geom& g, const sv_igemm_args& a, int es) {
    sv_igemm_args_g A;
    const int64_t xs = (int64_t)g.B * g.Hin * g.Win * g.ldx * es, os = (int64_t)g.B * g.Hout * g.Wout * g.ldo * es;
    for (int64_t grp = 0; grp < SV_MAX_GROUPS; ++grp) {
        sv_igemm_args r = a;
        if (grp > 0 && grp < sv_ngroups(a.groups)) {
            r.x = reinterpret_cast<const char*>(a.x) + grp * xs;
            r.out = reinterpret_cast<char*>(a.out) + grp * os;
            if (a.residual) r.residual = reinterpret_cast<const char*>(a.residual) + grp * os;
            if (a.ex) r.ex = reinterpret_cast<const char*>(a.ex) + grp * os;
            if (a.pro_scale) { r.pro_scale = a.pro_scale + grp * g.Cin; r.pro_shift = a.pro_shift + grp * g.Cin; }
            if (a.fold_stats) {
                r.fold_stats = a.fold_stats + grp * (int64_t)a.fold_replicas * 2 * g.Cin;
                r.fold_mean = a.fold_mean + grp * g.Cin;
                r.fold_rstd = a.fold_rstd + grp * g.Cin;
            }
            if (a.stats) r.stats = a.stats + grp * (int64_t)a.replicas * 2 * g.N;
            if (a.ex) {
                r.ex_scale = a.ex_scale + grp * g.N; r.ex_shift = a.ex_shift + grp * g.N;
                r.ex_mean = a.ex_mean + grp * g.N; r.ex_rstd = a.ex_rstd + grp * g.N;
                r.bsums = a.bsums + grp * (int64_t)a.replicas * 2 * g.N;
            }
        }
        A.g[grp] = r;
    }
    return A;
}
// the same for the weight-gradient parameter blocks (fields x, dy, pro_scale, pro_shift): the groups' operands follow
// each other, the gradient of the SHARED weights sums over the groups
template <typename P> struct sv_wg_g { P g[SV_MAX_GROUPS]; };
template <typename P>
inline sv_wg_g<P> sv_expand_wg(const sv_geom& g, const P& p, int groups, int es) {
    sv_wg_g<P> A;
    for (int64_t grp = 0; grp < SV_MAX_GROUPS; ++grp) {
        P r = p;
        if (grp > 0 && grp < groups) {
            r.x = reinterpret_cast<const char*>(p.x) + grp * ((int64_t)g.B * g.Hin * g.Win * g.ldx * es);
            r.dy = reinterpret_cast<const char*>(p.dy) + grp * ((int64_t)g.B * g.Hout * g.Wout * g.ldo * es);
            if (p.pro_scale) { r.pro_scale = p.pro_scale + grp * g.Cin; r.pro_shift = p.pro_shift + grp * g.Cin; }
        }
        A.g[grp] = r;
    }
    return A;
}

// BatchNorm finalisation folded into a consumer kernel (sv_igemm_args::fold_*): all 256 threads of the block sum the R replicas
// of (sum, sum of squares) of the C <= 64 channels (a fixed partition: thread tid takes channel tid % C and the replicas
// tid / C, + 256 / C, ...; the partial sums meet in LDS in index order), derive scale / shift -- sv_bn_finalize's arithmetic --
// into sc_out / sh_out (LDS, C floats each), and `writer` blocks also store scale / shift / mean / rstd to global memory.
// `scratch`: 2 * 256 DOUBLES of LDS (4 KB).  Ends with a barrier.
// The accumulators are doubles (sv_acc_t): sums of the producers' fp32 partial sums, exact in any order; mean and variance are
// formed in double (no E[x^2] - E[x]^2 cancellation in fp32) and rounded once -- the same arithmetic as bn_finalize_kernel.
__device__ __forceinline__ void sv_bn_moments(double s1, double s2, float count, float eps, float& mu, float& var, float& rs) {
    const double m = s1 / (double)count;
    double v = s2 / (double)count - m * m;
    v = v > 0.0 ? v : 0.0;
    mu = (float)m;
    var = (float)v;
    rs = rsqrtf(var + eps);
}
__device__ __forceinline__ void sv_bn_fold_block(const sv_igemm_args& a, int C, double* scratch, float* sc_out, float* sh_out,
                                                 bool writer) {
    const int tid = threadIdx.x, c = tid % C, part = tid / C, parts = 256 / C;
    double s1 = 0.0, s2 = 0.0;
    if (part < parts)
        for (int r = part; r < a.fold_replicas; r += parts) {
            s1 += a.fold_stats[(size_t)r * 2 * C + c];
            s2 += a.fold_stats[(size_t)r * 2 * C + C + c];
        }
    scratch[tid] = s1;
    scratch[256 + tid] = s2;
    __syncthreads();
    if (tid < C) {
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < parts; ++q) { t1 += scratch[q * C + tid]; t2 += scratch[256 + q * C + tid]; }
        float mu, var, rs;
        sv_bn_moments(t1, t2, a.fold_count, a.fold_eps, mu, var, rs);
        const float sc = a.fold_gamma[tid] * rs, sh = a.fold_beta[tid] - mu * sc;
        sc_out[tid] = sc;
        sh_out[tid] = sh;
        if (writer) {
            const_cast<float*>(a.pro_scale)[tid] = sc;
            const_cast<float*>(a.pro_shift)[tid] = sh;
            a.fold_mean[tid] = mu;
            a.fold_rstd[tid] = rs;
        }
    }
    __syncthreads();
}

// The same for the 512-thread kernels with register-resident weights (tconv.hip, sconv.hip): C = 32 / 64 / 128 channels, the
// coefficients as [C] pairs {scale, shift} (coef2, LDS).  `scratch`: 2 * 512 doubles of LDS (8 KB).  Ends with a barrier.
__device__ __forceinline__ void sv_bn_fold_block512(const sv_igemm_args& a, int C, double* scratch, float* coef2, bool writer) {
    const int tid = threadIdx.x, c = tid % C, part = tid / C, parts = 512 / C;
    double s1 = 0.0, s2 = 0.0;
    for (int r = part; r < a.fold_replicas; r += parts) {
        s1 += a.fold_stats[(size_t)r * 2 * C + c];
        s2 += a.fold_stats[(size_t)r * 2 * C + C + c];
    }
    scratch[tid] = s1;
    scratch[512 + tid] = s2;
    __syncthreads();
    if (tid < C) {
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < parts; ++q) { t1 += scratch[q * C + tid]; t2 += scratch[512 + q * C + tid]; }
        float mu, var, rs;
        sv_bn_moments(t1, t2, a.fold_count, a.fold_eps, mu, var, rs);
        const float sc = a.fold_gamma[tid] * rs, sh = a.fold_beta[tid] - mu * sc;
        coef2[2 * tid] = sc;
        coef2[2 * tid + 1] = sh;
        if (writer) {
            const_cast<float*>(a.pro_scale)[tid] = sc;
            const_cast<float*>(a.pro_shift)[tid] = sh;
            a.fold_mean[tid] = mu;
            a.fold_rstd[tid] = rs;
        }
    }
    __syncthreads();
}

// Interleaved tile order of the one-block-per-CU kernels (bwd3x3f.hip, fwd3x3f.hip) for ANY grid size: at its step k a block takes tile
// k * NC + slot of its group, and the blocks that share an XCD (the hardware deals blocks to the eight XCDs round-robin in linear
// block-id order, x fastest) own CONSECUTIVE slots, so that vertically adjacent tiles -- which share halo rows -- meet in one L2.
// (With NC a multiple of 8 this is conv3x3p_kernel's mapping; a batched launch of four groups on 248 blocks has NC = 62.)
__device__ __forceinline__ int sv_window_slot(int NC, int group, int bx) {
    const int l0 = group * NC;                                   // linear id of the group's first block
    const int xcd = (l0 + bx) & 7;
    int prefix = 0;
    for (int y = 0; y < 8; ++y) {
        const int first = (y - l0) & 7;                          // first bx of the group on XCD y
        const int cnt = first < NC ? (NC - first + 7) >> 3 : 0;
        if (y < xcd) prefix += cnt;
    }
    return prefix + ((bx - ((xcd - l0) & 7)) >> 3);
}

// host side ------------------------------------------------------------------------------------------
void sv_set_error(const char* fmt, ...);
bool sv_disabled(int kernel_bit);        // sv_set_option(SV_OPT_DISABLE_MASK, ...): a specialised kernel is switched off
int sv_wide_min_blocks();
bool sv_enabled(int kernel_bit);       // sv_set_option(SV_OPT_ENABLE_MASK, ...): a kernel that is OFF by default is switched on
bool sv_halo_all();
int sv_persistent_blocks();            // block budget of the persistent kernels: the launch's own (sv_igemm_args::block_budget,
                                       // sv_wgrad_args::block_budget) if it has one, else sv_set_option(SV_OPT_PERSISTENT_BLOCKS, ...)
// scope of one entry-point call on the calling thread: its launches see `budget` (> 0) as their block budget
struct SvBudgetScope {
    int old;
    explicit SvBudgetScope(int budget);
    ~SvBudgetScope();
};
bool sv_deterministic();               // sv_set_option(SV_OPT_DETERMINISTIC, 1): every accumulation in a fixed order
bool sv_det_stats();                   // ... 1 or 2: the BatchNorm statistics / backward sums in a fixed order (2: only those)
float* sv_det_scratch(size_t floats);  // deterministic mode: a slice of the library's scratch ring (nullptr + error text on failure)
enum { SV_FLAG_DET = 1 };              // sv_igemm_args::flags
// sv_igemm_query_blocks: the launch functions call sv_dry_run(grid) right before their launch; it returns true (and records
// the grid) when the calling thread is inside a query -- the caller then returns SV_OK without launching.  In deterministic
// mode it also checks the replica count of a real launch (returns true with *rc < 0 when it is too small).
bool sv_dry_run(int grid_x, const sv_igemm_args* a, int* rc);
bool sv_in_query();                    // the calling thread is inside sv_igemm_query_blocks (nothing may be launched)
// BatchNorm fold protocol (runtime.hip): sv_igemm announces a pending fold; a launcher whose kernel folds claims it (can =
// the kernel's own limits hold, e.g. fold_replicas <= 64) and passes fold_stats on, everybody else passes fold_stats = NULL;
// an unclaimed fold is materialised by the launch gate (sv_bn_finalize as a launch of its own)
void sv_fold_begin(const sv_geom* g, const sv_igemm_args* a, void* stream);
void sv_fold_end();
bool sv_fold_claim(bool can);
// sv_igemm_args::start_flag: the first block of every kernel of the family announces its start (see shotvae_hip.h)
__device__ __forceinline__ void sv_start_signal(const sv_igemm_args& a) {
    if (a.start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.start_flag, a.start_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define SV_LAUNCH_GATE(grid_x, a)                                \
    do {                                                         \
        int gate_rc_ = SV_OK;                                    \
        if (sv_dry_run((grid_x), (a), &gate_rc_)) return gate_rc_; \
    } while (0)
int sv_check_launch(const char* what);
void sv_prof_begin(hipStream_t s);
void sv_prof_end(hipStream_t s);
// brackets the launches of one entry point with the in-situ timing events (no-ops unless sv_prof_enable(1))
struct SvProfScope {
    hipStream_t s;
    explicit SvProfScope(void* st) : s((hipStream_t)st) { sv_prof_begin(s); }
    ~SvProfScope() { sv_prof_end(s); }
};
int sv_conv3x3_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_hwgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                  const void* dy, float* dw, float* ws, int64_t ws_elems, int groups, hipStream_t s, int* rc);
int sv_halo_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_tconvr_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_sconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_pconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_dconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_thconv_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_thwgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                   const void* dy, float* dw, int groups, hipStream_t s, int* rc);
// k4wgrad.hip: the thin 4x4 stride-2 layers of svhn_VAE (16 channels at 32x32 <-> 32 channels at 16x16)
int sv_k4wgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                   const void* dy, float* dw, float* ws, int64_t ws_elems, int groups, hipStream_t s, int* rc);
int sv_s2wgrad_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                   const void* dy, float* dw, int groups, hipStream_t s, int* rc);
int sv_conv3x3w_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc);
int sv_conv3x3x_try(const sv_geom* g, const sv_igemm_args* a, bool fwd, hipStream_t s, int* rc);
int sv_wgrad3x3_try(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift,
                    float pro_slope, const void* dy, float* dw, float* ws, int64_t ws_elems, int groups, hipStream_t s,
                    int* rc);

#define SV_REQUIRE(cond, code, ...)                \
    do {                                           \
        if (!(cond)) {                             \
            sv_set_error(__VA_ARGS__);             \
            return code;                           \
        }                                          \
    } while (0)
