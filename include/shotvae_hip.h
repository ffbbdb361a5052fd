/*
 * shotvae_hip.h  --  C ABI of libshotvae_hip.so: the MI355X (gfx950) compute path of the
 * SHOT-VAE training step.
 *
 * The reference (FengHZ/SHOT-VAE) is pure Python on torch.nn; it has no FFI.  Each entry point
 * below replaces the torch operator(s) that the cited reference lines dispatch (SURVEY.md §2.1
 * K1-K20).  Conventions for every function:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no allocation, no host sync,
 *     no exceptions -- safe to capture into a hipGraph;
 *   - returns 0 on success, <0 on error (SV_E_*); sv_last_error() gives the message;
 *   - activations are NHWC with an explicit channel stride, element type `dtype`
 *     (SV_F32 exact-fp32 MFMA mode for parity, SV_BF16 throughput mode); statistics,
 *     parameters, gradients of parameters and loss terms are always fp32.
 */
#ifndef SHOTVAE_HIP_H
#define SHOTVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SV_ABI_VERSION 8

enum { SV_F32 = 0, SV_BF16 = 1 };
/* ABI 6: the per-channel ACCUMULATORS of the BatchNorm statistics and backward sums (sv_igemm_args::stats / bsums / fold_stats,
 * sv_bn_finalize, sv_bn_branch::bsums, sv_pool_bwd, sv_bn_bwd_affine) are DOUBLES.  The partial sums of a launch's waves and
 * blocks meet there through atomic adds in an order that varies from run to run; as fp32 adds that order reached the results
 * (two repeats of the same bf16 step: 2e-5 relative in the forward scalars, amplified by 28 BatchNorm layers to a 0.97-0.98
 * cosine between the two gradients), as fp64 adds of fp32 partial sums it does not -- the additions are exact unless the
 * partial sums of one channel span more than ~2^29, so two runs agree bit for bit (measured) at no cost in time.            */
typedef double sv_acc_t;
enum { SV_OK = 0, SV_E_ARG = -1, SV_E_SHAPE = -2, SV_E_HIP = -3 };
enum { SV_MAX_TAPS = 16, SV_MAX_PHASES = 4 };

/* One sub-pixel phase of a (transposed / strided) convolution written as a gather-GEMM. */
typedef struct {
    int32_t ooy, oox;               /* output offset: oy = qy*osy + ooy                              */
    int32_t ntap;                   /* taps of this phase (0 => the phase's outputs are zero)        */
    int8_t dy[SV_MAX_TAPS];         /* input offset per tap: iy = qy*sy + dy[t] (zero outside)       */
    int8_t dx[SV_MAX_TAPS];
    int8_t torig[SV_MAX_TAPS];      /* tap index in the MASTER weight layout [N][T_orig][Cin]        */
    int64_t w_off;                  /* element offset of this phase's packed weights [N][ntap][Cin]  */
} sv_phase;

/* Geometry of one conv-like layer:  out[b, qy*osy+ooy, qx*osx+oox, n] =
 *     sum_{t,c} act(x[b, qy*sy+dy[t], qx*sx+dx[t], c]) * w[n][t][c]                                  */
typedef struct {
    int32_t B, Hin, Win, Cin, ldx;          /* input  tensor [B,Hin,Win,ldx],  Cin  % 16 == 0        */
    int32_t Hq, Wq, sy, sx;                 /* per-phase output grid and input stride                */
    int32_t Hout, Wout, N, ldo, osy, osx;   /* output tensor [B,Hout,Wout,ldo], N % 16 == 0          */
    int32_t T_orig;                         /* taps in the master weight layout                      */
    int32_t nphase;
    sv_phase phase[SV_MAX_PHASES];
} sv_geom;

/* ---- K1/K2/K3/K12/K13 forward and every dgrad: fused gather-GEMM on MFMA -----------------------
 * Replaces nn.Conv2d / nn.ConvTranspose2d forward (wideresnet.py:13-14,29-30,34-35,41-43;
 * decoder.py:13-58) fused with the preceding BatchNorm2d-apply + LeakyReLU/ReLU (wideresnet.py:
 * 27-28,32-33,39-40; decoder.py:19-20,...) as a load prologue, and with the residual add
 * (wideresnet.py:49) and the *next* BatchNorm's batch statistics as an epilogue.  With the
 * `ex` fields set it is the conv/convT input-gradient fused with the activation backward and the
 * two BatchNorm-backward reductions (autograd of the same lines).                                  */
typedef struct {
    const void* x;              /* input activations                                                 */
    const float* pro_scale;     /* [Cin] BN-apply scale (NULL: no prologue)                          */
    const float* pro_shift;     /* [Cin]                                                             */
    float pro_slope;            /* LeakyReLU slope in [0,1] (0 = ReLU)                                     */
    const void* w;              /* packed weights, element type = dtype                              */
    const float* bias;          /* [N] or NULL                                                       */
    const void* residual;       /* same layout as out, or NULL                                       */
    void* out;
    sv_acc_t* stats;            /* [R][2N] += (sum y, sum y^2) or NULL; block b adds to replica b % R  */
    const void* ex;             /* act-backward epilogue: raw tensor at the output positions / NULL  */
    const float* ex_scale;      /* [N] each                                                          */
    const float* ex_shift;
    const float* ex_mean;
    const float* ex_rstd;
    float ex_slope;
    sv_acc_t* bsums;            /* [R][2N] += (sum g, sum g*xhat)                                    */
    int32_t replicas;           /* R: power of two >= 1; spreads the per-channel atomics of the many
                                   blocks of a launch over R copies (consumers sum the copies)        */
    int32_t groups;             /* G >= 1 (0 = 1): BATCHED launch of G independent instances of the layer that
                                   share the weights and the bias -- the four forwards of a SHOT-VAE step
                                   (main_shot_vae.py:288,311,329,356) each with its OWN BatchNorm statistics.
                                   g->B is the batch of ONE group; x / out / residual / ex hold the groups back
                                   to back ([G][B][H][W][ld]); pro_scale / pro_shift are [G][Cin], the ex_*
                                   vectors [G][N], stats / bsums [G][R][2N].                            */
    int32_t block_budget;       /* > 0: block budget of THIS launch if it takes a persistent kernel (conv3x3p, halop),
                                   overriding SV_OPT_PERSISTENT_BLOCKS -- the paired backward gives a layer's weight and
                                   data gradient half the chip each without touching process-wide state; 0 = the option */
    int32_t flags;              /* reserved: written by the library (bit 0 = deterministic accumulation of this launch) */
    int32_t sparse_out;         /* 1: output positions of phases WITHOUT taps are left unwritten (and their epilogue operands
                                   unread) instead of zero-filled -- the data gradient of a stride-2 1x1 convolution is zero at
                                   three of four positions; its only consumer, sv_bn_bwd_apply with sv_bn_branch::sparse = 1,
                                   does not read them.  0: every output position is written.                              */
    int32_t reserved0;          /* 0 (ABI 5's ex_mode: the recomputing data gradient lost to the streaming pass twice and left the tree) */
    /* ABI 4: BatchNorm finalisation FOLDED into the consumer.  fold_stats != NULL: the prologue's BatchNorm has not been
       finalised yet -- fold_stats [R = fold_replicas][2 Cin] are the raw (sum, sum of squares) its producer accumulated over
       fold_count samples per channel; pro_scale / pro_shift (and fold_mean / fold_rstd) are then OUTPUT locations [Cin]:
       scale = gamma * rstd, shift = beta - mean * scale, exactly sv_bn_finalize's arithmetic.  A kernel that supports it
       (the persistent 3x3 kernel of the narrow body layers) lets every block derive the coefficients of its channels
       itself -- a few dozen loads at its start instead of a launch of its own between two layers -- and block 0 stores the
       four vectors for the backward pass / sv_bn_running_update; for every other kernel sv_igemm runs sv_bn_finalize first.
       Groups: fold_stats [G][R][2 Cin], the outputs [G][Cin].  (main_shot_vae.py has 33 BatchNorms per forward.)          */
    const sv_acc_t* fold_stats;
    const float* fold_gamma;
    const float* fold_beta;
    float* fold_mean;
    float* fold_rstd;
    float fold_count;
    float fold_eps;
    int32_t fold_replicas;
    int32_t reserved1;
    /* ABI 5: start signal.  start_flag != NULL: the first block of the launch stores start_value to *start_flag (device
       memory, 4 bytes) as soon as it starts -- i.e. when everything enqueued on the stream before this launch has completed.
       A second stream that waits for the value (sv_stream_wait_flag) is thereby forked off IN FRONT of this launch without
       an event in this stream's queue: the weight gradient beside its data gradient (an event record costs the recording
       queue ~6 us of idle time in front of the next kernel on MI355X, 34 times per step).                                  */
    uint32_t* start_flag;
    uint32_t start_value;
    int32_t reserved2;
} sv_igemm_args;

int sv_igemm(const sv_geom* g, int dtype, const sv_igemm_args* a, void* stream);
/* The affine coefficients of a BatchNorm backward from its two sums (sv_igemm_args::bsums of the data gradient behind it, [G][R][2C],
 * replicas summed in index order):  with A = gamma * rstd, m1 = sum g / count, m2 = sum g xhat / count
 *     scale_g[g][c] = A,   scale_x[g][c] = -A * m2 * rstd,   shift[g][c] = -A * m1 + A * m2 * rstd * mean
 * so that scale_g * g + scale_x * x + shift = A * (g - m1 - xhat * m2), the autograd of wideresnet.py:27,32 -- the operands
 * dy_scale / dy_scale2 / dy_shift of sv_bwd3x3_args, which forms that BatchNorm backward in its load path instead of a pass of its own
 * (sv_bn_bwd_apply: two reads and a write of the tensor between two layers) -- and, what sv_bn_bwd_apply does on the side,
 * dbeta[c] += sum g, dgamma[c] += sum g xhat over all groups (either may be NULL).
 * (ABI 6 carried the same fusion as sv_igemm_args::x2 / sv_wgrad_args::dy2, each launch of the pair forming it for itself: one pass
 *  saved, two consumers taxed, slower in the step -- docs/lab_notes_r05.md; removed in ABI 7.)                                      */
int sv_bn_bwd_affine(const sv_acc_t* bsums, int replicas, int C, float count, const float* gamma, const float* mean, const float* rstd,
                     float* dgamma, float* dbeta, float* scale_g, float* scale_x, float* shift, int groups, void* stream);
/* The grid (blocks in x) sv_igemm WOULD launch for these arguments under the current options; nothing is launched.  With
 * SV_OPT_DETERMINISTIC the per-channel accumulators (`stats` / `bsums`) need replicas >= 4 * blocks (next power of two):
 * every wave of every block then adds to a replica of its own.                                                        */
int sv_igemm_query_blocks(const sv_geom* g, int dtype, const sv_igemm_args* a, int* blocks);

/* ---- K19 weight gradient: dW[n][torig][c] += sum_m dy[m][n] * act(x[m,t][c]) --------------------
 * Replaces autograd's convolution_backward (weight part) for the same layers.  `splits` = number
 * of M ranges of the generic kernel (0 = choose).  `ws` is an optional caller-owned fp32 workspace
 * (ws_elems floats, contents irrelevant on entry): stride-1 3x3 layers publish per-block partial
 * slabs there with plain stores and reduce them in a second launch instead of contending on float
 * atomics; without it (or when it is too small) partials go to dw through float atomics.
 * groups (0 = 1): batched launch as in sv_igemm_args -- x / dy hold G instances back to back (g->B = one group's
 * batch), pro_scale / pro_shift are [G][Cin]; the gradient of the shared weights sums over the groups.              */
int sv_wgrad(const sv_geom* g, int dtype, const void* x, const float* pro_scale, const float* pro_shift,
             float pro_slope, const void* dy, float* dw, int splits, int use_tr, float* ws, int64_t ws_elems,
             int groups, void* stream);

/* The same launch with its arguments in a struct (ABI 3): adds the per-launch block budget of the persistent weight-gradient
 * kernels (0 = SV_OPT_PERSISTENT_BLOCKS).  sv_wgrad(...) == sv_wgrad_ex with block_budget = 0.                          */
typedef struct {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    float pro_slope;
    const void* dy;
    float* dw;
    int32_t splits;
    int32_t use_tr;
    float* ws;
    int64_t ws_elems;
    int32_t groups;
    int32_t block_budget;
} sv_wgrad_args;
int sv_wgrad_ex(const sv_geom* g, int dtype, const sv_wgrad_args* a, void* stream);

/* ---- fused backward of a narrow stride-1 3x3 convolution (ABI 7): data gradient + weight gradient in ONE launch ----------
 * Replaces, for the 32 -> 32 channel body convolutions of WideResNet-28-2 (wideresnet.py:29-35 under autograd; 8 / 16 / 32-pixel
 * maps) and the 64 -> 64 channel ones (16-pixel maps; C below is the channel count), the PAIR
 *     sv_igemm(geom_dgrad, x = dy, ex = raw input ...)   +   sv_wgrad_ex(geom_fwd, x = raw input, dy ...)
 * and, in the two-tensor form (dy2 != NULL), the sv_bn_bwd_apply pass in front of that pair too.  One persistent kernel stages a
 * tile of dy (with its halo) and of the raw input ONCE and feeds both products from the same LDS image: 3 tensor passes (4 in the
 * two-tensor form) where the pair makes 5 (8).  `g` is the layer's DATA-gradient geometry (one phase of nine taps); bf16 only.
 *   out[q][c]         = act'(x[q][c] * x_scale[c] + x_shift[c]) * sum_{t,n} dyeff[q + d(t)][n] * w[c][t][n]        (= sv_igemm's
 *                       ex epilogue, bit for bit), bsums[R][2C] += (sum out, sum out * (x - x_mean) * x_rstd)
 *   dw[n][torig(t)][c] += sum_q dyeff[q + d(t)][n] * act(x[q][c] * x_scale[c] + x_shift[c])
 *   dyeff = dy, or dy_scale[n] * dy + dy_scale2[n] * dy2 + dy_shift[n] (the coefficients of sv_bn_bwd_affine: the BatchNorm
 *           backward of the layer BEHIND the convolution; same expression and rounding as sv_igemm_args::x2 / sv_wgrad_args::dy2)
 * groups (0 = 1): as sv_igemm_args -- tensors [G][B][H][W][C], vectors [G][C], bsums [G][R][2C]; dw sums over the groups.
 * ws: caller-owned fp32 workspace of >= blocks * groups * 9 C^2 floats (blocks <= block_budget, default 256), contents irrelevant:
 * one partial slab per block, reduced into dw (+=, float atomics) by a second launch.  Not available under SV_OPT_DETERMINISTIC. */
typedef struct {
    const void* dy;             /* [G][B,H,W,32] gradient behind the convolution's output (or behind the BatchNorm after it)   */
    const void* dy2;            /* NULL, or that BatchNorm's raw input (= the convolution's own output)                         */
    const float* dy_scale;      /* [G][32] each; required with dy2                                                             */
    const float* dy_scale2;
    const float* dy_shift;
    const void* dy3;            /* NULL, or (with dy2) a third tensor ADDED to the two-tensor form: dyeff = dy_scale * dy + dy_scale2 *
                                   dy2 + dy_shift + dy3 -- for conv2 of a residual unit: (dy, dy2) = the data gradient behind the NEXT
                                   unit's norm1 and that BatchNorm's raw input (this unit's output), dy3 = the gradient arriving at the
                                   next unit's output: BatchNorm backward + skip connection of wideresnet.py:45-49 in the load path      */
    void* dy_out;               /* required with dy3: receives dyeff (the gradient at this unit's output: the previous unit's skip
                                   branch needs the tensor), every element written once                                                  */
    const void* x;              /* [G][B,H,W,32] RAW input of the BatchNorm in front of the convolution                        */
    const float* x_scale;       /* [G][32] each: that BatchNorm's scale / shift / mean / rstd (sv_bn_finalize)                 */
    const float* x_shift;
    const float* x_mean;
    const float* x_rstd;
    float x_slope;              /* LeakyReLU slope in [0, 1]                                                                   */
    const void* w;              /* the layer's data-gradient pack (sv_repack with geom_dgrad), bf16                            */
    void* out;                  /* [G][B,H,W,32]                                                                                */
    sv_acc_t* bsums;            /* [G][R][64]                                                                                   */
    int32_t replicas;
    int32_t groups;
    float* dw;                  /* master layout [32][9][32], += */
    float* ws;
    int64_t ws_elems;
    int32_t block_budget;       /* 0 = 256 */
    int32_t reserved0;
    /* ABI 8: the coefficients of the two- / three-tensor forms derived IN the launch (sv_bn_bwd_affine's arithmetic, every block from the
       raw sums; block 0 of a group adds dgamma / dbeta): one small launch less in front of every fused backward.  With fold_bsums set,
       dy_scale / dy_scale2 / dy_shift are not read (may be NULL). */
    const sv_acc_t* fold_bsums; /* NULL, or (with dy2) [G][fold_replicas][2C]: the backward sums of the BatchNorm BEHIND the convolution  */
    const float* fold_gamma;    /* [C]                                                                                               */
    const float* fold_mean;     /* [G][C]                                                                                            */
    const float* fold_rstd;     /* [G][C]                                                                                            */
    float* fold_dgamma;         /* [C], += ; may be NULL                                                                             */
    float* fold_dbeta;          /* [C], += ; may be NULL                                                                             */
    float fold_count;           /* elements per channel and group (B * H * W)                                                        */
    int32_t fold_replicas;
} sv_bwd3x3_args;
int sv_bwd3x3(const sv_geom* g, int dtype, const sv_bwd3x3_args* a, void* stream);

/* column sums: out[n] += sum_m y[m*ld + n]   (conv0 bias gradient)                                  */
int sv_colsum(int dtype, const void* y, int64_t M, int N, int ld, float* out, void* stream);

/* ---- K4 BatchNorm2d train-mode finalize (nn.BatchNorm2d semantics, eps/momentum explicit) -------
 * stats=[R][sum, sumsq] -> scale=gamma*rstd, shift=beta-mean*scale; saves mean/rstd; updates running
 * stats (unbiased var) unless running_mean is NULL.  groups (0 = 1): G sets of statistics [G][R][2C] of the same
 * BatchNorm (gamma / beta shared) -> outputs [G][C]; running_mean must then be NULL (sv_bn_running_update).   */
int sv_bn_finalize(const sv_acc_t* stats, int replicas, int C, float count, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var,
                   float* scale, float* shift, float* mean, float* rstd, int groups, void* stream);
/* Deferred running-statistics update of all nbn BatchNorms of ONE forward from the (mean, rstd) that
 * sv_bn_finalize saved (called with running_mean = NULL): table[bn] = {offset of the BN's
 * [scale|shift|mean|rstd] block (each `align`-padded) in bnbuf, running_mean offset, running_var offset
 * (both into bufs), C}; counts[bn] = samples per channel.  Lets the four forwards of a step run on several
 * streams while the momentum updates are still applied in the reference's order (1)(2)(3)(4).  groups (0 = 1): each
 * of the four arrays of a block is [groups][C] (the block padded to `align` as a whole); the groups' updates are
 * applied in group order.                                                                                  */
int sv_bn_running_update(const int32_t* table, const float* counts, int nbn, const float* bnbuf, float* bufs,
                         float eps, float momentum, int align, int groups, void* stream);
/* The same with an explicit order of the momentum updates: order[k] = the group whose statistics the k-th forward of the
 * reference left (HOST array of `groups` ints, a permutation; NULL = group order).  A batched launch may hold the forwards
 * in any order -- e.g. (1)(3)(2)(4), so that the two forwards whose reconstruction enters the loss are adjacent -- and the
 * running statistics still receive the updates as the reference applies them.                                         */
int sv_bn_running_update_ex(const int32_t* table, const float* counts, int nbn, const float* bnbuf, float* bufs,
                            float eps, float momentum, int align, int groups, const int32_t* order, void* stream);
/* eval-mode affine from running statistics (main_shot_vae.py:409-510 path)                         */
int sv_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, void* stream);
/* BatchNorm-apply + LeakyReLU / ReLU as a pass of its own: out = act(x * scale[c] + shift[c]), x / out [groups][M][C]
 * (contiguous channels, C % 8 == 0), scale / shift [groups][C].  The conv-like kernels fuse this into their load prologue;
 * for weight-heavy layers (the first ConvTranspose layers of the decoder: a few MB of activations against MBs of weights)
 * the prologue is re-applied once per output-channel tile and bounds the GEMM (VALU), so the step materialises it once and
 * the GEMM runs prologue-free.                                                                                        */
int sv_bn_act(int dtype, const void* x, const float* scale, const float* shift, float slope, int64_t M, int C, void* out,
              int groups, void* stream);
/* BatchNorm backward, second phase: dx = sum_br gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) (+res)
 * for one or two BN branches that share the input x; dgamma/dbeta accumulate (+=).  groups (0 = 1): G instances,
 * M = rows of ONE group; x / g / residual / dx [G][M][ld], mean / rstd [G][C], bsums [G][R][2C].               */
typedef struct {
    const void* g; const sv_acc_t* bsums; const float* gamma; float* dgamma; float* dbeta;
    int32_t replicas;           /* bsums is [replicas][2C]                                           */
    int32_t sparse;             /* wlog + 1 > 0: g is the data gradient of a STRIDE-2 layer written with sv_igemm_args::
                                   sparse_out -- defined at the even (row, column) positions of the 2^wlog-wide, 2^wlog-high
                                   maps only and taken as zero elsewhere (not read there); 0: dense;
                                   -(wlog + 1) < 0 (ABI 5): the same positions stored COMPACTLY, g [M / 4][ld] -- the data
                                   gradient of a stride-2 1x1 layer computed as a dense 1x1 product over the stride-2 grid
                                   (its raw tensor gathered with sv_gather_even)                                       */
} sv_bn_branch;
int sv_bn_bwd_apply(int dtype, int64_t M, int C, int ld, const void* x, const float* mean,
                    const float* rstd, float count, const sv_bn_branch* br, int nbranch,
                    const void* residual, void* dx, int groups, void* stream);

/* out [B][H/2][W/2][C] = in [B][2y][2x][C]: the even positions of an NHWC tensor (C a multiple of 8)                      */
int sv_gather_even(int dtype, const void* in, int B, int H, int W, int C, void* out, void* stream);

/* ---- K8 global average pool fused with the transition BN+LeakyReLU (vae.py:107,143) ------------- */
/* groups (0 = 1): B = ALL images, image b belongs to group b / (B / groups); coefficients [G][C], bsums [G][2C]     */
int sv_pool_fwd(int dtype, const void* x, const float* scale, const float* shift, float slope,
                int B, int HW, int C, int ld, float* feat, int groups, void* stream);
int sv_pool_bwd(int dtype, const void* x, const float* scale, const float* shift, float slope,
                const float* mean, const float* rstd, const float* dfeat, int B, int HW, int C, int ld,
                void* g, sv_acc_t* bsums, int groups, void* stream);

/* ---- K9 the three inference heads + LogSoftmax (vae.py:10-15,144-146) ----------------------------
 * W is [2*ldc+K][C] (rows: mean, log_sigma, disc), bias [2*ldc+K].                                   */
int sv_head_fwd(const float* feat, int B, int C, const float* W, const float* bias, int ldc, int K,
                float* mu, float* ls, float* la, void* stream);
/* dW/dbias accumulate (+=); dout_ws is a caller-owned [B][2*ldc+K] fp32 workspace                    */
int sv_head_bwd(const float* feat, int B, int C, const float* W, int ldc, int K, const float* la,
                const float* dmu, const float* dls, const float* dla, float* dfeat, float* dW,
                float* dbias, float* dout_ws, void* stream);

/* ---- K10/K11 reparameterisation sampler (vae.py:23-86) -------------------------------------------
 * mode 0: gumbel-softmax from u; 1: one-hot(label); 2: lam*onehot(label)+(1-lam)*onehot(label_mix).
 * latent is [B][Lpad] of `dtype` = [z | c | 0-pad]; csoft [B][K] fp32 keeps c for the backward.     */
/* lam_dev (optional, device scalar) overrides lam: keeps the step capturable into a hipGraph             */
int sv_sample_fwd(int dtype, const float* mu, const float* ls, const float* la, const float* eps,
                  const float* u, const int64_t* label, const int64_t* label_mix, float lam, const float* lam_dev, int mode,
                  float temperature, int B, int ldc, int K, int Lpad, void* latent, float* csoft,
                  void* stream);
/* dmu/dls/dla accumulate (+=) the sampler path of the latent gradient                              */
int sv_sample_bwd(int dtype, const void* dlatent, const float* ls, const float* eps, const float* csoft,
                  int mode, float temperature, int B, int ldc, int K, int Lpad,
                  float* dmu, float* dls, float* dla, void* stream);

/* ---- K14/K15 smooth-ELBO terms (lib/criterion.py:32-57) -----------------------------------------
 * out3 = [recon, KL_c, KL_d] already divided by B (and 2*sigma^2 for MSE); must be zeroed by the
 * caller.  x / x_rec are NCHW fp32 (API edge).                                                      */
int sv_elbo_fwd(const float* x, const float* x_rec, int64_t n_per_img, const float* mu, const float* ls,
                const float* la, int B, int ldc, int K, int bce, float x_sigma, float* out3, void* stream);
/* gout3: device pointer to the three upstream gradients; writes dx_rec, dmu, dls, dla (=, not +=)  */
int sv_elbo_bwd(const float* x, const float* x_rec, int64_t n_per_img, const float* mu, const float* ls,
                const float* la, int B, int ldc, int K, int bce, float x_sigma, const float* gout3,
                float* dx_rec, float* dmu, float* dls, float* dla, void* stream);

/* ---- K16 ClsCriterion (lib/criterion.py:97-108): out += -mean_b sum_c predict*label*weight ------ */
int sv_cls_fwd(const float* predict, const float* label, const float* weight, int B, int K, float* out,
               void* stream);
int sv_cls_bwd(const float* label, const float* weight, int B, int K, const float* gout, float* dpredict,
               void* stream);
/* posterior terms of main_shot_vae.py:319-321,359-361: out += (sum (mu-mt)^2 + sum (exp(ls)-st)^2)/B */
int sv_post_fwd(const float* mu, const float* ls, const float* mu_t, const float* sigma_t, int B, int D,
                float* out, void* stream);
int sv_post_bwd(const float* mu, const float* ls, const float* mu_t, const float* sigma_t, int B, int D,
                const float* gout, float* dmu, float* dls, void* stream);

/* ---- top-1 / top-k accuracy counts of valid() / test() (main_shot_vae.py:441-447, :493-499): for every row b the rank
 * of class label[b] among score[b][0..K) (descending; ties: lower index first); hits[0] += #(rank < 1),
 * hits[1] += #(rank < k).  score may be exp(disc_log_alpha) or disc_log_alpha itself (monotone).  hits: 2 floats,
 * accumulated (+=) across calls = across the batches of an epoch.                                              */
int sv_topk_hits(const float* score, const int64_t* label, int B, int K, int k, float* hits, void* stream);

/* ---- K17 mixup / label smoothing gather-lerp (lib/utils/mixup.py:22-25,36-39) --------------------
 * out[b] = lam*f(a[b]) + (1-lam)*f(a[index[b]]), f = exp when `exp_space` else identity.           */
int sv_mix_lerp(const float* a, const int64_t* index, float lam, const float* lam_dev, int B, int64_t row,
                int exp_space, float* out, void* stream);

/* random permutations from uniform keys (the device-side replacement of torch.randperm, mixup.py:20,34): perm[rank of
 * key i] = i within each of `batches` key vectors of length n (<= 16384); ties: the lower index first.                */
int sv_rank_permutation(const float* keys, int n, int batches, int64_t* perm, void* stream);

/* ---- fused loss stage of the SHOT-VAE step (main_shot_vae.py:289-323,340-363) ---------------------------------------
 * sv_shot_targets: the targets of the mixed forwards (2) and (4) in one launch -- label_smoothing / mixup_vae_data
 * (lib/utils/mixup.py:22-25,36-39) applied to the outputs of forwards (1) [mu_l, ls_l] and (3) [mu_u, ls_u, la_u]:
 *   sm_mu = lerp(mu_l, perm_l, lam_l), sm_sigma = lerp(exp(ls_l), ...), mx_mu / mx_sigma / mx_alpha likewise with
 *   (perm_u, lam_u), lab_mix = lam_l * onehot(label_l) + (1 - lam_l) * onehot(label_l[perm_l]) -- the ONE soft label
 *   that replaces the two ClsCriterion terms of :316-318 (the criterion is linear in its label).  lam_*_dev (optional
 *   device scalars) override lam_* (hipGraph capture).  All outputs fp32 [B][D] / [B][K].
 * sv_shot_compose: terms[0..9] = recon_l, KLc_l, KLd_l, recon_u, KLc_u, KLd_u, disc_post_l, cont_post_l, disc_post_u,
 *   cont_post_u (as written by sv_elbo_fwd / sv_cls_fwd / sv_post_fwd) -> terms[10] = loss_supervised, terms[11] =
 *   loss_unsupervised; coef[i] = d loss / d terms[i] (10 floats).
 * sv_shot_scale: gvec[i] = coef[i] * (upstream gradient of the loss term i belongs to): the `gout` operands of
 *   sv_elbo_bwd (gvec, gvec + 3), sv_cls_bwd (gvec + 6, gvec + 8) and sv_post_bwd (gvec + 7, gvec + 9).              */
typedef struct { float ew, kl_beta_c, kl_beta_d, cmi, dmi, pwm, ucw; } sv_shot_schedule;
int sv_shot_targets(const float* mu_l, const float* ls_l, const float* mu_u, const float* ls_u, const float* la_u,
                    const int64_t* label_l, const int64_t* perm_l, const int64_t* perm_u, float lam_l, const float* lam_l_dev,
                    float lam_u, const float* lam_u_dev, int B, int D, int K, float* sm_mu, float* sm_sigma, float* lab_mix,
                    float* mx_mu, float* mx_sigma, float* mx_alpha, void* stream);
int sv_shot_compose(float* terms, const sv_shot_schedule* sch, float* coef, void* stream);
int sv_shot_scale(const float* coef, const float* g_sup, const float* g_unsup, float* gvec, void* stream);

/* The whole loss stage of one grouped step in ONE call (the nine forward and six backward launches above, issued back to back
 * by the library: at ~5 us per kernel the stage is bound by the caller's per-launch host time otherwise).  Outputs of the four
 * batched forwards in the group order (1)(3)(2)(4): rec [2B][n_per_img] (groups (1), (3) only), mu / ls [4B][D], la [4B][K].
 * Upstream gradients 1 for both objectives (`(loss_supervised + loss_unsupervised).backward()`).
 * terms [12] must be ZEROED by the caller; coef [10], tgt [4BD + 2BK] and the four gradient tensors are plain outputs.    */
typedef struct {
    const float* rec; const float* mu; const float* ls; const float* la;
    const float* image_l; const float* image_u;
    const int64_t* label_l; const int64_t* perm_l; const int64_t* perm_u;
    float lam_l; const float* lam_l_dev; float lam_u; const float* lam_u_dev;
    int32_t B, D, K, bce; int64_t n_per_img; float x_sigma;
    sv_shot_schedule sch;
    float* terms; float* coef; float* tgt;
    float* d_rec; float* d_mu; float* d_ls; float* d_la;
} sv_shot_loss_args;
int sv_shot_loss_step(const sv_shot_loss_args* a, void* stream);

/* ABI 4: the same stage for a step whose forwards were NOT one batched launch of four equal groups -- the ragged last
 * batch of an epoch (main_shot_vae.py:280 zips a 4 000-label loader, 7 x 512 + 416, with the unlabelled one: B_l != B_u)
 * and --om (lib/utils/mixup.py:9-18: the pairing of forward (4) needs the outputs of forward (3)).  Every group has its own
 * pointers, order (1)(3)(2)(4) as above; groups (1), (2) have Bl rows, (3), (4) Bu rows; rec / d_rec: groups (1), (3).
 * tgt [2 Bl D + 2 Bu D + Bl K + Bu K].  sv_shot_targets2 = sv_shot_targets with the two batch sizes.                     */
typedef struct {
    const float* rec[2]; const float* mu[4]; const float* ls[4]; const float* la[4];
    const float* image_l; const float* image_u;
    const int64_t* label_l; const int64_t* perm_l; const int64_t* perm_u;
    float lam_l; const float* lam_l_dev; float lam_u; const float* lam_u_dev;
    int32_t Bl, Bu, D, K, bce, reserved0; int64_t n_per_img; float x_sigma;
    sv_shot_schedule sch;
    float* terms; float* coef; float* tgt;
    float* d_rec[2]; float* d_mu[4]; float* d_ls[4]; float* d_la[4];
} sv_shot_loss_args2;
int sv_shot_loss_step2(const sv_shot_loss_args2* a, void* stream);
int sv_shot_targets2(const float* mu_l, const float* ls_l, const float* mu_u, const float* ls_u, const float* la_u,
                     const int64_t* label_l, const int64_t* perm_l, const int64_t* perm_u, float lam_l, const float* lam_l_dev,
                     float lam_u, const float* lam_u_dev, int Bl, int Bu, int D, int K, float* sm_mu, float* sm_sigma, float* lab_mix,
                     float* mx_mu, float* mx_sigma, float* mx_alpha, void* stream);

/* ---- K18 optimal-match pairing (lib/utils/mixup.py:9-18,93-99): index[i] = argmin_{j!=rank0} ----
 * second-smallest entry of row i of the pairwise Gaussian-KL matrix.                               */
int sv_optimal_match(const float* mu, const float* ls, int B, int D, int64_t* index, void* stream);

/* ---- the one-stage smooth-ELBO conv-VAEs (BASELINE configs 1 / 5: smooth_vae_model/svhn_vae.py, mnist_vae.py) -----
 * sv_smooth_latent_fwd: everything between the fused head GEMM and the decoder (svhn_vae.py:137-208).  o [B][ldo] =
 *   [mean (Dc) | log-variance (Dc) | logits (Dd) | pad] of `dtype`; alpha = softmax(logits); z = mean + exp(logvar / 2) *
 *   eps in training mode, else mean; gs = Gumbel-softmax sample softmax((log(alpha + 1e-12) + g) / T), g = -log(-log(u +
 *   1e-12) + 1e-12) in training mode, else one-hot(argmax alpha); c = one-hot(label) when label != NULL, else gs.
 *   Outputs: mean, logvar [B][Dc], alpha, gs [B][Dd] (fp32), latent [B][Lpad] = [z | c | 0] of `dtype` (the decoder's
 *   input), latent32 [B][Dc + Dd] fp32 (the API's latent_sample).
 * sv_smooth_latent_bwd: d_o [B][ldo] from the latent's gradient dlat [B][Lpad] (`dtype`) and the loss' gradients
 *   dmean / dlogvar / dalpha (fp32, NULL = none); sample_path = 1 when c was the Gumbel-softmax sample.               */
int sv_smooth_latent_fwd(int dtype, const void* o, int ldo, const float* eps, const float* u, const int64_t* label,
                         float temperature, int training, int B, int Dc, int Dd, int Lpad, float* mean, float* logvar,
                         float* alpha, float* gs, void* latent, float* latent32, void* stream);
int sv_smooth_latent_bwd(int dtype, const void* dlat, int Lpad, const float* dmean, const float* dlogvar, const float* dalpha,
                         const float* logvar, const float* eps, const float* alpha, const float* gs, float temperature,
                         int training, int sample_path, int B, int Dc, int Dd, void* d_o, int ldo, void* stream);
/* reconstruction = tanh(f[..., :C]) of the decoder's NHWC output f [B][H][W][ld] as NCHW fp32 (svhn_vae.py:118-120), and
 * the backward d_f = d_out * (1 - out^2) in NHWC (channels >= C zero)                                                  */
int sv_tanh_to_nchw(int dtype, const void* f, int B, int C, int H, int W, int ld, float* out, void* stream);
int sv_tanh_to_nchw_bwd(int dtype, const float* d_out, const float* out, int B, int C, int H, int W, int ld, void* d_f, void* stream);
/* Trainer._loss_function (main_smooth_ELBO_svhn.py:228-310,312-335,368-388) in two launches: the raw reductions and the
 * composition.  terms [9] (zeroed by the caller): [0] num_pixels * MSE, [1] KL_c, [2] sum alpha log(alpha + 1e-12) / B
 * (KL_d = log D + [2]), [3] BCE(alpha, one-hot(label)) (label may be NULL), [4] the loss, [5..8] its four parts
 * (reconstruction, gamma_c |C_c - KL_c|, gamma_d |C_d - KL_d|, alpha_cls * BCE); coef [4] = d loss / d terms[0..3].
 * Capacities C = min((max - min) * steps / iters + min, max) (C_d also <= log D); steps_dev (optional device scalar)
 * replaces sch->steps (hipGraph replay).                                                                               */
typedef struct {
    float cont_min, cont_max, cont_iters, cont_gamma, disc_min, disc_max, disc_iters, disc_gamma, alpha_cls, steps;
} sv_smooth_schedule;
int sv_smooth_elbo_fwd(const float* data, const float* rec, int64_t n_per_img, const float* mean, const float* logvar,
                       const float* alpha, const int64_t* label, int B, int Dc, int Dd, const sv_smooth_schedule* sch,
                       const float* steps_dev, float* terms, float* coef, void* stream);
/* gradients of the loss (times the upstream gradient gout[0]) w.r.t. rec, mean, logvar, alpha (=, not +=)            */
int sv_smooth_elbo_bwd(const float* data, const float* rec, int64_t n_per_img, const float* mean, const float* logvar,
                       const float* alpha, const int64_t* label, int B, int Dc, int Dd, const float* coef, const float* gout,
                       float* d_rec, float* d_mean, float* d_logvar, float* d_alpha, void* stream);
/* torch.optim.Adam (no weight decay / amsgrad; main_smooth_ELBO_svhn.py:428) on a flat buffer: m, v = first / second
 * moments; step = 1-based update count (host) or step_dev (device scalar, hipGraph replay); g is scaled by grad_scale
 * first (1 / world after the gradient all-reduce).                                                                    */
int sv_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float step,
            const float* step_dev, float grad_scale, void* stream);

/* ---- K20 SGD(momentum, weight decay) on the flat parameter buffer (torch.optim.SGD semantics) ---- */
int sv_sgd(float* p, const float* g, float* v, int64_t n, float lr, float momentum, float weight_decay,
           float grad_scale, int first_step, void* stream);

/* ---- layout / packing helpers --------------------------------------------------------------------*/
/* NCHW fp32 [B,C,H,W] -> NHWC `dtype` [B,H,W,Cpad] (channels >= C zero-filled)                     */
int sv_nchw_to_nhwc(int dtype, const float* in, int B, int C, int H, int W, int Cpad, void* out, void* stream);
/* NHWC `dtype` [B,H,W,ld] (first C channels) -> NCHW fp32                                           */
int sv_nhwc_to_nchw(int dtype, const void* in, int B, int C, int H, int W, int ld, float* out, void* stream);
/* ---- K21 device-side input pipeline (lib/dataloader.py:58-70: Pad(4, reflect) -> RandomHorizontalFlip ->
 * RandomCrop(H) -> ToTensor) fused with the batch gather: sample b = data[index[b]] (uint8 HWC, [N][H][W][C]),
 * reflect-padded by `pad`, flipped if params[3b+2] != 0, cropped at (params[3b], params[3b+1]) in the padded image
 * (0 <= offset <= 2*pad), scaled by 1/255.  nhwc_cpad == 0: out = fp32 NCHW [B][C][H][W] (the model's API input);
 * nhwc_cpad > 0: out = `dtype` NHWC [B][H][W][nhwc_cpad], channels >= C zero (the stem convolution's input layout).
 * params == NULL: no augmentation (evaluation: ToTensor only).                                                  */
int sv_augment(int dtype, const uint8_t* data, const int64_t* index, const int32_t* params, int B, int H, int W, int C,
               int pad, int nhwc_cpad, void* out, void* stream);

/* master fp32 [N][T_orig][C] -> packed `dtype` per phase [n'][ntap][c'] (transpose swaps n and c)  */
int sv_repack(int dtype, const float* master, int N, int T_orig, int C, int transpose,
              const sv_geom* g, void* dst, void* stream);
/* ABI 4: the same packing straight from a tensor in ANOTHER layout -- element (n, tap, c) at src[n * sn + tap * st + c * sc],
 * zero beyond n_real / c_real (channel padding to the MFMA multiples): a torch Conv2d weight (OIHW: sn = c_real * T, st = 1,
 * sc = T), a ConvTranspose2d weight (IOHW: sn = T, st = 1, sc = n_real * T) or a Linear weight, without the intermediate
 * master copy (the smooth-ELBO models keep torch's own parameter tensors: smooth_vae_model/svhn_vae.py:62-132).          */
int sv_repack_strided(int dtype, const float* src, int n_real, int c_real, int64_t sn, int64_t st, int64_t sc, int N, int T_orig,
                      int C, int transpose, const sv_geom* g, void* dst, void* stream);

/* All packs of a network in one launch (68 sv_repack launches per optimizer step otherwise).  jobs: DEVICE array, one
 * entry per (layer, direction, phase) with taps, sorted by block0; a job owns ceil(size / 1024) consecutive blocks from
 * block0; total_blocks = their sum.  Offsets are in elements from master_base (fp32) / dst_base (`dtype`).             */
typedef struct {
    int64_t master_off, dst_off, size;      /* size = N * C * ntap elements of this phase's pack                     */
    int32_t N, T_orig, C, transpose, ntap, block0;
    int8_t torig[SV_MAX_TAPS];
} sv_repack_job;
int sv_repack_batch(int dtype, const float* master_base, const sv_repack_job* jobs, int njobs, int total_blocks,
                    void* dst_base, void* stream);

/* ABI 6: the same for parameters that live in torch's OWN layouts (the smooth-ELBO models keep nn.Conv2d / nn.ConvTranspose2d /
 * nn.Linear parameters): one launch gathers every weight pack -- and, with dtype SV_F32 into a float buffer, every padded bias
 * vector -- of a model straight from the parameter tensors, and one launch scatters the weight / bias gradients the kernels left
 * in master layout back into the parameters' .grad tensors (+=).  A job describes one (layer, direction, phase):
 *   element (n, tap t, c) of the layer  <->  ptr[n_hi * sn_hi + n_lo * sn_lo + torig[t] * st + c * sc],  n = n_hi * n_lo_count + n_lo
 * (n_lo_count > 1: a Linear layer whose output index is a permuted (c, y, x) flattening).  Gather: destination element i of the job
 * (pack order [n][tap][c], or [c][tap][n] when `transpose`) at dst_base[dst_off + i], zero beyond (n_real, c_real); with dst_ld the outer index has
 * its own stride (several jobs filling column ranges of one pack).  Scatter: the
 * master-layout gradient at src_base[dst_off + (n * ntap + t) * C + c] is ADDED to the parameter gradient element (torig[t] = t;
 * transpose = 1: the source at float offset dst_off is an array of DOUBLES -- the channel sums (sv_acc_t) of a data gradient's epilogue,
 * which are the bias gradient of the layer in front).
 * block0 = first block of the job; a job has ceil(size / 1024) blocks; jobs sorted by block0 (device array).                      */
typedef struct {
    float* ptr;                 /* the parameter (gather) / its gradient (scatter) */
    int64_t dst_off, size;
    int64_t dst_ld;             /* gather: destination stride of the OUTER index (n, or c when transposed); 0 = dense */
    int64_t sn_hi, sn_lo, st, sc;
    int32_t n_lo_count, N, C, ntap, transpose, n_real, c_real, block0;
    int8_t torig[SV_MAX_TAPS];
} sv_param_job;
int sv_param_gather(int dtype, const sv_param_job* jobs, int njobs, int total_blocks, void* dst_base, void* stream);
int sv_param_scatter_add(const sv_param_job* jobs, int njobs, int total_blocks, const float* src_base, void* stream);

/* ---- in-situ kernel timing (HIP events around launches of sv_igemm / sv_wgrad) -------------------
 * sv_prof_enable(1) starts recording; every launch is filed under the current tag (sv_prof_tag).
 * sv_prof_collect synchronises the device and returns, per tag, total milliseconds and launches.   */
int sv_prof_enable(int on);
int sv_prof_tag(int tag);
/* Launches the library issues on its own inside an entry point (the sv_bn_finalize of a folded sv_igemm whose kernel does not
 * derive the coefficients itself) are filed under `tag`; -1 (default): they are not recorded -- never as a second launch of
 * the layer's tag.                                                                                                        */
int sv_prof_nested_tag(int tag);
/* kind 0 = the folded BatchNorm finalisation (same as sv_prof_nested_tag), kind 1 = the materialised two-tensor prologue of
 * sv_igemm_args::x2 (the streaming launch sv_igemm issues for kernels that do not form it in their load path)             */
int sv_prof_nested_tag_kind(int kind, int tag);
int sv_prof_collect(int max_tags, double* ms, int* count);

/* ---- introspection (host only, no GPU): the compile-time "tile program" of the wide weight-gradient kernel.
 * halo_vectors = 3 or 4 (halo 16-byte vectors per thread; 4 for 32 x 32 maps).  items = int[180][3]: for the gap behind
 * every MFMA of one tile iteration (4 phases x 45) up to three item codes, 0 = none, 1000 + 41*vector + step = one
 * single-instruction step of the halo transform (step 40 = its LDS store), 2000 + k = dy DMA instruction k,
 * 3000 + v = global load of halo vector v (tile after next, into the second register set), 5000 + 4*v + d = the move of
 * dword d of vector v from the second register set into the first.  The kernel has no counted vmcnt wait (one vmcnt(0)
 * in front of its barrier, which follows gap 134): waits = int[5], [0..3] = 0, [4] = the gap of the last VMEM instruction
 * before that barrier.                                                                                      */
int sv_debug_wgrad_tile_program(int halo_vectors, int* items, int* waits);

/* The chunk program of the wide forward / data-gradient kernel (conv3x3x.hip): items = int[180][6], the gap behind every
 * MFMA of one 32-channel chunk (9 taps x 20).  Codes: 1000 + 41*vector + step (BatchNorm pass), 2000 + 3*slice + i (weight
 * DMA instruction; slice 0..3 = this chunk's taps 5..8, 4..8 = the next chunk's taps 0..4), 3000 + v (halo load),
 * 3500 + q (coefficient load), 4000 + v / 4500 (vmcnt waits), 5000 (stage flip).  waits = int[10]: vmcnt of the waits
 * before vectors 0..5, before the coefficients (each = the YOUNGER REGISTER LOADS only), and before the barriers after
 * taps 1, 4, 7 (each = the younger LDS-DMA instructions only: zero).                                               */
int sv_debug_conv_chunk_program(int* items, int* waits);

/* ---- dispatcher options (process-wide; set them between launches, not concurrently with them) --------------------
 * SV_OPT_DISABLE_MASK: OR of SV_K_* bits; a set bit routes the layers a specialised kernel would take to the next more
 * general one (sv_igemm: conv3x3x -> conv3x3w -> conv3x3 / conv3x3p / conv3x3m -> halo -> the generic gather-GEMM; sv_wgrad:
 * wgrad3x3w -> wgrad3x3 -> hwgrad -> the generic weight-gradient kernel).  Default 0.  The parity tests use it to compare every
 * specialised kernel with the general one on the same inputs (conv3x3x against conv3x3w bit for bit).
 * SV_OPT_WIDE_MIN_BLOCKS: minimum grid (blocks) for which the 256-pixel wide-tile kernels are chosen; default 256 (one
 * block per CU).  Tests set 1 to reach those kernels at small batch sizes.
 * SV_OPT_HALO_ALL: 1 = the LDS-halo gather-GEMM (halo.hip) takes every geometry it covers, not only the layers it is
 * faster on (tests: the kernel's whole range against the references).  Default 0.
 * SV_OPT_PERSISTENT_BLOCKS: block budget of the persistent narrow 3x3 kernels (conv3x3p, wgrad3x3), shared among the
 * groups of a batched launch.  Default 512 (two blocks per CU); tools/tune_blocks.py sweeps it.
 * SV_OPT_DETERMINISTIC: 1 = every floating-point accumulation of the library has a FIXED summation order, so that two runs
 * on the same inputs agree bit for bit (parity tests, reproducible training; since round 4 about 1.4x the step time of the
 * default mode on WRN-28-2 and 1.8x on WRN-28-10; the accumulator replicas grow with the grid):
 *   - BatchNorm statistics / BatchNorm-backward sums of the conv-like kernels: no LDS float atomics, every wave of every
 *     block adds its partial sums to a replica of its own, or (conv3x3w / conv3x3x) the block adds its waves' private sums in a
 *     fixed order (ONE adder per address; needs replicas >= 4 * blocks, see sv_igemm_query_blocks; sv_igemm fails with
 *     SV_E_ARG otherwise); the consumers sum the replicas in index order (sv_bn_bwd_apply after a 256 : 1 pre-pass);
 *   - weight gradients: partial slabs in the workspace + the ordered slab reduction (3x3 stride 1 and the tap-fused thin
 *     layers: a slab per block; the generic and the cooperative wide kernel: a zeroed slab per M range); without a workspace
 *     of at least two slabs ONE M range; the groups of a batched launch one after the other (a single adder per weight);
 *   - the small reductions (pool backward, column sums, loss terms) run in two passes: the blocks of the first own a slot
 *     each in a scratch ring the library allocates at the first use (32 MB; that first call must not be inside a stream
 *     capture), the second adds the slots in index order; the head weight gradient one slice after the other; BatchNorm
 *     dgamma / dbeta of a batched launch by one block, the groups in index order.
 * Default 0.
 * SV_OPT_ENABLE_MASK: OR of SV_K_* bits of kernels that are OFF by default (none at present: the 64 x 64-block narrow weight
 * gradient of round 4 lost to the 64 x 32 form inside the step twice and left the tree).                                      */
enum { SV_OPT_DISABLE_MASK = 0, SV_OPT_WIDE_MIN_BLOCKS = 1, SV_OPT_HALO_ALL = 2, SV_OPT_PERSISTENT_BLOCKS = 3,
       SV_OPT_DETERMINISTIC = 4, SV_OPT_ENABLE_MASK = 5 };
enum { SV_K_CONV3X3 = 1, SV_K_CONV3X3P = 2, SV_K_CONV3X3M = 4, SV_K_CONV3X3W = 8, SV_K_CONV3X3X = 16,
       SV_K_WGRAD3X3 = 32, SV_K_WGRAD3X3W = 64, SV_K_IGEMM_KV2 = 128, SV_K_HALO = 256, SV_K_HALOP = 512, SV_K_HWGRAD = 1024, SV_K_IGEMM_BIG = 2048, SV_K_WGRAD_WIDE = 4096, SV_K_IGEMM_ALIGNED = 8192, SV_K_IGEMM_DMA = 16384, SV_K_WGRAD_INCR = 32768, SV_K_WGRAD3X3M = 65536, SV_K_TCONVR = 131072, SV_K_TCONVR_EX = 262144, SV_K_SCONV = 524288, SV_K_PCONV = 8388608, SV_K_THCONV = 16777216, SV_K_THWGRAD = 67108864, SV_K_S2WGRAD = 134217728 };
int sv_set_option(int key, int value);
int sv_get_option(int key);          /* -1 for an unknown key */

/* Stream fork: everything enqueued on `from` so far happens-before whatever is enqueued on `to` afterwards (the step's weight
 * gradients run on a side stream beside the data gradients).  hipEventRecord + hipStreamWaitEvent on an event from a pool the
 * library owns; light != 0 creates the events with hipEventDisableSystemFence: the two streams are on one device, the
 * system-scope release an ordinary event performs in `from` (cache writeback + an idle gap in front of the next kernel, ~6 us per
 * fork on MI355X) is not needed for them.  Works under stream capture (the record becomes a graph edge).                  */
int sv_stream_fork(void* from, void* to, int light);
/* The device-side form of the fork (sv_igemm_args::start_flag).  sv_stream_flag_next: the flag word the library keeps for
 * `stream` (device memory, allocated and zeroed at the first call -- not inside a stream capture) and the next value of its
 * sequence; the caller passes both to the launch that is to signal and to sv_stream_wait_flag on the other stream.
 * sv_stream_wait_flag: enqueues a one-wave kernel on `stream` that returns once (int32)(*flag - value) >= 0; it gives up after
 * ~3 s (a signalling launch that never ran) and counts that in a STICKY counter in host-mapped memory: sv_flag_timeouts()
 * reads it without a copy or a synchronisation, so the host layer checks it on every step and raises (a weight gradient that
 * ran behind a failed wait read unfinished operands; the step is invalid).  sv_flag_timeouts_reset() clears it once the
 * caller has handled the condition (e.g. switched to sv_stream_fork).
 * NOT for environments that serialise kernel dispatch across streams (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL,
 * HIP_LAUNCH_BLOCKING): the waiting kernel may then be dispatched in front of the one it waits for -- fork with
 * sv_stream_fork there (the Python host layer checks the environment).                                                     */
int sv_stream_flag_next(void* stream, uint32_t** flag, uint32_t* value);
int sv_stream_wait_flag(void* stream, const uint32_t* flag, uint32_t value);
int sv_flag_timeouts(void);
int sv_flag_timeouts_reset(void);

int sv_version(void);
const char* sv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
