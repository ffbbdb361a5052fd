// ConvTranspose2d(4, 2, 1) forward with the weights REGISTER-resident (decoder.py:40-47, the 128 -> 64 layer, 8x8 -> 16x16;
// fused with the BatchNorm + ReLU in front of it -- decoder.py:37-38 -- as the load prologue and with the statistics of the
// BatchNorm behind it -- decoder.py:48 -- as the epilogue: the sv_igemm contract).  gfx950.
//
// Why a kernel of its own: this layer's 262 KB of weights fit no LDS-resident form, so the LDS-halo kernel (halo.hip) re-staged
// 128 KB of them per 128-position tile -- 105-112 us for 34 GFLOP and 100 MB of traffic, 0.12 of the layer's roofline, the
// largest single launch of the config-2 step behind the body.  The register file (512 KB per CU) does hold them:
//   * a block = eight waves = the four sub-pixel phases of the transposed convolution x the two 32-channel halves of the
//     output.  A wave keeps its slice of the weights -- [32 output channels][4 taps x 128 channels] bf16 = 32 KB = 128 VGPRs
//     per lane -- as the A operands of v_mfma_f32_32x32x16_bf16 for the lifetime of the (persistent) block: loaded once, 32 x
//     16 bytes per lane (two waves per SIMD: one wave's epilogue and staging run under the other's MFMAs; the whole matrix
//     of a phase in ONE wave, 256 VGPRs, spilled 170 registers next to the accumulators and the statistics);
//   * the input image (8 x 8 x 128, 16 KB) is staged ONCE per image for all four phases: BatchNorm + ReLU applied once per
//     element on the way into a zero-bordered 10 x 10 LDS image, stored as 8 PLANES of 16 channels ([k-step][pixel][32 B]:
//     the k-step of a B-fragment read is an immediate offset, a tap a per-lane base -- 8 address registers for 64 reads; the
//     rows are 12 pixels apart and the two 16-byte halves of a pixel's 32 bytes are swapped on odd rows, which makes every
//     ds_read_b128 lane group -- {0-3, 12-15, 20-27}, ...: x 0-3 of two rows and x 4-7 of the other two -- hit each 32-byte
//     slot of the 256-byte bank row twice, in different halves: conflict-free for every tap shift; the planes are padded
//     by 32 B so that the eight vectors a staging ds_write_b128 group stores land on 32 different banks); two images: the next one is loaded (registers) during the
//     MFMAs of this one and written behind them, one barrier per image;
//   * per image a wave runs 2 pixel tiles x 32 k-steps = 64 MFMAs on 64 ds_read_b128 (a tap is an LDS address offset, every
//     A fragment is a register) -- one read per MFMA, where the LDS array sustains two;
//   * epilogue out of the accumulators: per-lane partial sums of y and y^2 (fp32, over the block's images), bf16 stores of 4
//     channels (8 bytes) per lane -- the lane pair (l, l + 32) covers 16 contiguous bytes, the four stores of a pixel by each of
//     the two waves of a phase one whole 128-byte line; the sums meet across lanes and waves once per block and go to the double accumulators
//     (sv_acc_t) by one atomic add per channel and block.
// Same sv_geom / packed weights / sv_igemm_args contract as sv_igemm: a fast path inside it (SV_K_TCONVR disables).
#include <type_traits>

#include "common.h"

#ifndef SV_TCONVR_DBG
#define SV_TCONVR_DBG 0         // ablation switches (tools/probes/tconvr_ablate.sh): 1 no MFMA loop, 2 no output stores, 4 no next-image load / staging, 8 no statistics, 16 no LDS fragment reads
#endif
#ifndef SV_TCONVR_PIPE
#define SV_TCONVR_PIPE 1        // the epilogue of a pixel tile in the MFMA gaps of the other one (see the kernel)
#endif
#ifndef SV_TCONVR_PIPE_STAGE
#define SV_TCONVR_PIPE_STAGE 0  // 1: ... and the staging of the next image in the gaps of the second tile's MFMAs (measured WORSE: 46.5-48 vs 44.7 us -- the wait for the vectors and 16 spilled registers land inside the MFMA stream)
#endif
#ifndef SV_TCONVX16_TP
#define SV_TCONVX16_TP 2        // pixel tiles per pass of the 16 x 16 data-gradient kernel (1: 85.8 vs 83.8 us)
#endif
#ifndef SV_TCONVR_WMAP
#define SV_TCONVR_WMAP 0        // 1: (phase, tile) = (wave >> 1, wave & 1) instead of (wave & 3, wave >> 2)
#endif
#ifndef SV_TCONVR_KL
#define SV_TCONVR_KL (SV_TCONVR_PIPE ? 12 : 8)   // of a wave's 32 A fragments the last KL are read from LDS (the registers they would take spill otherwise)
#endif
#ifndef SV_TCONVR_PD
#define SV_TCONVR_PD 2          // the B fragments are requested this many groups ahead of their MFMAs
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN, int NOUT, int H>
struct tconvr_cfg {
    static constexpr int HP = H + 2, PITCH = 12, PLANE = HP * PITCH * 32 + 32, TILE = (CIN / 16) * PLANE;
    static constexpr int NT = NOUT / 32, MT = H * H / 32, KC = CIN / 16, KS = 4 * KC;
    static constexpr int NW = 4 * NT, NTH = 64 * NW;              // waves: (phase, 32-channel tile)
    static constexpr int VPT = H * H * (CIN / 8) / NTH;          // 16-byte vectors of an image per thread
    static constexpr int OFF_WSUM = 2 * TILE;                      // [4 phases][2][NOUT] floats
    static constexpr int OFF_COEF = OFF_WSUM + 4 * 2 * NOUT * 4;   // [2][CIN] floats: prologue scale, shift
    static constexpr int KL = SV_TCONVR_KL;                        // k-steps whose A fragments live in LDS instead of registers
    static constexpr int OFF_WLDS = OFF_COEF + 2 * CIN * 4;        // [NW][KL][64 lanes][16 B]
    static constexpr int LDS = OFF_WLDS + NW * KL * 1024;
    static_assert(2 * TILE < 65536, "the plane and image offsets are ds_read immediates");
    static_assert(H == 8, "pixel <-> lane mapping below: 4 rows of 8 per 32-pixel tile");
    static_assert((KS - KL) * 4 <= 128 && KL >= 0 && KL < KS, "a wave's weights in at most 128 VGPRs");
    static_assert(LDS <= 160 * 1024 && NW * 32 * 272 <= LDS, "LDS budget (the weight staging area of the start-up lies over everything)");
    static_assert(NTH == 512 && VPT >= 1, "eight waves");
};

// EX = true: the same geometry as a DATA GRADIENT -- the stride-2 3x3 convolution 64 -> 128 of the WideResNet (wideresnet.py:29-30,
// 41-43: dy 8x8x128 -> dx 16x16x64): phases of 1 / 2 / 2 / 4 taps (a phase's missing taps are skipped, the waves are dealt so
// that every SIMD gets 5 or 4 taps' worth of MFMAs), no load prologue, and sv_igemm's activation-backward epilogue: the raw
// tensor at the output positions (requested at the start of the interval with the store's own addressing, handed to the
// accumulator layout by the same v_permlane32_swap), g * act'(BatchNorm(x)) stored, sum g and sum g * xhat to bsums.
template <int CIN, int NOUT, int H, bool EX>
__global__ __launch_bounds__(512, 1) void tconvr_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    typedef tconvr_cfg<CIN, NOUT, H> C;
    constexpr int PITCH = C::PITCH, TILE = C::TILE, PLANE = C::PLANE, MT = C::MT, KC = C::KC, KS = C::KS, VPT = C::VPT, NTH = C::NTH, KL = C::KL, KR = KS - KL;
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave's phase and 32-channel tile (EX: phases 3 3 1 1 0 0 2 2 -- waves w and w + 4 share a SIMD: 4 + 1 and 2 + 2 taps)
    const int ph = EX ? (0x22001133 >> (4 * wave)) & 3 : (SV_TCONVR_WMAP ? wave >> 1 : wave & 3);
    const int nt = EX ? wave & 1 : (SV_TCONVR_WMAP ? wave & 1 : wave >> 2);
    const int q = lane & 31, h = lane >> 5;
    const sv_phase& P = g.phase[ph];
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    const int nimg = g.B;
    int img = blockIdx.x;
    const int ntap = EX ? __builtin_amdgcn_readfirstlane(P.ntap) : 4;

    // ---- the first image's vectors are requested before anything else
    const int sc = tid & 15;                 // this thread's channel chunk (channels 8 sc ..)
    bf16x8 xr[VPT];
    const int xoff = (tid >> 4) * CIN + 8 * sc;          // (a uniform image base + a 32-bit lane offset: no 64-bit address registers)
    auto request = [&](int im) {
        const bf16* const xi = X + (int64_t)im * (H * H * CIN);
#pragma unroll
        for (int i = 0; i < VPT; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(xi + xoff + (NTH / 16) * CIN * i);
    };
    if (img < nimg) request(img);

    // ---- weights of this wave's phase: A fragments (row = output channel 32 nt + q, k = 16 ks + 8 h ..)
    // (the last KL of them in this wave's own LDS slice, lane-linear: a conflict-free read at an immediate offset).
    // Fetched through LDS: a fragment load straight from global memory takes 32 bytes of each of 32 rows per instruction --
    // four instructions per 128-byte line, 8 waves x 32 KB against a 32 KB L1: measured 27 us of start-up per block (1 MB of
    // L2 requests per CU).  Instead every wave reads its [32 rows][256 B] slice of 8 k-steps in whole lines (16 lanes per
    // row), parks it in LDS (rows 272 B apart: the fragment read -- row = lane -- is conflict-free) and picks its fragments
    // up from there; the LDS ops of one wave execute in order, so no barrier is needed inside a wave's private region.
    bf16x8 wf[KR], wtail[KL];
    char* const wlds = smem + C::OFF_WLDS + wave * (KL * 1024) + lane * 16;
    {
        static_assert(KS % 8 == 0 && 4 * CIN * 2 == 1024, "weight staging: 8 k-steps = 256 B of a 1 KB row per pass");
        static_assert(KC == 8, "a pass = a tap");
        const int wrow = ntap * (CIN * 2);                  // bytes of a weight row [ntap][CIN]
        const char* const Wb = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(a.w) + P.w_off) + (32 * nt) * wrow;
        char* const wst = smem + wave * (32 * 272);
        const int vrow = lane >> 4, vcol = lane & 15;
#pragma unroll
        for (int pass = 0; pass < KS / 8; ++pass) {
            bf16x8 tmp[8];
            // (a tap the phase does not have: its fragments are never used -- the k loop skips the tap -- so the pass re-reads tap 0)
            const int wpass = (!EX || pass < ntap) ? pass : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) tmp[i] = *reinterpret_cast<const bf16x8*>(Wb + (vrow + 4 * i) * wrow + 256 * wpass + 16 * vcol);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<bf16x8*>(wst + (vrow + 4 * i) * 272 + 16 * vcol) = tmp[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8 f = *reinterpret_cast<const bf16x8*>(wst + q * 272 + (2 * j + h) * 16);
                const int ks = 8 * pass + j;
                if (ks < KR) wf[ks < KR ? ks : 0] = f;
                else wtail[ks >= KR ? ks - KR : 0] = f;
            }
        }
    }
    // ---- prologue coefficients of this thread's 8 channels
    // (kept in LDS, re-read per image: 16 registers that the MFMA loop needs)
    const bool has_pro = a.pro_scale != nullptr;
    float* const coef = reinterpret_cast<float*>(smem + C::OFF_COEF);
    const float slope = has_pro ? a.pro_slope : 1.f;
    __syncthreads();                              // every wave is done with the weight staging area (it lies over what follows)
#pragma unroll
    for (int j = 0; j < KL; ++j) *reinterpret_cast<bf16x8*>(wlds + j * 1024) = wtail[j];
    // (BatchNorm finalisation folded into this launch -- sv_igemm_args::fold_*: every block derives the coefficients itself, block 0
    //  of a group stores the four vectors; the scratch lies in the image area, zeroed below)
    if (!EX && a.fold_stats) sv_bn_fold_block512(a, CIN, reinterpret_cast<double*>(smem), coef, blockIdx.x == 0);
    else if (has_pro && tid < 2 * CIN) coef[tid] = (tid & 1) ? a.pro_shift[tid >> 1] : a.pro_scale[tid >> 1];      // [CIN] pairs {scale, shift}
    if (EX && tid < NOUT) {       // (EX: the area holds the epilogue's per-channel constants instead -- no prologue there)
        const float rs = a.ex_rstd[tid];
        reinterpret_cast<f32x4*>(coef)[tid] = f32x4{a.ex_scale[tid], a.ex_shift[tid], rs, -a.ex_mean[tid] * rs};     // xhat = x * rstd - mean * rstd
    }
    // ---- both LDS images zeroed once: the border stays zero (the padding of the convolution as data)
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * TILE / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    // staging destinations of this thread's vectors (pixel (tid >> 4) + 32 i of the bordered image, plane sc >> 1, half sc & 1)
    // (vector i: 32 pixels = 4 rows further -- same row parity, a constant offset)
    static_assert(NTH / 16 == 32, "the vectors of a thread are 4 image rows apart");
    int sdst;
    {
        const int p = tid >> 4, yy = (p >> 3) + 1, xx = (p & 7) + 1;
        sdst = (sc >> 1) * PLANE + (yy * PITCH + xx) * 32 + (((sc ^ yy) & 1) << 4);
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
        if (has_pro) {
            f32x4 s0, s1, t0, t1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(coef + 16 * sc + 4 * j);        // channels 8 sc + 2 j, + 1
                (j < 2 ? s0 : s1)[2 * (j & 1)] = c[0]; (j < 2 ? t0 : t1)[2 * (j & 1)] = c[1];
                (j < 2 ? s0 : s1)[2 * (j & 1) + 1] = c[2]; (j < 2 ? t0 : t1)[2 * (j & 1) + 1] = c[3];
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = bn_act8(xr[i], s0, s1, t0, t1, slope);
        } else {
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = xr[i];
        }
    };
    // B-fragment sources: pixel q of tile mt at tap t, channels 16 kc + 8 h ..  ->  rb[t] + mt * (4 rows) + kc * PLANE
    int rb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int yy = (q >> 3) + P.dy[t] + 1, xx = (q & 7) + P.dx[t] + 1;
        rb[t] = (yy * PITCH + xx) * 32 + (((h ^ yy) & 1) << 4);
    }
    // output positions of this lane's pixels (element offsets within an image)
    const int opix = (((q >> 3) * g.osy + P.ooy) * g.Wout + (q & 7) * g.osx + P.oox) * g.ldo + 32 * nt + 8 * h;
    const int otile = 4 * g.osy * g.Wout * g.ldo;           // tile mt: four grid rows further (uniform)
    const int64_t ostride = (int64_t)g.Hout * g.Wout * g.ldo;
    const bool want_stats = EX || a.stats != nullptr;
    const float ex_slope = EX ? a.ex_slope : 1.f;
    const bf16* __restrict__ EXP = reinterpret_cast<const bf16*>(a.ex);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 exr[EX ? MT : 1][2];                       // (EX) the raw tensor at this lane's two 16-byte store positions of each tile
    auto request_ex = [&](int im) {
        const bf16* const eimg = EXP + (int64_t)im * ostride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) exr[EX ? mt : 0][gp] = *reinterpret_cast<const u32x4*>(eimg + mt * otile + opix + 16 * gp);
    };
    float ps1[16], ps2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ps1[e] = ps2[e] = 0.f;

    // (the weights have landed HERE: without this the compiler places their counted waits -- vmcnt(23) ... vmcnt(0) -- inside the
    //  MFMA stream of the loop body, where vmcnt(0) also waits for the next image's loads and the previous image's stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
    __syncthreads();                              // the zeroed images
    if (img < nimg) stage(0);
    __syncthreads();

    // (diagnostic build, SV_TCONVR_DBG & 32: s_memtime stamps of block (0, 0), summed per segment and wave into the buffer
    //  sv_igemm_args::fold_mean points at -- tools/probes/tconvr_stamps.py; no stamp executes in the real kernel)
    unsigned long long tseg[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int seg) {
        if (SV_TCONVR_DBG & 32) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (seg >= 0) tseg[seg] += t - tprev;
            tprev = t;
        }
    };
    stamp(-1);
    // One image interval of a wave: request the next image (registers) -> 64 MFMAs on the staged image -> epilogue -> stage the
    // next image -> barrier.  Stamped (tools/probes/tconvr_stamps.sh), the vector work OUTSIDE the MFMA loop -- the epilogue's
    // ~150 VALU instructions, the staging's ~100 -- ran for as long as the loop itself and overlapped with nothing: VALU and MFMA
    // share a SIMD's vector issue, and a second wave's VALU stream got no slots beside the first wave's back-to-back MFMAs
    // (running the two waves of a SIMD in opposite order, epilogue-then-MFMA against MFMA-then-epilogue, changed nothing: 44.0
    // vs 43.4 us).  What does hide is vector work placed in the issue gaps of the wave's OWN MFMAs (an MFMA holds the vector
    // issue for 8 of its 32 cycles): SV_TCONVR_PIPE = 1, below.
    f32x16 acc[MT];
    // acc[mt][4 gq + e] = channel 32 nt + 8 gq + 4 h + e of pixel q of tile mt.  Stores widened to 16 bytes: v_permlane32_swap
    // hands the upper half-wave's channels 8 gq + 4 .. 7 to the lower one and the lower half-wave's channels 8 (gq + 1) .. + 3 to
    // the upper one -- lanes 0-31 then hold channels 8 gq .. 8 gq + 7, lanes 32-63 channels 8 gq + 8 .. 8 gq + 15: half the
    // store instructions, 64 contiguous bytes per pixel and wave.
    uint32_t pk[2][2];
    // element step e (0 .. 15) of tile mt for image im
    auto epi_step = [&](int mt, int e, int im) __attribute__((always_inline)) {
        const float v = acc[mt][e];
        if (want_stats && !(SV_TCONVR_DBG & 8)) {
            ps1[e] += v;
            ps2[e] += v * v;
        }
        if (e & 1) {
            typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bf16x2 pr = {(bf16)acc[mt][e - 1], (bf16)v};
            pk[(e >> 2) & 1][(e >> 1) & 1] = __builtin_bit_cast(uint32_t, pr);
        }
        if ((e & 7) == 7) {
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const auto r = __builtin_amdgcn_permlane32_swap(pk[0][e2], pk[1][e2], false, false);
                pk[0][e2] = r[0];
                pk[1][e2] = r[1];
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
            bf16* const oimg = O + (int64_t)im * ostride;
            if (!(SV_TCONVR_DBG & 2) || o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(oimg + mt * otile + opix + 16 * (e >> 3)) = o;
        }
    };
    // SV_TCONVR_PIPE: the staging of the next image too, in 8 steps -- element e of the thread's vectors (one channel: one
    // {scale, shift} pair from LDS, requested a step ahead) -- and the two LDS stores behind the last one
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 cq[2];
    auto stg_fetch = [&](int e) { cq[e & 1] = *reinterpret_cast<const f32x2*>(coef + 16 * sc + 2 * e); };
    auto stg_step = [&](int e, int buf) {
        if (has_pro) {
            const f32x2 c = cq[e & 1];
            if (e + 1 < 8) stg_fetch(e + 1);
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const float u = (float)xr[i][e] * c[0] + c[1];
                xr[i][e] = (bf16)fmaxf(u, u * slope);
            }
        }
        if (e == 7) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = xr[i];
        }
    };
    auto epilogue = [&](int mt, int im) {
#pragma unroll
        for (int e = 0; e < 16; ++e) epi_step(mt, e, im);
    };
    // (EX) activation-backward epilogue of image im, both tiles: per channel one {scale, shift, rstd, -mean * rstd} from LDS
    auto epilogue_ex = [&](int im) __attribute__((always_inline)) {
        bf16* const oimg = O + (int64_t)im * ostride;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            uint32_t xw[MT][2][2], ow[MT][2][2];        // [tile][gq = 2 gp + k][dword]: raw operand / packed result
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    // loaded: lanes 0-31 channels 16 gp + 0 .. 7, lanes 32-63 channels 16 gp + 8 .. 15; wanted: 8 gq + 4 h + e
                    // (lanes 32-63 of the first half <-> lanes 0-31 of the second: the lower lane of a pair keeps its first half and
                    //  receives the upper lane's first half as group 2 gp + 1; the upper lane receives the lower one's second half as
                    //  group 2 gp and keeps its own second half)
                    const auto r = __builtin_amdgcn_permlane32_swap(exr[EX ? mt : 0][gp][d], exr[EX ? mt : 0][gp][2 + d], false, false);
                    xw[mt][0][d] = r[0];
                    xw[mt][1][d] = r[1];
                }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int e0 = 4 * (2 * gp + k) + 2 * d;           // accumulator elements e0, e0 + 1 = this dword's two channels
                    const f32x4 c0 = reinterpret_cast<const f32x4*>(coef)[32 * nt + 8 * (2 * gp + k) + 4 * h + 2 * d];
                    const f32x4 c1 = reinterpret_cast<const f32x4*>(coef)[32 * nt + 8 * (2 * gp + k) + 4 * h + 2 * d + 1];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const uint32_t w = xw[mt][k][d];
                        const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
                        const float g0 = acc[mt][e0] * ((x0 * c0[0] + c0[1] > 0.f) ? 1.f : ex_slope);
                        const float g1 = acc[mt][e0 + 1] * ((x1 * c1[0] + c1[1] > 0.f) ? 1.f : ex_slope);
                        ps1[e0] += g0;
                        ps2[e0] += g0 * (x0 * c0[2] + c0[3]);
                        ps1[e0 + 1] += g1;
                        ps2[e0 + 1] += g1 * (x1 * c1[2] + c1[3]);
                        typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 pr = {(bf16)g0, (bf16)g1};
                        ow[mt][k][d] = __builtin_bit_cast(uint32_t, pr);
                    }
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto r = __builtin_amdgcn_permlane32_swap(ow[mt][0][d], ow[mt][1][d], false, false);
                    ow[mt][0][d] = r[0];
                    ow[mt][1][d] = r[1];
                }
                const u32x4 o = {ow[mt][0][0], ow[mt][0][1], ow[mt][1][0], ow[mt][1][1]};
                *reinterpret_cast<u32x4*>(oimg + mt * otile + opix + 16 * gp) = o;
            }
        }
    };
    // SV_TCONVR_PIPE: the two pixel tiles of an image run one after the other (32 MFMAs each on one accumulator set), and the
    // epilogue of the tile that has just finished rides in the MFMA gaps of the other one -- tile 1 of image i - 1 under tile 0
    // of image i (its accumulators stay live across the barrier), tile 0 of image i under tile 1: no second accumulator set.
    // bufc: LDS image of image im; prevc: tile 1 of image `prev` awaits its epilogue
    constexpr bool PIPE_STAGE = (SV_TCONVR_PIPE != 0) && (SV_TCONVR_PIPE_STAGE != 0);
    auto body = [&](auto bufc, auto prevc, int im, int prev) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr bool PREV = decltype(prevc)::value;
        const int nxt = im + gridDim.x;
        const bool has_next = nxt < nimg;
        if (has_next && !(SV_TCONVR_DBG & 4)) request(nxt);
        stamp(0);
        // the B fragments of step i + PD are requested before the MFMA(s) of step i; the scheduling barrier keeps the compiler
        // from hoisting all 64 reads to the front (256 registers, spilled)
        constexpr int PD = SV_TCONVR_PD, NB = PD + 1;           // request distance in k-steps, ring of NB fragment sets
        constexpr int TPS = SV_TCONVR_PIPE ? 1 : MT;            // pixel tiles per pass over the k-steps
#pragma unroll
        for (int pass = 0; pass < MT / TPS; ++pass) {
            bf16x8 bfr[NB][TPS], afr[NB];
            auto fetch = [&](int ks, bf16x8 (&dst)[TPS], bf16x8& adst) {
                const int t = ks / KC, kc = ks % KC;
#pragma unroll
                for (int i = 0; i < TPS; ++i)
                    dst[i] = *reinterpret_cast<const bf16x8*>(smem + rb[t] + (BUF * TILE + kc * PLANE + (pass * TPS + i) * (4 * PITCH * 32)));
                if (ks >= KR) adst = *reinterpret_cast<const bf16x8*>(wlds + (ks - KR) * 1024);
                if ((SV_TCONVR_DBG & 16) && ks >= PD) {     // (ablation: no LDS reads in the steady state -- stale fragments)
#pragma unroll
                    for (int i = 0; i < TPS; ++i) dst[i] = bfr[0][0];
                    adst = bfr[0][0];
                }
            };
#pragma unroll
            for (int i = 0; i < TPS; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[pass * TPS + i][e] = 0.f;
            if (!(SV_TCONVR_DBG & 1)) {
#pragma unroll
                for (int d = 0; d < PD; ++d) fetch(d, bfr[d % NB], afr[d % NB]);
            }
#pragma unroll
            for (int ks = 0; ks < ((SV_TCONVR_DBG & 1) ? 0 : KS); ++ks) {
                if (ks + PD < KS) fetch(ks + PD, bfr[(ks + PD) % NB], afr[(ks + PD) % NB]);
#pragma unroll
                for (int i = 0; i < TPS; ++i)
                    acc[pass * TPS + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks < KR ? wf[ks < KR ? ks : 0] : afr[ks % NB], bfr[ks % NB][i], acc[pass * TPS + i], 0, 0, 0);
                if (SV_TCONVR_PIPE && (ks & 1)) {      // one element of the other tile per two k-steps
                    if (pass == 1) epi_step(0, ks >> 1, im);
                    else if (PREV) epi_step(1, ks >> 1, prev);
                }
                if (PIPE_STAGE && pass == 1 && (ks & 3) == 0 && !(SV_TCONVR_DBG & 4)) {
                    // (second pass: the vectors requested at the start of the interval have had a whole pass to arrive)
                    if (has_next) {
                        if (ks == 0 && has_pro) stg_fetch(0);
                        stg_step(ks >> 2, BUF ^ 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(2);
        if (!SV_TCONVR_PIPE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) epilogue(mt, im);
        }
        stamp(3);
        if (has_next && !(SV_TCONVR_DBG & 4) && !PIPE_STAGE) stage(BUF ^ 1);
        stamp(4);
        __syncthreads();
        stamp(5);
    };
    // (EX) both tiles per k-step (a B-fragment pair per A fragment), taps beyond the phase's own skipped, the epilogue behind the loop
    auto body_ex = [&](auto bufc, int im) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        const int nxt = im + gridDim.x;
        const bool has_next = nxt < nimg;
        if (has_next) request(nxt);
        request_ex(im);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
        constexpr int PD = SV_TCONVR_PD, NB = PD + 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ntap) {
                bf16x8 bfr[NB][MT], afr[NB];
                auto fetch = [&](int kc, bf16x8 (&dst)[MT], bf16x8& adst) {
                    const int ks = KC * t + kc;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        dst[mt] = *reinterpret_cast<const bf16x8*>(smem + rb[t] + (BUF * TILE + kc * PLANE + mt * (4 * PITCH * 32)));
                    if (ks >= KR) adst = *reinterpret_cast<const bf16x8*>(wlds + (ks - KR) * 1024);
                };
#pragma unroll
                for (int d = 0; d < PD; ++d) fetch(d, bfr[d % NB], afr[d % NB]);
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const int ks = KC * t + kc;
                    if (kc + PD < KC) fetch(kc + PD, bfr[(kc + PD) % NB], afr[(kc + PD) % NB]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks < KR ? wf[ks < KR ? ks : 0] : afr[kc % NB], bfr[kc % NB][mt], acc[mt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        epilogue_ex(im);
        if (has_next) stage(BUF ^ 1);
        __syncthreads();
    };
    if constexpr (EX) {
        const int step = gridDim.x;
        while (img < nimg) {
            body_ex(std::integral_constant<int, 0>{}, img);
            img += step;
            if (img >= nimg) break;
            body_ex(std::integral_constant<int, 1>{}, img);
            img += step;
        }
    } else {
        static_assert(!SV_TCONVR_PIPE || (MT == 2 && KS == 32), "pipelined epilogue: 16 elements over 32 k-steps, two tiles");
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        const int step = gridDim.x;
        int last = -1;
        if (img < nimg) {
            body(T0{}, std::false_type{}, img, -1);
            last = img;
            img += step;
            while (img < nimg) {
                body(T1{}, std::true_type{}, img, last);
                last = img;
                img += step;
                if (img >= nimg) break;
                body(T0{}, std::true_type{}, img, last);
                last = img;
                img += step;
            }
        }
        if (SV_TCONVR_PIPE && last >= 0) epilogue(1, last);
    }
    if ((SV_TCONVR_DBG & 32) && a.fold_mean && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        unsigned long long* const dbg = reinterpret_cast<unsigned long long*>(a.fold_mean) + wave * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) dbg[i] = tseg[i];
    }

    // ---- statistics: 32 pixel lanes -> lanes 0 / 32, the four phases through LDS, one double atomic per channel and block
    if (want_stats) {
        float* const wsum = reinterpret_cast<float*>(smem + C::OFF_WSUM) + ph * 2 * NOUT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v1 = ps1[e], v2 = ps2[e];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                v1 += __shfl_xor(v1, o);
                v2 += __shfl_xor(v2, o);
            }
            if (q == 0) {
                const int n = 32 * nt + 8 * (e >> 2) + 4 * h + (e & 3);
                wsum[n] = v1;
                wsum[NOUT + n] = v2;
            }
        }
        __syncthreads();
        if (tid < 2 * NOUT) {
            const float* const ws = reinterpret_cast<const float*>(smem + C::OFF_WSUM);
            const float v = (ws[tid] + ws[2 * NOUT + tid]) + (ws[4 * NOUT + tid] + ws[6 * NOUT + tid]);
            double* const dst = (EX ? a.bsums : a.stats) + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * NOUT;
            atomicAdd(dst + tid, (double)v);
        }
    }
}

template <int CIN, int NOUT, int H, bool EX>
int launch_tconvr(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    typedef tconvr_cfg<CIN, NOUT, H> C;
    const int G = sv_ngroups(a->groups);
    int per = sv_persistent_blocks() / 2 / G;          // (the budget counts two blocks per CU; this kernel is one: 512 registers)
    if (per < 1) per = 1;
    if (per > g->B) per = g->B;
    // equal shares: every block the same number of images where the batch allows it
    const int rounds = (g->B + per - 1) / per;
    const int grid = (g->B + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tconvr_kernel<CIN, NOUT, H, EX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(tconvr)");
        optin = true;
    }
    sv_igemm_args b = *a;          // the forward form folds the BatchNorm finalisation of its prologue
    if (!sv_fold_claim(!EX && b.fold_stats != nullptr)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((tconvr_kernel<CIN, NOUT, H, EX>), dim3(grid, G), dim3(C::NTH), C::LDS, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(tconvr)");
}

// ---- the 32 <- 64 stride-2 3x3 data gradient (wideresnet.py:29-30, the first convolution of block 2: dy 16x16x64 -> dx 32x32x32)
// The same scheme at twice the map size and a quarter of the weights (36 KB: every A fragment a register, fetched straight from
// global memory): eight waves = four phases x the two halves of the phase's 16 x 16 grid (4 tiles of 2 rows x 16 pixels each, run
// as two passes of two tiles), dealt 3 3 1 1 0 0 2 2 as above.  LDS image: 4 planes of 16 channels, rows 24 pixels apart (a
// ds_read_b128 lane group -- x 0-3 and 12-15 of one row, x 4-11 of the other -- then hits every 32-byte slot twice, in different
// halves), 55 KB per image, two images.  Per image and block 335 KB move against 288 MFMAs: the layer is HBM-bound (335 MB,
// 63 us at the rate the streaming kernels reach), and the activation-backward epilogue (~9 VALU instructions per output element,
// 64 elements per thread and image) is the wave's longest stretch.
struct tconvx16_cfg {
    static constexpr int CIN = 64, NOUT = 32, H = 16, HP = H + 2, PITCH = 24;
    static constexpr int PLANE = HP * PITCH * 32 + 32, NPL = CIN / 16, TILE = NPL * PLANE;
    static constexpr int NTH = 512, VPT = H * H * (CIN / 8) / NTH;
    static constexpr int OFF_WSUM = 2 * TILE;                       // [8 waves][2][NOUT] floats
    static constexpr int OFF_CST = OFF_WSUM + 8 * 2 * NOUT * 4;     // [NOUT] x {scale, shift, rstd, -mean * rstd}
    static constexpr int LDS = OFF_CST + NOUT * 16;
    static_assert(VPT == 4 && LDS <= 160 * 1024, "staging / LDS budget");
    static_assert((NPL - 1) * PLANE + 3 * (2 * PITCH * 32) + HP * PITCH * 32 < 65536, "plane and tile offsets are ds_read immediates");
};

// FWD = true: the same loop as the FORWARD of the last decoder layer, ConvTranspose2d(64, 3 -> 16 padded, 4, 2, 1) at 16x16 -> 32x32
// (decoder.py:58): four 4-tap phases, BatchNorm + ReLU prologue applied while staging (its finalisation folded into the launch),
// 16 output channels (the upper half of the 32-row MFMA tile is zero weights), no epilogue fusion -- one 16-byte store per pixel.
// The LDS-halo kernel ran this layer at 1.5 TB/s of its 67 MB (two groups): 45 us.
template <bool FWD>
__global__ __launch_bounds__(512, 1) void tconvx16_kernel(const sv_geom g, const sv_igemm_args_g AG) {
    typedef tconvx16_cfg C;
    constexpr int CIN = C::CIN, NOUT = C::NOUT, H = C::H, PITCH = C::PITCH, PLANE = C::PLANE, TILE = C::TILE, NTH = C::NTH, VPT = C::VPT;
    constexpr int KC = CIN / 16;                                   // k-steps per tap
    const sv_igemm_args& a = AG.g[blockIdx.y];
    sv_start_signal(a);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = (0x22001133 >> (4 * wave)) & 3, th = wave & 1;   // phase, half of the phase's grid (tiles 4 th .. 4 th + 3)
    const int q = lane & 31, h = lane >> 5, r = q >> 4, x = q & 15;
    const sv_phase& P = g.phase[ph];
    const int ntap = __builtin_amdgcn_readfirstlane(P.ntap);
    const bf16* __restrict__ X = reinterpret_cast<const bf16*>(a.x);
    const bf16* __restrict__ EXP = reinterpret_cast<const bf16*>(a.ex);
    bf16* __restrict__ O = reinterpret_cast<bf16*>(a.out);
    const int nimg = g.B;
    int img = blockIdx.x;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    bf16x8 xr[VPT];
    auto request = [&](int im) {
        const bf16* const xi = X + (int64_t)im * (H * H * CIN);
#pragma unroll
        for (int i = 0; i < VPT; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(xi + tid * 8 + i * (NTH * 8));
    };
    if (img < nimg) request(img);
    // A fragments: row = output channel q, k = (tap t) 64 + 16 kc + 8 h ..
    bf16x8 wf[4 * KC];
    {
        const int qw = FWD ? (q & 15) : q;              // (FWD: 16 output channels -- rows 16 .. 31 of the tile get zero weights)
        const bf16* __restrict__ W = reinterpret_cast<const bf16*>(a.w) + P.w_off + qw * (ntap * CIN) + 8 * h;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                bf16x8 f = *reinterpret_cast<const bf16x8*>(W + (t < ntap ? t : 0) * CIN + 16 * kc);
                if (FWD && q >= 16) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (bf16)0.f;
                }
                wf[KC * t + kc] = f;
            }
    }
    float* const cst = reinterpret_cast<float*>(smem + C::OFF_CST);
    const bool has_pro = FWD && a.pro_scale != nullptr;
    const float slope = has_pro ? a.pro_slope : 1.f;
    if (!FWD && tid < NOUT) {
        const float rs = a.ex_rstd[tid];
        reinterpret_cast<f32x4*>(cst)[tid] = f32x4{a.ex_scale[tid], a.ex_shift[tid], rs, -a.ex_mean[tid] * rs};
    }
    // (FWD: the area holds the prologue's [CIN] pairs {scale, shift}; the folded finalisation's scratch lies in the image area,
    //  zeroed below)
    if (FWD && a.fold_stats) sv_bn_fold_block512(a, CIN, reinterpret_cast<double*>(smem), cst, blockIdx.x == 0);
    else if (has_pro && tid < 2 * CIN) cst[tid] = (tid & 1) ? a.pro_shift[tid >> 1] : a.pro_scale[tid >> 1];
    {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
        for (int i = tid; i < 2 * TILE / 16; i += NTH) *reinterpret_cast<bf16x8*>(smem + 16 * i) = z;
    }
    // staging: vector i of this thread = pixel (tid >> 3) + 64 i (4 rows further: same row parity), chunk sc = tid & 7
    const int sc = tid & 7;
    int sdst;
    {
        const int p = tid >> 3, yy = (p >> 4) + 1, xx = (p & 15) + 1;
        sdst = (sc >> 1) * PLANE + (yy * PITCH + xx) * 32 + (((sc ^ yy) & 1) << 4);
    }
    auto stage = [&](int buf) __attribute__((always_inline)) {
        if (has_pro) {
            f32x4 s0, s1, t0, t1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(cst + 16 * sc + 4 * j);        // channels 8 sc + 2 j, + 1
                (j < 2 ? s0 : s1)[2 * (j & 1)] = c[0]; (j < 2 ? t0 : t1)[2 * (j & 1)] = c[1];
                (j < 2 ? s0 : s1)[2 * (j & 1) + 1] = c[2]; (j < 2 ? t0 : t1)[2 * (j & 1) + 1] = c[3];
            }
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = bn_act8(xr[i], s0, s1, t0, t1, slope);
        } else {
#pragma unroll
            for (int i = 0; i < VPT; ++i) *reinterpret_cast<bf16x8*>(smem + buf * TILE + sdst + i * (4 * PITCH * 32)) = xr[i];
        }
    };
    // B fragments: pixel (row 8 th + 2 mt + r, column x) at tap t, channels 16 kc + 8 h ..  ->  rb[t] + mt (2 rows) + kc PLANE
    int rb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int yy = 8 * th + r + P.dy[t] + 1, xx = x + P.dx[t] + 1;
        rb[t] = (yy * PITCH + xx) * 32 + (((h ^ yy) & 1) << 4);
    }
    const int opix = (((8 * th + r) * g.osy + P.ooy) * g.Wout + x * g.osx + P.oox) * g.ldo + 8 * h;
    const int otile = 2 * g.osy * g.Wout * g.ldo;
    const int64_t ostride = (int64_t)g.Hout * g.Wout * g.ldo;
    const float ex_slope = FWD ? 1.f : a.ex_slope;
    float ps1[16], ps2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ps1[e] = ps2[e] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the weights are here (no counted waits for them inside the loop)
    __syncthreads();
    if (img < nimg) stage(0);
    __syncthreads();

    constexpr int TP = SV_TCONVX16_TP, NPASS = 4 / TP;      // tiles per pass over the taps
    f32x16 acc[TP];
    u32x4 exr[TP][2];                              // [tile of the pass][16-byte half]: the raw tensor at this lane's store positions
    auto request_ex = [&](int im, int pass) __attribute__((always_inline)) {
        const bf16* const eimg = EXP + (int64_t)im * ostride;
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) exr[i][gp] = *reinterpret_cast<const u32x4*>(eimg + (TP * pass + i) * otile + opix + 16 * gp);
    };
    auto epilogue_ex = [&](int im, int pass) __attribute__((always_inline)) {
        bf16* const oimg = O + (int64_t)im * ostride;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            uint32_t xw[TP][2][2], ow[TP][2][2];
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto rr = __builtin_amdgcn_permlane32_swap(exr[i][gp][d], exr[i][gp][2 + d], false, false);
                    xw[i][0][d] = rr[0];
                    xw[i][1][d] = rr[1];
                }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int e0 = 4 * (2 * gp + k) + 2 * d;
                    const f32x4 c0 = reinterpret_cast<const f32x4*>(cst)[8 * (2 * gp + k) + 4 * h + 2 * d];
                    const f32x4 c1 = reinterpret_cast<const f32x4*>(cst)[8 * (2 * gp + k) + 4 * h + 2 * d + 1];
#pragma unroll
                    for (int i = 0; i < TP; ++i) {
                        const uint32_t w = xw[i][k][d];
                        const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
                        const float g0 = acc[i][e0] * ((x0 * c0[0] + c0[1] > 0.f) ? 1.f : ex_slope);
                        const float g1 = acc[i][e0 + 1] * ((x1 * c1[0] + c1[1] > 0.f) ? 1.f : ex_slope);
                        ps1[e0] += g0;
                        ps2[e0] += g0 * (x0 * c0[2] + c0[3]);
                        ps1[e0 + 1] += g1;
                        ps2[e0 + 1] += g1 * (x1 * c1[2] + c1[3]);
                        typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 pr = {(bf16)g0, (bf16)g1};
                        ow[i][k][d] = __builtin_bit_cast(uint32_t, pr);
                    }
                    __builtin_amdgcn_sched_barrier(0);       // (the 16 constant reads of an epilogue are not all hoisted to its top: 64 registers)
                }
#pragma unroll
            for (int i = 0; i < TP; ++i) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto rr = __builtin_amdgcn_permlane32_swap(ow[i][0][d], ow[i][1][d], false, false);
                    ow[i][0][d] = rr[0];
                    ow[i][1][d] = rr[1];
                }
                const u32x4 o = {ow[i][0][0], ow[i][0][1], ow[i][1][0], ow[i][1][1]};
                *reinterpret_cast<u32x4*>(oimg + (TP * pass + i) * otile + opix + 16 * gp) = o;
            }
        }
    };
    auto body = [&](auto bufc, int im) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        const int nxt = im + gridDim.x;
        const bool has_next = nxt < nimg;
        if (has_next) request(nxt);
        // (the pass loop is NOT unrolled: with one copy of the epilogue per pass in the body the register allocator spilled 120-250
        //  registers -- the statistics accumulators across the copies --, with one copy in a loop none)
#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            int rbb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) rbb[t] = rb[t] + BUF * TILE + pass * (TP * 2 * PITCH * 32);
            __builtin_amdgcn_sched_barrier(0);       // (nothing of a pass moves into another one)
            if (!FWD) request_ex(im, pass);   // (a pass of MFMAs ahead of its use; the partner wave covers the rest of the latency)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            constexpr int PD = SV_TCONVR_PD, NB = PD + 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < ntap) {
                    bf16x8 bfr[NB][TP];
                    auto fetch = [&](int kc, bf16x8 (&dst)[TP]) __attribute__((always_inline)) {
#pragma unroll
                        for (int i = 0; i < TP; ++i)
                            dst[i] = *reinterpret_cast<const bf16x8*>(smem + rbb[t] + (kc * PLANE + i * (2 * PITCH * 32)));
                    };
#pragma unroll
                    for (int d = 0; d < PD; ++d) fetch(d, bfr[d % NB]);
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) {
                        if (kc + PD < KC) fetch(kc + PD, bfr[(kc + PD) % NB]);
#pragma unroll
                        for (int i = 0; i < TP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[KC * t + kc], bfr[kc % NB][i], acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!FWD) epilogue_ex(im, pass);
            else {
                // acc[i][4 gq + e] = channel 8 gq + 4 h + e (gq < 2: 16 channels) of pixel q of tile i: one 16-byte store
                bf16* const oimg = O + (int64_t)im * ostride;
#pragma unroll
                for (int i = 0; i < TP; ++i) {
                    uint32_t pk[2][2];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                            const bf16x2 pr = {(bf16)acc[i][4 * k + 2 * d], (bf16)acc[i][4 * k + 2 * d + 1]};
                            pk[k][d] = __builtin_bit_cast(uint32_t, pr);
                        }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const auto rr = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
                        pk[0][d] = rr[0];
                        pk[1][d] = rr[1];
                    }
                    const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                    *reinterpret_cast<u32x4*>(oimg + (TP * pass + i) * otile + opix) = o;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) stage(BUF ^ 1);
        __syncthreads();
    };
    {
        const int step = gridDim.x;
        while (img < nimg) {
            body(std::integral_constant<int, 0>{}, img);
            img += step;
            if (img >= nimg) break;
            body(std::integral_constant<int, 1>{}, img);
            img += step;
        }
    }
    // ---- sums: 32 pixel lanes -> lanes 0 / 32, the eight waves through LDS, one double atomic per channel and block
    if (!FWD) {
        float* const wsum = reinterpret_cast<float*>(smem + C::OFF_WSUM) + wave * 2 * NOUT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v1 = ps1[e], v2 = ps2[e];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                v1 += __shfl_xor(v1, o);
                v2 += __shfl_xor(v2, o);
            }
            if (q == 0) {
                const int n = 8 * (e >> 2) + 4 * h + (e & 3);
                wsum[n] = v1;
                wsum[NOUT + n] = v2;
            }
        }
        __syncthreads();
        if (tid < 2 * NOUT) {
            const float* const ws = reinterpret_cast<const float*>(smem + C::OFF_WSUM) + tid;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += ws[w * 2 * NOUT];
            atomicAdd(a.bsums + (size_t)(blockIdx.x & (a.replicas - 1)) * 2 * NOUT + tid, (double)v);
        }
    }
}

template <bool FWD>
int launch_tconvx16(const sv_geom* g, const sv_igemm_args* a, hipStream_t s) {
    typedef tconvx16_cfg C;
    const int G = sv_ngroups(a->groups);
    int per = sv_persistent_blocks() / 2 / G;
    if (per < 1) per = 1;
    if (per > g->B) per = g->B;
    const int rounds = (g->B + per - 1) / per;
    const int grid = (g->B + rounds - 1) / rounds;
    static bool optin = false;
    if (!optin) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tconvx16_kernel<FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            return sv_check_launch("hipFuncSetAttribute(tconvx16)");
        optin = true;
    }
    sv_igemm_args b = *a;          // the forward form folds the BatchNorm finalisation of its prologue
    if (!sv_fold_claim(FWD && b.fold_stats != nullptr)) b.fold_stats = nullptr;
    a = &b;
    SV_LAUNCH_GATE(grid, a);
    sv_prof_begin(s);
    hipLaunchKernelGGL((tconvx16_kernel<FWD>), dim3(grid, G), dim3(C::NTH), C::LDS, s, *g, sv_expand_groups(*g, *a, 2));
    sv_prof_end(s);
    return sv_check_launch("sv_igemm(tconvx16)");
}

}  // namespace

// Returns 1 and sets *rc when the launch is a ConvTranspose2d(4, 2, 1) forward this kernel covers.
int sv_tconvr_try(const sv_geom* g, int dtype, const sv_igemm_args* a, hipStream_t s, int* rc) {
    if (sv_disabled(SV_K_TCONVR) || dtype != SV_BF16) return 0;
    if (a->bias || a->residual || a->sparse_out) return 0;
    if (a->ex && (a->stats || a->pro_scale || sv_disabled(SV_K_TCONVR_EX))) return 0;
    if ((a->flags & SV_FLAG_DET) && (a->stats || a->ex)) return 0;            // (fixed-order statistics: the LDS-halo kernel's per-wave slots)
    if (g->nphase != 4 || g->sy != 1 || g->sx != 1 || g->osy != 2 || g->osx != 2) return 0;
    if ((int64_t)g->Hout * g->Wout * g->ldo >= ((int64_t)1 << 31)) return 0;
    if (a->ex && g->Hin == 16 && g->Win == 16 && g->Hq == 16 && g->Wq == 16 && g->Hout == 32 && g->Wout == 32 && g->Cin == 64 && g->ldx == 64 &&
        g->N == 32 && g->ldo % 4 == 0) {
        for (int p = 0; p < 4; ++p) {
            const sv_phase& P = g->phase[p];
            if (P.ntap > 4 || P.ntap < 1 || P.ooy < 0 || P.ooy > 1 || P.oox < 0 || P.oox > 1) return 0;
            for (int t = 0; t < P.ntap; ++t)
                if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
        }
        *rc = launch_tconvx16<false>(g, a, s);
        return 1;
    }
    // the last decoder layer's forward: ConvTranspose2d(64, 16 (3 padded), 4, 2, 1) at 16x16, no epilogue fusion
    if (!a->ex && !a->stats && g->Hin == 16 && g->Win == 16 && g->Hq == 16 && g->Wq == 16 && g->Hout == 32 && g->Wout == 32 && g->Cin == 64 &&
        g->ldx == 64 && g->N == 16 && g->ldo % 8 == 0) {
        for (int p = 0; p < 4; ++p) {
            const sv_phase& P = g->phase[p];
            if (P.ntap != 4 || P.ooy < 0 || P.ooy > 1 || P.oox < 0 || P.oox > 1) return 0;
            for (int t = 0; t < 4; ++t)
                if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
        }
        *rc = launch_tconvx16<true>(g, a, s);
        return 1;
    }
    if (g->Hin != 8 || g->Win != 8 || g->Hq != 8 || g->Wq != 8 || g->Hout != 16 || g->Wout != 16) return 0;
    if (g->Cin != 128 || g->ldx != 128 || g->N != 64 || g->ldo % 4 != 0) return 0;
    for (int p = 0; p < 4; ++p) {
        const sv_phase& P = g->phase[p];
        if (P.ntap > 4 || P.ntap < (a->ex ? 1 : 4) || P.ooy < 0 || P.ooy > 1 || P.oox < 0 || P.oox > 1) return 0;
        for (int t = 0; t < P.ntap; ++t)
            if (P.dy[t] < -1 || P.dy[t] > 1 || P.dx[t] < -1 || P.dx[t] > 1) return 0;
    }
    if ((int64_t)g->Hout * g->Wout * g->ldo >= ((int64_t)1 << 31)) return 0;
    *rc = a->ex ? launch_tconvr<128, 64, 8, true>(g, a, s) : launch_tconvr<128, 64, 8, false>(g, a, s);
    return 1;
}
