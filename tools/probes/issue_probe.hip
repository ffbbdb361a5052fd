// Single-wave issue probe (gfx950): how many cycles does one wave per SIMD need for a stream of MFMAs with other
// instructions between them?  Answers whether VALU / LDS / SALU instructions of the SAME wave hide behind its MFMAs.
//   hipcc --offload-arch=gfx950 -O2 issue_probe.hip -o issue_probe && ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#define VALU(r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(c));
#define PKF(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(c2));
#define SALU(r) asm volatile("s_add_u32 %0, %0, 1" : "+s"(r) :: "scc");
#define DSR(r) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(laddr));
#define DSW() asm volatile("ds_write_b128 %0, %1" :: "v"(laddr), "v"(wv));
#define NOP() asm volatile("s_nop 0");
#define MFMA32(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(big[i]) : "v"(a), "v"(b));
#define DSR128(r) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(laddr));
#define GLD(r) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(goff), "s"(gsrc) : "memory");
#define DMA() asm volatile("s_add_u32 m0, %0, 4096\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lbase), "v"(goff), "s"(gsrc) : "memory", "scc");

template <int P>
__global__ __launch_bounds__(256, 1) void probe(float* out, uint32_t* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    f32x4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.f + threadIdx.x); b[j] = (__bf16)0.5f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, c = 0.999f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, c2 = {0.999f, 0.999f};
    uint32_t s0 = 0, s1 = 0;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    s16x4 d0, d1;
    f32x4 wv = {1.f, 2.f, 3.f, 4.f};
    const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    const uint32_t goff = threadIdx.x * 16;
    const float* gsrc = out + (blockIdx.x & 7) * 4096;
    const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 1024);
    f32x4 g0, g1;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define G(i)                                                                                \
        MFMA(i)                                                                             \
        if (P == 1) { VALU(v0) }                                                            \
        if (P == 2) { VALU(v0) VALU(v1) }                                                   \
        if (P == 3) { VALU(v0) VALU(v1) VALU(v2) }                                          \
        if (P == 4) { VALU(v0) VALU(v1) VALU(v2) VALU(v3) }                                 \
        if (P == 5) { DSR(d0) }                                                             \
        if (P == 6) { DSR(d0) DSR(d1) }                                                     \
        if (P == 7) { DSR(d0) VALU(v0) VALU(v1) }                                           \
        if (P == 8) { SALU(s0) SALU(s1) }                                                   \
        if (P == 9) { PKF(p0) }                                                             \
        if (P == 10) { PKF(p0) PKF(p1) }                                                    \
        if (P == 11) { DSW() }                                                              \
        if (P == 12) { NOP() NOP() }                                                        \
        if (P == 14 && (i & 3) == 0) { GLD(g0) }                                            \
        if (P == 15 && (i & 3) == 0) { DMA() }                                              \
        if (P == 16 && (i & 1) == 0) { GLD(g0) }                                            \
        if (P == 17 && (i & 1) == 0) { DMA() }                                              \
        if (P == 13) { VALU(v0) VALU(v1) VALU(v2) VALU(v3) VALU(v0) VALU(v1) VALU(v2) VALU(v3) }
        if (P < 20) { G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7) }
#define H(i)                                                                                \
        MFMA32(i)                                                                           \
        if (P == 21) { VALU(v0) VALU(v1) }                                                  \
        if (P == 22) { VALU(v0) VALU(v1) VALU(v2) VALU(v3) }                                \
        if (P == 23) { VALU(v0) VALU(v1) VALU(v2) VALU(v3) VALU(v0) VALU(v1) }              \
        if (P == 24) { VALU(v0) VALU(v1) VALU(v2) VALU(v3) VALU(v0) VALU(v1) VALU(v2) VALU(v3) } \
        if (P == 25) { DSR128(q0) }                                                         \
        if (P == 26) { DSR128(q0) DSR128(q1) }                                              \
        if (P == 27) { DSR128(q0) VALU(v0) VALU(v1) }                                       \
        if (P == 28) { DSR128(q0) VALU(v0) VALU(v1) VALU(v2) VALU(v3) }                     \
        if (P == 29 && (i & 1) == 0) { DMA() }                                              \
        if (P == 30) { DMA() }
        if (P >= 20) { H(0) H(1) H(2) H(3) H(0) H(1) H(2) H(3) }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+v"(g0), "+v"(g1) :: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = q0[0] + q1[1] + big[0][0] + big[1][1] + big[2][2] + big[3][3] + g0[0] + g1[1] + v0 + v1 + v2 + v3 + p0[0] + p1[1] + (float)s0 + (float)s1 + (float)d0[0] + (float)d1[1];
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = (uint32_t)(t1 - t0);
}

template <int P>
void run(const char* name, float* out, uint32_t* cyc) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(probe<P>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<P>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 4, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += x; m /= h.size();
    printf("%-44s %7.2f memtime ticks / MFMA   (%.1f us kernel => %.2f ns / MFMA)\n", name, m / (iters * 8.0), ms * 1e3, ms * 1e6 / (iters * 8.0));
}

int main() {
    float* out; uint32_t* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 4);
    run<0>("MFMA only", out, cyc);
    run<1>("MFMA + 1 VALU", out, cyc);
    run<2>("MFMA + 2 VALU", out, cyc);
    run<3>("MFMA + 3 VALU", out, cyc);
    run<4>("MFMA + 4 VALU", out, cyc);
    run<13>("MFMA + 8 VALU", out, cyc);
    run<9>("MFMA + 1 v_pk_fma_f32", out, cyc);
    run<10>("MFMA + 2 v_pk_fma_f32", out, cyc);
    run<5>("MFMA + 1 ds_read_b64_tr_b16", out, cyc);
    run<6>("MFMA + 2 ds_read_b64_tr_b16", out, cyc);
    run<7>("MFMA + 1 ds_read_b64_tr_b16 + 2 VALU", out, cyc);
    run<11>("MFMA + 1 ds_write_b128", out, cyc);
    run<8>("MFMA + 2 SALU", out, cyc);
    run<12>("MFMA + 2 s_nop", out, cyc);
    run<14>("MFMA, every 4th + global_load_dwordx4", out, cyc);
    run<15>("MFMA, every 4th + LDS-DMA (3 instr)", out, cyc);
    run<16>("MFMA, every 2nd + global_load_dwordx4", out, cyc);
    run<17>("MFMA, every 2nd + LDS-DMA (3 instr)", out, cyc);
    printf("-- v_mfma_f32_32x32x16_bf16 (32 cycles of matrix pipe), per MFMA:\n");
    run<20>("MFMA32 only", out, cyc);
    run<21>("MFMA32 + 2 VALU", out, cyc);
    run<22>("MFMA32 + 4 VALU", out, cyc);
    run<23>("MFMA32 + 6 VALU", out, cyc);
    run<24>("MFMA32 + 8 VALU", out, cyc);
    run<25>("MFMA32 + 1 ds_read_b128", out, cyc);
    run<26>("MFMA32 + 2 ds_read_b128", out, cyc);
    run<27>("MFMA32 + 1 ds_read_b128 + 2 VALU", out, cyc);
    run<28>("MFMA32 + 1 ds_read_b128 + 4 VALU", out, cyc);
    run<29>("MFMA32, every 2nd + LDS-DMA (3 instr)", out, cyc);
    run<30>("MFMA32 + LDS-DMA (3 instr)", out, cyc);
    return 0;
}
