// Two waves per SIMD (gfx950): does one wave's VALU / LDS stream run under the other wave's MFMAs?
// Block = 512 threads: waves 0-3 (one per SIMD) run stream A, waves 4-7 stream B; both timed with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { S_MFMA16 = 0, S_MFMA32 = 1, S_VALU = 2, S_DSR = 3, S_IDLE = 4, S_MIX = 5 };

template <int KIND>
__device__ __forceinline__ float stream(int iters, uint32_t laddr) {
    f32x4 acc[8];
    f32x16 big[4];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.f + threadIdx.x); b[j] = (__bf16)0.5f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, c = 0.999f;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    f32x4 d0 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (KIND == S_MFMA16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        } else if (KIND == S_MFMA32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(big[i]) : "v"(a), "v"(b));
        } else if (KIND == S_VALU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(c));
            }
        } else if (KIND == S_DSR) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(laddr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == S_MIX) {       // the shape of a conv step: 8 MFMAs, then 16 VALU + 4 LDS reads
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(c));
                asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(laddr));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float s = v0 + v1 + v2 + v3 + d0[0];
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    for (int i = 0; i < 4; ++i) s += big[i][0];
    return s;
}

template <int KA, int KB>
__global__ __launch_bounds__(512, 1) void pair(float* out, uint32_t* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16 + ((threadIdx.x >> 6) & 7) * 1024;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    float s;
    if (wave < 4) s = stream<KA>(iters, laddr);
    else s = stream<KB>(iters, laddr);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = (uint32_t)(t1 - t0);
}

template <int KA, int KB>
void run(const char* name, float* out, uint32_t* cyc) {
    const int iters = 2000, blocks = 256;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((pair<KA, KB>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 4, hipMemcpyDeviceToHost);
    double ma = 0, mb = 0;
    for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? ma : mb) += h[i * 8 + w];
    printf("%-58s A %8.1f  B %8.1f cycles per iteration\n", name, ma / (blocks * 4.0 * iters), mb / (blocks * 4.0 * iters));
}

int main() {
    float* out; uint32_t* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 4);
    printf("per iteration: MFMA16 = 8 x 16x16x32 (128 cycles of matrix pipe), MFMA32 = 4 x 32x32x16 (128), VALU = 32 v_fma, DSR = 8 ds_read_b128, MIX = 8 MFMA16 + 16 VALU + 4 ds_read_b128\n");
    run<S_MFMA16, S_IDLE>("A: MFMA16            B: idle", out, cyc);
    run<S_MFMA16, S_MFMA16>("A: MFMA16            B: MFMA16", out, cyc);
    run<S_MFMA32, S_MFMA32>("A: MFMA32            B: MFMA32", out, cyc);
    run<S_VALU, S_IDLE>("A: VALU              B: idle", out, cyc);
    run<S_MFMA16, S_VALU>("A: MFMA16            B: VALU", out, cyc);
    run<S_MFMA32, S_VALU>("A: MFMA32            B: VALU", out, cyc);
    run<S_DSR, S_IDLE>("A: DSR               B: idle", out, cyc);
    run<S_MFMA16, S_DSR>("A: MFMA16            B: DSR", out, cyc);
    run<S_MIX, S_IDLE>("A: MIX               B: idle", out, cyc);
    run<S_MIX, S_MIX>("A: MIX               B: MIX", out, cyc);
    return 0;
}
