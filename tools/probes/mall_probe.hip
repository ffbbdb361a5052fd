// Does the order in which a consumer walks a tensor that was JUST written matter (memory-side cache, 256 MB)?
//   producer: persistent blocks write their contiguous range front to back; consumer reads (a) the same way, (b) each range back to
//   front, (c) flat grid-stride ascending, (d) flat descending.  hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void produce(f4* dst, size_t n4, float v) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, b0 = (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += 256) { const size_t j = b0 + i; if (j < n4) dst[j] = f4{v, v, v, v}; }
}
__global__ __launch_bounds__(256) void consume(const f4* src, size_t n4, int mode, float* out) {
    float s = 0.f;
    if (mode < 2) {
        const size_t per = (n4 + gridDim.x - 1) / gridDim.x, b0 = (size_t)blockIdx.x * per;
        for (size_t i = threadIdx.x; i < per; i += 256) {
            const size_t k = mode == 0 ? i : per - 1 - i, j = b0 + k;
            if (j < n4) { const f4 q = __builtin_nontemporal_load(src + j); s += q[0] + q[1] + q[2] + q[3]; }
        }
    } else {
        const size_t gsz = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += gsz) {
            const size_t j = mode == 2 ? i : n4 - 1 - i;
            const f4 q = __builtin_nontemporal_load(src + j); s += q[0] + q[1] + q[2] + q[3];
        }
    }
    if (s == 12345.678f) *out = s;
}
int main() {
    const size_t sizes[] = {64, 134, 268, 536};
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : sizes) {
        const size_t n4 = mb * 1000000 / 16;
        f4 *a, *other; hipMalloc(&a, n4 * 16); hipMalloc(&other, (size_t)600 * 1000000);
        for (int mode = 0; mode < 4; ++mode)
            for (int cold = 0; cold < 2; ++cold) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, a, n4, (float)rep);
                    if (cold) hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, 0, other, (size_t)600 * 1000000 / 16, 1.f);   // evict
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(consume, dim3(1024), dim3(256), 0, 0, a, n4, mode, out);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("%4zu MB  mode %d (%s) %s: %7.1f us  %6.2f TB/s\n", mb, mode,
                       mode == 0 ? "ranges, same order" : mode == 1 ? "ranges, reversed  " : mode == 2 ? "flat ascending    " : "flat descending   ",
                       cold ? "after 600 MB of other writes" : "right after the producer    ", best * 1e3, mb * 1e6 / (best * 1e-3) / 1e12);
            }
        hipFree(a); hipFree(other);
    }
    return 0;
}
