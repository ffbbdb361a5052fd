// In-kernel shader clock at a point of the step: one wave spins for ~20 us and reports d(s_memtime) / d(s_memrealtime) (shader cycles
// per 100 MHz tick).  Built on the GPU box by tools/probes/clock_probe.sh into a library of its own (no entry point of the product).
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 2000) r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
extern "C" int sv_clock_probe(unsigned long long* out, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
