// Development aid: what slows a one-wave "tree-like" workgroup down beside the evaluator kernel -- dependent global loads (memory
// latency), dependent vector arithmetic (issue slots between MFMAs), dependent LDS reads?  Built on the GPU box by
// tools/contention_probe.py:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libcontention.so tools/probe/contention_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(64) void chase_kernel(const uint32_t *ring, uint32_t n_ring, int steps, uint32_t *out) {
    uint32_t i = (blockIdx.x * 2654435761u) % n_ring;
    for (int s = 0; s < steps; s++) i = ring[i];                       // one dependent round trip per step (every lane the same address)
    if (threadIdx.x == 0) out[blockIdx.x] = i;
}
__global__ __launch_bounds__(64) void valu_kernel(int steps, double *out) {
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.999999;
    for (int s = 0; s < steps; s++) { x = x * y + 1e-12; x = x * y + 1e-12; x = x * y + 1e-12; x = x * y + 1e-12; }   // dependent f64 chain
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void lds_kernel(int steps, uint32_t *out) {
    __shared__ uint32_t t[512];
    for (int i = threadIdx.x; i < 512; i += 64) t[i] = (i * 37 + 11) & 511;
    __syncthreads();
    uint32_t i = threadIdx.x;
    for (int s = 0; s < steps; s++) i = t[i];                          // dependent LDS reads
    out[blockIdx.x * 64 + threadIdx.x] = i;
}
__global__ __launch_bounds__(64) void salu_kernel(int steps, uint32_t *out) {
    uint32_t x = blockIdx.x | 1u;
    for (int s = 0; s < steps; s++) { x = x * 1664525u + 1013904223u; x ^= x >> 7; }                   // wave-uniform: scalar unit
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
extern "C" {
int probe_chase(const void *ring, unsigned n_ring, int steps, void *out, int blocks, void *stream) {
    hipLaunchKernelGGL(chase_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (const uint32_t *)ring, n_ring, steps, (uint32_t *)out);
    return (int)hipGetLastError();
}
int probe_valu(int steps, void *out, int blocks, void *stream) {
    hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, steps, (double *)out);
    return (int)hipGetLastError();
}
int probe_lds(int steps, void *out, int blocks, void *stream) {
    hipLaunchKernelGGL(lds_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, steps, (uint32_t *)out);
    return (int)hipGetLastError();
}
int probe_salu(int steps, void *out, int blocks, void *stream) {
    hipLaunchKernelGGL(salu_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, steps, (uint32_t *)out);
    return (int)hipGetLastError();
}
}
