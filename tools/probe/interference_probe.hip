// Development aid: what does ONE instruction of a tree wave cost the evaluator kernel that runs beside it?  2048 one-wave workgroups at issue
// priority 2 (as advance_kernel runs), each a loop of one instruction kind; tools/interference_probe.py launches them back to back on one
// stream, net_forward_kernel on another, and divides the evaluator's slow-down by the instructions a wave executed per evaluator launch.
#include <hip/hip_runtime.h>
#include <cstdint>

#define PRIO(p) do { if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 3) __builtin_amdgcn_s_setprio(3); } while (0)

__global__ __launch_bounds__(64) void k_valu_i32(int steps, int prio, uint32_t *out) {
    PRIO(prio);
    uint32_t x = threadIdx.x;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void k_valu_f64(int steps, int prio, uint32_t *out) {
    PRIO(prio);
    double x = 1.0 + threadIdx.x * 1e-9;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(0.999999), "v"(1e-12));
    }
    out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)x;
}
__global__ __launch_bounds__(64) void k_salu(int steps, int prio, uint32_t *out) {
    PRIO(prio);
    uint32_t x = blockIdx.x | 1u;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x));
    }
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ __launch_bounds__(64) void k_lds(int steps, int prio, uint32_t *out) {
    PRIO(prio);
    __shared__ uint32_t t[512];
    for (int i = threadIdx.x; i < 512; i += 64) t[i] = (i * 37 + 11) & 511;
    __syncthreads();
    uint32_t i = threadIdx.x;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int k = 0; k < 16; k++) i = t[i];
    }
    out[blockIdx.x * 64 + threadIdx.x] = i;
}
// every lane its own line (a gather of 64 lines per instruction), dependent
__global__ __launch_bounds__(64) void k_gather(const uint32_t *ring, uint32_t mask, int steps, int prio, uint32_t *out) {
    PRIO(prio);
    uint32_t i = (blockIdx.x * 64 + threadIdx.x) * 2654435761u & mask;
    for (int s = 0; s < steps; s++) i = ring[i] & mask;
    out[blockIdx.x * 64 + threadIdx.x] = i;
}
// the whole wave one line, dependent (a node block header)
__global__ __launch_bounds__(64) void k_uniform_load(const uint32_t *ring, uint32_t mask, int steps, int prio, uint32_t *out) {
    PRIO(prio);
    uint32_t i = (blockIdx.x * 2654435761u) & mask;
    for (int s = 0; s < steps; s++) i = ring[i] & mask;
    if (threadIdx.x == 0) out[blockIdx.x] = i;
}
// contiguous 512-byte stores (a node block's arrays), not waited for
__global__ __launch_bounds__(64) void k_store(uint32_t *buf, uint32_t mask, int steps, int prio, uint32_t *out) {
    PRIO(prio);
    uint32_t base = (blockIdx.x * 2654435761u) & mask & ~127u;
    for (int s = 0; s < steps; s++) { buf[(base + threadIdx.x) & mask] = s; base = (base * 1664525u + 1013904223u) & mask & ~127u; }
    if (threadIdx.x == 0) out[blockIdx.x] = base;
}
// nothing but being resident: a wave that sleeps
__global__ void k_sleep(int steps, int prio, uint32_t *out) {
    PRIO(prio);
    for (int s = 0; s < steps; s++) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) out[blockIdx.x] = steps;
}
extern "C" int probe_sleep(int blocks, int threads, int steps, int prio, void *out, void *stream) {
    hipLaunchKernelGGL(k_sleep, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, steps, prio, (uint32_t *)out);
    return (int)hipGetLastError();
}
extern "C" int probe_launch(int kind, const void *ring, unsigned mask, int steps, int prio, void *out, int blocks, void *stream) {
    hipStream_t st = (hipStream_t)stream; uint32_t *o = (uint32_t *)out;
    switch (kind) {
    case 0: hipLaunchKernelGGL(k_valu_i32, dim3(blocks), dim3(64), 0, st, steps, prio, o); break;
    case 1: hipLaunchKernelGGL(k_valu_f64, dim3(blocks), dim3(64), 0, st, steps, prio, o); break;
    case 2: hipLaunchKernelGGL(k_salu, dim3(blocks), dim3(64), 0, st, steps, prio, o); break;
    case 3: hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, st, steps, prio, o); break;
    case 4: hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(64), 0, st, (const uint32_t *)ring, mask, steps, prio, o); break;
    case 5: hipLaunchKernelGGL(k_uniform_load, dim3(blocks), dim3(64), 0, st, (const uint32_t *)ring, mask, steps, prio, o); break;
    case 6: hipLaunchKernelGGL(k_store, dim3(blocks), dim3(64), 0, st, (uint32_t *)ring, mask, steps, prio, o); break;
    case 7: hipLaunchKernelGGL(k_sleep, dim3(blocks), dim3(64), 0, st, steps, prio, o); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
