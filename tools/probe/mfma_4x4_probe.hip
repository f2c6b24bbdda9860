// Development aid: lane mapping of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per instruction), found by experiment:
// A = lane id, B = 100 * lane id, C = 0  ->  D[reg] of every lane.     hipcc --offload-arch=gfx950 -O2 -o /tmp/p4 tools/probe/mfma_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, unsigned long long *cyc) {
    const int lane = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 100.0f * (float)(lane + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[lane * 4 + r] = c[r];
    // throughput: 256 dependent-free instructions on four accumulators
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)lane, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < 64; i++) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[256 + lane] = a0[0] + a1[1] + a2[2] + a3[3];
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    float *d; unsigned long long *c;
    hipMalloc(&d, 4096); hipMalloc(&c, 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c);
    float h[320]; unsigned long long hc;
    hipMemcpy(h, d, 1280, hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane++) {
        printf("lane %2d:", lane);
        for (int r = 0; r < 4; r++) {
            // D = a * b with a = (la + 1), b = 100 (lb + 1): recover la, lb
            const double v = h[lane * 4 + r] / 100.0;
            int la = -1, lb = -1;
            for (int x = 1; x <= 64 && la < 0; x++) for (int y = 1; y <= 64; y++) if ((double)x * y == v && (x - 1) / 4 == (y - 1) / 4 && (x - 1) / 4 == lane / 4) { la = x - 1; lb = y - 1; break; }
            printf("  reg%d = A[lane %2d] * B[lane %2d]", r, la, lb);
        }
        printf("\n");
    }
    printf("256 instructions: %llu ticks of s_memtime (100 MHz) = %.1f ns each\n", hc, hc * 10.0 / 256);
    return 0;
}
