// Development aid: what one SIMD sustains with v_mfma_f32_16x16x32_bf16 (16 cycles nominal) alone and beside VALU / LDS work.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16_probe tools/probe/mfma_bf16_probe.hip && /tmp/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// per k-step: NACC*6 MFMAs (three pieces of A against three of B, six products); optionally 3*NACC ds_read_b128 of the next A pieces and
// NVALU dependent-free VALU ops per MFMA (the split of an epilogue, modelled as and/sub pairs)
template <int NACC, bool LDSR, int NVALU>
__global__ __launch_bounds__(512) void probe(const float *__restrict__ wsrc, float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[3 * 224 * 20];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 224 * 20; i += blockDim.x) lds[i] = 0x3c003c00u + (unsigned)((i * 7 + 3) % 13);
    __syncthreads();
    const int l15 = lane & 15, q = lane >> 4;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    const u32x4 *bp = reinterpret_cast<const u32x4 *>(wsrc) + lane;
    u32x4 b[3] = {bp[0], bp[64], bp[128]};
    u32x4 a[NACC][3];
    int addr[NACC];
    for (int i = 0; i < NACC; i++) {
        addr[i] = ((i * 16 + l15) * 20 + 4 * q);
        for (int p = 0; p < 3; p++) a[i][p] = *reinterpret_cast<const u32x4 *>(&lds[p * 224 * 20 + addr[i]]);
    }
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = wsrc[lane + 64 * i];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kb = 0; kb < iters; kb++) {
        u32x4 an[NACC][3];
        for (int i = 0; i < NACC; i++)
            for (int p = 0; p < 3; p++) {
                an[i][p] = a[i][p];
                if (LDSR) an[i][p] = *reinterpret_cast<const u32x4 *>(&lds[p * 224 * 20 + addr[i] + (kb & 1) * 16 + ((kb & 7) >> 1) * 20]);
            }
        static const int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small products first
#pragma unroll
        for (int j = 0; j < 6; j++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][PA[j]]), __builtin_bit_cast(bf16x8, b[PB[j]]), acc[i], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NVALU; u++) {
                    float &x = v[(j * NACC + i + u) & 7];
                    const float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
                    x = (x - h) * 1.0009765625f + h;      // and + sub + fma : 3 VALU
                }
            }
        }
        for (int i = 0; i < NACC; i++) for (int p = 0; p < 3; p++) a[i][p] = an[i][p];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < NACC; i++) s += acc[i];
    float vs = 0;
    for (int i = 0; i < 8; i++) vs += v[i];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3] + vs;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (tid >> 6)] = t1 - t0;
}

template <int NACC, bool LDSR, int NVALU>
void run(const char *name, int threads, const float *w, float *out, unsigned long long *cyc) {
    const int iters = 512, grid = 256;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((probe<NACC, LDSR, NVALU>), dim3(grid), dim3(threads), 0, 0, w, out, cyc, iters);
        hipDeviceSynchronize();
    }
    const int nw = grid * threads / 64;
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (auto v : h) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
    mean /= nw;
    const double mfma_per_wave = iters * NACC * 6.0;
    const int waves_per_simd = threads / 256;
    printf("%-40s waves/SIMD %d  cycles/MFMA/wave %.1f (max %.1f)  pipe use %.1f%% (16 cyc per MFMA per SIMD)  VALU/MFMA %d\n", name, waves_per_simd,
           mean / mfma_per_wave, mx / mfma_per_wave, 100.0 * 16.0 * mfma_per_wave * waves_per_simd / mx, 3 * NVALU);
}

int main() {
    float *w, *out; unsigned long long *cyc;
    hipMalloc(&w, 1 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<float> hw(1 << 18);
    for (size_t i = 0; i < hw.size(); i++) hw[i] = (float)((i * 31 + 7) % 17) * 0.003f;
    hipMemcpy(w, hw.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<4, false, 0>("4 acc, registers only", threads, w, out, cyc);
        run<2, false, 0>("2 acc, registers only", threads, w, out, cyc);
        run<1, false, 0>("1 acc (dependent chain)", threads, w, out, cyc);
        run<4, true, 0>("4 acc + 3 ds_read_b128 per A", threads, w, out, cyc);
        run<2, true, 0>("2 acc + 3 ds_read_b128 per A", threads, w, out, cyc);
        run<4, false, 1>("4 acc + 3 VALU per MFMA", threads, w, out, cyc);
        run<4, false, 2>("4 acc + 6 VALU per MFMA", threads, w, out, cyc);
        run<4, true, 1>("4 acc + ds_read + 3 VALU per MFMA", threads, w, out, cyc);
    }
    return 0;
}
