// Development aid: what one SIMD's fp32 matrix pipe sustains under the instruction mixes of net_forward_kernel.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probe/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Cfg { int lds_reads, cndmask, gload, branchy; };

// ITER k-steps; per k-step NACC*4 MFMAs (16x16x4 f32), optionally NACC ds_read_b128 (next step's A), 4*NACC v_cndmask,
// one 1-KB global load (B two steps ahead), a wave-uniform branch around an extra MFMA
template <int NACC, bool LDSR, bool CND, bool GLD, bool BR, bool M32>
__global__ __launch_bounds__(512) void probe(const float *__restrict__ wsrc, float *out, unsigned long long *cyc, int iters, unsigned mask, int kb0, int kb1) {
    __shared__ __attribute__((aligned(16))) float lds[224 * 40];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 224 * 40; i += blockDim.x) lds[i] = (float)((i * 7 + 3) % 13) * 0.01f;
    __syncthreads();
    const int l15 = lane & 15, q = lane >> 4;
    f32x4 acc[NACC], accx = {0, 0, 0, 0};
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(wsrc) + lane;
    f32x4 b0 = bp[0], b1 = bp[64];
    f32x4 a[NACC];
    int addr[NACC];
    for (int i = 0; i < NACC; i++) { addr[i] = ((i * 16 + l15) * 40 + 4 * q); a[i] = *reinterpret_cast<const f32x4 *>(&lds[addr[i]]); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kb = 0; kb < iters; kb++) {
        f32x4 b2 = b1;
        if (GLD) b2 = bp[(size_t)((kb + 2) & 15) * 64];
        f32x4 an[NACC];
        for (int i = 0; i < NACC; i++) {
            an[i] = a[i];
            if (LDSR) an[i] = *reinterpret_cast<const f32x4 *>(&lds[addr[i] + ((kb & 7) - 3) * 40 + (kb & 1) * 16 + 160]);
            if (CND) { const bool ok = (mask >> ((kb + i) & 31)) & 1u; if (!ok) an[i] = f32x4{0, 0, 0, 0}; }
        }
        const bool extra = BR && (kb >= kb0) && (kb < kb1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b0[j], acc[i], 0, 0, 0);
            }
            if (extra) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][j], b0[j], accx, 0, 0, 0);
        }
        for (int i = 0; i < NACC; i++) a[i] = an[i];
        b0 = b1; b1 = b2;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = accx;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (tid >> 6)] = t1 - t0;
}

template <int NACC, bool LDSR, bool CND, bool GLD, bool BR>
void run(const char *name, int threads, const float *w, float *out, unsigned long long *cyc) {
    const int iters = 512, grid = 256;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((probe<NACC, LDSR, CND, GLD, BR, false>), dim3(grid), dim3(threads), 0, 0, w, out, cyc, iters, 0x7FFFFFFFu, 128, 256);
        hipDeviceSynchronize();
    }
    const int nw = grid * threads / 64;
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (auto v : h) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
    mean /= nw;
    const double mfma_per_wave = iters * NACC * 4.0 + (BR ? 128 * 4.0 : 0.0);
    const int waves_per_simd = threads / 256;
    printf("%-44s waves/SIMD %d  cycles/MFMA/wave %.1f (max %.1f)  pipe use %.1f%% (32 cyc per MFMA per SIMD)\n", name, waves_per_simd,
           mean / mfma_per_wave, mx / mfma_per_wave, 100.0 * 32.0 * mfma_per_wave * waves_per_simd / mx);
}

int main() {
    float *w, *out; unsigned long long *cyc;
    hipMalloc(&w, 1 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<float> hw(1 << 18);
    for (size_t i = 0; i < hw.size(); i++) hw[i] = (float)((i * 31 + 7) % 17) * 0.003f;
    hipMemcpy(w, hw.data(), 1 << 20, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<4, false, false, false, false>("4 acc, registers only", threads, w, out, cyc);
        run<3, false, false, false, false>("3 acc, registers only", threads, w, out, cyc);
        run<2, false, false, false, false>("2 acc, registers only", threads, w, out, cyc);
        run<1, false, false, false, false>("1 acc (dependent chain)", threads, w, out, cyc);
        run<3, true, false, false, false>("3 acc + ds_read_b128 A", threads, w, out, cyc);
        run<3, true, true, false, false>("3 acc + ds_read A + cndmask", threads, w, out, cyc);
        run<3, true, true, true, false>("3 acc + ds_read A + cndmask + global B", threads, w, out, cyc);
        run<3, true, true, true, true>("3 acc + ds_read + cndmask + global + branch", threads, w, out, cyc);
        run<3, false, true, false, false>("3 acc + cndmask only", threads, w, out, cyc);
        run<3, false, false, true, false>("3 acc + global B only", threads, w, out, cyc);
        run<3, false, false, false, true>("3 acc + branch only", threads, w, out, cyc);
        run<7, true, false, true, false>("7 acc + ds_read A + global B", threads, w, out, cyc);
    }
    return 0;
}
