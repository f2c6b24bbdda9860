// ccsp_net.hip -- the policy/value network (graph of model.py:58-145) as ONE
// fused HIP kernel for gfx950: a workgroup carries 8 positions through the whole network with every
// activation resident in LDS, each layer an fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains,
// the reference's own arithmetic type) GEMM whose B operand (weights, BatchNorm folded in at load time)
// streams from L2 in a pre-packed per-lane order.  Replaces ~90 MIOpen/elementwise launches per forward.
//
//   rows = position * 25 + (r * 5 + c)    (200 rows per workgroup, padded to 13 tiles of 16)
//   stem   3x3 valid 7->64      : A = implicit im2col of the 7x7x7 planes      K = 9 taps x 8 (7 + zero pad)
//   block  1x1 64->32, 3x3 same 32->32 (implicit im2col with zero halo), 1x1 32->64 + residual, ReLU each
//   policy 1x1 64->16, flatten (h, w, c) 400 -> dense 294 logits               (M = positions)
//   value  1x1 64->1, flatten 25 -> dense 32 ReLU -> dense 1 tanh              (plain FMA, tiny)
//   epilogue: p = float64 softmax(logits) (utils.softmax, utils.py:187-192), v float32
//
// Packed weight order of a GEMM layer with K = 16*KB, N = 16*NT:  [nt][kb][lane][j]  =
// W[k = 16 kb + 4 (lane >> 4) + j][n = 16 nt + (lane & 15)]  -- one 16-byte load per lane per 16 k.
// Within a 16-k block the MFMA k-slot q = lane >> 4 therefore carries k = 4q + j in step j; A is read
// with the same mapping (one ds_read_b128 per lane per 16 k).
#include <cmath>
#include <cstring>
#include <vector>
#include "ccsp_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NB = 8;                    // positions per workgroup
constexpr int ROWS = NB * 25;            // 200
constexpr int MT = 13;                   // 16-row tiles (208 rows, 8 of them padding)
constexpr int MTP = 14;                  // tiles allocated: waves that split 13 tiles unevenly compute one phantom tile
                                         // (rows 208..223) rather than branch around MFMAs
#ifndef CCSP_NET_LDX
#define CCSP_NET_LDX 68
#define CCSP_NET_LDY 36
#endif
constexpr int LDX = CCSP_NET_LDX;        // row stride of the 64-channel buffer (floats): 16-byte aligned, bank-spread
constexpr int LDY = CCSP_NET_LDY;        // row stride of the 32-channel buffers
constexpr int LDI = 12;                  // input planes: 7 channels + zeros; 12 spreads eight consecutive cells over all banks
constexpr int NPOL = 294, NPOL_PAD = 304;

// ---- packed weight blob layout (floats) ----------------------------------------------------------------
struct Layout {
    int stem_w, stem_b;
    int l1_w[9], l1_b[9], l2_w[9], l2_b[9], l3_w[9], l3_b[9];
    int pc_w, pc_b, pf_w, pf_b;
    int vc_w, vc_b, f1_w, f1_b, f2_w, f2_b;
    int total;
};

constexpr Layout make_layout() {
    Layout L{};
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    L.stem_w = take(4 * 5 * 256); L.stem_b = take(64);
    for (int i = 0; i < 9; i++) {
        L.l1_w[i] = take(2 * 4 * 256); L.l1_b[i] = take(32);
        L.l2_w[i] = take(2 * 18 * 256); L.l2_b[i] = take(32);
        L.l3_w[i] = take(4 * 2 * 256); L.l3_b[i] = take(64);
    }
    L.pc_w = take(1 * 4 * 256); L.pc_b = take(16);
    L.pf_w = take(19 * 25 * 256); L.pf_b = take(NPOL_PAD);
    L.vc_w = take(64); L.vc_b = take(1);
    L.f1_w = take(25 * 32); L.f1_b = take(32);
    L.f2_w = take(32); L.f2_b = take(1);
    L.total = o;
    return L;
}
constexpr Layout LAY = make_layout();

// plain (Keras-order, BatchNorm already folded) input of ccsp_net_pack: offsets in floats
constexpr int PLAIN_TOTAL = 4032 + 64 + 9 * (2048 + 32 + 9216 + 32 + 2048 + 64) + 1024 + 16 + 117600 + 294 + 64 + 1 + 800 + 32 + 32 + 1;
static_assert(PLAIN_TOTAL == 244920, "249852 parameters minus the 4 x 1233 BatchNorm values folded away");

struct Smem {
    float x[MTP * 16 * LDX];             // 64-channel trunk activations (60.9 KB)
    float y1[MTP * 16 * LDY];            // 32-channel (32.3 KB); the stem's input planes and the policy conv output alias it
    float y2[MTP * 16 * LDY];            // 32-channel (32.3 KB); logits / value scratch alias it
    float part[2][4][256];               // partial sums of the k-split row tile 12 of the 32-column layers (8 KB)
};

__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
}

// One GEMM layer for one wave: output tiles (mt0 .. mt0+NMT) x (one 16-column tile nt), K = 16*KB.
// afrag(mt, kb, i) -> the lane's four A values of k-block kb for row tile mt (i = mt - mt0, a compile-time
// slot for per-tile precomputed data);  epi(mt, acc) consumes a tile.
// No tile is ever skipped: a wave whose share ends past tile 12 computes phantom rows (allocated, never read).
template <int NMT, typename AFrag, typename Epi>
__device__ __forceinline__ void gemm_tiles(const float *__restrict__ wpacked, int nt, int KB, int mt0,
                                           AFrag afrag, Epi epi) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NMT];
#pragma unroll
    for (int i = 0; i < NMT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(wpacked) + (size_t)nt * KB * 64 + lane;
    f32x4 b = bp[0];
    for (int kb = 0; kb < KB; kb++) {
        const f32x4 bnext = (kb + 1 < KB) ? bp[(size_t)(kb + 1) * 64] : b;
        f32x4 a[NMT];
#pragma unroll
        for (int i = 0; i < NMT; i++) a[i] = afrag(mt0 + i, kb, i);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < NMT; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b[j], acc[i], 0, 0, 0);
        }
        b = bnext;
    }
#pragma unroll
    for (int i = 0; i < NMT; i++) epi(mt0 + i, acc[i]);
}

// The 32-column layers have 13 x 2 = 26 tile jobs for 8 waves.  Instead of four jobs on every wave (two of the
// 32 slots duplicated, four phantom), a wave takes THREE full row tiles of its column tile and a quarter of the
// k-range of row tile 12 (xmt) of the same column tile -- same weight stream, 3.25 jobs' worth of MFMAs instead of
// 4.  The quarter's raw sums go to Smem::part and are added up in a fixed order after the layer's barrier.
template <int NMT, typename AFrag, typename Epi>
__device__ __forceinline__ void gemm_tiles_split(const float *__restrict__ wpacked, int nt, int KB, int mt0, int xmt, int kpart,
                                                 AFrag afrag, Epi epi, float *part /* [256] of this (nt, kpart) */) {
    const int lane = threadIdx.x & 63;
    const int kb0 = (KB * kpart) >> 2, kb1 = (KB * (kpart + 1)) >> 2;
    f32x4 acc[NMT], accx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NMT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(wpacked) + (size_t)nt * KB * 64 + lane;
    // software pipeline: weights two k-blocks ahead (L2 latency), activations one k-block ahead (LDS latency)
    f32x4 b0 = bp[0], b1 = bp[(size_t)(KB > 1 ? 1 : 0) * 64];
    f32x4 a[NMT], ax;
#pragma unroll
    for (int i = 0; i < NMT; i++) a[i] = afrag(mt0 + i, 0, i);
    ax = afrag(xmt, 0, NMT);
    for (int kb = 0; kb < KB; kb++) {
        const int k2 = kb + 2 < KB ? kb + 2 : KB - 1, k1 = kb + 1 < KB ? kb + 1 : KB - 1;
        const f32x4 b2 = bp[(size_t)k2 * 64];
        f32x4 an[NMT];
#pragma unroll
        for (int i = 0; i < NMT; i++) an[i] = afrag(mt0 + i, k1, i);
        const f32x4 axn = afrag(xmt, k1, NMT);
        const bool extra = (kb >= kb0) & (kb < kb1);                    // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < NMT; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b0[j], acc[i], 0, 0, 0);
            if (extra) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[j], b0[j], accx, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NMT; i++) a[i] = an[i];
        ax = axn; b0 = b1; b1 = b2;
    }
#pragma unroll
    for (int i = 0; i < NMT; i++) epi(mt0 + i, acc[i]);
    *reinterpret_cast<f32x4 *>(&part[lane * 4]) = accx;                 // D-fragment order: [lane][reg]
}

// after the barrier: row tile 12 of a 32-column layer = ((part0 + part1) + part2) + part3 + bias, ReLU
__device__ __forceinline__ void reduce_split_tile(const float (*part)[4][256], const float *__restrict__ bias, float *y, int xmt) {
    const int tid = threadIdx.x;
    if (tid < 512) {
        const int nt = tid >> 8, e = tid & 255, lane = e >> 2, reg = e & 3;
        const int row = xmt * 16 + 4 * (lane >> 4) + reg, col = nt * 16 + (lane & 15);
        const float v = ((part[nt][0][e] + part[nt][1][e]) + part[nt][2][e]) + part[nt][3][e] + bias[col];
        y[row * LDY + col] = v > 0.f ? v : 0.f;
    }
}

// D fragment -> rows: lane holds column (lane & 15) of rows 16 mt + 4 (lane >> 4) + reg
template <typename F>
__device__ __forceinline__ void for_each_out(int mt, const f32x4 &acc, F f) {
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, r0 = mt * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) f(r0 + reg, col, acc[reg]);
}

#ifdef CCSP_STAMPS                        // diagnostic build only (tools/stamps_net.py): s_memtime at the layer boundaries of workgroup 0
__device__ unsigned long long net_stamps[64];
#define NET_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) net_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NET_STAMP(i) do { } while (0)
#endif

constexpr int NTH = 512;                 // 8 waves per workgroup = 2 per SIMD (one workgroup per CU: 133 KB of LDS)

__global__ __launch_bounds__(NTH) void net_forward_kernel(const float *__restrict__ W, const float *__restrict__ planes, int n,
                                                          float *__restrict__ logits_out, double *__restrict__ p_out,
                                                          float *__restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem &S = *reinterpret_cast<Smem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: tile choices and k-ranges become scalar branches
    const int q = lane >> 4, l15 = lane & 15;
    const long long s0 = (long long)blockIdx.x * NB;              // first position of this workgroup
    const int here = (int)((n - s0) < NB ? (n - s0) : NB);

    // ---- input planes -> LDS [NB*49][8] (channel 7 = 0), aliasing y1 ---------------------------------
    float *in = S.y1;
    for (int i = tid; i < NB * 49 * LDI; i += NTH) {
        const int s = i / (49 * LDI), rem = i % (49 * LDI), cell = rem / LDI, ch = rem % LDI;
        in[i] = (ch < 7 && s < here) ? planes[(s0 + s) * 343 + cell * 7 + ch] : 0.0f;
    }
    __syncthreads();
    NET_STAMP(0);
#ifdef CCSP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) net_stamps[60] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- stem: 3x3 valid, K = 9 taps x 8 -> 5 k-blocks of 2 taps (10th tap = zero weights) -----------
    {
        auto afrag = [&](int mt, int kb, int) -> f32x4 {
            const int row = mt * 16 + l15;
            const int s = row / 25, pos = row % 25, r = pos / 5, c = pos % 5;
            const int tap = kb * 2 + (q >> 1);
            const int dr = tap / 3, dc = tap % 3;
            const bool ok = (row < ROWS) & (tap < 9);
            const int src = s * 49 + (r + dr) * 7 + (c + dc);       // padding rows / the 10th tap read on inside y1 and are zeroed
            f32x4 a = *reinterpret_cast<const f32x4 *>(&in[src * LDI + (q & 1) * 4]);
            if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
            return a;
        };
        const int nt = wave & 3, mt0 = (wave >> 2) * 7;        // 4 column tiles x 2 row halves (tile 13 phantom)
        const float bv = W[LAY.stem_b + nt * 16 + l15];        // this lane's output column
        auto epi = [&](int mt, const f32x4 &acc) {
            for_each_out(mt, acc, [&](int row, int col, float v) {
                const float o = v + bv;
                S.x[row * LDX + nt * 16 + col] = o > 0.f ? o : 0.f;
            });
        };
        gemm_tiles<7>(W + LAY.stem_w, nt, 5, mt0, afrag, epi);
    }
    __syncthreads();
    NET_STAMP(1);

    // 3x3 layers: per row tile of this wave, once for all nine blocks: the row's address and which of the 9 taps stay
    // inside its 5x5 map (slots 0-2: the wave's full tiles; slot 3: row tile 12, of which it computes a quarter of the
    // k-range)
    int rowaddr[4]; uint32_t tapmask[4];
    {
        const int mt0 = 3 * (wave >> 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (i < 3 ? mt0 + i : 12) * 16 + l15;
            const int pos = row % 25, r = pos / 5, c = pos % 5;
            uint32_t m = 0;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int dr = t / 3 - 1, dc = t % 3 - 1;
                if ((r + dr >= 0) & (r + dr < 5) & (c + dc >= 0) & (c + dc < 5)) m |= 1u << t;
            }
            tapmask[i] = row < ROWS ? m : 0u;
            rowaddr[i] = row * LDY + 4 * q;
        }
    }

    // ---- nine bottleneck residual blocks (model.py:120-145) ------------------------------------------
    for (int blk = 0; blk < 9; blk++) {
        {   // 1x1 64 -> 32: 2 column tiles x 4 row groups of three tiles + a quarter of tile 12's k-range each
            const int nt = wave & 1, qr = wave >> 1, mt0 = 3 * qr;
            auto afrag = [&](int mt, int kb, int) -> f32x4 {
                return *reinterpret_cast<const f32x4 *>(&S.x[(mt * 16 + l15) * LDX + kb * 16 + 4 * q]);
            };
            const float bv = W[LAY.l1_b[blk] + nt * 16 + l15];
            auto epi = [&](int mt, const f32x4 &acc) {
                for_each_out(mt, acc, [&](int row, int col, float v) {
                    const float o = v + bv;
                    S.y1[row * LDY + nt * 16 + col] = o > 0.f ? o : 0.f;
                });
            };
            gemm_tiles_split<3>(W + LAY.l1_w[blk], nt, 4, mt0, 12, qr, afrag, epi, S.part[nt][qr]);
        }
        __syncthreads();
        reduce_split_tile(S.part, W + LAY.l1_b[blk], S.y1, 12);
        __syncthreads();
        NET_STAMP(2 + 3 * blk);
        {   // 3x3 same 32 -> 32: k-block kb = tap (kb >> 1), channels 16 (kb & 1) ..; zero halo outside the 5x5 map
            const int nt = wave & 1, qr = wave >> 1, mt0 = 3 * qr;
            auto afrag = [&](int, int kb, int i) -> f32x4 {
                const int tap = kb >> 1;                                        // wave-uniform
                const int toff = ((tap / 3 - 1) * 5 + (tap % 3 - 1)) * LDY + (kb & 1) * 16;
                const bool ok = (tapmask[i] >> tap) & 1u;
                // a tap outside the 5x5 map still reads ITS OWN shifted address (inside Smem: the tail of x or y1's
                // phantom rows) and is zeroed afterwards: a common dummy address would collide with the lane that
                // owns those banks in every 8-lane group of the ds_read_b128
                f32x4 a = *reinterpret_cast<const f32x4 *>(&S.y1[rowaddr[i] + toff]);
                if (!ok) a = f32x4{0.f, 0.f, 0.f, 0.f};
                return a;
            };
            const float bv = W[LAY.l2_b[blk] + nt * 16 + l15];
            auto epi = [&](int mt, const f32x4 &acc) {
                for_each_out(mt, acc, [&](int row, int col, float v) {
                    const float o = v + bv;
                    S.y2[row * LDY + nt * 16 + col] = o > 0.f ? o : 0.f;
                });
            };
            gemm_tiles_split<3>(W + LAY.l2_w[blk], nt, 18, mt0, 12, qr, afrag, epi, S.part[nt][qr]);
            NET_STAMP(32 + blk);                             // diagnostic: wave 0 done with its share of the 3x3 layer
#ifdef CCSP_STAMPS
            if (blockIdx.x == 0 && lane == 0 && blk == 4) {
                net_stamps[44 + wave] = __builtin_amdgcn_s_memtime();
                net_stamps[52 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID: wave slot, SIMD, CU...
            }
#endif
        }
        __syncthreads();
        reduce_split_tile(S.part, W + LAY.l2_b[blk], S.y2, 12);
        __syncthreads();
        NET_STAMP(3 + 3 * blk);
        {   // 1x1 32 -> 64 + residual: 4 column tiles x 2 row halves
            const int nt = wave & 3, mt0 = (wave >> 2) * 7;
            auto afrag = [&](int mt, int kb, int) -> f32x4 {
                return *reinterpret_cast<const f32x4 *>(&S.y2[(mt * 16 + l15) * LDY + kb * 16 + 4 * q]);
            };
            const float bv = W[LAY.l3_b[blk] + nt * 16 + l15];
            auto epi = [&](int mt, const f32x4 &acc) {
                for_each_out(mt, acc, [&](int row, int col, float v) {
                    float *px = &S.x[row * LDX + nt * 16 + col];
                    const float o = v + bv + *px;                               // add([x, block_input]) then ReLU
                    *px = o > 0.f ? o : 0.f;
                });
            };
            gemm_tiles<7>(W + LAY.l3_w[blk], nt, 2, mt0, afrag, epi);
        }
        __syncthreads();
        NET_STAMP(4 + 3 * blk);
    }

    // ---- policy head: 1x1 64 -> 16 (+ReLU) into pc[row][16] (contiguous = [position][400]), aliasing y1 ----
    float *pc = S.y1;                                    // [224][16] floats; only rows < 200 are read
    {
        const int mt0 = wave < 6 ? 2 * wave : 12;            // two tiles per wave; waves 6 and 7 both take tiles 12, 13
        auto afrag = [&](int mt, int kb, int) -> f32x4 {
            return *reinterpret_cast<const f32x4 *>(&S.x[(mt * 16 + l15) * LDX + kb * 16 + 4 * q]);
        };
        const float bv = W[LAY.pc_b + l15];
        auto epi = [&](int mt, const f32x4 &acc) {
            for_each_out(mt, acc, [&](int row, int col, float v) {
                const float o = v + bv;
                pc[row * 16 + col] = o > 0.f ? o : 0.f;
            });
        };
        gemm_tiles<2>(W + LAY.pc_w, 0, 4, mt0, afrag, epi);
    }
    // ---- value head, part 1: 1x1 64 -> 1 (+ReLU) per row, into y2[0..199] -----------------------------
    float *vc = S.y2;                                    // [200]
    float *lg = S.y2 + 256;                              // logits [NB][NPOL_PAD]
    if (tid < ROWS) {
        const float *wv = W + LAY.vc_w;
        float acc = W[LAY.vc_b];
        const float *xr = &S.x[tid * LDX];
#pragma unroll 8
        for (int k = 0; k < 64; k++) acc += xr[k] * wv[k];
        vc[tid] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    NET_STAMP(29);

    // ---- policy dense 400 -> 294: M = positions (one 16-row tile, rows >= NB are zero), 19 column tiles ---
    // A wave owns column tiles wave, wave + 8, wave + 16 and runs them TOGETHER: one A fragment per k-block feeds
    // three independent accumulators, and the weights (470 KB, streamed from L2 by every workgroup) are fetched
    // two k-blocks ahead on three streams -- with one tile at a time the layer waited for one load per 4 MFMAs.
    {
        constexpr int PD = 2, KBP = 25;
        const float *bias = W + LAY.pf_b;
        const f32x4 *bp[3];
        bool valid[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int nt = wave + 8 * i;
            valid[i] = nt < 19;
            bp[i] = reinterpret_cast<const f32x4 *>(W + LAY.pf_w) + (size_t)(valid[i] ? nt : 0) * KBP * 64 + lane;
        }
        f32x4 acc[3], bq[PD][3];
#pragma unroll
        for (int i = 0; i < 3; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < PD; d++)
#pragma unroll
            for (int i = 0; i < 3; i++) bq[d][i] = bp[i][(size_t)d * 64];
        const float *arow = &pc[(l15 < NB ? l15 : 0) * 400 + 4 * q];
        for (int kb = 0; kb < KBP; kb++) {
            f32x4 a = *reinterpret_cast<const f32x4 *>(arow + kb * 16);
            if (l15 >= NB) a = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 bn[3];
            const int kn = kb + PD < KBP ? kb + PD : KBP - 1;
#pragma unroll
            for (int i = 0; i < 3; i++) bn[i] = bp[i][(size_t)kn * 64];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bq[0][i][j], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int d = 0; d + 1 < PD; d++) bq[d][i] = bq[d + 1][i];
                bq[PD - 1][i] = bn[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int nt = wave + 8 * i;
            if (valid[i])
                for_each_out(0, acc[i], [&](int row, int col, float v) {
                    if (row < NB) lg[row * NPOL_PAD + nt * 16 + col] = v + bias[nt * 16 + col];
                });
        }
    }
    // ---- value head, part 2: dense 25 -> 32 ReLU (thread = (position, unit)), then 32 -> 1 tanh ---------
    float *h1 = S.y2 + 256 + NB * NPOL_PAD;              // [NB][32]
    if (tid < NB * 32) {
        const int s = tid >> 5, u = tid & 31;            // 8 positions x 32 units = 256 threads
        const float *w1 = W + LAY.f1_w;                  // [25][32] (Keras [in][out])
        float acc = W[LAY.f1_b + u];
#pragma unroll 5
        for (int i = 0; i < 25; i++) acc += vc[s * 25 + i] * w1[i * 32 + u];
        h1[s * 32 + u] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    NET_STAMP(30);
    if (tid < here) {
        const float *w2 = W + LAY.f2_w;
        float acc = W[LAY.f2_b];
        for (int u = 0; u < 32; u++) acc += h1[tid * 32 + u] * w2[u];
        v_out[s0 + tid] = tanhf(acc);
    }
    // ---- logits out + float64 softmax (utils.softmax, utils.py:187-192): one position per wave ---
    for (int s = wave; s < here; s += NTH / 64) {
        const float *row = lg + s * NPOL_PAD;
        float mx = -INFINITY;
        for (int i = lane; i < NPOL; i += 64) mx = fmaxf(mx, row[i]);
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
        double e[5];
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = lane + 64 * j;
            e[j] = i < NPOL ? exp((double)row[i] - (double)mx) : 0.0;
            sum += e[j];
        }
        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = lane + 64 * j;
            if (i < NPOL) {
                if (p_out) p_out[(s0 + s) * NPOL + i] = e[j] / sum;
                if (logits_out) logits_out[(s0 + s) * NPOL + i] = row[i];
            }
        }
    }
    NET_STAMP(31);
#ifdef CCSP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) net_stamps[61] = __builtin_amdgcn_s_memrealtime();
#endif
}

// pack one GEMM weight matrix W[K][N] (row-major, K x N valid, zero padded) into [nt][kb][lane][j]
void pack_gemm(const std::vector<float> &Wkn, int K, int N, int KB, int NT, float *out) {
    for (int nt = 0; nt < NT; nt++)
        for (int kb = 0; kb < KB; kb++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 4; j++) {
                    const int k = 16 * kb + 4 * (lane >> 4) + j, nn = 16 * nt + (lane & 15);
                    out[((nt * KB + kb) * 64 + lane) * 4 + j] = (k < K && nn < N) ? Wkn[(size_t)k * N + nn] : 0.0f;
                }
}

}  // namespace

extern "C" {

int ccsp_net_plain_size(void) { return PLAIN_TOTAL; }
int ccsp_net_packed_size(void) { return LAY.total; }

// Host: plain = the network's parameters in Keras order with each BatchNormalization folded into the
// convolution in front of it: stem kernel HWIO (3,3,7,64), bias 64; per block kernel (64,32), bias, kernel
// HWIO (3,3,32,32), bias, kernel (32,64), bias; policy conv (64,16), bias; policy dense (400,294) with rows
// in Keras' (h, w, c) flatten order, bias 294; value conv (64), bias 1; dense_1 (25,32), bias 32; value dense
// (32), bias 1.  packed = the layout net_forward_kernel reads (ccsp_net_packed_size() floats).
int ccsp_net_pack(const float *plain, float *packed) {
    if (!plain || !packed) return CCSP_EINVAL;
    memset(packed, 0, sizeof(float) * (size_t)LAY.total);
    const float *p = plain;
    {   // stem: k = tap * 8 + ch (ch 7 = zero pad), tap = dr * 3 + dc; plain HWIO index ((dr*3+dc)*7 + ch)*64 + o
        std::vector<float> Wkn(80 * 64, 0.0f);
        for (int tap = 0; tap < 9; tap++)
            for (int ch = 0; ch < 7; ch++)
                for (int o = 0; o < 64; o++) Wkn[(size_t)(tap * 8 + ch) * 64 + o] = p[(tap * 7 + ch) * 64 + o];
        pack_gemm(Wkn, 80, 64, 5, 4, packed + LAY.stem_w);
        p += 4032;
        memcpy(packed + LAY.stem_b, p, 64 * sizeof(float)); p += 64;
    }
    for (int b = 0; b < 9; b++) {
        { std::vector<float> Wkn(p, p + 64 * 32); pack_gemm(Wkn, 64, 32, 4, 2, packed + LAY.l1_w[b]); p += 2048; }
        memcpy(packed + LAY.l1_b[b], p, 32 * sizeof(float)); p += 32;
        { std::vector<float> Wkn(p, p + 288 * 32); pack_gemm(Wkn, 288, 32, 18, 2, packed + LAY.l2_w[b]); p += 9216; }   // HWIO: k = tap*32 + ch
        memcpy(packed + LAY.l2_b[b], p, 32 * sizeof(float)); p += 32;
        { std::vector<float> Wkn(p, p + 32 * 64); pack_gemm(Wkn, 32, 64, 2, 4, packed + LAY.l3_w[b]); p += 2048; }
        memcpy(packed + LAY.l3_b[b], p, 64 * sizeof(float)); p += 64;
    }
    { std::vector<float> Wkn(p, p + 64 * 16); pack_gemm(Wkn, 64, 16, 4, 1, packed + LAY.pc_w); p += 1024; }
    memcpy(packed + LAY.pc_b, p, 16 * sizeof(float)); p += 16;
    { std::vector<float> Wkn(p, p + 400 * 294); pack_gemm(Wkn, 400, 294, 25, 19, packed + LAY.pf_w); p += 117600; }      // row (h*5+w)*16 + c = pc's own order
    memcpy(packed + LAY.pf_b, p, 294 * sizeof(float)); p += 294;
    memcpy(packed + LAY.vc_w, p, 64 * sizeof(float)); p += 64;
    packed[LAY.vc_b] = *p; p += 1;
    memcpy(packed + LAY.f1_w, p, 800 * sizeof(float)); p += 800;
    memcpy(packed + LAY.f1_b, p, 32 * sizeof(float)); p += 32;
    memcpy(packed + LAY.f2_w, p, 32 * sizeof(float)); p += 32;
    packed[LAY.f2_b] = *p; p += 1;
    return (p - plain) == PLAIN_TOTAL ? CCSP_OK : CCSP_EINVAL;
}

// Device: Model.predict (model.py:21-24) for a batch.  packed: device copy of ccsp_net_pack's output; planes
// [n][7][7][7] f32; logits [n][294] f32 (or NULL); p [n][294] f64 softmax (or NULL); v [n] f32.
int ccsp_net_forward(const float *packed, const float *planes, int n, float *logits, double *p, float *v, void *stream) {
    if (n < 0 || (n > 0 && (!packed || !planes || !v))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    // the 133 KB dynamic-LDS opt-in is a per-device property of the kernel: once per device ordinal, not per process
    static bool attr_set[64] = {false};
    int dev = 0;
    CCSP_HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        CCSP_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(net_forward_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem)));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int grid = (n + NB - 1) / NB;
    hipLaunchKernelGGL(net_forward_kernel, dim3(grid), dim3(NTH), sizeof(Smem), (hipStream_t)stream, packed, planes, n, logits, p, v);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

#ifdef CCSP_STAMPS
int ccsp_debug_net_stamps(unsigned long long *out) {
    CCSP_HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(net_stamps), 64 * sizeof(unsigned long long)));
    return CCSP_OK;
}
#endif

}  // extern "C"
