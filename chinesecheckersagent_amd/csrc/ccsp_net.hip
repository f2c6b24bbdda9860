// ccsp_net.hip -- the policy/value network (graph of model.py:58-145) as ONE
// fused HIP kernel for gfx950: a workgroup carries 8 positions through the whole network with every
// activation resident in LDS, each layer an fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains,
// the reference's own arithmetic type) GEMM whose weights (BatchNorm folded in at load time) stream from L2
// in a pre-packed per-lane order.  Replaces ~90 MIOpen/elementwise launches per forward.
//
//   rows = position * 25 + (r * 5 + c), or -- <8,8> and <4,4>, round 5 -- cell-major (Cfg::CELLMAJOR): 200 rows per workgroup of 8
//   positions, padded to 13 tiles of 16
//   stem   3x3 valid 7->64      : A = implicit im2col of the 7x7x7 planes      K = 9 taps x 8 (7 + zero pad)
//   block  1x1 64->32, 3x3 same 32->32 (implicit im2col with zero halo), 1x1 32->64 + residual, ReLU each
//   policy 1x1 64->16, flatten (h, w, c) 400 -> dense 294 logits               (M = positions)
//   value  1x1 64->1, flatten 25 -> dense 32 ReLU -> dense 1 tanh              (plain FMA, tiny)
//   epilogue: p = float64 softmax(logits) (utils.softmax, utils.py:187-192), v float32
//
// Inner loops carry NO vector-ALU instruction besides the MFMAs: on gfx950 the fp32 MFMA shares the SIMD's fp32 lanes
// with ordinary VALU work, and every v_cndmask / v_add between MFMAs was measured to cost 10-15 cycles of matrix pipe
// (tools/probe/mfma_probe.hip: 99.7 % of the pipe on bare MFMAs, 67 % with one v_cndmask per MFMA).  Hence: the 3x3
// layers read their input from a copy with a ZERO HALO instead of masking taps (and, with cell-major rows, skip the k-blocks of an edge
// tile that would read nothing but halo), weights come through buffer loads whose addresses are scalar (SGPR offset + immediate), and
// the k-split's extra MFMAs sit behind one scalar branch per k-block or, where the layer is specialised per row group, behind none.
// The weights are the MFMA's FIRST operand (tile_out below): a lane ends up with four consecutive channels of one row.
// An evaluation is a function of the position alone: every output is formed by the same chains in the same order
// whatever the batch size, the slot in the batch or the row tile (gemm_tiles_split).
//
// Packed weight order of a GEMM layer with K = 16*KB, N = 16*NT:  [nt][kb][lane][j]  =
// W[k = 16 kb + 4 (lane >> 4) + j][n = 16 nt + (lane & 15)]  -- one 16-byte load per lane per 16 k.
// Within a 16-k block the MFMA k-slot q = lane >> 4 therefore carries k = 4q + j in step j; A is read
// with the same mapping (one ds_read_b128 per lane per 16 k).
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>
#include "ccsp_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CCSP_NET_LDX
#define CCSP_NET_LDX 68
#define CCSP_NET_LDY 36
#endif
constexpr int LDX = CCSP_NET_LDX;        // row stride of the 64-channel buffer (floats): 16-byte aligned, bank-spread
constexpr int LDY = CCSP_NET_LDY;        // row stride of the 32-channel buffers
constexpr int LDI = 12;                  // input planes: 7 channels + zeros; 12 spreads eight consecutive cells over all banks
constexpr int NPOL = 294, NPOL_PAD = 304;
#ifndef CCSP_NET_LDS_BIAS_ALL
#define CCSP_NET_LDS_BIAS_ALL 0          // 1: the trunk's biases through LDS in <8,8> too (see bias4 in the kernel)
#endif
#ifndef CCSP_NET_PDG
#define CCSP_NET_PDG 1                   // policy dense: groups of four k the weight loads run ahead
#endif
// k-segments (accumulation chains) every output of a layer kind is summed from, ((c0 + c1) + c2) + ... in fixed order -- the same in
// every workgroup shape.  A float32 FMA chain over K terms drifts like sqrt(K) roundings; cutting it into segments of 16-48 k costs
// NSEG - 1 vector adds per output and no MFMA.  MEASURED in round 4 on 4096 self-play positions x 3 weight files (3.6 M logits,
// profiles/r4_n1_wide_variants.json): two segments in the stem and the first 1x1 take the logits further than 1e-5 from the float64
// restatement from 215 to 82 and the mean distance down by a tenth, but NOT the maximum (2.0e-5 -> 2.2e-5; it wanders between 1.8e-5
// and 2.7e-5 over six segmentations) -- and the adds take the fp32 lanes the MFMA shares: +0.8 % kernel time.  Not adopted: all 1.
#ifndef CCSP_NET_SEG_STEM
#define CCSP_NET_SEG_STEM 1              // stem 3x3, K = 80 (5 k-blocks)
#endif
#ifndef CCSP_NET_SEG_L1
#define CCSP_NET_SEG_L1 1                // blocks' first 1x1, K = 64 (4 k-blocks)
#endif
#ifndef CCSP_NET_SEG_L2
#define CCSP_NET_SEG_L2 4                // blocks' 3x3, K = 288 (18 k-blocks); a multiple of the waves sharing the last row tile (4)
#endif
#ifndef CCSP_NET_SEG_L3
#define CCSP_NET_SEG_L3 1                // blocks' last 1x1, K = 32 (2 k-blocks)
#endif
#ifndef CCSP_NET_SEG_PC
#define CCSP_NET_SEG_PC 1                // policy 1x1, K = 64 (4 k-blocks)
#endif

// ---- packed weight blob layout (floats) ----------------------------------------------------------------
struct Layout {
    int trunk_b;
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
    // the trunk's biases stand TOGETHER (stem 64 | per block 32, 32, 64 | policy conv 16: the order of Smem::bias, BIAS_* below), so that
    // a workgroup copies them into LDS with one 16-byte load per thread
    L.trunk_b = take(64 + 9 * 128 + 16);
    L.stem_w = take(4 * 5 * 256); L.stem_b = L.trunk_b;
    for (int i = 0; i < 9; i++) {
        L.l1_w[i] = take(2 * 4 * 256); L.l1_b[i] = L.trunk_b + 64 + i * 128;
        L.l2_w[i] = take(2 * 18 * 256); L.l2_b[i] = L.trunk_b + 64 + i * 128 + 32;
        L.l3_w[i] = take(4 * 2 * 256); L.l3_b[i] = L.trunk_b + 64 + i * 128 + 64;
    }
    L.pc_w = take(1 * 4 * 256); L.pc_b = L.trunk_b + 64 + 9 * 128;
    L.pf_w = take(5 * 100 * 256); L.pf_b = take(NPOL_PAD);      // [tile of 64 columns][group of 4 k][lane = column][k]
    L.vc_w = take(64); L.vc_b = take(1);
    L.f1_w = take(25 * 32); L.f1_b = take(32);
    L.f2_w = take(32); L.f2_b = take(1);
    L.total = o;
    return L;
}
constexpr Layout LAY = make_layout();
// The nine blocks' entries are equally spaced in the blob: inside the block loop an offset is l*_x[0] + blk * BLK_STRIDE -- scalar
// arithmetic -- instead of a look-up of the table in constant memory (an s_load and an `s_waitcnt lgkmcnt(0)` -- which also waits
// for every LDS read in flight -- in the middle of a layer's MFMA stream, six times per block).
constexpr int BLK_STRIDE = LAY.l1_w[1] - LAY.l1_w[0];
constexpr bool layout_is_regular() {
    for (int i = 0; i < 9; i++)
        if (LAY.l1_w[i] != LAY.l1_w[0] + i * BLK_STRIDE || LAY.l2_w[i] != LAY.l2_w[0] + i * BLK_STRIDE || LAY.l3_w[i] != LAY.l3_w[0] + i * BLK_STRIDE) return false;
    return true;
}
static_assert(layout_is_regular(), "the blocks' weights are BLK_STRIDE floats apart");
// The trunk's biases in LDS (Smem::bias): stem 64 | per block: first 1x1 32, 3x3 32, last 1x1 64 | policy conv 16.  An epilogue that
// reads its bias from global memory waits with `s_waitcnt vmcnt(0)` -- for the bias AND for the next layer's weight k-blocks that were
// requested a moment earlier (an L2 round trip, exposed in the wave whose epilogue nothing hides: seen in the listing, round 5).
constexpr int BIAS_STEM = 0, BIAS_BLK = 64, BIAS_PER_BLK = 128, BIAS_L1 = 0, BIAS_L2 = 32, BIAS_L3 = 64, BIAS_PC = BIAS_BLK + 9 * BIAS_PER_BLK, BIAS_N = BIAS_PC + 16;
static_assert(LAY.stem_b == LAY.trunk_b + BIAS_STEM && LAY.l1_b[3] == LAY.trunk_b + BIAS_BLK + 3 * BIAS_PER_BLK + BIAS_L1 && LAY.l2_b[8] == LAY.trunk_b + BIAS_BLK + 8 * BIAS_PER_BLK + BIAS_L2 &&
              LAY.l3_b[0] == LAY.trunk_b + BIAS_BLK + BIAS_L3 && LAY.pc_b == LAY.trunk_b + BIAS_PC && BIAS_N % 4 == 0 && LAY.trunk_b % 4 == 0, "Smem::bias is the blob's trunk_b region");

// plain (Keras-order, BatchNorm already folded) input of ccsp_net_pack: offsets in floats
constexpr int PLAIN_TOTAL = 4032 + 64 + 9 * (2048 + 32 + 9216 + 32 + 2048 + 64) + 1024 + 16 + 117600 + 294 + 64 + 1 + 800 + 32 + 32 + 1;
static_assert(PLAIN_TOTAL == 244920, "249852 parameters minus the 4 x 1233 BatchNorm values folded away");

// 3x3 input with a SHARED zero halo: position s, cell (r, c) of the 5x5 map sits at row PAD0 + 36 s + 6 r + c -- every map row
// is followed by one zero cell (the right neighbour of column 4 AND the left neighbour of the next row's column 0), every map by
// one zero row of six (the bottom halo of this map AND the top halo of the next).  Neighbour (dr, dc) = + 6 dr + dc.
constexpr int PAD0 = 7, PADPOS = 36;
// A workgroup of NW waves carries NB positions (rows = position * 25 + cell, MT tiles of 16 rows).  Four shapes are built:
//   <8, 8>: 200 rows = 13 tiles (4 % padding), one 132-KB workgroup per CU, two waves per SIMD from the SAME workgroup --
//           at every barrier both are out of matrix work at once;
//   <4, 4>: 100 rows = 7 tiles (12 % padding), 78 KB, four waves: a workgroup is done in 0.65 of the time -- the shape of batches that
//           do not fill the GPU (up to 1024 positions), whose launch is as long as one workgroup;
//   <2, 8>: 50 rows = 4 tiles (22 % padding), 53 KB, eight waves with ONE tile job each in the 32-column layers: the shape of
//           the smallest batches (one game of selfplay(), the arena's few games per GPU), where only latency counts.
// Tile shares: 64-column layers -- wave & 3 = column tile, F64 (second half: F64B) row tiles from M64 * (wave >> 2) (until round 5 the two halves of <8, 8> both
// compute tile 6); 32-column layers -- wave & 1 = column tile, F32 row tiles from F32 * (wave >> 1) plus, where the tiles do not
// divide (XT), a 1/NSPLIT share of the k-range of the last tile MT - 1.  Every shape forms every output by the same chains in the
// same order: a position's result does not depend on the shape.
//   <1, 8>: 25 rows = 2 tiles, 53 KB (the heads' scratch for four positions), eight waves on ONE position (round 5): every tile job of the 3x3 layers is shared by TWO waves
//           (half the k-range = two of the four segments each), the 1x1 layers have a job for every wave or every second one -- half the
//           matrix work per CU of <2, 8>: the shape of the batches whose whole launch is ONE workgroup per CU and nothing but latency
//           (one game of selfplay(), the arena's and config 5's few dozen slots per GPU: up to 256 positions).
#ifndef CCSP_NET_SKIP_DEAD_TAPS
#define CCSP_NET_SKIP_DEAD_TAPS 1
#endif
#ifndef CCSP_NET_KB_FENCE
#define CCSP_NET_KB_FENCE 1
#endif
#ifndef CCSP_NET_CM4
#define CCSP_NET_CM4 1
#endif
// Cell-major rows (<8, 8> and <4, 4>): a row tile holds 16 / NB cells x NB positions (row 16 t + NB h + s = position s of the tile's h-th
// cell), the last tile the 25th cell (+ padding rows).  The ring of 16 border cells is cut into four runs of four, one per map edge
// (0 top, 1 right, 2 bottom, 3 left): an EDGE TILE is cells of one run -- <8, 8>: two tiles per edge, <4, 4>: one.
//   <8, 8>: row group g = wave >> 1 of the 32-column layers owns tiles 3 g, 3 g + 1 (edge g) and 3 g + 2 (a pair of interior cells);
//   <4, 4>: row group g owns tiles 3 g, 3 g + 1 (edges 2 g and 2 g + 1) and 3 g + 2 (four interior cells).
constexpr unsigned char CELL8[25] = {0, 1, 2, 3, 6, 7,   4, 9, 14, 19, 11, 12,   24, 23, 22, 21, 16, 17,   20, 15, 10, 5, 8, 13,   18};
constexpr unsigned char CELL4[25] = {0, 1, 2, 3,  4, 9, 14, 19,  6, 7, 8, 11,   24, 23, 22, 21,  20, 15, 10, 5,  12, 13, 16, 17,   18};
constexpr int on_edge(int cell, int e) { return e == 0 ? cell / 5 == 0 : (e == 1 ? cell % 5 == 4 : (e == 2 ? cell / 5 == 4 : cell % 5 == 0)); }
constexpr bool interior(int cell) { return !on_edge(cell, 0) && !on_edge(cell, 1) && !on_edge(cell, 2) && !on_edge(cell, 3); }
constexpr bool cells_ok() {
    bool seen8[25] = {}, seen4[25] = {};
    for (int i = 0; i < 25; i++) {
        if (CELL8[i] > 24 || seen8[CELL8[i]] || CELL4[i] > 24 || seen4[CELL4[i]]) return false;
        seen8[CELL8[i]] = true; seen4[CELL4[i]] = true;
    }
    for (int g = 0; g < 4; g++)
        for (int i = 0; i < 6; i++)
            if (i < 4 ? !on_edge(CELL8[6 * g + i], g) : !interior(CELL8[6 * g + i])) return false;
    for (int g = 0; g < 2; g++)
        for (int i = 0; i < 12; i++)
            if (i < 8 ? !on_edge(CELL4[12 * g + i], 2 * g + i / 4) : !interior(CELL4[12 * g + i])) return false;
    return interior(CELL8[24]) && interior(CELL4[24]);
}
static_assert(cells_ok(), "CELL8 / CELL4: permutations; the edge tiles on their edges, the other tiles interior");
template <int NB>
constexpr unsigned long long cell_pack(int from) {
    unsigned long long v = 0;
    for (int i = 0; i < 12 && from + i < 25; i++) v |= (unsigned long long)(NB == 8 ? CELL8 : CELL4)[from + i] << (5 * i);
    return v;
}
template <int NB>
__device__ __forceinline__ int cell_of(int idx) {                // CELLn[idx] without a memory access (5-bit fields of three constants)
    const unsigned long long w = idx < 12 ? cell_pack<NB>(0) : (idx < 24 ? cell_pack<NB>(12) : cell_pack<NB>(24));
    return (int)((w >> (5 * (idx < 12 ? idx : (idx < 24 ? idx - 12 : 0)))) & 31);
}
// the k-blocks (bit kb: tap kb >> 1 = 3 (dr + 1) + (dc + 1)) of a 3x3 layer that read nothing but halo for a tile on edge g
constexpr unsigned EDGE_DEAD[4] = {0x0003Fu /* dr = -1 */, 0x30C30u /* dc = +1 */, 0x3F000u /* dr = +1 */, 0x030C3u /* dc = -1 */};
// Cell-major zero-halo input of the 3x3 layers: cell (r, c) of position s sits at row (6 r + c + CM_PAD) * NB + s -- the same shared halo
// per map row as below, all NB positions of a cell slot together; neighbour (dr, dc) = + (6 dr + dc) * NB rows.  Only the slots a
// LIVE tap can reach exist: from (-1, 3) (tap (-1, -1) of cell (0, 4), a right-edge tile) to (5, 1) (tap (+1, +1) of cell (4, 0)).
constexpr int CM_PAD = 3, CM_SLOTS = 35;

template <int NBv, int NWv>
struct Cfg {
    static constexpr int NB = NBv, NW = NWv, NTH = NWv * 64, ROWS = NBv * 25, MT = (NBv * 25 + 15) / 16;
    static constexpr int NSPLIT = NWv / 2;                       // row groups of the 32-column layers = waves sharing the k-range of tile MT - 1
    static constexpr bool ALLSPLIT = MT < NSPLIT;                // fewer row tiles than row groups (<1, 8>): EVERY tile of the 3x3 layers is k-split
    static constexpr int NSH = ALLSPLIT ? NSPLIT / MT : NSPLIT;  // waves sharing the k-range of a split tile
    static constexpr int XTILES = ALLSPLIT ? MT : 1;             // row tiles whose 3x3 outputs reach the next layer as partial sums (Smem::part)
    static constexpr int F32 = MT / NSPLIT;                      // full row tiles per wave in the 32-column layers (3, 3, 1; 0 when ALLSPLIT)
    static constexpr int F32A = F32 > 0 ? F32 : 1;               // (array extents: at least one slot)
    static constexpr int XT = ALLSPLIT ? 0 : MT - F32 * NSPLIT;  // 1: a last row tile shared by the NSPLIT waves of a column tile; 0: none
    static constexpr int NH = NWv / 4;                           // row halves of the 64-column layers
    static constexpr int F64 = (MT + NH - 1) / NH;               // row tiles per wave of the first half in the 64-column layers (7, 7, 2, 1)
    static constexpr int M64 = NH == 2 ? F64 : 0;                // first tile of the second half
    static constexpr int F64B = NH == 2 ? MT - F64 : F64;        // its row tiles per wave (<8, 8>: 6 -- the two waves of a SIMD, one of each half, have
                                                                 // 13 tile jobs between them; until round 5 both halves computed tile 6: 14)
    static constexpr bool PADFULL = F32 * NSPLIT * 16 > ROWS;    // the "full" tiles of the 32-column layers hold padding rows
    static constexpr int HB = NBv < 4 ? 4 : NBv;                 // positions the heads are laid out for (the 4 x 4 MFMA carries four at a time)
    static constexpr int NSEG = CCSP_NET_SEG_L2;                 // k-segments every output of a 3x3 layer is summed from (gemm_tiles_split): the
                                                                 // SAME in every shape, so that both shapes compute a position with the same bits
    // <8, 8> (round 5): rows in CELL-MAJOR order -- a row tile is two cells x eight positions (CELL8), so that a tile whose two cells lie on
    // the same edge of the 5 x 5 map has three taps of the 3x3 layers that fall off the map for EVERY one of its rows: their k-blocks
    // multiply nothing but halo zeros and are not issued (EDGE_DEAD; a skipped product is an exact zero: the sums keep their bits).
    static constexpr bool CELLMAJOR = (NBv == 8 || (NBv == 4 && CCSP_NET_CM4)) && CCSP_NET_SKIP_DEAD_TAPS;
    static constexpr int NE = CELLMAJOR ? 2 : 0;                 // a wave's first NE tiles of a 3x3 layer are edge tiles
    static constexpr unsigned dead_of(int g, int slot) { return NBv == 8 ? EDGE_DEAD[g] : EDGE_DEAD[2 * g + slot]; }   // row group g, tile slot 0 / 1
    // row group -> its share of the shared tile's k-range.  <8, 8> (one segment each; segments: k-blocks 0-3, 4-8, 9-12, 13-17): the
    // segment most of whose k-blocks are dead for the group's edge -- the shared tile's MFMAs fall where the wave has the fewest others
    static constexpr int xseg_of(int g) { return NBv == 8 ? (g == 2 ? 3 : (g == 3 ? 2 : g)) : g; }
    static constexpr int PADROWS_ = CELLMAJOR ? CM_SLOTS * NBv : PAD0 + NBv * PADPOS + 1, PADROWS_HEADS = (HB * 400 + LDY - 1) / LDY;
    static constexpr int PADROWS = PADROWS_ > PADROWS_HEADS ? PADROWS_ : PADROWS_HEADS;   // (the policy conv output of HB positions aliases y1)
    static constexpr int INROWS = NBv * 49 + 56;                 // staged input planes + what padding rows / the zero-weight 10th tap reach
    static constexpr int NTW = (19 + NWv - 1) / NWv;             // policy dense: column tiles per wave
#ifndef CCSP_NET_PDG_SMALL
#define CCSP_NET_PDG_SMALL 1
#endif
    // policy dense: groups of four k the weight loads run ahead.  A batch that fills the GPU streams the layer's 470 KB per workgroup at
    // what L2 delivers (one group ahead is enough; more was slower); the small shapes' few workgroups wait for L2 LATENCY instead.
    static constexpr int PDG = NBv <= 2 ? CCSP_NET_PDG_SMALL : CCSP_NET_PDG;
    static_assert((XT == 0 || XT == 1) && (F32 >= 1 || ALLSPLIT) && F32 <= 3 && F64 <= 7 && MT <= 13 && (NWv == 4 || NWv == 8), "tile shares");
    static_assert(!ALLSPLIT || (NSPLIT % MT == 0 && NSEG % NSH == 0 && F64 == 1 && F64B == 1 && M64 + 1 == MT), "<1, 8>: two waves per 3x3 tile job, one 64-column job per wave");
};

template <typename C>
struct Smem {
    float y1[C::PADROWS * LDY];          // 32-channel 1x1 output = 3x3 input, zero halo; the policy conv output aliases it
    static constexpr int Y2A = C::MT * 16 * LDY, Y2B = C::INROWS * LDI, Y2C = 256 + C::HB * (NPOL_PAD + 32);
    static constexpr int Y2N = Y2A > Y2B ? (Y2A > Y2C ? Y2A : Y2C) : (Y2B > Y2C ? Y2B : Y2C);
    float y2[Y2N];                       // 32-channel 3x3 output; the stem's input planes and the logits / value scratch alias it
    // (y1 and y2 FIRST: a DS instruction's immediate offset is 16 bits -- the cell-major taps of <8,8> reach 16 KB past a lane's base, and
    // with the 57-KB trunk buffer in front of y1 a third of them no longer fitted: an extra address register per tile, 174 -> 197 VGPRs)
    float x[C::MT * 16 * LDX > 4 * C::HB * 320 ? C::MT * 16 * LDX : 4 * C::HB * 320];   // 64-channel trunk activations; the policy dense
                                         // layer's partial sums [4][HB][320] alias it
    float part[2 * C::XTILES][C::NSEG][256];   // partial sums (one per k-segment) of the k-split row tiles of the 3x3 layers: [tile * 2 + column tile]
    float bias[(C::NB < 8 || CCSP_NET_LDS_BIAS_ALL) ? BIAS_N : 4];  // the trunk's biases (see BIAS_*): read by the epilogues through LDS in the small shapes
    static_assert(C::INROWS * LDI <= Y2N, "the staged input planes alias y2");
    static_assert(C::MT * 16 * 16 <= C::PADROWS * LDY && C::HB * 400 <= C::PADROWS * LDY, "the policy conv output aliases y1");
    static_assert(256 + C::HB * (NPOL_PAD + 32) <= Y2N, "logits and value scratch alias y2");
};
#ifndef CCSP_NET_NO_LDS_ASSERT              // (layout experiments)
static_assert(sizeof(Smem<Cfg<8, 8>>) + 3 * 6900 <= 160 * 1024, "<8,8>: one evaluator workgroup per CU plus three tree-kernel workgroups");
#endif
static_assert(sizeof(Smem<Cfg<4, 4>>) + 3 * 6900 <= 160 * 1024, "<4,4>: an evaluator workgroup per CU plus three tree-kernel workgroups (two per CU no longer fit: the shape carries batches of at most one workgroup per CU)");
static_assert(sizeof(Smem<Cfg<2, 8>>) <= 64 * 1024, "<2,8>: a small workgroup");
static_assert(sizeof(Smem<Cfg<1, 8>>) <= 64 * 1024, "<1,8>: a small workgroup");

// A per-lane LDS base address the compiler cannot see through.  A DS instruction carries a 16-bit immediate offset; clang folds the
// workgroup-relative offset of a Smem member and the per-tile constant into it where the sum fits -- and where it does not (the trunk
// buffer behind the first 64 KB: 70272 + tile * 4352), it materialises ONE ADDRESS REGISTER PER TILE in front of the block loop
// (22 VGPRs, measured with tools/vgpr_liveness.py).  Behind lds_opaque the member's offset sits in the lane's base register and the
// per-tile constant alone in the immediate.
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
__device__ __forceinline__ lds_float *lds_opaque(float *p) {
    unsigned a = (unsigned)(size_t)(lds_float *)p;
    asm volatile("" : "+v"(a));
    return (lds_float *)(size_t)a;
}
__device__ __forceinline__ f32x4 lds_load4(const lds_float *p, int off) { return *reinterpret_cast<const lds_f32x4 *>(p + off); }
__device__ __forceinline__ void lds_store4(lds_float *p, int off, const f32x4 &v) { *reinterpret_cast<lds_f32x4 *>(p + off) = v; }

// the packed weights as a buffer resource: loads take a scalar byte offset (+ the lane's 16 bytes), no vector address math
struct WBuf {
    __amdgpu_buffer_rsrc_t r;
    int voff;                            // lane * 16
    __device__ __forceinline__ f32x4 load(int float_off) const {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, float_off * 4, 0));
    }
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
// slot for per-tile precomputed data);  epi(mt, acc, i) consumes a tile (i = mt - mt0, static after unrolling) (bias, residual, ReLU: the bias is added AFTER the
// sum as Keras does -- starting the accumulator from it costs accuracy: 2.6e-5 instead of 1.5e-5 worst logit).  wbase = float offset of the layer's packed weights.
// No tile is ever skipped: where two waves share an odd number of tiles, both compute the middle one.
// Register discipline of both loops (every index is static after unrolling): the A fragments of k-block kb live in
// buffer kb & 1 and those of kb + 1 are read into the OTHER buffer, the weight ring slot refilled during k-block kb is the
// one k-block kb - 1 used -- a load never targets a register an MFMA issued a moment ago still reads (the compiler would
// otherwise pad such write-after-read hazards with s_nop).
// A layer's FIRST weight k-blocks arrive in `pre` (NPRE of them, requested by the previous layer before its epilogue and
// its barrier: an L2 round trip is a sixth of a 1x1 layer's MFMA time); next() is called once the last k-block's MFMAs
// are issued and requests the following layer's.
#ifndef CCSP_NET_NPRE
#define CCSP_NET_NPRE 3
#endif
constexpr int NPREMAX = CCSP_NET_NPRE;
template <int KB> struct Pre { static constexpr int N = KB < NPREMAX ? KB : NPREMAX; };
template <int KBN>
__device__ __forceinline__ void prefetch(const WBuf &wb, int wbase, int nt, f32x4 (&pre)[NPREMAX]) {
#pragma unroll
    for (int d = 0; d < Pre<KBN>::N; d++) pre[d] = wb.load(wbase + nt * KBN * 256 + d * 256);
}

template <int KB, int NSEG>
__device__ __forceinline__ constexpr int seg_of(int kb) {               // the k-segment k-block kb belongs to
    int seg = 0;
    for (int c = 1; c < NSEG; c++) seg += kb >= (KB * c) / NSEG ? 1 : 0;
    return seg;
}
template <int NMT, int KB, int NSEG, typename AFrag, typename Next, typename Epi>
__device__ __forceinline__ void gemm_tiles(const WBuf &wb, int wbase, int nt, int mt0, f32x4 (&pre)[NPREMAX], AFrag afrag, Next next, Epi epi) {
    constexpr int NPRE = Pre<KB>::N;
    constexpr int PB = NPRE + 1 < KB ? NPRE + 1 : KB;                    // weight ring: PB - 1 k-blocks in flight (L2 latency)
    static_assert(NSEG >= 1 && NSEG <= KB, "a segment is at least one k-block");
    f32x4 acc[NMT][NSEG];
#pragma unroll
    for (int i = 0; i < NMT; i++)
#pragma unroll
        for (int c = 0; c < NSEG; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int w0 = wbase + nt * KB * 256;
    f32x4 bq[PB];
#pragma unroll
    for (int d = 0; d < NPRE; d++) bq[d] = pre[d];
    f32x4 a[2][NMT];
#pragma unroll
    for (int i = 0; i < NMT; i++) a[0][i] = afrag(mt0 + i, 0, i);
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        int seg = 0;                                                    // the segment k-block kb belongs to (static after unrolling)
#pragma unroll
        for (int c = 1; c < NSEG; c++) seg += kb >= (KB * c) / NSEG ? 1 : 0;
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < NMT; i++) a[(kb + 1) & 1][i] = afrag(mt0 + i, kb + 1, i);   // one k-block ahead (LDS latency)
        }
        if (kb + PB - 1 < KB && kb + PB - 1 >= NPRE) bq[(kb + PB - 1) % PB] = wb.load(w0 + (kb + PB - 1) * 256);
        const f32x4 b = bq[kb % PB];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < NMT; i++)
                acc[i][seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][i][j], acc[i][seg], 0, 0, 0);   // D^T: see tile_out
        }
    }
    next();
#pragma unroll
    for (int i = 0; i < NMT; i++) {
        f32x4 sum = acc[i][0];
#pragma unroll
        for (int c = 1; c < NSEG; c++) sum = sum + acc[i][c];
        epi(mt0 + i, sum, i);
    }
}

// The 32-column layers have 13 x 2 = 26 tile jobs for 8 waves.  Instead of four jobs on every wave (two of the
// 32 slots duplicated, four phantom), a wave takes THREE full row tiles of its column tile and a quarter of the
// k-range of row tile 12 (xmt) of the same column tile -- same weight stream, 3.25 jobs' worth of MFMAs instead of
// 4.  The quarter's raw sums go to Smem::part and are added up in a fixed order after the layer's barrier.
// EVERY output of these layers -- full tiles too -- is the fixed-order sum ((c0 + c1) + c2) + c3 of NSEG = 4 accumulation chains
// over four segments of the k-range (in every workgroup shape: the NSH waves that share the last tile take NSEG / NSH segments each), so that a row's arithmetic does not depend on which tile (hence which slot of the
// batch) it sits in: an evaluation is a function of the position alone, whatever the batch size, the slot or the sharding.
// NE > 0 (<8, 8>, cell-major rows): the wave's first NE tiles lie on one edge of the map; DEAD0 / DEAD1 (tile slot 0 / 1) have bit kb set for the k-blocks whose
// tap falls off the map for every row of such a tile -- neither their MFMAs nor their LDS reads exist.  The chains that remain are the
// ones every other shape forms, minus exact zeros.  STATIC: dead mask and k-share are template constants -- the kernel holds one copy of
// the layer per row group and picks it with ONE branch per layer; the layer itself is straight-line code, the shared tile's MFMAs
// interleaved with the others'.  (Round 5, A/B on one box: the same skipping behind scalar branches per k-block -- three of them, taken
// or not -- gave 7 of the 12 us that the straight-line form gives.)
template <int NMT, int KB, int NSEG, int NSH, int NE, bool STATIC, unsigned DEAD0, unsigned DEAD1, int KPARTC, typename AFrag, typename Next, typename Epi>
__device__ __forceinline__ void gemm_tiles_split(const WBuf &wb, int wbase, int nt, int mt0, int xmt, int kpart_rt, f32x4 (&pre)[NPREMAX],
                                                 AFrag afrag, Next next, Epi epi, lds_float *part /* this lane's piece of segment kpart * SPW of this nt: [NSEG][256], + lane * 4 */) {
    constexpr int NPRE = Pre<KB>::N;
    constexpr int PB = NPRE + 1 < KB ? NPRE + 1 : KB;
    constexpr int SPW = NSEG / NSH;                                     // segments of the shared tile that one of its NSH waves computes
    static_assert(NSEG % NSH == 0, "the waves sharing the last tile take whole segments");
    static_assert(NE >= 0 && NE <= 2 && NE < NMT && (NE == 0 || STATIC), "at least one tile without dead taps; dead taps are compile-time");
    const int kpart = STATIC ? KPARTC : kpart_rt;
    constexpr auto is_live = [](int i, int kb) { return i >= NE || !(((i == 0 ? DEAD0 : DEAD1) >> kb) & 1u); };     // tile slot i in k-block kb
    f32x4 acc[NMT][NSEG], accx[SPW];
#pragma unroll
    for (int i = 0; i < NMT; i++)
#pragma unroll
        for (int c = 0; c < NSEG; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < SPW; c++) accx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int w0 = wbase + nt * KB * 256;
    f32x4 bq[PB];
#pragma unroll
    for (int d = 0; d < NPRE; d++) bq[d] = pre[d];
    f32x4 a[2][NMT + 1];
#pragma unroll
    for (int i = 0; i < NMT; i++)
        if (is_live(i, 0)) a[0][i] = afrag(mt0 + i, 0, i);
    if (!STATIC || seg_of<KB, NSEG>(0) / SPW == kpart) a[0][NMT] = afrag(xmt, 0, NMT);
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        int seg = 0;                                                    // the segment k-block kb belongs to (static after unrolling)
#pragma unroll
        for (int c = 1; c < NSEG; c++) seg += kb >= (KB * c) / NSEG ? 1 : 0;
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < NMT; i++)
                if (is_live(i, kb + 1)) a[(kb + 1) & 1][i] = afrag(mt0 + i, kb + 1, i);
            if (!STATIC || (seg_of<KB, NSEG>(kb + 1)) / SPW == kpart) a[(kb + 1) & 1][NMT] = afrag(xmt, kb + 1, NMT);
        }
        if (kb + PB - 1 < KB && kb + PB - 1 >= NPRE) bq[(kb + PB - 1) % PB] = wb.load(w0 + (kb + PB - 1) * 256);
        const f32x4 b = bq[kb % PB];
        if constexpr (STATIC) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int i = 0; i < NMT; i++)
                    if (is_live(i, kb)) acc[i][seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][i][j], acc[i][seg], 0, 0, 0);   // D^T: see tile_out
                if (seg / SPW == kpart) accx[seg % SPW] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][NMT][j], accx[seg % SPW], 0, 0, 0);
            }
#if CCSP_NET_KB_FENCE
            __builtin_amdgcn_sched_barrier(0);                           // the layer is ONE basic block: without a fence per k-block the max-ILP scheduler
                                                                        // pulls the loads of many k-blocks to the front (206 VGPRs)
#endif
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int i = 0; i < NMT; i++)
                    acc[i][seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][i][j], acc[i][seg], 0, 0, 0);   // D^T: see tile_out
            }
            if (seg / SPW == kpart) {                                   // wave-uniform: ONE scalar branch per k-block
#pragma unroll
                for (int j = 0; j < 4; j++)
                    accx[seg % SPW] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][NMT][j], accx[seg % SPW], 0, 0, 0);
            }
        }
    }
    next();
#pragma unroll
    for (int i = 0; i < NMT; i++) {
        f32x4 sum = acc[i][0];
#pragma unroll
        for (int c = 1; c < NSEG; c++) sum = sum + acc[i][c];            // the order in which the shared tile's partial sums are added up by its consumer
        epi(mt0 + i, sum, i);
    }
    if (kpart >= 0) {                                                   // (kpart < 0: a shape without a shared tile)
#pragma unroll
        for (int c = 0; c < SPW; c++) lds_store4(part, c * 256, accx[c]);   // D-fragment order: [segment][lane][reg]
    }
}

// <1, 8> (Cfg::ALLSPLIT): a wave's whole share of a 3x3 layer is a 1/NSH part of ONE tile's k-range -- whole segments of the same NSEG
// chains as everywhere else (kpart * SPW .. + SPW - 1; the segments of every part are equally long: asserted), their raw sums to `part`
// for the next layer to add up in the fixed order.  The part's k-blocks start at a wave-uniform RUNTIME index: afrag(kb) and the weight
// offset take it as a scalar; all of the part's weight k-blocks are requested up front by ksplit_load (KB / NSH registers of four: the
// caller issues it a layer early, so that the L2 round trip is over when the barrier in front of this layer opens), the activation
// fragments run one k-block ahead.
template <int KB, int NSH>            // the weight k-blocks of part `kpart` of column tile nt, requested (by the caller: a layer EARLY, behind no barrier)
__device__ __forceinline__ void ksplit_load(const WBuf &wb, int wbase, int nt, int kpart, f32x4 (&bq)[KB / NSH]) {
    const int w0 = wbase + (nt * KB + kpart * (KB / NSH)) * 256;
#pragma unroll
    for (int j = 0; j < KB / NSH; j++) bq[j] = wb.load(w0 + j * 256);
}
template <int KB, int NSEG, int NSH, typename AFrag>
__device__ __forceinline__ void gemm_ksplit(const f32x4 (&bq)[KB / NSH], int kpart, AFrag afrag, lds_float *part /* as in gemm_tiles_split */) {
    constexpr int SPW = NSEG / NSH, KBP = KB / NSH;
    static_assert(NSEG % NSH == 0 && KB % NSH == 0, "whole segments, equal parts");
    constexpr auto seg_lo = [](int c) { return (KB * c) / NSEG; };
    static_assert([&]() { for (int p = 1; p < NSH; p++) for (int c = 0; c <= SPW; c++) if (seg_lo(p * SPW + c) - seg_lo(p * SPW) != seg_lo(c) - seg_lo(0)) return false; return true; }(),
                  "every part's segments have the lengths of the first part's");
    static_assert(seg_lo(SPW) == KBP, "a part is KB / NSH k-blocks");
    const int lane = threadIdx.x & 63;
    const int kb0 = kpart * KBP;                                        // wave-uniform
    f32x4 accx[SPW];
#pragma unroll
    for (int c = 0; c < SPW; c++) accx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[2];
    a[0] = afrag(kb0);
#pragma unroll
    for (int j = 0; j < KBP; j++) {
        int seg = 0;                                                    // the segment (of this part) k-block j belongs to: static after unrolling
#pragma unroll
        for (int c = 1; c < SPW; c++) seg += j >= seg_lo(c) ? 1 : 0;
        if (j + 1 < KBP) a[(j + 1) & 1] = afrag(kb0 + j + 1);
        const f32x4 b = bq[j];
#pragma unroll
        for (int q = 0; q < 4; q++) accx[seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[q], a[j & 1][q], accx[seg], 0, 0, 0);   // D^T: see tile_out
    }
#pragma unroll
    for (int c = 0; c < SPW; c++) lds_store4(part, c * 256, accx[c]);   // D-fragment order: [segment][lane][reg]
}

// The same share-out for the SHORT 32-column layers (the blocks' first 1x1, K = 64): there a reduction pass and its two
// barriers cost more than the imbalance they remove, so the last row tile is not k-split but computed whole by the wave of
// each column tile whose share comes last (kpart == NSPLIT - 1).  No tile of these layers is split, so every output is one
// plain accumulation chain over k -- the same arithmetic for every row without the segment sums of gemm_tiles_split.
// MINE_C: 0 / 1 = the answer is a template constant (the caller holds a copy of the layer for either: straight-line code, the last
// tile's MFMAs interleaved with the others'), -1 = decided by kpart at run time (one scalar branch per k-block).
template <int NMT, int KB, int NSPLIT, int NSEG, int MINE_C, typename AFrag, typename Next, typename Epi, typename EpiX>
__device__ __forceinline__ void gemm_tiles_last(const WBuf &wb, int wbase, int nt, int mt0, int xmt, int kpart, f32x4 (&pre)[NPREMAX],
                                                AFrag afrag, Next next, Epi epi, EpiX epix) {
    constexpr int NPRE = Pre<KB>::N;
    constexpr int PB = NPRE + 1 < KB ? NPRE + 1 : KB;
    static_assert(NSEG >= 1 && NSEG <= KB, "a segment is at least one k-block");
    const bool mine = MINE_C < 0 ? kpart == NSPLIT - 1 : MINE_C != 0;   // wave-uniform
    f32x4 acc[NMT + 1][NSEG];
#pragma unroll
    for (int i = 0; i <= NMT; i++)
#pragma unroll
        for (int c = 0; c < NSEG; c++) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int w0 = wbase + nt * KB * 256;
    f32x4 bq[PB];
#pragma unroll
    for (int d = 0; d < NPRE; d++) bq[d] = pre[d];
    f32x4 a[2][NMT + 1];
#pragma unroll
    for (int i = 0; i < NMT; i++) a[0][i] = afrag(mt0 + i, 0, i);
    if (MINE_C != 0) a[0][NMT] = afrag(xmt, 0, NMT);
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        int seg = 0;                                                    // static after unrolling
#pragma unroll
        for (int c = 1; c < NSEG; c++) seg += kb >= (KB * c) / NSEG ? 1 : 0;
        if (kb + 1 < KB) {
#pragma unroll
            for (int i = 0; i < NMT; i++) a[(kb + 1) & 1][i] = afrag(mt0 + i, kb + 1, i);
            if (MINE_C != 0) a[(kb + 1) & 1][NMT] = afrag(xmt, kb + 1, NMT);
        }
        if (kb + PB - 1 < KB && kb + PB - 1 >= NPRE) bq[(kb + PB - 1) % PB] = wb.load(w0 + (kb + PB - 1) * 256);
        const f32x4 b = bq[kb % PB];
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < NMT; i++)
                acc[i][seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][i][j], acc[i][seg], 0, 0, 0);   // D^T: see tile_out
            if (MINE_C > 0) acc[NMT][seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][NMT][j], acc[NMT][seg], 0, 0, 0);
        }
        if (MINE_C < 0 && mine) {                                       // ONE scalar branch per k-block
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc[NMT][seg] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[kb & 1][NMT][j], acc[NMT][seg], 0, 0, 0);
        }
    }
    next();
#pragma unroll
    for (int i = 0; i <= NMT; i++) {
        f32x4 sum = acc[i][0];
#pragma unroll
        for (int c = 1; c < NSEG; c++) sum = sum + acc[i][c];
        if (i < NMT) epi(mt0 + i, sum, i);
        else if (mine) epix(sum);
    }
}

// ReLU as ONE instruction (v_med3_f32 v, 0, +inf): `v > 0 ? v : 0` costs a canonicalising v_max plus the v_max itself
__device__ __forceinline__ float relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff()); }

// The layer GEMMs pass the WEIGHTS as the MFMA's first operand and the activations as its second: the instruction then
// produces the transposed tile, and a lane holds, for ONE activation row (16 mt + (lane & 15)), FOUR CONSECUTIVE output
// channels (16 nt + 4 (lane >> 4) + 0..3) -- one 16-byte LDS store per tile (and one 16-byte read for the residual) where
// the untransposed fragment (a column of four rows) needs four 4-byte ones.  The packed weight order serves both ways
// round: lane (l15, q) holds W[k = 4 q + j][n = l15] for either operand slot.
//
// D fragment of the policy dense layer (untransposed): lane holds column (lane & 15) of rows 16 mt + 4 (lane >> 4) + reg
template <typename F>
__device__ __forceinline__ void for_each_out(int mt, const f32x4 &acc, F f) {
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, r0 = mt * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int reg = 0; reg < 4; reg++) f(r0 + reg, col, acc[reg]);
}

#ifdef CCSP_STAMPS                        // diagnostic build only (tools/stamps_net.py): s_memtime at the layer boundaries of workgroup 0
__device__ unsigned long long net_stamps[64];
__device__ unsigned long long net_stamps2[8][8];          // [wave][point] inside block 4's last 1x1 layer
#define NET_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) net_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define NET_STAMP2(pt) do { if (blockIdx.x == 0 && lane == 0 && blk == 4) net_stamps2[wave][pt] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NET_STAMP(i) do { } while (0)
#define NET_STAMP2(pt) do { } while (0)
#endif

// REQ (the free-running path's hand-off, ccsp_net_forward_requests): the input is the batch of REQUEST records -- the position, 64 bytes
// instead of 1372 of float32 planes: utils.to_model_input (C1) happens in the input phase below -- and the answer is compact: p_out[i][j] =
// the softmax entry of the j-th legal move of request i (`moves`), CCSP_REQUEST_MOVES doubles per row, k of them written.
template <typename C, bool REQ>
__global__ __launch_bounds__(C::NTH) void net_forward_kernel(const float *__restrict__ W, const float *__restrict__ planes, int n,
                                                             float *__restrict__ logits_out, double *__restrict__ p_out,
                                                             float *__restrict__ v_out, const uint16_t *__restrict__ moves) {
    constexpr int NB = C::NB, NTH = C::NTH, ROWS = C::ROWS, MT = C::MT, NSPLIT = C::NSPLIT, PADROWS = C::PADROWS, INROWS = C::INROWS,
                  NW = C::NW, F32 = C::F32, F32A = C::F32A, F64 = C::F64, F64B = C::F64B, M64 = C::M64, XT = C::XT, HB = C::HB;
    constexpr bool AS = C::ALLSPLIT;                              // <1, 8>: every 3x3 tile job is shared by NSH waves (see Cfg)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem<C> &S = *reinterpret_cast<Smem<C> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: tile choices and k-ranges become scalar branches
    const int q = lane >> 4, l15 = lane & 15;
    const long long s0 = (long long)blockIdx.x * NB;              // first position of this workgroup
    const int here = (int)((n - s0) < NB ? (n - s0) : NB);

    WBuf wb;
    wb.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, LAY.total * 4, 0x00020000);
    wb.voff = lane * 16;

#ifdef CCSP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) net_stamps[62] = __builtin_amdgcn_s_memtime();
#endif
    // ---- input planes -> LDS [INROWS][LDI] (channels 7.. and the cells past the 8 positions = 0), aliasing y2 ----------
    // The workgroup's positions are ONE contiguous run of here * 343 floats (16-byte aligned: NB * 343 * 4 is a multiple of 16 for
    // even NB): coalesced 16-byte loads, issued before the staging area is zeroed, then every element goes to its (cell, channel) slot
    // of the 12-float rows.  (Was: one 4-byte load and a handful of divisions per LDS element, pad channels included.)
    float *in = S.y2;                                          // (dead again before the first 3x3 layer writes y2)
    // y1, the zero-halo input of the 3x3 layers (only interior cells are ever written again), is cleared in the same phase
    static_assert((PADROWS * LDY) % 4 == 0, "y1 is cleared 16 bytes at a time");
    for (int i = tid * 4; i < PADROWS * LDY; i += NTH * 4) *reinterpret_cast<f32x4 *>(&S.y1[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NB < 8 || CCSP_NET_LDS_BIAS_ALL)            // (see bias4 below)
    for (int i = tid; i < BIAS_N / 4; i += NTH)               // the trunk's biases -> LDS: 308 16-byte pieces (visible behind the input phase's barrier)
        *reinterpret_cast<f32x4 *>(&S.bias[4 * i]) = *reinterpret_cast<const f32x4 *>(W + LAY.trunk_b + 4 * i);
    const ccsp_request *rq = reinterpret_cast<const ccsp_request *>(planes) + s0;      // (REQ)
    if constexpr (REQ) {
        // C1 in place: the staging area zeroed, then one thread per (position, checker) writes the checker's id + 1 at the cells it
        // stands on now / one / two plies ago (ccsp_scatter_checker: the un-swapping of utils.py:135-155), one thread per (position, cell)
        // the player-two flag (utils.py:157-158).  A row that asks for nothing stays zero.
        static_assert((INROWS * LDI) % 4 == 0, "the staging area is cleared 16 bytes at a time");
        for (int i = tid * 4; i < INROWS * LDI; i += NTH * 4) *reinterpret_cast<f32x4 *>(&in[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        struct PlaneWriter {                                   // img[cell * 7 + ch] = val -> in[(s * 49 + cell) * LDI + ch]
            float *base;
            struct Ref { float *p; __device__ __forceinline__ void operator=(uint8_t val) const { *p = (float)val; } };
            __device__ __forceinline__ Ref operator[](int i) const { const int cell = i / 7; return Ref{base + cell * LDI + (i - 7 * cell)}; }
        };
        for (int t = tid; t < here * 12; t += NTH) {
            const int s = t / 12, k = t - 12 * s;
            if (rq[s].kind == 0) continue;
            const ccsp_sr st = ccsp_load_sr(&rq[s].state);
            ccsp_scatter_checker(st, (int)rq[s].player, k, PlaneWriter{in + s * 49 * LDI});
        }
        for (int t = tid; t < here * 49; t += NTH) {
            const int s = t / 49, cell = t - 49 * s;
            if (rq[s].kind != 0 && rq[s].player == 2) in[(s * 49 + cell) * LDI + 6] = 1.0f;
        }
    } else if constexpr ((NB * 343) % 4 != 0) {                // (<2, 8>: 686 floats per workgroup, not a multiple of four: the plain way)
        for (int i = tid; i < INROWS * LDI; i += NTH) {
            const int cellg = i / LDI, ch = i % LDI, s = cellg / 49, cell = cellg % 49;
            in[i] = (ch < 7 && s < here) ? planes[(s0 + s) * 343 + cell * 7 + ch] : 0.0f;
        }
    } else {
        constexpr int NV = (NB * 343) / 4, PER = (NV + NTH - 1) / NTH;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(planes + s0 * 343);
        const int nreal = here * 343;
        f32x4 v[PER];
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int e = (tid + r * NTH) * 4;
            v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e + 3 < nreal) v[r] = src[tid + r * NTH];
            else if (e < nreal) { for (int j = 0; j < 4; j++) if (e + j < nreal) v[r][j] = planes[s0 * 343 + e + j]; }
        }
        static_assert((INROWS * LDI) % 4 == 0, "the staging area is cleared 16 bytes at a time");
        for (int i = tid * 4; i < INROWS * LDI; i += NTH * 4) *reinterpret_cast<f32x4 *>(&in[i]) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PER; r++) {
            const int e = (tid + r * NTH) * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned ee = (unsigned)(e + j);
                if ((int)ee < nreal) {
                    const unsigned cellg = __umulhi(ee, 613566757u);       // ee / 7 (exact for ee < 2^31: 2^32 / 7 rounded up)
                    in[cellg * LDI + (ee - 7u * cellg)] = v[r][j];
                }
            }
        }
    }
#ifdef CCSP_EXP_NET_INPUT_DELAY     // experiment (round 6): an input phase longer by ~N x 30 ns (what generating the move lists here would add)
    __builtin_amdgcn_s_sleep(CCSP_EXP_NET_INPUT_DELAY);
#endif
    __syncthreads();
    NET_STAMP(0);
#ifdef CCSP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) net_stamps[60] = __builtin_amdgcn_s_memrealtime();
#endif

    const int nt2 = wave & 1, qr = wave >> 1, mt3 = AS ? qr % MT : F32 * qr;    // this wave's share of the 32-column layers (see below; AS: its ONE row tile)
    // its share of the last tile's k-range (none in a shape without one).  Cell-major: the segment most of whose k-blocks are dead for the
    // wave's edge tiles (XSEG) -- the shared tile's MFMAs fall where the wave has the fewest others.
    const int kshare = XT ? (C::CELLMAJOR ? (qr == 0 ? C::xseg_of(0) : (qr == 1 ? C::xseg_of(1) : (qr == 2 ? C::xseg_of(2) : C::xseg_of(3)))) : qr) : -1;
    const int kh = AS ? qr / MT : 0;                             // AS: its part of the tile's k-range in the 3x3 layers; part 0 computes the tile in the first 1x1
    f32x4 pre[NPREMAX];                                          // the next layer's first weight k-blocks, in flight across barriers
    // row -> (position s of the workgroup, cell pos = 5 r + c); a padding row gives some valid pair (its results are never read)
    constexpr bool CM = C::CELLMAJOR;
    auto row_sp = [&](int row, int &s, int &pos) {
        if constexpr (CM) { const int t = row >> 4, l = row & 15; pos = cell_of<NB>(t < MT - 1 ? t * (16 / NB) + l / NB : 24); s = l % NB; }
        else { if (row >= ROWS) row -= 25; s = row / 25; pos = row % 25; }
    };
    auto y1_at = [&](int s, int pos, bool top_left) -> int {      // element offset in y1 of the cell (top_left: of its tap (-1, -1))
        if constexpr (CM) return ((6 * (pos / 5) + pos % 5 + CM_PAD - (top_left ? 7 : 0)) * NB + s) * LDY + 4 * q;
        else return ((top_left ? 0 : PAD0) + s * PADPOS + (pos / 5) * 6 + (pos % 5)) * LDY + 4 * q;
    };
    constexpr int TAPROW = (CM ? NB : 1) * LDY;                    // one cell to the right in y1 (one map row down: 6 of them)
    // this lane's four output channels (idx: BIAS_*).  Through LDS in the shapes whose launch is one workgroup's LATENCY (A/B on one box,
    // round 5: <4,4> 75.0 -> 73.5 us, <2,8> 46.5 -> 44.7, <1,8> 36.4 -> 34.7); from global memory in <8,8>, where the epilogues that no
    // longer wait run INTO the partner wave's MFMA stream and the launch of 2048 positions got 1 % longer (118.4 -> 119.5 us)
    constexpr bool LDS_BIAS = NB < 8 || CCSP_NET_LDS_BIAS_ALL;
    auto bias4 = [&](int idx) -> f32x4 {
        if constexpr (LDS_BIAS) return *reinterpret_cast<const f32x4 *>(&S.bias[idx + 4 * q]);
        else return *reinterpret_cast<const f32x4 *>(W + LAY.trunk_b + idx + 4 * q);
    };
    auto relu4 = [](const f32x4 &v) -> f32x4 { return f32x4{relu(v[0]), relu(v[1]), relu(v[2]), relu(v[3])}; };

    // The trunk's tiles have the same owner in the stem and in every block's last layer (column tile wave & 3, row tiles 6 (wave >> 2)
    // .. + 6): the owner keeps its seven tiles in registers as well, so that the residual add needs no LDS read.
    f32x4 xr[F64];
    lds_float *const xown = lds_opaque(&S.x[(((wave >> 2) * M64) * 16 + l15) * LDX + (wave & 3) * 16 + 4 * q]);   // this lane's piece of its first tile
    // ---- stem: 3x3 valid, K = 9 taps x 8 -> 5 k-blocks of 2 taps (10th tap = zero weights) -----------
    // A row's nine taps are plain offsets from its top-left input cell (valid convolution); rows past the 200th and the
    // zero-weight 10th tap read staged zeros.  No masks in the loop.
    {
        const int nt = wave & 3, mt0 = (wave >> 2) * M64;      // 4 column tiles x row halves: <8,8> tiles 0-6 and 7-12
        int sbase[F64];
#pragma unroll
        for (int i = 0; i < F64; i++) {
            const int row = (mt0 + i) * 16 + l15;
            int s = row / 25, pos = row % 25;                  // (rows past the last: staged zeros)
            if constexpr (CM) row_sp(row, s, pos);
            sbase[i] = (s * 49 + (pos / 5) * 7 + (pos % 5)) * LDI + (q & 1) * 4;
        }
        int tapoff[5];                                         // lane groups q = 0,1 carry tap 2 kb, q = 2,3 tap 2 kb + 1
#pragma unroll
        for (int kb = 0; kb < 5; kb++) {
            const int tap = kb * 2 + (q >> 1);
            tapoff[kb] = ((tap / 3) * 7 + tap % 3) * LDI;
        }
        auto afrag = [&](int, int kb, int i) -> f32x4 {
            return *reinterpret_cast<const f32x4 *>(&in[sbase[i] + tapoff[kb]]);
        };
        const f32x4 bv = bias4(BIAS_STEM + nt * 16);
        auto epi = [&](int, const f32x4 &acc, int i) {
            xr[i] = relu4(acc + bv);
            lds_store4(xown, i * 16 * LDX, xr[i]);
        };
        prefetch<5>(wb, LAY.stem_w, nt, pre);
        auto next = [&]() { prefetch<4>(wb, LAY.l1_w[0], nt2, pre); };
        if (F64B == F64 || (wave >> 2) == 0) gemm_tiles<F64, 5, CCSP_NET_SEG_STEM>(wb, LAY.stem_w, nt, mt0, pre, afrag, next, epi);
        else gemm_tiles<F64B, 5, CCSP_NET_SEG_STEM>(wb, LAY.stem_w, nt, mt0, pre, afrag, next, epi);
    }
    __syncthreads();
    NET_STAMP(1);

    // 32-column layers: a wave owns row tiles 3 qr .. 3 qr + 2 of column tile nt, and a quarter of tile 12's k-range.
    // Per slot (0-2: the full tiles; 3: tile 12), once for all nine blocks:
    //   y1p[i]    in y1, the row's TOP-LEFT tap (zero-halo copy): tap (dr, dc) is + (dr*6 + dc) * TAPROW
    //   (the interior cell of this lane's row -- 1x1 epilogue -> 3x3 input -- is y1p[i] + CELL0: no second address register per tile)
    const lds_float *y1p[F32A + 1];                            // (every per-lane LDS base of the block loop: lds_opaque -- see there)
    bool prow_ok[F32A];                                        // (PADFULL shapes: is this lane's row of the tile a real cell?)
#pragma unroll
    for (int i = 0; i < F32A + 1; i++) {
        int s, pos;                                            // padding rows of the last tile: any valid cell (results never read)
        row_sp((i < F32A ? mt3 + i : MT - 1) * 16 + l15, s, pos);
        y1p[i] = lds_opaque(&S.y1[y1_at(s, pos, true)]);       // = cell (r - 1, c - 1): PAD0 - 7 = 0
        if (i < F32A) prow_ok[i] = (mt3 + i) * 16 + l15 < ROWS;   // <8,8>, <4,4>: always a real cell
    }
    constexpr int CELL0 = 7 * TAPROW;                          // the cell itself from its top-left tap: one map row down, one cell to the right
    const lds_float *const xin = lds_opaque(&S.x[(mt3 * 16 + l15) * LDX + 4 * q]);          // the first 1x1 layer's input rows: this wave's tiles ...
    const lds_float *const xinx = lds_opaque(&S.x[((MT - 1) * 16 + l15) * LDX + 4 * q]);    // ... and the last tile
    lds_float *const y2w = lds_opaque(&S.y2[(mt3 * 16 + l15) * LDY + nt2 * 16 + 4 * q]);                 // the 3x3 layer's output: this wave's tiles
    const lds_float *const y2r = lds_opaque(&S.y2[(((wave >> 2) * M64) * 16 + l15) * LDY + 4 * q]);      // ... as the last 1x1 layer reads it
    lds_float *const partw = lds_opaque(&S.part[AS ? mt3 * 2 + nt2 : nt2][(AS ? kh : (XT ? kshare : 0)) * (C::NSEG / C::NSH)][lane * 4]);
    const lds_float *const partr = lds_opaque(&S.part[AS ? ((wave >> 2) * M64) * 2 : 0][0][lane * 4]);

    // ---- nine bottleneck residual blocks (model.py:120-145) ------------------------------------------
    f32x4 bq2[AS ? 18 / C::NSH : 1];                             // (AS: this wave's weight k-blocks of the block's 3x3 layer)
    for (int blk = 0; blk < 9; blk++) {
        const int wo = blk * BLK_STRIDE, bb = BIAS_BLK + blk * BIAS_PER_BLK;     // this block's weights in the blob / biases in LDS
        int n2o = nt2 * 16;                                                      // (opaque per block: the sum y1p[i] + n2o is formed where it is used --
        asm volatile("" : "+s"(n2o));                                            //  one v_add per tile and layer -- not kept in a register per tile)
        {   // 1x1 64 -> 32: 2 column tiles x 4 row groups of three tiles + a quarter of tile 12's k-range each
            auto afrag = [&](int, int kb, int i) -> f32x4 {                     // (slot i: row tile mt3 + i; slot F32A: the last tile)
                return i < F32A ? lds_load4(xin, i * 16 * LDX + kb * 16) : lds_load4(xinx, kb * 16);
            };
            const f32x4 bv = bias4(bb + BIAS_L1 + nt2 * 16);
            auto epi = [&](int, const f32x4 &acc, int i) {
                const lds_float *at = i == 0 ? y1p[0] : (i == 1 ? y1p[F32A > 1 ? 1 : 0] : y1p[F32A > 2 ? 2 : 0]);
                if constexpr (C::PADFULL) {
                    if (!(i == 0 ? prow_ok[0] : (i == 1 ? prow_ok[F32A > 1 ? 1 : 0] : prow_ok[F32A > 2 ? 2 : 0]))) return;
                }
                lds_store4(const_cast<lds_float *>(at), CELL0 + n2o, relu4(acc + bv));
            };
            auto epix = [&](const f32x4 &acc) {                                 // the last tile: its real rows only
                if (l15 < ROWS - (MT - 1) * 16) lds_store4(const_cast<lds_float *>(y1p[F32A]), CELL0 + n2o, relu4(acc + bv));
            };
            if constexpr (AS) ksplit_load<18, C::NSH>(wb, (LAY.l2_w[0] + wo), nt2, kh, bq2);      // the 3x3 layer's weights: in flight across this layer
            if constexpr (AS) {
                // one tile job per (row tile, column tile): the wave with part 0 of the tile's k-range in the 3x3 layer computes it whole
                // here (a plain chain over k like every other shape's); the 3x3 layer requests its own weights
                auto epi1 = [&](int, const f32x4 &acc, int) {
                    if (prow_ok[0]) lds_store4(const_cast<lds_float *>(y1p[0]), CELL0 + n2o, relu4(acc + bv));
                };
                if (kh == 0) gemm_tiles<1, 4, CCSP_NET_SEG_L1>(wb, (LAY.l1_w[0] + wo), nt2, mt3, pre, afrag, []() {}, epi1);
            } else
            {
                auto next = [&]() { prefetch<18>(wb, (LAY.l2_w[0] + wo), nt2, pre); };
                if constexpr (CM && NB < 8) {                    // a copy of the layer with the last tile, one without (gemm_tiles_last).  A/B on one
                                                                 // box, round 5: <4,4> 68.3 -> 67.7 us per 1024 positions; <8,8> (two waves per SIMD) 103.8 -> 105.3: not there
                    if (kshare == NSPLIT - 1) gemm_tiles_last<F32A, 4, NSPLIT, CCSP_NET_SEG_L1, 1>(wb, (LAY.l1_w[0] + wo), nt2, mt3, MT - 1, kshare, pre, afrag, next, epi, epix);
                    else gemm_tiles_last<F32A, 4, NSPLIT, CCSP_NET_SEG_L1, 0>(wb, (LAY.l1_w[0] + wo), nt2, mt3, MT - 1, kshare, pre, afrag, next, epi, epix);
                } else
                    gemm_tiles_last<F32A, 4, NSPLIT, CCSP_NET_SEG_L1, -1>(wb, (LAY.l1_w[0] + wo), nt2, mt3, MT - 1, kshare, pre, afrag, next, epi, epix);
            }
        }
        __syncthreads();
        NET_STAMP(2 + 3 * blk);
        {   // 3x3 same 32 -> 32: k-block kb = tap (kb >> 1), channels 16 (kb & 1) ..; the halo supplies the zeros
            auto afrag = [&](int, int kb, int i) -> f32x4 {
                const int tap = kb >> 1;                                        // compile-time after unrolling
                return lds_load4(y1p[i], ((tap / 3) * 6 + tap % 3) * TAPROW + (kb & 1) * 16);
            };
            const f32x4 bv = bias4(bb + BIAS_L2 + nt2 * 16);
            auto epi = [&](int, const f32x4 &acc, int i) { lds_store4(y2w, i * 16 * LDY, relu4(acc + bv)); };
            if constexpr (AS) {
                auto afrag_rt = [&](int kb) -> f32x4 {                           // (kb: wave-uniform, known at run time)
                    const int tap = kb >> 1;
                    return lds_load4(y1p[0], ((tap / 3) * 6 + tap % 3) * TAPROW + (kb & 1) * 16);
                };
                gemm_ksplit<18, C::NSEG, C::NSH>(bq2, kh, afrag_rt, partw);
                prefetch<2>(wb, (LAY.l3_w[0] + wo), wave & 3, pre);
            } else
            {
                auto next = [&]() { prefetch<2>(wb, (LAY.l3_w[0] + wo), wave & 3, pre); };
                if constexpr (CM) {                               // one straight-line copy of the layer per row group (= map edge)
                    auto run = [&](auto G) {
                        constexpr int g = decltype(G)::value;
                        gemm_tiles_split<F32A, 18, C::NSEG, NSPLIT, C::NE, true, C::dead_of(g, 0), C::dead_of(g, 1), C::xseg_of(g)>(wb, (LAY.l2_w[0] + wo), nt2, mt3, MT - 1, 0, pre, afrag, next, epi, partw);
                    };
                    if (qr == 0) run(std::integral_constant<int, 0>{});
                    else if (NSPLIT == 2 || qr == 1) run(std::integral_constant<int, 1>{});
                    else if (qr == 2) run(std::integral_constant<int, NSPLIT == 2 ? 1 : 2>{});
                    else run(std::integral_constant<int, NSPLIT == 2 ? 1 : 3>{});
                } else
                    gemm_tiles_split<F32A, 18, C::NSEG, NSPLIT, 0, false, 0u, 0u, 0>(wb, (LAY.l2_w[0] + wo), nt2, mt3, MT - 1, kshare, pre, afrag, next, epi, partw);
            }
            NET_STAMP(32 + blk);                             // diagnostic: wave 0 done with its share of the 3x3 layer
#ifdef CCSP_STAMPS
            if (blockIdx.x == 0 && lane == 0 && blk == 4) {
                net_stamps[44 + wave] = __builtin_amdgcn_s_memtime();
                net_stamps[52 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID: wave slot, SIMD, CU...
            }
#endif
        }
        __syncthreads();
        NET_STAMP(3 + 3 * blk);
        {   // 1x1 32 -> 64 + residual: 4 column tiles x 2 row halves
            const int nt = wave & 3, half = wave >> 2, mt0 = half * M64;     // <8,8>: tiles 0-6 and 7-12
            // The 3x3 layer's k-split tile (MT - 1) is consumed straight from its partial sums -- no reduction pass, no two extra
            // barriers: part[nt2][c][lane * 4 + j] is, for THIS lane's (row, k-slot), exactly what an activation fragment of
            // k-block nt2 holds, so the waves whose share ends with that tile form ((c0 + c1) + c2) + c3 + bias, ReLU in registers.
            NET_STAMP2(0);
            const bool has_x = AS || (XT && half == C::NH - 1);             // wave-uniform (AS: EVERY tile of the 3x3 layer arrives as partial sums)
            constexpr int XSLOT = F64B - 1;                                 // the tile slot of the last tile in the wave that has it
            const int xt = AS ? mt0 : 0;                                    // which of the Smem::part tiles
            f32x4 ax[2];
            if (has_x) {
#pragma unroll
                for (int kb = 0; kb < 2; kb++) {
                    f32x4 v = lds_load4(partr, kb * C::NSEG * 256);
#pragma unroll
                    for (int c = 1; c < C::NSEG; c++) v = v + lds_load4(partr, (kb * C::NSEG + c) * 256);
                    ax[kb] = relu4(v + bias4(bb + BIAS_L2 + kb * 16));
                }
            }
            auto afrag = [&](int mt, int kb, int i) -> f32x4 {
                if (i == XSLOT && has_x) return ax[kb];
                return lds_load4(y2r, i * 16 * LDY + kb * 16);
            };
            const f32x4 bv = bias4(bb + BIAS_L3 + nt * 16);
            auto epi = [&](int mt, const f32x4 &acc, int i) {
                xr[i] = relu4(acc + bv + xr[i]);                                // add([x, block_input]) then ReLU; the input from registers
                lds_store4(xown, i * 16 * LDX, xr[i]);
            };
            NET_STAMP2(1);
            auto next = [&]() {
                NET_STAMP2(2);
                // (no branch: the same three loads either way -- the next block's first 1x1 or, behind the last block, the policy conv)
                prefetch<4>(wb, blk < 8 ? LAY.l1_w[0] + wo + BLK_STRIDE : LAY.pc_w, blk < 8 ? nt2 : 0, pre);
            };
            if (F64B == F64 || half == 0) gemm_tiles<F64, 2, CCSP_NET_SEG_L3>(wb, (LAY.l3_w[0] + wo), nt, mt0, pre, afrag, next, epi);
            else gemm_tiles<F64B, 2, CCSP_NET_SEG_L3>(wb, (LAY.l3_w[0] + wo), nt, mt0, pre, afrag, next, epi);
            NET_STAMP2(3);
        }
        __syncthreads();
        NET_STAMP2(4);
        NET_STAMP(4 + 3 * blk);
    }

    // ---- policy head: 1x1 64 -> 16 (+ReLU) into pc[row][16] (contiguous = [position][400]), aliasing y1 ----
    float *pc = S.y1;                                    // [208][16] floats; only rows < 200 are read
    {
        const int mt0 = 2 * wave < MT - 2 ? 2 * wave : MT - 2;   // two tiles per wave; the last waves share the last two (plain stores)
        const lds_float *const xpc = lds_opaque(&S.x[(mt0 * 16 + l15) * LDX + 4 * q]);
        auto afrag = [&](int, int kb, int i) -> f32x4 { return lds_load4(xpc, i * 16 * LDX + kb * 16); };
        const f32x4 bv = bias4(BIAS_PC);
        int pcat[2];                                         // [position][cell][16]; padding rows go behind the 200th
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (mt0 + i) * 16 + l15;
            int s, pos;
            row_sp(row, s, pos);
            pcat[i] = (CM ? (row < ROWS ? s * 25 + pos : row) : row) * 16 + 4 * q;
        }
        auto epi = [&](int, const f32x4 &acc, int i) {
            *reinterpret_cast<f32x4 *>(&pc[pcat[i]]) = relu4(acc + bv);
        };
        gemm_tiles<2, 4, CCSP_NET_SEG_PC>(wb, LAY.pc_w, 0, mt0, pre, afrag, []() {}, epi);
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
        int s, pos;
        row_sp(tid, s, pos);
        vc[CM ? s * 25 + pos : tid] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    NET_STAMP(29);

    // ---- policy dense 400 -> 294 on v_mfma_f32_4x4x1_16B_f32: sixteen independent 4 x 4 outer products per instruction (64 FLOP per
    // cycle like the other fp32 MFMAs).  A 16 x 16 tile would hold the workgroup's 8 (or 4) positions in 16 rows -- half (three
    // quarters) of every instruction wasted.  Here block b of an instruction is (4 positions) x (output columns 4b .. 4b+3) for ONE k:
    // lane l supplies the activation of position l & 3 and the weight of column 64 T + l (tools/probe/mfma_4x4_probe.hip: D[reg r] of
    // lane l = A[lane 4 (l >> 2) + r] * B[lane l]) and ends up with its column's logits of four positions; eight positions = two
    // instructions on the same weight register.  No row of any tile is empty.
    // Work split: wave = (k-quarter kq, tile half th): 25 groups of four k for the column tiles of its half (3 + 2 of the five 64-wide
    // tiles; both waves of a SIMD together 5), every weight crosses L2 -> registers exactly once per workgroup (16 bytes per lane and
    // group, scalar-addressed, two groups ahead).  The four k-quarters' partial sums meet in LDS (the trunk buffer is dead by now) and
    // every logit is ((q0 + q1) + q2) + q3 + bias -- the same chains for every position, whatever its slot or the workgroup shape.
    float *part = S.x;                                   // [4][HB][320] partial sums
    {
        constexpr int GPQ = 25, NHALF = NW / 4, TPW = (5 + NHALF - 1) / NHALF, PG = HB / 4, PDG = C::PDG;
        const int kq = wave & 3, th = wave >> 2;
        int woff[TPW];
        bool valid[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int T = th * TPW + t;
            valid[t] = T < 5;
            woff[t] = LAY.pf_w + ((valid[t] ? T : 0) * 100 + kq * GPQ) * 256;
        }
        f32x4 acc[TPW][PG], bq[PDG + 1][TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int g = 0; g < PG; g++) acc[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < PDG; d++)
#pragma unroll
            for (int t = 0; t < TPW; t++) bq[d][t] = wb.load(woff[t] + d * 256);
        const float *arow = &pc[(lane & 3) * 400 + kq * GPQ * 4];
        f32x4 a[2][PG];
#pragma unroll
        for (int g = 0; g < PG; g++) a[0][g] = *reinterpret_cast<const f32x4 *>(arow + g * 1600);
#pragma unroll
        for (int kg = 0; kg < GPQ; kg++) {
            if (kg + 1 < GPQ) {
#pragma unroll
                for (int g = 0; g < PG; g++) a[(kg + 1) & 1][g] = *reinterpret_cast<const f32x4 *>(arow + g * 1600 + (kg + 1) * 4);
            }
            if (kg + PDG < GPQ) {
#pragma unroll
                for (int t = 0; t < TPW; t++) bq[(kg + PDG) % (PDG + 1)][t] = wb.load(woff[t] + (kg + PDG) * 256);
            }
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                if (t + 1 < TPW || valid[t]) {                          // only a wave's LAST tile can be empty
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int g = 0; g < PG; g++)
                            acc[t][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[kg & 1][g][j], bq[kg % (PDG + 1)][t][j], acc[t][g], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            if (valid[t]) {
                const int T = th * TPW + t;
#pragma unroll
                for (int g = 0; g < PG; g++)
#pragma unroll
                    for (int r = 0; r < 4; r++) part[(kq * HB + 4 * g + r) * 320 + T * 64 + lane] = acc[t][g][r];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < NB * NPOL; i += NTH) {
        const int s = i / NPOL, col = i - s * NPOL;
        const float q0 = part[(0 * HB + s) * 320 + col], q1 = part[(1 * HB + s) * 320 + col], q2 = part[(2 * HB + s) * 320 + col],
                    q3 = part[(3 * HB + s) * 320 + col];
        lg[s * NPOL_PAD + col] = ((q0 + q1) + q2) + q3 + W[LAY.pf_b + col];
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
        if (!REQ || rq[tid].kind != 0) v_out[s0 + tid] = tanhf(acc);
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
        if constexpr (REQ) {
            // the compact answer: the float64 softmax entries of the request's k legal moves, in the order of its move row.  Every lane
            // forms its five entries as before (the same e / sum), leaves them in LDS (the trunk buffer is dead; 296 doubles per wave) and
            // lane j picks entry moves[j]: DS operations of one wave complete in order, no barrier needed.
            double *pd = reinterpret_cast<double *>(S.x) + wave * 296;
            if (rq[s].kind != 0) {
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int i = lane + 64 * j;
                    if (i < NPOL) pd[i] = e[j] / sum;
                }
                __builtin_amdgcn_wave_barrier();
                // (k and the move entries come from caller-owned buffers: k is clamped to the row, an entry that is no action index -- a row
                // the engine has not written yet -- answers 0.0 instead of reading another wave's area)
                const int k = (int)rq[s].k < CCSP_MAX_MOVES ? (int)rq[s].k : CCSP_MAX_MOVES;
                const uint16_t *mrow = moves + (s0 + s) * CCSP_REQUEST_MOVES;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int j = lane + 64 * h;
                    if (j < k) { const int a = mrow[j] & 0x1FF; p_out[(s0 + s) * CCSP_REQUEST_MOVES + j] = a < NPOL ? pd[a] : 0.0; }
                }
                __builtin_amdgcn_wave_barrier();                // (the next position of this wave overwrites pd)
            }
        } else {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = lane + 64 * j;
            if (i < NPOL) {
                if (p_out) p_out[(s0 + s) * NPOL + i] = e[j] / sum;
                if (logits_out) logits_out[(s0 + s) * NPOL + i] = row[i];
            }
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

template <typename C, bool REQ = false>
static int launch_net(const float *packed, const float *planes, int n, float *logits, double *p, float *v, void *stream, bool *attr_set,
                      const uint16_t *moves = nullptr) {
    // the dynamic-LDS opt-in is a per-device property of the kernel: once per device ordinal, not per process
    int dev = 0;
    CCSP_HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        CCSP_HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(net_forward_kernel<C, REQ>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem<C>)));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int grid = (n + C::NB - 1) / C::NB;
    hipLaunchKernelGGL((net_forward_kernel<C, REQ>), dim3(grid), dim3(C::NTH), sizeof(Smem<C>), (hipStream_t)stream, packed, planes, n, logits, p, v, moves);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

static int g_net_shape = 0;                   // positions per workgroup: 8 (one workgroup per CU), 4 (two per CU), 0 = by batch size
#ifndef CCSP_NET_SMALL
#define CCSP_NET_SMALL 1024                   // batches up to this many positions run in the <4, 4> shape (see ccsp_net_forward)
#define CCSP_NET_TINY 512                     // ... and up to this many in the <2, 8> shape
#endif
#ifndef CCSP_NET_ONE
#define CCSP_NET_ONE 256                      // ... and up to this many (one workgroup per CU) in the <1, 8> shape
#endif
static int pick_shape(int n) { return n <= CCSP_NET_ONE ? 1 : (n <= CCSP_NET_TINY ? 2 : (n <= CCSP_NET_SMALL ? 4 : 8)); }

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
    for (int T = 0; T < 5; T++)                                            // policy dense: row k = (h*5+w)*16 + c = pc's own order
        for (int g = 0; g < 100; g++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 4; j++) {
                    const int k = 4 * g + j, col = 64 * T + lane;
                    packed[LAY.pf_w + ((T * 100 + g) * 64 + lane) * 4 + j] = col < 294 ? p[(size_t)k * 294 + col] : 0.0f;
                }
    p += 117600;
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
// Test / measurement hook: pick the workgroup shape of ccsp_net_forward (8 or 4 positions per workgroup; anything else
// restores the default).  Both shapes compute every position with the same arithmetic in the same order: results are identical.
int ccsp_debug_net_shape(int positions_per_workgroup) {
    g_net_shape = (positions_per_workgroup == 1 || positions_per_workgroup == 2 || positions_per_workgroup == 4 || positions_per_workgroup == 8) ? positions_per_workgroup : 0;
    return g_net_shape;
}

int ccsp_net_forward(const float *packed, const float *planes, int n, float *logits, double *p, float *v, void *stream) {
    if (n < 0 || (n > 0 && (!packed || !planes || !v))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    static bool attr8[64] = {false}, attr4[64] = {false}, attr2[64] = {false}, attr1[64] = {false};
    // All shapes compute a position with the same arithmetic in the same order (bit-identical results), so the choice is one of speed
    // only.  While a batch does not fill the 256 CUs its launch is as long as ONE workgroup (tools/bench_net_sizes.py): 116-119 us in
    // <8, 8> for 1 .. 2048 positions, 75-77 us in <4, 4> up to 1024 (4 positions on 4 waves), 46-48 us in <2, 8> up to 512 (2 positions on 8
    // waves: one tile job per wave in the 32-column layers) -- the arena's 24 games, one game of selfplay(), config 5's small cohorts.
    // Large batches want <8, 8>: the policy dense layer's weights cross L2 once per 8 positions.
    const int shape = g_net_shape ? g_net_shape : pick_shape(n);
    if (shape == 1) return launch_net<Cfg<1, 8>>(packed, planes, n, logits, p, v, stream, attr1);
    if (shape == 2) return launch_net<Cfg<2, 8>>(packed, planes, n, logits, p, v, stream, attr2);
    if (shape == 4) return launch_net<Cfg<4, 4>>(packed, planes, n, logits, p, v, stream, attr4);
    return launch_net<Cfg<8, 8>>(packed, planes, n, logits, p, v, stream, attr8);
}

int ccsp_net_forward_requests(const float *packed, const ccsp_request *req, const uint16_t *moves, int n, double *pk, float *v, void *stream) {
    if (n < 0 || (n > 0 && (!packed || !req || !moves || !pk || !v))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    static bool attr8[64] = {false}, attr4[64] = {false}, attr2[64] = {false}, attr1[64] = {false};
    const float *in = reinterpret_cast<const float *>(req);
    const int shape = g_net_shape ? g_net_shape : pick_shape(n);
    if (shape == 1) return launch_net<Cfg<1, 8>, true>(packed, in, n, nullptr, pk, v, stream, attr1, moves);
    if (shape == 2) return launch_net<Cfg<2, 8>, true>(packed, in, n, nullptr, pk, v, stream, attr2, moves);
    if (shape == 4) return launch_net<Cfg<4, 4>, true>(packed, in, n, nullptr, pk, v, stream, attr4, moves);
    return launch_net<Cfg<8, 8>, true>(packed, in, n, nullptr, pk, v, stream, attr8, moves);
}

#ifdef CCSP_STAMPS
int ccsp_debug_net_stamps(unsigned long long *out) {
    const bool more = out[63] == 0x5eedULL;            // (the caller passes 128 words and says so)
    CCSP_HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(net_stamps), 64 * sizeof(unsigned long long)));
    if (more) CCSP_HIPCHK(hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(net_stamps2), 64 * sizeof(unsigned long long)));
    return CCSP_OK;
}
#endif

}  // extern "C"
