// ccsp_engine.hip -- GPU-resident batched self-play: one 64-lane wavefront owns one game.
//
// Replaces (per game slot, for thousands of slots at once) the reference's
//   selfplay.selfplay        selfplay.py:11-80    ply loop + end-of-ply rules
//   selfplay.make_random_move selfplay.py:83-104  opening plies
//   selfplay.make_move       selfplay.py:107-133  root expansion + Dirichlet noise + search
//   MCTS.moveToLeaf          MCTS.py:49-76        -> wave_select()
//   MCTS.expandAndBackUp     MCTS.py:79-118       -> wave_expand() + wave_backup()
//   MCTS.search              MCTS.py:121-153      -> wave_finish_ply()
//
// Data layout in HBM.  Per slot a byte pool holds the search tree of the current
// ply as node blocks, each block = one expanded node:
//     +0   ccsp_state state (32 B)          the node's position
//     +32  u32 K, u32 player                edges / player to move
//     +40  u32 shadow, f32 v                tree reuse (ccsp_advance): offset/8 of the SAME position's block in the previous ply's
//                                           tree (0 = none); the evaluator's value of this position
//     +48  f64 P[K] | f64 W[K] | u32 N[K] | u32 child[K] | u16 mv[K]
// child[j] = 0 (leaf not expanded yet) | 0xFFFFFFFF (the move wins: terminal leaf) |
//            (block offset/8) << 7 | K_child.  A level of selection is therefore ONE round of
// coalesced loads: lane j reads edge j (and j+64), and the chosen lane's child word already holds
// the next block's address and width.  Sum(N) of a node is not reduced: it equals the visit count of
// the edge that leads to it minus one (root: the simulation index).
//
// Bit-exactness: f64 arithmetic in the reference's operation order, compiled with
// -ffp-contract=off; sqrt(N_sum) and N**(1/tau) come from host-built tables (libm) indexed by
// integers; ties are resolved by the closed form of the running-max rule (H2).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ccsp_common.h"

namespace {

constexpr uint32_t CHILD_LEAF = 0u;
constexpr uint32_t CHILD_TERMINAL = 0xFFFFFFFFu;
constexpr int BLOCK_HDR = 48;
constexpr int MAX_BLOCK_BYTES = (BLOCK_HDR + 26 * CCSP_MAX_MOVES + 7) & ~7;   // 3328
constexpr int CCSP_EVAL_CACHED = 5;     // internal (ccsp_advance): priors and value copied from the previous ply's tree

// A game slot: 16 x u64 in memory (SlotMem), plain scalars in registers (Slot).  Words:
//  0-3 root position | 4 game id | 5 draw-stream prefix | 6 result row | 7 expansions
//  8 ply, n_hist | 9 useless, pool_used | 10 root_k, sim
//  11 bytes: player, status, det_tau, n_hm, progress[0], progress[1], player_turn, opening_left
//  12-13 destinations of Board.hist_moves (board.py:246-248), oldest first, one byte each
//  14 "searching" flag of the fused path (begin -> sims -> end) | 15 spare
struct SlotMem { uint64_t w[16]; };
static_assert(sizeof(SlotMem) == 128, "SlotMem must stay 128 bytes");

struct Slot {
    ccsp_sr st;                   // root position of the ply being decided
    uint64_t game;                // global game id
    uint64_t hgame;               // draw-stream prefix of (seed, game)
    uint64_t index;               // row of the result table
    uint64_t expansions;
    uint32_t ply;                 // plies played (opening plies included)
    uint32_t n_hist;              // MCTS plies logged = len(play_history)
    int32_t useless;              // num_useless_moves (selfplay.py:50-55)
    uint32_t pool_used;           // bytes of the pool in use
    uint32_t root_k;              // edges of the root block
    uint32_t sim;                 // simulations done in this ply (stepped path)
    uint32_t player;              // to move
    uint32_t status;
    uint32_t det_tau;
    uint32_t n_hm;                // entries in hm
    uint32_t progress0, progress1;   // player_progresses
    uint32_t player_turn;
    uint32_t opening_left;        // random opening plies still to play (INITIAL_RANDOM_MOVES, selfplay.py:32)
    uint64_t hm0, hm1;
};

// what the simulation loop needs of a slot (kept small: it lives in SGPRs across the hot loop)
struct SimCtx {
    uint64_t hgame;
    uint32_t ply, root_k, player, pool_used;
    uint32_t nsum_bias;           // Sum(N) at the root = simulation index - bias: 0 after selfplay's root
                                  // pre-expansion (selfplay.py:117), 1 in the arena where simulation 0 expands the root
};

struct Pending {                  // select -> expand_backup hand-off of the stepped path (64 bytes); on the free-running path the
                                  // record IS the request the evaluator reads (ccsp_request of ccsp.h: same layout)
    ccsp_sr leaf;                 // leaf position
    uint32_t kind;                // 0 none, 1 expand, 2 terminal; free-running: 1 a leaf / 3, 4 a ply's root asks the evaluator
    uint32_t depth;
    uint32_t link_off;            // byte offset in the pool of the child word to link (or ~0u for the root)
    uint32_t leaf_player;
    uint32_t k;                   // free-running: legal moves of `leaf` = entries of the slot's row of the move table and of the answer
    uint32_t walk_c;              // free-running, kind 0: a selection given up at the deadline goes on in front of this child word ...
    uint32_t walk_at;             // ... at level (low 16 bits; 0 = no such walk) with Sum(N) (high 16 bits) ...
    uint32_t walk_edges;          // ... having scanned this many edges so far (the counter of the byte model)
};
static_assert(sizeof(Pending) == 64 && sizeof(Pending) == sizeof(ccsp_request), "Pending must stay 64 bytes");
static_assert(offsetof(Pending, kind) == offsetof(ccsp_request, kind) && offsetof(Pending, leaf_player) == offsetof(ccsp_request, player) &&
              offsetof(Pending, k) == offsetof(ccsp_request, k), "ccsp_request is the public face of Pending");
constexpr int REQ_MV = CCSP_REQUEST_MOVES;     // row stride of the request move table and of the compact answer (entries)
constexpr uint16_t MV_WINS = 0x8000;           // move-table entry: the move wins (terminal leaf)

constexpr int CCSP_DBG_STRIDE = CCSP_DEBUG_WORDS_PER_SLOT;
struct Params {
    SlotMem *slots;
    Pending *pend;
    uint8_t *pool;
    uint8_t *pool2;               // ccsp_enable_tree_reuse: the second tree pool (a ply is searched in one, the previous ply's tree stays in the other)
    uint64_t pool_stride;
    uint64_t *path;
    uint32_t path_stride;
    const double *sqrt_tab;
    const double *pow_tab;
    const double *rcp_tab;        // 1/i, max(sims + 2, RCP_N) entries
    unsigned long long *counters;
    unsigned long long *dbg;      // [n_slots][CCSP_DBG_STRIDE] diagnostic cycle sums of advance_kernel per slot (CCSP_ADVANCE_DEBUG; ccsp_debug_read adds them up)
    uint32_t *stepacc;            // [n_slots][8] per-slot tallies of the stepped path (flushed once per ply: no
                                  // global atomics inside the per-simulation kernels)
    unsigned long long *visit_hist;
    ccsp_state *log_state;
    ccsp_sample_meta *log_meta;
    double *log_pi;
    unsigned long long log_cap;
    unsigned long long *log_count;
    ccsp_game_result *results;
    unsigned long long max_games;
    uint64_t first_game, stride, seed;
    int n_slots, sims, randomised, auto_restart, max_plies;
    int arena, arena_det_tau, enforce_move_limit;   // next-3: Game.start / AiPlayer semantics (game.py, player.py)
    int greedy, gen, stuck_limit;                   // next-4: GreedyPlayer seats (CCSP_GREEDY_*), data-generator mode, its ply cap
};

constexpr int RCP_N = 422;          // reciprocal table entries kept in LDS by the fused simulation kernel: searches of up to 420
                                    // simulations (BASELINE's 400) divide through it; 422 doubles = the 3376 bytes the ply-end data of
                                    // the same union needs, so that the per-wave LDS stays at 6.9 KB and THREE tree-kernel workgroups
                                    // fit beside an evaluator workgroup (ccsp_net.hip)

// The line tables of ccsp_rules.h as the engine's one-wave workgroups keep them in LDS: both senses of a hop in ONE byte (two 3-bit
// landing positions, 7 = none) -- 1340 bytes instead of 2236.  With the evaluator's 137-KB workgroup on a CU every byte of a tree
// workgroup decides how many of them fit beside it (advance_kernel: 2.2 KB = ten; 3.1 KB was seven).
struct alignas(16) EngineLines {           // (1344 bytes with its tail padding: copied as 84 x 16 bytes)
    uint8_t lp[CCSP_NCELL][4];          // [cell][axis] = line << 3 | position
    uint8_t cell[CCSP_NLINES][8];       // [line][position]
    uint8_t base[CCSP_NLINES + 5];      // off-board bits of each line's pattern
    uint8_t hopp[128][7];               // [pattern][position]: low nibble = landing in sense -, high nibble = in sense +
};
static_assert(sizeof(EngineLines) % 4 == 0, "copied as dwords");
static constexpr EngineLines make_engine_lines() {
    const ccsp_line_tables t = ccsp_make_lines();
    EngineLines e{};
    for (int c = 0; c < CCSP_NCELL; c++) for (int a = 0; a < 4; a++) e.lp[c][a] = t.lp[c][a];
    for (int l = 0; l < CCSP_NLINES; l++) for (int q = 0; q < 8; q++) e.cell[l][q] = t.cell[l][q];
    for (int l = 0; l < CCSP_NLINES + 5; l++) e.base[l] = t.base[l];
    for (int pat = 0; pat < 128; pat++) for (int q = 0; q < 7; q++) e.hopp[pat][q] = (uint8_t)(t.hop[pat][q][0] | (t.hop[pat][q][1] << 4));
    return e;
}
static __device__ const EngineLines ENGINE_LINES_DEV = make_engine_lines();
// one wave copies the tables: two 16-byte loads per lane, both in flight before either is waited for (as a loop of dwords the copy was
// six dependent round trips at the head of every launch of every stepped kernel)
__device__ __forceinline__ void load_engine_lines(EngineLines *lds, int tid) {
    constexpr int N16 = (int)(sizeof(EngineLines) / 16);
    static_assert(sizeof(EngineLines) % 16 == 0 && N16 > 64 && N16 <= 128, "two passes of 64 lanes x 16 bytes");
    const uint4 *src = reinterpret_cast<const uint4 *>(&ENGINE_LINES_DEV);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    const int second = tid + 64 < N16 ? tid + 64 : tid;
    const uint4 a = src[tid], b = src[second];
    dst[tid] = a;
    if (tid + 64 < N16) dst[tid + 64] = b;
}

struct Lds {                      // per-wave scratch (one wave per workgroup)
    EngineLines T;                // line tables, copied from device constant data
    uint32_t lines[32];           // occupancy pattern of the 27 board lines for the position being expanded
    uint8_t lists[6][24];
    union __attribute__((aligned(4))) {
        uint8_t stack[6][96];     // depth-first stacks of wave_movegen (<= 6 pushes per visited sub-lattice cell)
        uint8_t img[CCSP_PLANES + 1];   // wave_encode's byte image (never alive while a move list is being generated)
    };
    uint8_t cnt[8];
    // LAST: a kernel that neither ends plies nor draws root noise nor starts games nor divides through the table (advance_kernel)
    // allocates the struct only up to here (LDS_LIGHT bytes) -- 2.2 KB instead of 5.6: ten of its workgroups fit beside an evaluator
    // workgroup, not three
    union {
        struct {                  // ply begin / end: pi vector, Dirichlet draws
            double pi[CCSP_NUM_ACTIONS];
            double gam[CCSP_MAX_MOVES + 2];
        };
        double rcp[RCP_N];        // simulation loop: 1/i, i < RCP_N (pick_edge); refilled at the start of every ply's simulations
        uint8_t cells[CCSP_NCELL + 7];   // slot_start_game's shuffle (Board(randomised=True)): after the ply's pi is logged and sampled
    };
};
constexpr size_t LDS_LIGHT = (offsetof(Lds, pi) + 15) & ~(size_t)15;

struct Tally {                    // wave-uniform counters, flushed once per kernel (named scalars: never indexed)
    unsigned long long expansions, terminal_sims, sims, plies, mcts_plies, games_won, games_discarded,
        sum_depth, sum_children, select_edges, samples, errors;
};

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }
// the lane id through an instruction the compiler cannot hoist: everything wave_movegen derives from the lane (group, direction, axis,
// strides ...) is then computed WHERE the generator runs.  advance_kernel generates moves behind its simulation loop; with the plain
// lane id those per-lane constants are hoisted to the top of the kernel, live across the loop and spill to scratch (24-56 dwords per lane:
// every one four cache lines of traffic beside the evaluator).
__device__ __forceinline__ int lane_id_here() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}


// src must be wave-uniform
__device__ __forceinline__ uint32_t bcast32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src) {
    const uint32_t lo = bcast32((uint32_t)v, src), hi = bcast32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v);
}
__device__ __forceinline__ ccsp_sr uni_sr(const ccsp_sr &s) {
    ccsp_sr r; r.occ0 = uni64(s.occ0); r.occ1 = uni64(s.occ1); r.a = uni64(s.a); r.b = uni64(s.b); return r;
}

// max over the 64 lanes, same value returned in every lane.  All lanes must be active, no NaNs.
// The doubles are compared as order-preserving integer keys, high word first and then the low words of the
// lanes that hold the largest high word: two 32-bit reductions whose steps are single DPP-fused v_max_u32
// (xor 1, xor 2, mirror within 8, mirror within 16, row_bcast:15, row_bcast:31; lane 63 holds the result)
// instead of 64-bit compare/select chains.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t x) {
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);   // 0 = identity of max
    return y > x ? y : x;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    x = dpp_max_u32<0xB1, 0xF>(x);         // quad_perm [1,0,3,2]
    x = dpp_max_u32<0x4E, 0xF>(x);         // quad_perm [2,3,0,1]
    x = dpp_max_u32<0x141, 0xF>(x);        // row_half_mirror
    x = dpp_max_u32<0x140, 0xF>(x);        // row_mirror
    x = dpp_max_u32<0x142, 0xA>(x);        // row_bcast:15 into rows 1 and 3
    x = dpp_max_u32<0x143, 0xC>(x);        // row_bcast:31 into rows 2 and 3
    return bcast32(x, 63);
}
__device__ __forceinline__ double wave_max_f64(double x) {
    const uint64_t b = ccsp_to_bits(x);
    const uint32_t h = (uint32_t)(b >> 32), l = (uint32_t)b;
    const uint32_t neg = (uint32_t)((int32_t)h >> 31);                 // all ones for negative values
    const uint32_t kh = h ^ (neg | 0x80000000u), kl = l ^ neg;         // unsigned order of (kh, kl) = order of the doubles
    const uint32_t mh = wave_max_u32(kh);
    const uint32_t ml = wave_max_u32(kh == mh ? kl : 0u);
    const uint32_t back = (mh & 0x80000000u) ? 0u : 0xFFFFFFFFu;
    return ccsp_from_bits(((uint64_t)(mh ^ (back | 0x80000000u)) << 32) | (uint64_t)(ml ^ back));
}

// wave_max_f64 that stops after the high words when exactly one lane holds the largest: `eq_high` = the lanes with the largest high
// word, `unique` = there is one (then mx is not computed: the caller reads that lane's value)
struct KeyMax { uint64_t eq_high; bool unique; double mx; };
__device__ __forceinline__ KeyMax wave_max_f64_key(double x) {
    const uint64_t b = ccsp_to_bits(x);
    const uint32_t h = (uint32_t)(b >> 32), l = (uint32_t)b;
    const uint32_t neg = (uint32_t)((int32_t)h >> 31);
    const uint32_t kh = h ^ (neg | 0x80000000u), kl = l ^ neg;
    const uint32_t mh = wave_max_u32(kh);
    KeyMax r;
    r.eq_high = __ballot(kh == mh);
    r.unique = (r.eq_high & (r.eq_high - 1)) == 0;
    r.mx = 0.0;
    if (!r.unique) {
        const uint32_t ml = wave_max_u32(kh == mh ? kl : 0u);
        const uint32_t back = (mh & 0x80000000u) ? 0u : 0xFFFFFFFFu;
        r.mx = ccsp_from_bits(((uint64_t)(mh ^ (back | 0x80000000u)) << 32) | (uint64_t)(ml ^ back));
    }
    return r;
}

// the r-th (0-based) set bit of a 128-bit wave-uniform mask (lo = entries 0..63, hi = 64..127):
// every lane ranks its own bit, the matching lane is found by ballot
__device__ __forceinline__ int nth_set_bit(uint64_t lo, uint64_t hi, int r) {
    const int lane = lane_id();
    const uint64_t below = (1ULL << lane) - 1;
    const int nlo = ccsp_popc64(lo);
    const bool hit_lo = ((lo >> lane) & 1) && ccsp_popc64(lo & below) == r;
    const bool hit_hi = ((hi >> lane) & 1) && nlo + ccsp_popc64(hi & below) == r;
    const uint64_t b_lo = __ballot(hit_lo), b_hi = __ballot(hit_hi);
    return b_lo ? ccsp_ctz64(b_lo) : 64 + ccsp_ctz64(b_hi);
}

// ---- node block addressing --------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_bytes(int k) { return (uint32_t)((BLOCK_HDR + 26 * k + 7) & ~7); }
__device__ __forceinline__ double *blk_P(uint8_t *b, int) { return reinterpret_cast<double *>(b + BLOCK_HDR); }
__device__ __forceinline__ double *blk_W(uint8_t *b, int k) { return reinterpret_cast<double *>(b + BLOCK_HDR + 8 * k); }
__device__ __forceinline__ uint32_t *blk_N(uint8_t *b, int k) { return reinterpret_cast<uint32_t *>(b + BLOCK_HDR + 16 * k); }
__device__ __forceinline__ uint32_t *blk_child(uint8_t *b, int k) { return reinterpret_cast<uint32_t *>(b + BLOCK_HDR + 20 * k); }
__device__ __forceinline__ uint16_t *blk_mv(uint8_t *b, int k) { return reinterpret_cast<uint16_t *>(b + BLOCK_HDR + 24 * k); }

// Experiment switch (round 6): CCSP_TREE_NT = 1 gives the free-running path's block WRITES (wave_expand_answer,
// wave_copy_block) a streaming cache policy, 2 also the old tree's reads of wave_copy_block and the selection's edge loads -- to see whether
// the tree waves' traffic costs the evaluator beside them its weights in L2.  Default 0: measured, no gain (profiles/r6_pipeline_ab.txt).
#ifndef CCSP_TREE_NT
#define CCSP_TREE_NT 0
#endif
template <typename T> __device__ __forceinline__ void st_blk(T *p, T v) { if (CCSP_TREE_NT >= 1) __builtin_nontemporal_store(v, p); else *p = v; }
template <typename T> __device__ __forceinline__ T ld_old(const T *p) { return CCSP_TREE_NT >= 2 ? __builtin_nontemporal_load(p) : *p; }

__device__ __forceinline__ uint64_t path_entry(uint32_t off8, int k, int j) { return ((uint64_t)off8 << 16) | ((uint64_t)k << 8) | (uint64_t)j; }

// ---- B2-B4 for one position, wave-cooperative ----------------------------------------------------------
// Lane = (checker g = lane/8, direction d = lane%8; 6 x 6 lanes work).  The six checkers run the
// reference's depth-first hop search (board.py:166-211) in lock-step, one visited cell per iteration: the six
// direction lanes of a checker evaluate their mirror hop from the checker's current cell at once (one
// lookup in the line tables of ccsp_rules.h: HOP[line pattern][position][sense]) and a ballot ranks the legal
// landings onto the checker's stack in LDS (see the loop).
// Walk cells never coincide with hop landings (different sub-lattice), so they need no visited bit.
// result: lds.lists / lds.cnt; returns K (wave-uniform)
// `pick` (rollouts only): called once after the first iteration with the 6-bit mask of checkers that can move at
// all (a walk, or a hop from where they stand); it names ONE checker and only that checker's search is carried on --
// rule S1 (selfplay.py:95-98) needs one checker's list, not all six.  NoPick = every checker, the full list.
struct NoPick { __device__ __forceinline__ int operator()(uint32_t) const { return -1; } };

template <typename Pick, bool HERE = false>
__device__ __forceinline__ int wave_movegen_impl(Lds &lds, const ccsp_sr &st, int player, Pick pick) {
    const int lane = HERE ? lane_id_here() : lane_id();
    const int grp = lane >> 3, dir = lane & 7;
    const bool act = (grp < 6) & (dir < 6);
    const int g = grp < 6 ? grp : 0, d = dir < 6 ? dir : 0;
    const EngineLines &T = lds.T;
    __syncthreads();                                   // previous users of lists/cnt/lines are done
    // occupancy patterns of the 27 lines: off-board bits preset, then one bit per checker and axis
    if (lane < CCSP_NLINES) lds.lines[lane] = T.base[lane];
    __syncthreads();
    if (lane < 12) {
        const int cell = ccsp_sr_pos(st, lane);
#pragma unroll
        for (int a = 0; a < 3; a++) { const int lp = T.lp[cell][a]; atomicOr(&lds.lines[lp >> 3], 1u << (lp & 7)); }
    }
    __syncthreads();
    const int origin = ccsp_sr_pos(st, (player - 1) * 6 + g);
    const int axis = d % 3, sense = ((d >= 1) & (d <= 3)) ? 1 : 0;
    const int olp = T.lp[origin][axis];
    const int oline = olp >> 3, opos = olp & 7;
    const uint32_t omask = ~(1u << opos);              // the moving checker is lifted off its lines (board.py:158)
    // walks, direction order (board.py:149-155)
    const int np = opos + (sense ? 1 : -1);
    const bool walk = act & (np >= 0) & (np <= 6) & (((lds.lines[oline] >> (np & 7)) & 1u) == 0);
    const int nb = T.cell[oline][np & 7];
    const uint32_t wm = (uint32_t)(__ballot(walk) >> (8 * grp)) & 0x3Fu;
    if (walk) lds.lists[g][__popc(wm & ((1u << d) - 1u))] = (uint8_t)nb;
    int n = __popc(wm);
    // hops (board.py:166-211): explicit-stack depth-first search.  Popping a cell that is still unvisited
    // visits it (= the recursive call), evaluates its six mirror hops at once and pushes the legal, unvisited
    // landings with the FIRST direction on top; a popped cell that was reached through another branch in the
    // meantime is dropped (= the `not in hops` test the caller's loop makes when it gets to that direction).
    // That is the pre-order of the recursion, in about V+1 steps for V hop cells instead of one step per
    // tree edge in each direction.
    // row/column of a cell without division: r = (37 * cell) >> 8 is exact for cell < 56
    auto row_of = [](int cell) { return (int)(__umul24((unsigned)cell, 37u) >> 8); };
    const int stride = axis == 0 ? 7 : (axis == 1 ? 1 : 8);          // cell-index step per line position
    // line index of a cell on this lane's axis = l0 + lr * row + lc * col   (c | 7 + r | 20 + r - c)
    const int l0 = axis == 0 ? 0 : (axis == 1 ? 7 : 20), lr = axis == 0 ? 0 : 1, lc = axis == 0 ? 1 : (axis == 1 ? 0 : -1);
    uint64_t visited = 0;
    int sp = 1;
    bool alive = grp < 6;
    if (act && dir == 0) lds.stack[g][0] = (uint8_t)origin;
    __builtin_amdgcn_wave_barrier();
    bool first = true;
    while (__any(alive)) {
        const int x = lds.stack[g][sp > 0 ? sp - 1 : 0];
        const bool fresh = alive & (((visited >> x) & 1) == 0);
        if (fresh) visited |= 1ULL << x;
        const int r = row_of(x), c = x - 7 * r;
        const int line = l0 + lr * r + lc * c;
        const int pos = axis == 0 ? r : (axis == 1 ? c : (r < c ? r : c));
        uint32_t pat = lds.lines[line];
        pat = line == oline ? (pat & omask) : pat;
        const int hp = (T.hopp[pat][pos] >> (4 * sense)) & 7;
        const int land = x + (hp - pos) * stride;                       // = T.cell[line][hp] when hp < 7
        const bool ok = act & fresh & (hp < 7) & (((visited >> (land & 63)) & 1) == 0);
        const uint32_t m = (uint32_t)(__ballot(ok) >> (8 * grp)) & 0x3Fu;
        if (fresh & (x != origin) & (dir == 0)) lds.lists[g][n] = (uint8_t)x;
        n += (fresh & (x != origin)) ? 1 : 0;
        if (ok) lds.stack[g][sp - 1 + __popc(m >> (d + 1))] = (uint8_t)land;
        sp = alive ? sp - 1 + __popc(m) : 0;
        alive = sp > 0;
        if (first) {
            first = false;
            const uint64_t can = __ballot((grp < 6) & ((n > 0) | (sp > 0)));         // lane 8 g speaks for checker g
            uint32_t hm = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) hm |= (uint32_t)((can >> (8 * i)) & 1) << i;
            const int only = pick(hm);
            if (only >= 0) alive = alive & (grp == only);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (act && dir == 0) lds.cnt[g] = (uint8_t)n;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) total += lds.cnt[i];
    return total;
}

template <bool HERE = false>
__device__ __forceinline__ int wave_movegen(Lds &lds, const ccsp_sr &st, int player) {
    return wave_movegen_impl<NoPick, HERE>(lds, st, player, NoPick());
}

// entry j of the flattened move list -> (checker id, destination)
__device__ __forceinline__ void move_of(const Lds &lds, int j, int &id, int &dest) {
    int base = 0; id = 0;
    bool go = true;                                     // stop at the first checker whose list holds entry j
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int c = lds.cnt[i];
        go = go && (j >= base + c);
        if (go) { base += c; id = i + 1; }
    }
    dest = lds.lists[id][j - base];
}

// ---- evaluators of the fused path --------------------------------------------------------------
struct EvalCtx {
    int kind;
    const double *p_row;          // external: p[slot][294]
    float v_ext;                  // external / cached: the position's value (kept in the block header)
    const double *p_edges;        // cached: P[K] of the same position's block in the previous ply's tree, by edge index
    uint32_t shadow;              // cached: that block's offset / 8 (kept in the new block's header: its children find theirs through it)
};

__device__ __forceinline__ double prior_of(const EvalCtx &ev, const ccsp_sr &st, int player, uint64_t key, int id, int dest) {
    switch (ev.kind) {
        case CCSP_EVAL_HASH: return ccsp_hash_prior(key, id * CCSP_NCELL + dest);
        case CCSP_EVAL_FORWARD: return ccsp_forward_prior(st, player, id, dest);
        case CCSP_EVAL_EXTERNAL: return ev.p_row[id * CCSP_NCELL + dest];
        default: return 1.0 / 294.0;
    }
}

// config 2b: random playout from `st` (player to move `player`), S1's sampling rule, <= 64 plies
__device__ __forceinline__ float wave_rollout(Lds &lds, ccsp_sr st, int player, uint64_t hgame, uint32_t ply, uint32_t sim_key) {
    const int leaf_player = player;
    uint32_t counter = 0;
    for (int step = 0; step < 64; step++) {
        // rule S1 (selfplay.py:95-98): a uniform checker among those that can move (drawn again while it cannot), then
        // a uniform destination of it -- only that checker's hop search is run to the end
        int id = -1;
        auto pick = [&](uint32_t can_move) -> int {
            if (can_move == 0) return 6;                    // nobody moves: no search to finish
            for (;;) {
                id = (int)ccsp_choice(ccsp_rng_from(hgame, ply, sim_key, counter++, CCSP_P_ROLLOUT), 6);
                if ((can_move >> id) & 1) return id;
            }
        };
        wave_movegen_impl(lds, st, player, pick);
        if (id < 0) return 0.0f;
        const int t = (int)ccsp_choice(ccsp_rng_from(hgame, ply, sim_key, counter++, CCSP_P_ROLLOUT), lds.cnt[id]);
        const int dest = lds.lists[id][t];
        st = ccsp_place(st, player, id, dest);
        const int w = ccsp_check_win(st.occ0, st.occ1);
        if (w) return w == leaf_player ? 1.0f : -1.0f;
        player = 3 - player;
    }
    return 0.0f;
}

__device__ __forceinline__ float value_of(const EvalCtx &ev, Lds &lds, const ccsp_sr &st, int player, uint64_t key,
                                          uint64_t hgame, uint32_t ply, uint32_t sim_key) {
    switch (ev.kind) {
        case CCSP_EVAL_HASH: return ccsp_hash_value(key);
        case CCSP_EVAL_FORWARD: return ccsp_forward_value(st, player);
        case CCSP_EVAL_ROLLOUT: return wave_rollout(lds, st, player, hgame, ply, sim_key);
        case CCSP_EVAL_EXTERNAL: return ev.v_ext;
        default: return 0.0f;
    }
}

// ---- T3 (expansion half): create the node block of `st` -----------------------------------------
// MCTS.py:93-109.  Children are NOT materialised (the reference deep-copies a Board per child,
// MCTS.py:104): an edge keeps its move, and a child's position is rebuilt from its parent's when
// the child is expanded.  Whether the move wins (leaf.check_win(), MCTS.py:81) is decided here,
// once, and kept in the child word.  Returns K; the block is at pool + off.
// `root_noise`: apply selfplay.py:121-124 to the priors before they are stored.
__device__ __forceinline__ int wave_expand(Lds &lds, SimCtx &sl, uint8_t *pool, const ccsp_sr &st, int player,
                           const EvalCtx &ev, uint64_t key, bool root_noise, uint32_t &off_out) {
    const int lane = lane_id();
    const int K = wave_movegen(lds, st, player);
    const uint32_t off = sl.pool_used;
    off_out = off;
    if (K == 0) return 0;                               // stays a leaf (MCTS.py:95-109 adds no edge)
    uint8_t *b = pool + off;
    sl.pool_used = off + block_bytes(K);
    if (lane == 0) {
        ccsp_store_sr(reinterpret_cast<ccsp_state *>(b), st);
        reinterpret_cast<uint32_t *>(b + 32)[0] = (uint32_t)K;
        reinterpret_cast<uint32_t *>(b + 32)[1] = (uint32_t)player;
        reinterpret_cast<uint32_t *>(b + 32)[2] = ev.shadow;
        reinterpret_cast<float *>(b + 32)[3] = ev.v_ext;
    }
    double pr[2] = {0.0, 0.0};
    int idxs[2] = {0, 0};
    uint32_t ch[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) {
            int id, dest;
            move_of(lds, j, id, dest);
            idxs[h] = id * CCSP_NCELL + dest;
            pr[h] = ev.kind == CCSP_EVAL_CACHED ? ev.p_edges[j] : prior_of(ev, st, player, key, id, dest);
            // leaf.check_win() after this move (MCTS.py:81, board.py:89-111): only the mover's bitboard changes
            const int from = ccsp_sr_pos(st, (player - 1) * 6 + id);
            const uint64_t flip = (1ULL << from) | (1ULL << dest);
            const uint64_t o1 = player == 1 ? st.occ0 ^ flip : st.occ0, o2 = player == 2 ? st.occ1 ^ flip : st.occ1;
            ch[h] = ccsp_check_win(o1, o2) ? CHILD_TERMINAL : CHILD_LEAF;
        }
    }
    if (root_noise) {                                   // selfplay.py:121-124
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = lane + 64 * h;
            if (j < K) lds.gam[j] = ccsp_gamma_small(sl.hgame, sl.ply, (uint32_t)j, CCSP_DIRICHLET_ALPHA);
        }
        __syncthreads();
        if (lane == 0) {
            double s = 0.0;
            for (int j = 0; j < K; j++) s = s + lds.gam[j];
            lds.gam[CCSP_MAX_MOVES] = s;
        }
        __syncthreads();
        const double s = lds.gam[CCSP_MAX_MOVES];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = lane + 64 * h;
            if (j < K) {
                const double noise = (s == 0.0) ? 1.0 / (double)K : lds.gam[j] / s;
                double p = pr[h];
                p = p * (1. - CCSP_DIR_NOISE_FACTOR);
                p = p + CCSP_DIR_NOISE_FACTOR * noise;
                pr[h] = p;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) {
            blk_P(b, K)[j] = pr[h];
            blk_W(b, K)[j] = 0.0;
            blk_N(b, K)[j] = 0u;
            blk_child(b, K)[j] = ch[h];
            blk_mv(b, K)[j] = (uint16_t)idxs[h];
        }
    }
    return K;
}

// ---- T3 on the free-running path, split around the evaluator: the move list when the request is MADE, the block when the answer COMES ----
// Request time: Board.get_valid_moves of the position to be evaluated (MCTS.py:95) -> the slot's row of the request move table: action
// index (utils.encode_checker_index) | MV_WINS where leaf.check_win() holds after the move (MCTS.py:81, decided once, here).  The
// evaluator's epilogue reads the row and answers with the priors of exactly these moves.  Returns K.
template <bool HERE = false>      // HERE: see lane_id_here (advance_kernel)
__device__ __forceinline__ int wave_request_moves(Lds &lds, const ccsp_sr &st, int player, uint16_t *mv_out) {
    const int K = wave_movegen<HERE>(lds, st, player);
    const int lane = HERE ? lane_id_here() : lane_id();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) {
            int id, dest;
            move_of(lds, j, id, dest);
            const int from = ccsp_sr_pos(st, (player - 1) * 6 + id);
            const uint64_t flip = (1ULL << from) | (1ULL << dest);
            const uint64_t o1 = player == 1 ? st.occ0 ^ flip : st.occ0, o2 = player == 2 ? st.occ1 ^ flip : st.occ1;
            mv_out[j] = (uint16_t)((id * CCSP_NCELL + dest) | (ccsp_check_win(o1, o2) ? MV_WINS : 0));
        }
    }
    return K;
}

// Answer time: the node block of `st` (MCTS.py:97-109) from the request's move row and the compact answer pk[j] = p[action index of move j]
// -- no move generation, no line tables, no LDS (NOISE: the root's Dirichlet noise, selfplay.py:121-124, draws through lds.gam).
template <bool NOISE, bool HERE = false>       // HERE: see lane_id_here
__device__ __forceinline__ int wave_expand_answer(Lds *lds, SimCtx &sl, uint8_t *pool, const ccsp_sr &st, int player, int K,
                                                  const uint16_t *mv, const double *pk, float v, uint32_t &off_out) {
    const int lane = HERE ? lane_id_here() : lane_id();
    const uint32_t off = sl.pool_used;
    off_out = off;
    if (K == 0) return 0;                               // stays a leaf (MCTS.py:95-109 adds no edge)
    uint8_t *b = pool + off;
    sl.pool_used = off + block_bytes(K);
    if (lane == 0) {
        ccsp_store_sr(reinterpret_cast<ccsp_state *>(b), st);
        reinterpret_cast<uint32_t *>(b + 32)[0] = (uint32_t)K;
        reinterpret_cast<uint32_t *>(b + 32)[1] = (uint32_t)player;
        reinterpret_cast<uint32_t *>(b + 32)[2] = 0u;                  // no shadow: the position was not in the previous ply's tree
        reinterpret_cast<float *>(b + 32)[3] = v;
    }
    double pr[2] = {0.0, 0.0};
    uint32_t m[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) { pr[h] = pk[j]; m[h] = mv[j]; }
    }
    if (NOISE) {                                        // selfplay.py:121-124 (the arithmetic of wave_expand)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = lane + 64 * h;
            if (j < K) lds->gam[j] = ccsp_gamma_small(sl.hgame, sl.ply, (uint32_t)j, CCSP_DIRICHLET_ALPHA);
        }
        __syncthreads();
        if (lane == 0) {
            double s = 0.0;
            for (int j = 0; j < K; j++) s = s + lds->gam[j];
            lds->gam[CCSP_MAX_MOVES] = s;
        }
        __syncthreads();
        const double s = lds->gam[CCSP_MAX_MOVES];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = lane + 64 * h;
            if (j < K) {
                const double noise = (s == 0.0) ? 1.0 / (double)K : lds->gam[j] / s;
                double p = pr[h];
                p = p * (1. - CCSP_DIR_NOISE_FACTOR);
                p = p + CCSP_DIR_NOISE_FACTOR * noise;
                pr[h] = p;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) {
            st_blk(&blk_P(b, K)[j], pr[h]);
            st_blk(&blk_W(b, K)[j], 0.0);
            st_blk(&blk_N(b, K)[j], 0u);
            st_blk(&blk_child(b, K)[j], (m[h] & MV_WINS) ? CHILD_TERMINAL : CHILD_LEAF);
            st_blk(&blk_mv(b, K)[j], (uint16_t)(m[h] & 0x7FFFu));
        }
    }
    return K;
}

// ---- T2: MCTS.moveToLeaf --------------------------------------------------------------------------
struct Leaf {
    int kind;                     // 1 expand, 2 terminal; 0 = the walk was given up after `depth` levels (ccsp_advance's deadline)
    int depth;
    uint32_t next_c, next_nsum;   // kind 0: the child word of the node the walk stopped in front of and its Sum(N) -- where it goes on
    uint32_t parent_shadow;       // SHADOW only: header word "shadow" of the block the last edge leaves from
    int parent_k, sel;            // that block's edge count and the edge taken
    uint32_t link_off;            // pool offset of the child word of the last edge
    ccsp_sr st;                   // leaf position (kind 1)
    int player;                   // player to move at the leaf
};

// MCTS.py:56-72 at one node with Sum(N) > 0: the edge moveToLeaf takes.  WIDE = the node has more than 64 edges
// (two per lane); the common narrow node skips the second half altogether.
struct Pick { int sel; uint32_t c, n, mv; uint64_t w; };

template <bool WIDE, bool RCP>
__device__ __forceinline__ Pick pick_edge(const double *__restrict__ sqrt_tab, const double *rcp, const SimCtx &sl, uint8_t *b, int K,
                                          uint32_t nsum, uint32_t sim, int level) {
    constexpr int H = WIDE ? 2 : 1;
    const int lane = lane_id();
    const double sq = sqrt_tab[nsum];               // np.sqrt(N_sum), MCTS.py:62
    double qu[H], wv[H]; uint32_t n[H], ch[H], mv[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
        // (the second half -- nodes of more than 64 edges, rare -- takes the lane id through lane_id_here(): its offsets are then computed
        // when needed instead of living in four vector registers for the whole kernel)
        const int j = h ? lane_id_here() + 64 : lane;
        qu[h] = -INFINITY; wv[h] = 0.0; n[h] = 0; ch[h] = 0; mv[h] = 0;
        if (j < K) {
            // the five arrays through scalar base + 32-bit lane offset: the offsets are pinned inside the loop (an empty asm), or their
            // zero-extensions are hoisted out of it as 64-bit pairs and every load gets a 64-bit vector address computation of its own
            uint32_t o8 = 8u * (uint32_t)j, o4 = 4u * (uint32_t)j, o2 = 2u * (uint32_t)j;
            asm volatile("" : "+v"(o8), "+v"(o4), "+v"(o2));
            const uint8_t *e = b + BLOCK_HDR;
            const double p = ld_old(reinterpret_cast<const double *>(e + o8));
            const double w = ld_old(reinterpret_cast<const double *>(e + 8 * K + o8));
            wv[h] = w;
            n[h] = ld_old(reinterpret_cast<const uint32_t *>(e + 16 * K + o4));
            ch[h] = ld_old(reinterpret_cast<const uint32_t *>(e + 20 * K + o4));
            mv[h] = ld_old(reinterpret_cast<const uint16_t *>(e + 24 * K + o2));
            double U, Q;
            if (RCP) {                                                               // same quotients, fewer instructions
                const double dn = (double)n[h];
                U = ccsp_div_by_table(CCSP_C_PUCT * p * sq, 1. + dn, rcp[n[h] + 1]);
                Q = n[h] ? ccsp_div_by_table(w, dn, rcp[n[h]]) : 0.0;
            } else {
                U = CCSP_C_PUCT * p * sq / (1. + (double)n[h]);                     // left to right, MCTS.py:62
                Q = n[h] ? w / (double)n[h] : 0.0;                                   // MCTS.py:89/118
            }
            qu[h] = Q + U;
        }
    }
    // running max with an epsilon tie list (MCTS.py:65-69), closed form:
    // m = first index of the maximum; ties = {m} + {j > m : |QU_j - QU_m| < eps}
    double mx; int m;
    if (!WIDE) {
        // the usual node: ONE lane holds the largest high word of the order-preserving key -- that lane is m and its value the maximum
        // (11 vector instructions fewer than the second reduction over the low words, which runs only when high words tie)
        const KeyMax km = wave_max_f64_key(qu[0]);
        if (km.unique) { m = ccsp_ctz64(km.eq_high); mx = ccsp_from_bits(bcast64(ccsp_to_bits(qu[0]), m)); }
        else { mx = km.mx; m = ccsp_ctz64(__ballot(qu[0] == mx)); }
    } else {
        mx = wave_max_f64(qu[0] > qu[H - 1] ? qu[0] : qu[H - 1]);
        const uint64_t eq_lo = __ballot(qu[0] == mx), eq_hi = __ballot(qu[H - 1] == mx);
        m = eq_lo ? ccsp_ctz64(eq_lo) : 64 + ccsp_ctz64(eq_hi);
    }
    uint64_t tie_lo = __ballot(lane > m && fabs(qu[0] - mx) < CCSP_EPSILON);
    uint64_t tie_hi = WIDE ? __ballot(lane + 64 > m && fabs(qu[H - 1] - mx) < CCSP_EPSILON) : 0;
    Pick pk;
    pk.sel = m;
    if (tie_lo | tie_hi) {                          // rare: more than one edge within epsilon of the maximum
        if (m < 64) tie_lo |= 1ULL << m; else tie_hi |= 1ULL << (m - 64);
        const int cnt = ccsp_popc64(tie_lo) + ccsp_popc64(tie_hi);
        const int r = (int)ccsp_choice(ccsp_rng_from(sl.hgame, sl.ply, sim, (uint32_t)level, CCSP_P_SELECT), (uint32_t)cnt);   // MCTS.py:72
        pk.sel = nth_set_bit(tie_lo, tie_hi, r);
    }
    const int sl_lane = pk.sel & 63;
    const bool hi = WIDE && pk.sel >= 64;
    pk.c = bcast32(hi ? ch[H - 1] : ch[0], sl_lane);
    pk.n = bcast32(hi ? n[H - 1] : n[0], sl_lane);
    pk.mv = bcast32(hi ? mv[H - 1] : mv[0], sl_lane);
    pk.w = bcast64(ccsp_to_bits(hi ? wv[H - 1] : wv[0]), sl_lane);
    return pk;
}

#ifndef CCSP_ADVANCE_DEADLINE_CODE
#define CCSP_ADVANCE_DEADLINE_CODE 1
#endif
// `give_up_at` (ccsp_advance): a value of the 100 MHz clock (its low 32 bits) past which the walk is abandoned at the next level (Leaf.kind 0; nothing has been
// changed: the caller selects again in its next call); 0 = never
// `from` (ccsp_advance): a walk given up earlier goes on where it stopped -- at level from->level, in front of the node of child word
// from->c with Sum(N) from->nsum; the path so far is in path[0 .. level) (nothing in the tree has changed in between: the slot is its
// only writer, and the draws of a level are keyed by (ply, simulation, level)).
struct WalkFrom { uint32_t c, nsum; int level; };
template <bool RCP, bool REGPATH = RCP, bool SHADOW = false>   // RCP: divisions through a table of reciprocals; REGPATH: the caller keeps the path's first 64 levels in registers; SHADOW: tree reuse (ccsp_advance)
__device__ __forceinline__ Leaf wave_select(const double *__restrict__ sqrt_tab, const double *rcp, const SimCtx &sl, uint8_t *pool, uint64_t *path, uint32_t sim,
                                            uint64_t &mypath, double &myW, uint32_t &myN, uint32_t &select_edges, uint32_t give_up_at = 0,
                                            const WalkFrom from = WalkFrom{0u, 0u, 0}) {
    const int lane = lane_id();
    uint32_t off = 0;
    int K = (int)sl.root_k;
    uint32_t nsum = sim - sl.nsum_bias;                 // Sum(N) over the root's edges
    int level = 0;
    int player = sl.player;
    if (from.level != 0) {                              // (rare: the lanes of the levels already walked fetch what the walk left in them)
        off = (from.c >> 7) << 3; K = (int)(from.c & 127); nsum = from.nsum; level = from.level;
        if (level & 1) player = 3 - player;
        // (the lanes of the levels already walked hold nothing: the caller fetches their path entries and statistics after the walk,
        // walk_rejoin below; the path stores here leave their entries in memory alone)
    }
    const int lane0 = from.level < 64 ? from.level : 64;
    Leaf out;
    out.next_c = 0; out.next_nsum = 0;
    for (;;) {
        uint8_t *b = pool + off;
        int sel;
        uint32_t c_sel, n_sel, mv_sel;
        uint64_t w_sel;
        ccsp_sr st;
        bool have_st = false;
        uint32_t hdr_shadow = 0;
        if (SHADOW) hdr_shadow = uni32(reinterpret_cast<const uint32_t *>(b + 32)[2]);   // rides with this level's loads; read at the leaf only
        if (nsum == 0) {
            // First visit of this node: every edge has N = 0, so U = c*P*sqrt(0)/(1+0) = 0 and Q = 0 for all of
            // them -- the running-max rule (MCTS.py:65-69) keeps ALL K edges and random.choice picks uniformly
            // (SURVEY.md H3).  No statistics need to be read: draw the index, fetch that edge's child and move.
            sel = K > 1 ? (int)ccsp_choice(ccsp_rng_from(sl.hgame, sl.ply, sim, (uint32_t)level, CCSP_P_SELECT), (uint32_t)K) : 0;
            c_sel = uni32(blk_child(b, K)[sel]);
            mv_sel = uni32((uint32_t)blk_mv(b, K)[sel]);
            st = ccsp_load_sr(reinterpret_cast<const ccsp_state *>(b));     // nearly always the last node of the path:
            have_st = true;                                                  // its position comes with the same round trip
            n_sel = 0; w_sel = 0;
            select_edges += (uint32_t)K;
        } else {
        const Pick pk = K > 64 ? pick_edge<true, RCP>(sqrt_tab, rcp, sl, b, K, nsum, sim, level)
                               : pick_edge<false, RCP>(sqrt_tab, rcp, sl, b, K, nsum, sim, level);
        sel = pk.sel; c_sel = pk.c; n_sel = pk.n; mv_sel = pk.mv; w_sel = pk.w;
        select_edges += (uint32_t)K;
        }
        const uint64_t entry = path_entry(off >> 3, K, sel);
        if (level < 64) { if (lane == level) { mypath = entry; myW = ccsp_from_bits(w_sel); myN = n_sel; } }   // backup needs no reload
        if (level >= 64 && lane == 0) path[level] = entry;                 // (the first 64 levels: in `mypath`; written out below where the caller does not keep them)
        level++;
        if (c_sel != CHILD_LEAF && c_sel != CHILD_TERMINAL) {       // descend (MCTS.py:74)
            if (CCSP_ADVANCE_DEADLINE_CODE && SHADOW && give_up_at != 0 && (int32_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - give_up_at) > 0) {
                // given up between two levels: the path so far goes to memory, the caller records where the walk goes on
                if (lane >= lane0 && lane < (level < 64 ? level : 64)) path[lane] = mypath;
                out.kind = 0; out.depth = level; out.next_c = c_sel; out.next_nsum = n_sel - 1;
                return out;
            }
            off = (c_sel >> 7) << 3;
            K = (int)(c_sel & 127);
            nsum = n_sel - 1;
            player = 3 - player;
            continue;
        }
        out.depth = level;
        // the path for the kernel (or call) that backs this leaf up: ONE store of the lanes' entries instead of one per level from lane 0 -- a
        // store between two levels' loads makes the next level's wait (the counter runs in order) wait for the store's acknowledgement too
        if (!REGPATH && lane >= lane0 && lane < (level < 64 ? level : 64)) path[lane] = mypath;
        out.parent_shadow = hdr_shadow; out.parent_k = K; out.sel = sel;
        out.link_off = off + BLOCK_HDR + 20 * K + 4 * sel;
        out.player = 3 - player;
        // the position is read at the last node of the path only: a load at every level would put a second
        // memory round trip in front of each level's edge loads
        if (!have_st) st = ccsp_load_sr(reinterpret_cast<const ccsp_state *>(b));
        if (c_sel == CHILD_TERMINAL) { out.kind = 2; out.st = st; }
        else { out.kind = 1; out.st = uni_sr(ccsp_place(st, player, (int)mv_sel / CCSP_NCELL, (int)mv_sel % CCSP_NCELL)); }
        return out;
    }
}

// a resumed walk (WalkFrom) has reached its leaf: the lanes of the levels walked in the EARLIER call fetch their path entries and the
// statistics of their edges, as if the walk had been made in one go
__device__ __forceinline__ void walk_rejoin(uint8_t *pool, const uint64_t *path, int from_level, uint64_t &mypath, double &myW, uint32_t &myN) {
    const int lane = lane_id_here();
    if (lane < (from_level < 64 ? from_level : 64)) {
        mypath = path[lane];
        uint8_t *pb = pool + ((mypath >> 16) << 3);
        const int pk_ = (int)((mypath >> 8) & 0xFF), pj = (int)(mypath & 0xFF);
        myN = blk_N(pb, pk_)[pj]; myW = blk_W(pb, pk_)[pj];
    }
}

// ---- T3 (backup half): MCTS.py:83-90 (terminal) and 112-118 ---------------------------------------
// `have_stats`: the W and N of the first 64 path edges were kept in registers by wave_select (fused path)
template <bool HERE = false>       // HERE: see lane_id_here
__device__ __forceinline__ void wave_backup(uint8_t *pool, const uint64_t *path, uint64_t mypath, double myW, uint32_t myN,
                                            bool have_stats, int depth, bool terminal, float v) {
    const int lane = HERE ? lane_id_here() : lane_id();
    for (int i0 = 0; i0 < depth; i0 += 64) {
        const int i = i0 + lane;
        if (i < depth) {
            const uint64_t e = (i0 == 0) ? mypath : path[i];
            uint8_t *b = pool + ((e >> 16) << 3);
            const int K = (int)((e >> 8) & 0xFF), j = (int)(e & 0xFF);
            // the edge at depth i was played by the player to move at depth i; the leaf sits at `depth`
            const bool same = ((depth - i) & 1) == 0;                 // edge.currPlayer == leafNode.currPlayer
            double add;
            if (terminal) add = (double)(1 * (same ? -1 : 1));        // MCTS.py:87-89
            else add = (double)v * (double)(same ? 1 : -1);           // MCTS.py:115-117
            const bool reg = have_stats && i0 == 0;
            const uint32_t n0 = reg ? myN : blk_N(b, K)[j];
            const double w0 = reg ? myW : blk_W(b, K)[j];
            blk_N(b, K)[j] = n0 + 1u;
            blk_W(b, K)[j] = w0 + add;
        }
    }
}

// ---- C1 for one position into planes[343] (stepped path) --------------------------------------------
__device__ __forceinline__ void wave_encode(Lds &lds, const ccsp_sr &st, int player, float *out) {
    // the byte image of the 343 values (ids 1..6 scattered by twelve lanes, the player-two flag in channel 6 of every cell), then
    // four bytes -> four floats -> one 16-byte store per lane and pass: 86 dwords, two passes (beside the evaluator every vector
    // instruction of a tree wave waits for a gap between MFMAs: 35 of them here instead of a hundred)
    const int lane = lane_id();
    static_assert(sizeof(lds.img) >= 344 && (CCSP_PLANES + 1) % 4 == 0, "the image is cleared and read as 86 dwords");
    uint32_t *img4 = reinterpret_cast<uint32_t *>(&lds.img[0]);
    __syncthreads();
    img4[lane] = 0u;
    if (lane < 22) img4[64 + lane] = 0u;
    __syncthreads();
    if (player == 2 && lane < CCSP_NCELL) lds.img[lane * 7 + 6] = 1;             // utils.py:157-158
    if (lane < 12) ccsp_scatter_checker(st, player, lane, &lds.img[0]);
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int d = lane + 64 * pass;                                          // dword 0 .. 85 = values 4 d .. 4 d + 3
        if (d < 85) {
            const uint32_t w = img4[d];
            struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; } f;          // (a slot's 1372-byte row is 4-byte aligned, not 16)
            f.x = (float)(w & 0xFFu); f.y = (float)((w >> 8) & 0xFFu); f.z = (float)((w >> 16) & 0xFFu); f.w = (float)(w >> 24);
            *reinterpret_cast<F4 *>(out + 4 * d) = f;
        } else if (d == 85) {                                                    // values 340, 341, 342 (343 is not a multiple of four)
            const uint32_t w = img4[85];
            out[340] = (float)(w & 0xFFu); out[341] = (float)((w >> 8) & 0xFFu); out[342] = (float)((w >> 16) & 0xFFu);
        }
    }
}

// ---- game bookkeeping ----------------------------------------------------------------------------------
__device__ __forceinline__ void slot_start_game(const Params &P, Lds &lds, Slot &sl, unsigned long long index) {
    const int lane = lane_id();
    const uint64_t game = P.first_game + index * P.stride;
    sl.game = game; sl.index = index; sl.hgame = ccsp_rng_game(P.seed, game);
    sl.expansions = 0; sl.ply = 0; sl.n_hist = 0; sl.useless = 0; sl.pool_used = 0; sl.root_k = 0; sl.sim = 0;
    sl.player = 1; sl.status = CCSP_ST_RUNNING; sl.det_tau = 0; sl.n_hm = 0;
    sl.progress0 = sl.progress1 = 0; sl.player_turn = 0; sl.hm0 = sl.hm1 = 0;
    sl.opening_left = P.arena ? 0u : (uint32_t)CCSP_INITIAL_RANDOM_MOVES;      // Game.start has no random opening
    if (P.gen) sl.opening_left = (P.greedy & CCSP_GREEDY_RANDOM_START) ? (uint32_t)CCSP_INITIAL_RANDOM_MOVES : 0u;   // data_generators.py:31
    if (P.arena) sl.det_tau = (uint32_t)P.arena_det_tau;                        // Game(tree_tau=...) (game.py:9)
    uint8_t pos[12] = {42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4};       // Board.__init__ (board.py:42-46)
    if (P.randomised) {                                                      // board.py:61-85 via spec.pick_distinct
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < CCSP_NCELL; i++) lds.cells[i] = (uint8_t)i;
            for (int i = 0; i < 12; i++) {
                const int j = i + (int)ccsp_choice(ccsp_rng_from(sl.hgame, 0, (uint32_t)i, 0, CCSP_P_INIT), (uint32_t)(CCSP_NCELL - i));
                const uint8_t t = lds.cells[i]; lds.cells[i] = lds.cells[j]; lds.cells[j] = t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 12; i++) pos[i] = lds.cells[i];
    }
    ccsp_sr s; s.occ0 = 0; s.occ1 = 0; s.a = 0; s.b = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        if (i < 6) s.occ0 |= 1ULL << pos[i]; else s.occ1 |= 1ULL << pos[i];
        if (i < 8) s.a |= (uint64_t)pos[i] << (8 * i); else s.b |= (uint64_t)pos[i] << (8 * (i - 8));
    }
    s.b |= 0xFFFFFFFFULL << 32;                                              // no history yet
    sl.st = s;
}

// next-3: Game.start's loop body after decide_move (game.py:64-91): place, winner first, then the ring of the last
// 16 destinations with ITS repetition test (full ring, <= 3 distinct destinations of the mover), then the plain
// move count limit when enforce_move_limit.  `useless` holds num_moves.
__device__ __forceinline__ void slot_finish(const Params &P, Lds &lds, Slot &sl, int status, Tally &tl);

// auto_restart: slot g plays the game indices g, g + n_slots, g + 2 n_slots, ... below max_games.  No shared counter: which game
// a slot plays next does not depend on which other slot finished first, so a run is reproducible whatever the number of plies a
// launch carries (fused_plies_kernel) and however the waves are scheduled.
__device__ __forceinline__ void slot_restart(const Params &P, Lds &lds, Slot &sl) {
    const unsigned long long idx = sl.index + (unsigned long long)P.n_slots;
    if (idx < P.max_games) slot_start_game(P, lds, sl, idx);
    else sl.status = CCSP_ST_IDLE;
}

__device__ __forceinline__ void slot_after_move_arena(const Params &P, Lds &lds, Slot &sl, int id, int dest, Tally &tl) {
    const ccsp_sr ns = ccsp_place(sl.st, (int)sl.player, id, dest);
    sl.st = ns;
    sl.ply += 1;                                                             // total_moves
    tl.plies += 1;
    int status = CCSP_ST_RUNNING;
    const int w = ccsp_check_win(ns.occ0, ns.occ1);
    if (w) status = w;                                                       // game.py:70-71
    else {
        if (sl.n_hm == CCSP_TOTAL_HIST_MOVES) {                              // game.py:73-75
            sl.hm0 = (sl.hm0 >> 8) | (sl.hm1 << 56);
            sl.hm1 = sl.hm1 >> 8;
            sl.n_hm = CCSP_TOTAL_HIST_MOVES - 1;
        }
        {
            const int i = (int)sl.n_hm;
            if (i < 8) sl.hm0 = (sl.hm0 & ~(0xFFULL << (8 * i))) | ((uint64_t)dest << (8 * i));
            else sl.hm1 = (sl.hm1 & ~(0xFFULL << (8 * (i - 8)))) | ((uint64_t)dest << (8 * (i - 8)));
            sl.n_hm += 1;
        }
        uint64_t dests = 0;                                                  // game.py:78-82
        for (int i = (int)sl.n_hm - 1; i >= 0; i -= 2) {
            const int d = i < 8 ? (int)((sl.hm0 >> (8 * i)) & 0xFF) : (int)((sl.hm1 >> (8 * (i - 8))) & 0xFF);
            dests |= 1ULL << d;
        }
        if (sl.n_hm == CCSP_TOTAL_HIST_MOVES && ccsp_popc64(dests) <= CCSP_UNIQUE_DEST_LIMIT) status = CCSP_ST_DISCARD_REPETITION;
        else {
            sl.useless += 1;                                                 // num_moves (game.py:84)
            if (P.enforce_move_limit && sl.useless >= CCSP_PROGRESS_MOVE_LIMIT) status = CCSP_ST_DISCARD_NO_PROGRESS;   // 86-89
            else if ((int)sl.ply >= P.max_plies) status = CCSP_ST_ERROR;
            sl.player = 3 - sl.player;                                       // swap_players
            sl.player_turn = 1 - sl.player_turn;
        }
    }
    sl.status = (uint32_t)status;
    if (status != CCSP_ST_RUNNING) slot_finish(P, lds, sl, status, tl);
}

// a finished game (arena / greedy generator): result row, counters, the slot's next game
__device__ __forceinline__ void slot_finish(const Params &P, Lds &lds, Slot &sl, int status, Tally &tl) {
    const bool won = status == CCSP_ST_WON_P1 || status == CCSP_ST_WON_P2, bad = status == CCSP_ST_ERROR;
    tl.games_won += won ? 1ULL : 0ULL;
    tl.errors += bad ? 1ULL : 0ULL;
    tl.games_discarded += (!won && !bad) ? 1ULL : 0ULL;
    if (lane_id() == 0 && sl.index < P.max_games) {
        const uint32_t reward = (uint32_t)(uint8_t)(int8_t)(status == CCSP_ST_WON_P1 ? 1 : (status == CCSP_ST_WON_P2 ? -1 : 0));
        const uint64_t w0 = (uint64_t)(uint32_t)status | ((uint64_t)reward << 8) | ((uint64_t)(sl.ply & 0xFFFF) << 16) | ((uint64_t)sl.n_hist << 32);
        *reinterpret_cast<ulonglong2 *>(P.results + sl.index) = make_ulonglong2(w0, sl.expansions);
    }
    if (P.auto_restart && !bad) slot_restart(P, lds, sl);      // after an ERROR (sample log full, ply cap) the slot stays out of play
}

// selfplay.py:38-74 after a ply was played: `moved` = (from, to) by sl.player on sl.st -> updates sl
__device__ __forceinline__ void slot_after_move(const Params &P, Lds &lds, Slot &sl, int id, int dest, Tally &tl) {
    const ccsp_sr ns = ccsp_place(sl.st, (int)sl.player, id, dest);
    sl.st = ns;
    sl.ply += 1;
    tl.plies += 1;
    // Board.hist_moves: deque of the last 16 moves (board.py:246-248); only destinations are read
    if (sl.n_hm == CCSP_TOTAL_HIST_MOVES) {                                  // popleft
        sl.hm0 = (sl.hm0 >> 8) | (sl.hm1 << 56);
        sl.hm1 = sl.hm1 >> 8;
        sl.n_hm = CCSP_TOTAL_HIST_MOVES - 1;
    }
    {
        const int i = (int)sl.n_hm;
        if (i < 8) sl.hm0 = (sl.hm0 & ~(0xFFULL << (8 * i))) | ((uint64_t)dest << (8 * i));
        else sl.hm1 = (sl.hm1 & ~(0xFFULL << (8 * (i - 8)))) | ((uint64_t)dest << (8 * (i - 8)));
        sl.n_hm += 1;
    }
    // repetition rule, selfplay.py:40-47
    int n_cur = 0; uint64_t dests = 0;
    for (int i = (int)sl.n_hm - 1; i >= 0; i -= 2) {
        n_cur++;
        const int d = i < 8 ? (int)((sl.hm0 >> (8 * i)) & 0xFF) : (int)((sl.hm1 >> (8 * (i - 8))) & 0xFF);
        dests |= 1ULL << d;
    }
    int status = CCSP_ST_RUNNING;
    if (n_cur * 2 >= CCSP_TOTAL_HIST_MOVES && ccsp_popc64(dests) <= CCSP_UNIQUE_DEST_LIMIT) status = CCSP_ST_DISCARD_REPETITION;
    if (status == CCSP_ST_RUNNING) {
        const int pt = (int)sl.player_turn;
        const int pe = ccsp_progress(ns, pt + 1);                             // selfplay.py:50
        const int best = (int)(pt ? sl.progress1 : sl.progress0);
        if (pe > best) {                                                      // int(n * 5 / 6), selfplay.py:52
            sl.useless = (sl.useless * 5) / 6;
            if (pt) sl.progress1 = (uint32_t)pe; else sl.progress0 = (uint32_t)pe;
        } else sl.useless += 1;
        sl.player_turn = (uint32_t)(1 - pt);
        sl.player = 3 - sl.player;
        if ((int)sl.n_hist + CCSP_INITIAL_RANDOM_MOVES > CCSP_TOTAL_MOVES_TILL_TAU0) sl.det_tau = 1;   // selfplay.py:62-65
        const int w = ccsp_check_win(ns.occ0, ns.occ1);
        if (w) status = w;                                                    // selfplay.py:67-69
        else if (sl.useless >= CCSP_PROGRESS_MOVE_LIMIT) status = CCSP_ST_DISCARD_NO_PROGRESS;   // 72-74
        else if ((int)sl.ply >= P.max_plies) status = CCSP_ST_ERROR;
    }
    sl.status = (uint32_t)status;
    if (status != CCSP_ST_RUNNING) {
        const bool won = status == CCSP_ST_WON_P1 || status == CCSP_ST_WON_P2, bad = status == CCSP_ST_ERROR;
        tl.games_won += won ? 1ULL : 0ULL;                 // (arithmetic, not branches: keeps the tally in registers)
        tl.errors += bad ? 1ULL : 0ULL;
        tl.games_discarded += (!won && !bad) ? 1ULL : 0ULL;
        if (lane_id() == 0 && sl.index < P.max_games) {
            const uint32_t reward = (uint32_t)(uint8_t)(int8_t)(status == CCSP_ST_WON_P1 ? 1 : (status == CCSP_ST_WON_P2 ? -1 : 0));   // utils.py:34-44
            const uint64_t w0 = (uint64_t)(uint32_t)status | ((uint64_t)reward << 8) | ((uint64_t)(sl.ply & 0xFFFF) << 16) | ((uint64_t)sl.n_hist << 32);
            *reinterpret_cast<ulonglong2 *>(P.results + sl.index) = make_ulonglong2(w0, sl.expansions);
        }
        if (P.auto_restart && !bad) slot_restart(P, lds, sl);
    }
}

// S1: selfplay.make_random_move (selfplay.py:83-104)
// ---- next-4: GreedyPlayer (player.py:67-129) ------------------------------------------------------------
// is the player to move a GreedyPlayer seat?
__device__ __forceinline__ bool greedy_to_move(const Params &P, const Slot &sl) {
    if (P.gen) return true;
    uint32_t seats = (uint32_t)P.greedy & 3u;
    if ((P.greedy & CCSP_GREEDY_ALTERNATE) && (sl.game & 1)) seats = ((seats & 1u) << 1) | (seats >> 1);   // ai_vs_greedy.py:47-48
    return ((seats >> (sl.player - 1)) & 1u) != 0;
}
// is the GreedyPlayer to move the stochastic variant (player.py:68, stochastic=True)?
__device__ __forceinline__ bool stochastic_to_move(const Params &P, const Slot &sl) {
    uint32_t seats = ((uint32_t)P.greedy >> 4) & 3u;
    if ((P.greedy & CCSP_GREEDY_ALTERNATE) && (sl.game & 1)) seats = ((seats & 1u) << 1) | (seats >> 1);
    return ((seats >> (sl.player - 1)) & 1u) != 0;
}
// no search this ply: a random opening ply or a GreedyPlayer's move
__device__ __forceinline__ bool no_search(const Params &P, const Slot &sl) { return sl.opening_left > 0 || greedy_to_move(P, sl); }

__device__ __forceinline__ int human_row(int cell) {                   // board_utils.np_index_to_human_coord()[0]
    const int r = (int)(__umul24((unsigned)cell, 37u) >> 8);
    return 8 * r - cell + 7;                                            // row - col + BOARD_WIDTH, col = cell - 7 row
}

// decide_move(training=True) on the move list wave_movegen left in LDS (K entries): the set of filtered best
// moves as a 128-bit mask over list positions; returns its size.
//   player.py:100-110  keep the moves of maximum forward distance (rows of the human view; player one moves up)
//   player.py:112-115  of those, only the ones that start on the row of the last (rear-most) checker
__device__ __forceinline__ int wave_greedy_best(const Lds &lds, const ccsp_sr &st, int player, int K, uint64_t &f_lo, uint64_t &f_hi) {
    const int lane = lane_id();
    uint32_t dist[2], key[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        dist[h] = 0; key[h] = 0;
        if (j < K) {
            int id, dest;
            move_of(lds, j, id, dest);
            const int s = human_row(ccsp_sr_pos(st, (player - 1) * 6 + id)), e = human_row(dest);
            const int d = player == 1 ? s - e : e - s;
            dist[h] = (uint32_t)(d + 32);                                // 1 .. 63: 0 is "no move"
            key[h] = (uint32_t)((player == 1 ? s : 14 - s) + 1);        // rear-most checker = largest key
        }
    }
    const uint32_t dmax = wave_max_u32(dist[0] > dist[1] ? dist[0] : dist[1]);
    const bool b0 = dist[0] == dmax && dist[0] != 0, b1 = dist[1] == dmax && dist[1] != 0;
    const uint32_t k0 = b0 ? key[0] : 0u, k1 = b1 ? key[1] : 0u;
    const uint32_t kmax = wave_max_u32(k0 > k1 ? k0 : k1);
    f_lo = __ballot(b0 && key[0] == kmax);
    f_hi = __ballot(b1 && key[1] == kmax);
    return ccsp_popc64(f_lo) + ccsp_popc64(f_hi);
}

// GreedyPlayer(stochastic=True).decide_move (player.py:77-97) on the move list wave_movegen left in LDS (K entries): the moves that
// go forward are drawn with probability dist / sum(dist) -- float64 priors, cumulative sums in list order, the first one whose
// cdf / cdf[last] exceeds u (np.random.choice(len, p=prior) through spec.sample_index) -- or, if none goes forward, one of all the
// moves uniformly (random.choice(backward_moves), player.py:92).  Returns the list position of the move.  The cumulative sums are a
// serial float64 chain in list order (lane 0, once per ply of such a seat).
__device__ __forceinline__ int wave_greedy_stochastic(Lds &lds, const ccsp_sr &st, int player, int K, uint64_t u64) {
    const int lane = lane_id();
    __syncthreads();
    int nf = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        int d = 0;
        if (j < K) {
            int id, dest;
            move_of(lds, j, id, dest);
            const int s = human_row(ccsp_sr_pos(st, (player - 1) * 6 + id)), e = human_row(dest);
            d = player == 1 ? s - e : e - s;                             // player.py:83-85
        }
        lds.gam[j] = d > 0 ? (double)d : 0.0;                            // forward moves carry their distance, the others 0
        nf += ccsp_popc64(__ballot(d > 0));
    }
    __syncthreads();
    if (nf == 0) return (int)ccsp_choice(u64, (uint32_t)K);              // every move is a "backward" move, in list order
    if (lane == 0) {
        double sum = 0.0;
        for (int j = 0; j < K; j++) sum = sum + lds.gam[j];              // integers: exact
        double last = 0.0;
        for (int j = 0; j < K; j++) if (lds.gam[j] > 0.0) last = last + lds.gam[j] / sum;
        const double u = (double)(u64 >> 11) * 1.1102230246251565e-16;
        double c = 0.0; int pick = -1, lastf = 0;
        for (int j = 0; j < K; j++) {
            if (!(lds.gam[j] > 0.0)) continue;
            lastf = j;
            c = c + lds.gam[j] / sum;
            if (c / last > u) { pick = j; break; }
        }
        lds.cnt[6] = (uint8_t)(pick >= 0 ? pick : lastf);
    }
    __syncthreads();
    return (int)lds.cnt[6];
}

// GreedyDataGenerator.generate_play after a greedy ply (data_generators.py:57-69)
__device__ __forceinline__ void slot_after_move_gen(const Params &P, Lds &lds, Slot &sl, int id, int dest, Tally &tl) {
    const ccsp_sr ns = ccsp_place(sl.st, (int)sl.player, id, dest);
    sl.st = ns;
    sl.ply += 1;
    tl.plies += 1;
    int status = CCSP_ST_RUNNING;
    const int w = ccsp_check_win(ns.occ0, ns.occ1);
    if (w) status = w;                                                       // 61-63
    else {
        sl.useless += 1;                                                     // plies without a winner: the clock of line 66
        if (sl.useless > P.stuck_limit) {                                    // 66-67: play_history[:AVERAGE_TOTAL_MOVE], draw
            status = CCSP_ST_DISCARD_NO_PROGRESS;
            if (sl.n_hist > CCSP_AVERAGE_TOTAL_MOVE) sl.n_hist = CCSP_AVERAGE_TOTAL_MOVE;
        } else if ((int)sl.ply >= P.max_plies) status = CCSP_ST_ERROR;
        else { sl.player = 3 - sl.player; sl.player_turn = 1 - sl.player_turn; }   // 69
    }
    sl.status = (uint32_t)status;
    if (status != CCSP_ST_RUNNING) slot_finish(P, lds, sl, status, tl);
}

// one GreedyPlayer ply: Game.start's seat (one draw among the filtered best moves, player.py:122) or a ply of
// the data generator (the sample row gets pi = 1/len on those moves first, data_generators.py:44-55)
__device__ __forceinline__ void wave_greedy_ply(const Params &P, Lds &lds, Slot &sl, Tally &tl) {
    const int lane = lane_id();
    const int K = wave_movegen(lds, sl.st, (int)sl.player);
    if (!P.gen && K > 0 && stochastic_to_move(P, sl)) {                 // Game.start's seat with GreedyPlayer(stochastic=True)
        const int j = wave_greedy_stochastic(lds, sl.st, (int)sl.player, K, ccsp_rng_from(sl.hgame, sl.ply, 0, 0, CCSP_P_GREEDY));
        int id, dest;
        move_of(lds, j, id, dest);
        slot_after_move_arena(P, lds, sl, id, dest, tl);
        return;
    }
    uint64_t f_lo, f_hi;
    const int cnt = K > 0 ? wave_greedy_best(lds, sl.st, (int)sl.player, K, f_lo, f_hi) : 0;
    if (cnt == 0) { sl.status = CCSP_ST_ERROR; tl.errors += 1; return; }
    if (P.gen) {
        unsigned long long row = 0;
        if (lane == 0) row = atomicAdd(P.log_count, 1ULL);
        row = uni64(row);
        if (row < P.log_cap) {
            if (lane == 0) {
                ccsp_store_sr(P.log_state + row, sl.st);
                *reinterpret_cast<ulonglong2 *>(P.log_meta + row) = make_ulonglong2(sl.game, (uint64_t)sl.ply | ((uint64_t)sl.player << 32));
            }
            double *dst = P.log_pi + row * CCSP_NUM_ACTIONS;
            for (int i = lane; i < CCSP_NUM_ACTIONS; i += 64) dst[i] = 0.0;
            __syncthreads();
            const double share = 1.0 / (double)cnt;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
                if (((h ? f_hi : f_lo) >> lane) & 1) { int id, dest; move_of(lds, j, id, dest); dst[id * CCSP_NCELL + dest] = share; }
            }
        }
        tl.samples += (row < P.log_cap) ? 1ULL : 0ULL;
        sl.n_hist += 1;
        if (row >= P.log_cap) { sl.status = CCSP_ST_ERROR; slot_finish(P, lds, sl, CCSP_ST_ERROR, tl); return; }   // log full: see wave_finish_ply
    }
    const int r = cnt > 1 ? (int)ccsp_choice(ccsp_rng_from(sl.hgame, sl.ply, 0, 0, CCSP_P_GREEDY), (uint32_t)cnt) : 0;
    const int j = nth_set_bit(f_lo, f_hi, r);
    int id, dest;
    move_of(lds, j, id, dest);
    if (P.gen) slot_after_move_gen(P, lds, sl, id, dest, tl);
    else slot_after_move_arena(P, lds, sl, id, dest, tl);
}

__device__ __forceinline__ void wave_opening_ply(const Params &P, Lds &lds, Slot &sl, Tally &tl) {
    const int K = wave_movegen(lds, sl.st, (int)sl.player);
    if (K == 0) { sl.status = CCSP_ST_ERROR; tl.errors += 1; return; }
    uint32_t counter = 0;
    int id;
    for (;;) {
        id = (int)ccsp_choice(ccsp_rng_from(sl.hgame, sl.ply, counter++, 0, CCSP_P_OPENING), 6);
        if (lds.cnt[id] > 0) break;
    }
    const int t = (int)ccsp_choice(ccsp_rng_from(sl.hgame, sl.ply, counter++, 0, CCSP_P_OPENING), lds.cnt[id]);
    const int dest = lds.lists[id][t];
    sl.opening_left -= 1;
    if (P.gen) {                                                     // data_generators.py:38-40: no winner check, no rules
        sl.st = ccsp_place(sl.st, (int)sl.player, id, dest);
        sl.ply += 1; tl.plies += 1;
        sl.player = 3 - sl.player; sl.player_turn = 1 - sl.player_turn;
        if ((int)sl.ply >= P.max_plies) { sl.status = CCSP_ST_ERROR; tl.errors += 1; }
        return;
    }
    slot_after_move(P, lds, sl, id, dest, tl);
}

// A game that was started in this slot DURING this call (auto-restart) plays its random opening plies at once -- they need
// no search (selfplay.py:32-33) -- so that the slot searches again in the very next ply instead of sitting out six calls
// (7 % of a slot's time at ~80 plies per game).  Per-game results are unchanged: draws are keyed by (game id, ply).
__device__ __forceinline__ void fast_forward_opening(const Params &P, Lds &lds, Slot &sl, uint64_t game_before, Tally &tl) {
    if (!P.auto_restart || P.arena || P.gen || sl.game == game_before) return;
    while (sl.status == CCSP_ST_RUNNING && sl.opening_left > 0) wave_opening_ply(P, lds, sl, tl);
}

// NumPy pairwise summation of 294 float64 (np.sum at MCTS.py:137) on 32 lanes: NumPy's order is 294 -> (72 + 72) + (72 + 78),
// each block = 8 strided accumulators (nine terms each) combined as
// ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), the last block's six left-over terms added to ITS result one after the other, then
// (b0 + b1) + (b2 + b3).  Lane 8 b + k runs accumulator k of block b; the pair sums go through lane exchanges in exactly that
// shape (IEEE addition commutes: which of a pair's lanes holds which term does not matter).  Same bits as the serial form, a ninth of
// its length -- at a ply boundary the wave sits beside an evaluator launch, where a dependent f64 addition takes five times as long.
__device__ __forceinline__ double wave_pairwise_294(const double *a) {
    const int lane = lane_id();
    const int blk = (lane >> 3) & 3, k = lane & 7;
    const double *base = a + 72 * blk;
    double r = base[k];
#pragma unroll
    for (int g = 1; g < 9; g++) r += base[8 * g + k];
    auto xchg = [](double x, int m) -> double {
        const uint64_t u = ccsp_to_bits(x);
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)u, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(u >> 32), m);
        return ccsp_from_bits(((uint64_t)hi << 32) | lo);
    };
    r = r + xchg(r, 1);                                   // r0 + r1, r2 + r3, r4 + r5, r6 + r7
    r = r + xchg(r, 2);                                   // (r0 + r1) + (r2 + r3), (r4 + r5) + (r6 + r7)
    r = r + xchg(r, 4);                                   // the block's sum, in all eight lanes of the block
    if (blk == 3) {                                       // 294 = 3 x 72 + 78: the last block's tail
#pragma unroll
        for (int i = 72; i < 78; i++) r += base[i];
    }
    r = r + xchg(r, 8);                                   // b0 + b1, b2 + b3
    r = r + xchg(r, 16);                                  // (b0 + b1) + (b2 + b3)
    return ccsp_from_bits(uni64(ccsp_to_bits(r)));        // lane 0's copy (lanes 32 .. 63 computed the same)
}

// T4 + end of make_move: pi from the root's visit counts, action sampling, sample-log row, Board.place
// chosen_child (optional): the child word of the root edge that was played (tree reuse: the next ply's root in THIS tree)
__device__ __forceinline__ void wave_finish_ply(const Params &P, Lds &lds, Slot &sl, uint8_t *pool, Tally &tl, uint32_t *chosen_child = nullptr) {
    const int lane = lane_id();
    const int K = (int)sl.root_k;
    uint8_t *b = pool;
    __syncthreads();
    for (int i = lane; i < CCSP_NUM_ACTIONS; i += 64) lds.pi[i] = 0.0;
    __syncthreads();
    uint32_t mvs[2] = {0xFFFFu, 0xFFFFu};
    bool overflow = false;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) {
            const uint32_t n = blk_N(b, K)[j];
            mvs[h] = blk_mv(b, K)[j];
            // pow(N, 1/tau) (MCTS.py:132): tau = 1 -> N exactly; tau = 0.01 -> host libm table of N**100
            const double pw = sl.det_tau ? P.pow_tab[n] : (double)n;
            overflow = overflow || (pw > 1.7976931348623157e308);
            lds.pi[mvs[h]] = pw;
            atomicAdd(&P.visit_hist[mvs[h]], (unsigned long long)n);
        }
    }
    // N**100 leaves float64 from N = 1210 on: Python's pow(int, float) raises OverflowError there (MCTS.py:132) and the
    // reference's worker dies with it.  Here the game ends with status ERROR (counted; its slot stays out of play) instead of
    // handing back pi = inf / inf.  Searches of up to 1209 simulations cannot get there.
    if (__ballot(overflow)) { sl.status = CCSP_ST_ERROR; slot_finish(P, lds, sl, CCSP_ST_ERROR, tl); return; }
    __syncthreads();
    const double s = wave_pairwise_294(lds.pi);                               // np.sum, MCTS.py:137
    for (int i = lane; i < CCSP_NUM_ACTIONS; i += 64) lds.pi[i] = lds.pi[i] / s;
    __syncthreads();
    // play_history.append((root.state, pi)) (selfplay.py:128) -> one row of the sample log
    unsigned long long row = 0;
    if (lane == 0) row = atomicAdd(P.log_count, 1ULL);
    row = uni64(row);
    if (row < P.log_cap) {
        if (lane == 0) {
            ccsp_store_sr(P.log_state + row, sl.st);
            ulonglong2 m = make_ulonglong2(sl.game, (uint64_t)sl.ply | ((uint64_t)sl.player << 32));
            *reinterpret_cast<ulonglong2 *>(P.log_meta + row) = m;
        }
        double *dst = P.log_pi + row * CCSP_NUM_ACTIONS;
        for (int i = lane; i < CCSP_NUM_ACTIONS; i += 64) dst[i] = lds.pi[i];
    }
    __syncthreads();
    // np.random.choice(294, p=pi) stand-in (spec.sample_index): cumsum, / last, first cdf > u.  pi is zero outside the root's K edges
    // and x + 0.0 = x: the running sum only moves at the edges' entries, and the first index that passes the test is one of them.
    // So: the K edges ranked by action index (a 294-bit mask in LDS, popcounts below each one's bit), their values laid out in that
    // order, ONE chain of K additions instead of 294 (lane 0), then every lane tests its own entries and a ballot finds the first
    // that passes -- the serial scan's pick, at an eighth of its length.
    uint32_t *mask = &lds.lines[0];                       // 10 words (the line patterns are not alive at a ply's end)
    uint16_t *sorted_a = reinterpret_cast<uint16_t *>(&lds.stack[0][0]);       // [K] action index by rank (the hop stacks are not alive either)
    double *sorted_c = &lds.gam[0];                       // [K] value, then running sum, by rank
    if (lane < 10) mask[lane] = 0u;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) if (lane + 64 * h < K) atomicOr(&mask[mvs[h] >> 5], 1u << (mvs[h] & 31));
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (lane + 64 * h < K) {
            const int w = (int)(mvs[h] >> 5);
            int r = __popc(mask[w] & ((1u << (mvs[h] & 31)) - 1u));
            for (int i = 0; i < w; i++) r += __popc(mask[i]);
            sorted_a[r] = (uint16_t)mvs[h];
            sorted_c[r] = lds.pi[mvs[h]];
        }
    }
    __syncthreads();
    if (lane == 0) {
        double c = 0.0;
        for (int r = 0; r < K; r++) { c = c + sorted_c[r]; sorted_c[r] = c; }
    }
    __syncthreads();
    int pick = CCSP_NUM_ACTIONS - 1;
    {
        const double u = (double)(ccsp_rng_from(sl.hgame, sl.ply, 0, 0, CCSP_P_SAMPLE) >> 11) * 1.1102230246251565e-16;
        const double last = sorted_c[K - 1];
        const uint64_t hit_lo = __ballot(lane < K && sorted_c[lane < K ? lane : 0] / last > u);
        const uint64_t hit_hi = __ballot(lane + 64 < K && sorted_c[lane + 64 < K ? lane + 64 : 0] / last > u);
        if (hit_lo) pick = (int)sorted_a[ccsp_ctz64(hit_lo)];
        else if (hit_hi) pick = (int)sorted_a[64 + ccsp_ctz64(hit_hi)];
    }
    const int cid = pick / CCSP_NCELL, cdest = pick % CCSP_NCELL;
    const uint32_t pick_idx = (uint32_t)pick;
    const uint64_t f_lo = __ballot(mvs[0] == pick_idx), f_hi = __ballot(mvs[1] == pick_idx);
    const bool found = (f_lo | f_hi) != 0;                                                        // MCTS.py:141-151
    if (chosen_child && found) *chosen_child = uni32(blk_child(b, K)[f_lo ? ccsp_ctz64(f_lo) : 64 + ccsp_ctz64(f_hi)]);
    tl.samples += (row < P.log_cap) ? 1ULL : 0ULL;
    sl.n_hist += 1;
    tl.mcts_plies += 1;
    // the sample log is full: the row is lost, so the game must not be handed back as if it were whole (its rows are
    // labelled by alternation from the first one, utils.py:64-72) -- it ends with status ERROR and counts as an error
    if (row >= P.log_cap) { sl.status = CCSP_ST_ERROR; slot_finish(P, lds, sl, CCSP_ST_ERROR, tl); return; }
    if (!found) { sl.status = CCSP_ST_ERROR; tl.errors += 1; return; }
    if (P.arena) slot_after_move_arena(P, lds, sl, cid, cdest, tl);
    else slot_after_move(P, lds, sl, cid, cdest, tl);
}

__device__ __forceinline__ void tally_add(const Params &P, int which, unsigned long long v) {
    if (v) atomicAdd(&P.counters[which], v);
}
__device__ __forceinline__ void tally_flush(const Params &P, const Tally &tl) {
    if (lane_id() == 0) {
        tally_add(P, CCSP_CNT_EXPANSIONS, tl.expansions); tally_add(P, CCSP_CNT_TERMINAL_SIMS, tl.terminal_sims);
        tally_add(P, CCSP_CNT_SIMS, tl.sims); tally_add(P, CCSP_CNT_PLIES, tl.plies);
        tally_add(P, CCSP_CNT_MCTS_PLIES, tl.mcts_plies); tally_add(P, CCSP_CNT_GAMES_WON, tl.games_won);
        tally_add(P, CCSP_CNT_GAMES_DISCARDED, tl.games_discarded); tally_add(P, CCSP_CNT_SUM_DEPTH, tl.sum_depth);
        tally_add(P, CCSP_CNT_SUM_CHILDREN, tl.sum_children); tally_add(P, CCSP_CNT_SELECT_EDGES, tl.select_edges);
        tally_add(P, CCSP_CNT_SAMPLES, tl.samples); tally_add(P, CCSP_CNT_ERRORS, tl.errors);
    }
}

__device__ __forceinline__ void tally_zero(Tally &tl) {
    tl.expansions = tl.terminal_sims = tl.sims = tl.plies = tl.mcts_plies = tl.games_won = tl.games_discarded = 0;
    tl.sum_depth = tl.sum_children = tl.select_edges = tl.samples = tl.errors = 0;
}

__device__ __forceinline__ Slot load_slot(const SlotMem *p) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
    const ulonglong2 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], w5 = q[5], w6 = q[6];
    Slot s;
    s.st.occ0 = uni64(w0.x); s.st.occ1 = uni64(w0.y); s.st.a = uni64(w1.x); s.st.b = uni64(w1.y);
    s.game = uni64(w2.x); s.hgame = uni64(w2.y); s.index = uni64(w3.x); s.expansions = uni64(w3.y);
    const uint64_t a = uni64(w4.x), b = uni64(w4.y), c = uni64(w5.x), d = uni64(w5.y);
    s.ply = (uint32_t)a; s.n_hist = (uint32_t)(a >> 32);
    s.useless = (int32_t)(uint32_t)b; s.pool_used = (uint32_t)(b >> 32);
    s.root_k = (uint32_t)c; s.sim = (uint32_t)(c >> 32);
    s.player = (uint32_t)(d & 0xFF); s.status = (uint32_t)((d >> 8) & 0xFF); s.det_tau = (uint32_t)((d >> 16) & 0xFF);
    s.n_hm = (uint32_t)((d >> 24) & 0xFF); s.progress0 = (uint32_t)((d >> 32) & 0xFF); s.progress1 = (uint32_t)((d >> 40) & 0xFF);
    s.player_turn = (uint32_t)((d >> 48) & 0xFF); s.opening_left = (uint32_t)((d >> 56) & 0xFF);
    s.hm0 = uni64(w6.x); s.hm1 = uni64(w6.y);
    return s;
}
__device__ __forceinline__ void store_slot(SlotMem *p, const Slot &s) {
    if (lane_id() == 0) {
        ulonglong2 *q = reinterpret_cast<ulonglong2 *>(p);
        q[0] = make_ulonglong2(s.st.occ0, s.st.occ1);
        q[1] = make_ulonglong2(s.st.a, s.st.b);
        q[2] = make_ulonglong2(s.game, s.hgame);
        q[3] = make_ulonglong2(s.index, s.expansions);
        q[4] = make_ulonglong2((uint64_t)s.ply | ((uint64_t)s.n_hist << 32), (uint64_t)(uint32_t)s.useless | ((uint64_t)s.pool_used << 32));
        const uint64_t d = (uint64_t)s.player | ((uint64_t)s.status << 8) | ((uint64_t)s.det_tau << 16) | ((uint64_t)s.n_hm << 24) |
                           ((uint64_t)s.progress0 << 32) | ((uint64_t)s.progress1 << 40) | ((uint64_t)s.player_turn << 48) |
                           ((uint64_t)s.opening_left << 56);
        q[5] = make_ulonglong2((uint64_t)s.root_k | ((uint64_t)s.sim << 32), d);
        q[6] = make_ulonglong2(s.hm0, s.hm1);
    }
}
// advance_kernel's forms: the record comes in through the SCALAR cache (two s_load_dwordx16: no vector loads, none of the 30
// v_readfirstlane a per-lane load of wave-uniform data needs -- beside an evaluator launch every vector instruction of a tree wave waits
// for a gap between MFMAs, scalar ones do not), and only the words a search changes go back.  The record was written by an EARLIER
// launch (the scalar cache is invalidated when a kernel starts); nothing in the calling kernel may have stored to it before.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t pair64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
__device__ __forceinline__ Slot load_slot_scalar(const SlotMem *p, uint64_t &w15) {
    u32x16 a, b;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
    Slot s;
    s.st.occ0 = pair64(a[0], a[1]); s.st.occ1 = pair64(a[2], a[3]); s.st.a = pair64(a[4], a[5]); s.st.b = pair64(a[6], a[7]);
    s.game = pair64(a[8], a[9]); s.hgame = pair64(a[10], a[11]); s.index = pair64(a[12], a[13]); s.expansions = pair64(a[14], a[15]);
    s.ply = b[0]; s.n_hist = b[1]; s.useless = (int32_t)b[2]; s.pool_used = b[3]; s.root_k = b[4]; s.sim = b[5];
    const uint64_t d = pair64(b[6], b[7]);
    s.player = (uint32_t)(d & 0xFF); s.status = (uint32_t)((d >> 8) & 0xFF); s.det_tau = (uint32_t)((d >> 16) & 0xFF);
    s.n_hm = (uint32_t)((d >> 24) & 0xFF); s.progress0 = (uint32_t)((d >> 32) & 0xFF); s.progress1 = (uint32_t)((d >> 40) & 0xFF);
    s.player_turn = (uint32_t)((d >> 48) & 0xFF); s.opening_left = (uint32_t)((d >> 56) & 0xFF);
    s.hm0 = pair64(b[8], b[9]); s.hm1 = pair64(b[10], b[11]);
    w15 = pair64(b[14], b[15]);
    return s;
}
__device__ __forceinline__ void store_slot_search(SlotMem *p, const Slot &s) {      // expansions, pool_used, sim: what a simulation changes
    if (lane_id() == 0) {
        p->w[7] = s.expansions;
        p->w[9] = (uint64_t)(uint32_t)s.useless | ((uint64_t)s.pool_used << 32);
        p->w[10] = (uint64_t)s.root_k | ((uint64_t)s.sim << 32);
    }
}
__device__ __forceinline__ Pending load_pending_scalar(const Pending *p) {
    u32x16 a;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(p) : "memory");
    Pending pd;
    pd.leaf.occ0 = pair64(a[0], a[1]); pd.leaf.occ1 = pair64(a[2], a[3]); pd.leaf.a = pair64(a[4], a[5]); pd.leaf.b = pair64(a[6], a[7]);
    pd.kind = a[8]; pd.depth = a[9]; pd.link_off = a[10]; pd.leaf_player = a[11];
    pd.k = a[12]; pd.walk_c = a[13]; pd.walk_at = a[14]; pd.walk_edges = a[15];
    return pd;
}
__device__ __forceinline__ Slot empty_slot() {
    Slot s;
    s.st.occ0 = s.st.occ1 = s.st.a = s.st.b = 0; s.game = s.hgame = s.index = s.expansions = 0;
    s.ply = s.n_hist = 0; s.useless = 0; s.pool_used = s.root_k = s.sim = 0;
    s.player = 1; s.status = CCSP_ST_IDLE; s.det_tau = s.n_hm = s.progress0 = s.progress1 = s.player_turn = s.opening_left = 0;
    s.hm0 = s.hm1 = 0;
    return s;
}

// ---- kernels ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void reset_kernel(Params P) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    Slot sl = empty_slot();
    if ((unsigned long long)g < P.max_games) slot_start_game(P, lds, sl, (unsigned long long)g);
    store_slot(P.slots + g, sl);
    if (lane_id() == 0) { P.pend[g].kind = 0; P.slots[g].w[14] = 0; P.slots[g].w[15] = 0; }
}

// ---- fused path (built-in evaluator): a ply = three phases -- begin, simulations, end -- which hand the game over through its slot
// record, so that the simulation loop carries only a compact context in registers (word 14 of the record = "searching" flag between
// the phases).  fused_plies_kernel runs them ply after ply in one launch; the three kernels launch one phase each.

// (1) opening move (selfplay.py:32-33), or root expansion + Dirichlet noise (selfplay.py:114-124)
__device__ __forceinline__ void fused_begin_core(const Params &P, Lds &lds, int g, Slot &sl, int evaluator) {   // lds.T is loaded
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    Tally tl; tally_zero(tl);
    uint32_t searching = 0;
    if (sl.opening_left > 0) wave_opening_ply(P, lds, sl, tl);
    else if (greedy_to_move(P, sl)) wave_greedy_ply(P, lds, sl, tl);
    else {
        EvalCtx ev; ev.kind = evaluator; ev.p_row = nullptr; ev.v_ext = 0.0f; ev.p_edges = nullptr; ev.shadow = 0;
        SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = 0; cx.player = sl.player; cx.pool_used = 0; cx.nsum_bias = (uint32_t)P.arena;
        const uint64_t rkey = evaluator == CCSP_EVAL_HASH ? ccsp_state_key(sl.st, (int)sl.player) : 0;
        // (the root's value is backed up along an empty path, selfplay.py:117: nothing to compute)
        if (P.arena && sl.ply > CCSP_TOTAL_MOVES_TILL_TAU0) sl.det_tau = 1;        // player.py:152-155
        uint32_t off;
        const int K = wave_expand(lds, cx, pool, sl.st, (int)sl.player, ev, rkey, !P.arena, off);   // arena: no Dirichlet noise
        sl.root_k = (uint32_t)K; sl.pool_used = cx.pool_used; sl.sim = 0;
        tl.expansions += 1; tl.sum_children += (unsigned long long)K; sl.expansions += 1;
        if (K == 0) { sl.status = CCSP_ST_ERROR; tl.errors += 1; }             // assert, selfplay.py:118
        else searching = 1;
    }
    store_slot(P.slots + g, sl);
    if (lane_id() == 0) P.slots[g].w[14] = searching;
    tally_flush(P, tl);
}

__global__ __launch_bounds__(64) void fused_begin_kernel(Params P, int evaluator) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING) return;
    load_engine_lines(&lds.T, lane_id());
    __syncthreads();
    fused_begin_core(P, lds, g, sl, evaluator);
}

// (2) the simulations (MCTS.py:123-125): select -> evaluate -> expand -> backup, `sims` times
__device__ __forceinline__ void fused_sims_core(const Params &P, Lds &lds, int g, int evaluator) {   // lds.T is loaded
    const int lane = lane_id();
    SlotMem *sm = P.slots + g;
    SimCtx cx;
    {
        const uint64_t w8 = uni64(sm->w[8]), w9 = uni64(sm->w[9]), w10 = uni64(sm->w[10]), w11 = uni64(sm->w[11]);
        cx.hgame = uni64(sm->w[5]); cx.ply = (uint32_t)w8; cx.pool_used = (uint32_t)(w9 >> 32);
        cx.root_k = (uint32_t)w10; cx.player = (uint32_t)(w11 & 0xFF); cx.nsum_bias = (uint32_t)P.arena;
    }
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    uint64_t *path = P.path + (uint64_t)g * P.path_stride;
    const double *sqrt_tab = P.sqrt_tab;
    const uint32_t sims = (uint32_t)P.sims;
    const bool use_rcp = P.sims + 2 <= RCP_N;              // every N and 1 + N of this ply indexes the LDS table
    for (int i = lane; i < RCP_N; i += 64) lds.rcp[i] = P.rcp_tab[i];
    __syncthreads();
    EvalCtx ev; ev.kind = evaluator; ev.p_row = nullptr; ev.v_ext = 0.0f; ev.p_edges = nullptr; ev.shadow = 0;
    uint32_t n_exp = 0, n_term = 0, sum_depth = 0, sum_children = 0, select_edges = 0;
#ifdef CCSP_STAMPS            // diagnostic build only (tools/stamps.py): cycles per phase, summed per wave
    unsigned long long t_sel = 0, t_exp = 0, t_bak = 0, t0, t1;
#define STAMP(x) do { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(x) do { } while (0)
#endif
    for (uint32_t sim = cx.nsum_bias; sim < sims; sim++) {       // arena: simulation 0 was the root expansion
        uint64_t mypath = 0; double myW = 0.0; uint32_t myN = 0;
#ifdef CCSP_STAMPS
        STAMP(t0);
#endif
        // issue priority: a wave in its expansion (a long dependent chain of LDS look-ups with few instructions) goes before the
        // waves that are selecting (wide, independent arithmetic between memory round trips): +6 % on config 2a, measured
        __builtin_amdgcn_s_setprio(0);
        const Leaf lf = use_rcp ? wave_select<true>(sqrt_tab, lds.rcp, cx, pool, path, sim, mypath, myW, myN, select_edges)
                                : wave_select<false>(sqrt_tab, nullptr, cx, pool, path, sim, mypath, myW, myN, select_edges);
        __builtin_amdgcn_s_setprio(2);
#ifdef CCSP_STAMPS
        STAMP(t1); t_sel += t1 - t0; t0 = t1;
#endif
        sum_depth += (uint32_t)lf.depth;
        float v = 0.0f;
        if (lf.kind == 1) {
            const uint64_t key = evaluator == CCSP_EVAL_HASH ? ccsp_state_key(lf.st, lf.player) : 0;
            v = value_of(ev, lds, lf.st, lf.player, key, cx.hgame, cx.ply, sim + 1);
            uint32_t noff;
            const int k = wave_expand(lds, cx, pool, lf.st, lf.player, ev, key, false, noff);
            if (k > 0 && lane == 0) *reinterpret_cast<uint32_t *>(pool + lf.link_off) = ((noff >> 3) << 7) | (uint32_t)k;
            n_exp += 1; sum_children += (uint32_t)k;
        } else n_term += 1;
#ifdef CCSP_STAMPS
        STAMP(t1); t_exp += t1 - t0; t0 = t1;
#endif
        wave_backup(pool, path, mypath, myW, myN, true, lf.depth, lf.kind == 2, v);
        __syncthreads();                                  // this simulation's stores before the next one's loads
#ifdef CCSP_STAMPS
        STAMP(t1); t_bak += t1 - t0;
#endif
    }
#ifdef CCSP_STAMPS
    if (lane == 0) { atomicAdd(&P.counters[12], t_sel); atomicAdd(&P.counters[13], t_exp); atomicAdd(&P.counters[14], t_bak); }
#endif
    if (lane == 0) {
        sm->w[9] = (sm->w[9] & 0xFFFFFFFFULL) | ((uint64_t)cx.pool_used << 32);
        sm->w[7] += n_exp;                                              // expansions spent on this game
        tally_add(P, CCSP_CNT_EXPANSIONS, n_exp); tally_add(P, CCSP_CNT_TERMINAL_SIMS, n_term);
        tally_add(P, CCSP_CNT_SIMS, sims - cx.nsum_bias); tally_add(P, CCSP_CNT_SUM_DEPTH, sum_depth);
        tally_add(P, CCSP_CNT_SUM_CHILDREN, sum_children); tally_add(P, CCSP_CNT_SELECT_EDGES, select_edges);
    }
}

__global__ __launch_bounds__(64, 4) void fused_sims_kernel(Params P, int evaluator) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    if (uni64(P.slots[g].w[14]) != 1) return;
    load_engine_lines(&lds.T, lane_id());
    __syncthreads();
    fused_sims_core(P, lds, g, evaluator);
}

// (3) pi, sampling, sample-log row, Board.place, end-of-ply rules (MCTS.py:127-153, selfplay.py:38-74)
__device__ __forceinline__ void fused_end_core(const Params &P, Lds &lds, int g, Slot &sl) {              // lds.T is loaded
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    Tally tl; tally_zero(tl);
    const uint64_t game0 = sl.game;
    wave_finish_ply(P, lds, sl, pool, tl);
    fast_forward_opening(P, lds, sl, game0, tl);
    store_slot(P.slots + g, sl);
    if (lane_id() == 0) P.slots[g].w[14] = 0;
    tally_flush(P, tl);
}

__global__ __launch_bounds__(64) void fused_end_kernel(Params P) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    if (uni64(P.slots[g].w[14]) != 1) return;
    Slot sl = load_slot(P.slots + g);
    load_engine_lines(&lds.T, lane_id());        // (the opening plies of a restarted game generate moves)
    __syncthreads();
    fused_end_core(P, lds, g, sl);
}

// The same three phases for `n_plies` plies of ONE game in ONE launch: a wave carries its game from ply to ply without waiting for
// the slowest game of every ply (a ply's launch lasts as long as its slowest search; the average wave is done a quarter earlier).
__global__ __launch_bounds__(64, 4) void fused_plies_kernel(Params P, int evaluator, int n_plies) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    load_engine_lines(&lds.T, lane_id());
    __syncthreads();
    for (int i = 0; i < n_plies; i++) {
        Slot sl = load_slot(P.slots + g);
        if (sl.status != CCSP_ST_RUNNING) return;
        fused_begin_core(P, lds, g, sl, evaluator);
        __syncthreads();                                   // this phase's stores before the next one's loads
        if (uni64(P.slots[g].w[14]) != 1) continue;
        fused_sims_core(P, lds, g, evaluator);
        __syncthreads();
        sl = load_slot(P.slots + g);
        fused_end_core(P, lds, g, sl);
        __syncthreads();
    }
}

// stepped path, phase 1: root planes out (or nothing for slots in their opening plies)
__global__ __launch_bounds__(64) void ply_begin_kernel(Params P, float *planes) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING || no_search(P, sl)) return;
    wave_encode(lds, sl.st, sl.player, planes + (uint64_t)g * CCSP_PLANES);
}

// stepped path, phase 2: root expansion with the evaluator's (p, v) + Dirichlet noise
__global__ __launch_bounds__(64) void root_expand_kernel(Params P, const double *p, const float *v) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING || no_search(P, sl)) return;
    load_engine_lines(&lds.T, lane_id());
    __syncthreads();
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    Tally tl; tally_zero(tl);
    EvalCtx ev; ev.kind = CCSP_EVAL_EXTERNAL; ev.p_row = p + (uint64_t)g * CCSP_NUM_ACTIONS; ev.v_ext = v[g]; ev.p_edges = nullptr; ev.shadow = 0;
    SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = 0; cx.player = sl.player; cx.pool_used = 0; cx.nsum_bias = (uint32_t)P.arena;
    sl.sim = (uint32_t)P.arena;                                 // arena: this expansion IS simulation 0
    if (P.arena && sl.ply > CCSP_TOTAL_MOVES_TILL_TAU0) sl.det_tau = 1;
    uint32_t off;
    const int K = wave_expand(lds, cx, pool, sl.st, (int)sl.player, ev, 0, !P.arena, off);
    sl.root_k = (uint32_t)K; sl.pool_used = cx.pool_used;
    tl.expansions += 1; tl.sum_children += (unsigned long long)K; sl.expansions += 1;
    if (K == 0) { sl.status = CCSP_ST_ERROR; tl.errors += 1; }
    store_slot(P.slots + g, sl);
    tally_flush(P, tl);
}

// stepped path, phase 3: selection; leaf planes out; hand-off record for expand_backup
__device__ __forceinline__ void select_core(const Params &P, Lds &lds, int g, float *planes) {
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING || no_search(P, sl) || sl.sim >= (uint32_t)P.sims) {
        if (lane_id() == 0) P.pend[g].kind = 0;
        return;
    }
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    uint64_t *path = P.path + (uint64_t)g * P.path_stride;
    uint64_t mypath = 0; double myW = 0.0; uint32_t myN = 0;
    SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = sl.root_k; cx.player = sl.player; cx.pool_used = sl.pool_used; cx.nsum_bias = (uint32_t)P.arena;
    uint32_t select_edges = 0;
    // (IEEE divisions here: the fused kernel's table of reciprocals costs a 3.4-KB LDS fill per launch in this short kernel, and read
    // straight from global memory -- wave_select<true, false>(P.sqrt_tab, P.rcp_tab, ...), round 3 -- it changes nothing measurable)
    const Leaf lf = wave_select<false>(P.sqrt_tab, nullptr, cx, pool, path, sl.sim, mypath, myW, myN, select_edges);
    if (lane_id() == 0) {
        uint32_t *a = P.stepacc + (size_t)g * 8;
        a[2] += 1u; a[3] += (uint32_t)lf.depth; a[5] += select_edges; a[1] += lf.kind == 1 ? 0u : 1u;
        ulonglong2 *q = reinterpret_cast<ulonglong2 *>(P.pend + g);
        q[0] = make_ulonglong2(lf.st.occ0, lf.st.occ1);
        q[1] = make_ulonglong2(lf.st.a, lf.st.b);
        q[2] = make_ulonglong2((uint64_t)(uint32_t)lf.kind | ((uint64_t)(uint32_t)lf.depth << 32),
                               (uint64_t)lf.link_off | ((uint64_t)(uint32_t)lf.player << 32));
    }
    if (lf.kind == 1) wave_encode(lds, lf.st, lf.player, planes + (uint64_t)g * CCSP_PLANES);
}

__global__ __launch_bounds__(64) void select_kernel(Params P, float *planes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_LIGHT];      // (no pi / gam / rcp / cells in these kernels)
    Lds &lds = *reinterpret_cast<Lds *>(lds_raw);
    __builtin_amdgcn_s_setprio(2);       // beside an evaluator launch (stepped path): the short tree kernels go first, the next evaluator launch waits for them
    select_core(P, lds, blockIdx.x, planes);
}

// stepped path, phase 4: expansion with (p, v) + backup
__device__ __forceinline__ void expand_backup_core(const Params &P, Lds &lds, int g, const double *p, const float *v) {
    Pending pd;
    {
        const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(P.pend + g);
        const ulonglong2 a = q[0], b = q[1], c = q[2];
        pd.leaf.occ0 = uni64(a.x); pd.leaf.occ1 = uni64(a.y); pd.leaf.a = uni64(b.x); pd.leaf.b = uni64(b.y);
        const uint64_t c0 = uni64(c.x), c1 = uni64(c.y);
        pd.kind = (uint32_t)c0; pd.depth = (uint32_t)(c0 >> 32); pd.link_off = (uint32_t)c1; pd.leaf_player = (uint32_t)(c1 >> 32);
    }
    if (pd.kind == 0) return;
    Slot sl = load_slot(P.slots + g);
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    const uint64_t *path = P.path + (uint64_t)g * P.path_stride;
    float val = 0.0f;
    if (pd.kind == 1) {
        load_engine_lines(&lds.T, lane_id());
        __syncthreads();
        EvalCtx ev; ev.kind = CCSP_EVAL_EXTERNAL; ev.p_row = p + (uint64_t)g * CCSP_NUM_ACTIONS; ev.v_ext = v[g]; ev.p_edges = nullptr; ev.shadow = 0;
        val = ev.v_ext;
        uint32_t noff;
        SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = sl.root_k; cx.player = sl.player; cx.pool_used = sl.pool_used; cx.nsum_bias = (uint32_t)P.arena;
        const int k = wave_expand(lds, cx, pool, pd.leaf, (int)pd.leaf_player, ev, 0, false, noff);
        sl.pool_used = cx.pool_used;
        if (k > 0 && lane_id() == 0) *reinterpret_cast<uint32_t *>(pool + pd.link_off) = ((noff >> 3) << 7) | (uint32_t)k;
        if (lane_id() == 0) { uint32_t *a = P.stepacc + (size_t)g * 8; a[0] += 1u; a[4] += (uint32_t)k; }
        sl.expansions += 1;
    }
    __syncthreads();
    const int lane = lane_id();
    const uint64_t mypath = (lane < (int)pd.depth) ? path[lane] : 0;
    wave_backup(pool, path, mypath, 0.0, 0u, false, (int)pd.depth, pd.kind == 2, val);
    sl.sim += 1;
    store_slot(P.slots + g, sl);
    if (lane_id() == 0) P.pend[g].kind = 0;            // consumed: a repeated call is a no-op
}

__global__ __launch_bounds__(64) void expand_backup_kernel(Params P, const double *p, const float *v) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_LIGHT];      // (no pi / gam / rcp / cells in these kernels)
    Lds &lds = *reinterpret_cast<Lds *>(lds_raw);
    __builtin_amdgcn_s_setprio(2);       // beside an evaluator launch (stepped path): the short tree kernels go first, the next evaluator launch waits for them
    expand_backup_core(P, lds, blockIdx.x, p, v);
}

// phases 4 and 3 of consecutive simulations in one launch: expansion + backup of simulation s, then the selection of s + 1 (the
// same wave, the same game) -- inside a ply the stepped loop is [evaluate -> this] instead of [evaluate -> expand_backup -> select]
__global__ __launch_bounds__(64) void expand_backup_select_kernel(Params P, const double *p, const float *v, float *planes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_LIGHT];      // (no pi / gam / rcp / cells in these kernels)
    Lds &lds = *reinterpret_cast<Lds *>(lds_raw);
    __builtin_amdgcn_s_setprio(2);
    expand_backup_core(P, lds, blockIdx.x, p, v);
    __syncthreads();                                   // the slot record, the pool and the hand-off record: stores before loads
    select_core(P, lds, blockIdx.x, planes);
}

// Tree reuse: the node block of a position the previous ply's tree holds (`ob`, at offset shadow << 3 of the other pool) re-created in
// the current tree -- what wave_expand writes for that position with the evaluator's answer, without generating the moves again:
// same position = same move list in the same order, same priors, same won-leaf marks; statistics start from zero (MCTS.py:97-109).
__device__ __forceinline__ int wave_copy_block(SimCtx &sl, uint8_t *pool, const uint8_t *ob, uint32_t shadow, uint32_t &off_out) {
    const int lane = lane_id_here();                    // (see lane_id_here: nothing of this block lives outside it)
    const uint4 h0 = *reinterpret_cast<const uint4 *>(ob), h1 = *reinterpret_cast<const uint4 *>(ob + 16), h2 = *reinterpret_cast<const uint4 *>(ob + 32);
    const int K = (int)uni32(h2.x);
    const uint32_t off = sl.pool_used;
    off_out = off;
    uint8_t *b = pool + off;
    sl.pool_used = off + block_bytes(K);
    if (lane == 0) {
        *reinterpret_cast<uint4 *>(b) = h0; *reinterpret_cast<uint4 *>(b + 16) = h1;
        *reinterpret_cast<uint4 *>(b + 32) = make_uint4(h2.x, h2.y, shadow, h2.w);          // K, player, shadow, v
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int j = lane + 64 * h;
        if (j < K) {
            st_blk(&blk_P(b, K)[j], ld_old(&reinterpret_cast<const double *>(ob + BLOCK_HDR)[j]));
            st_blk(&blk_W(b, K)[j], 0.0);
            st_blk(&blk_N(b, K)[j], 0u);
            st_blk(&blk_child(b, K)[j], ld_old(&reinterpret_cast<const uint32_t *>(ob + BLOCK_HDR + 20 * K)[j]) == CHILD_TERMINAL ? CHILD_TERMINAL : CHILD_LEAF);
            st_blk(&blk_mv(b, K)[j], ld_old(&reinterpret_cast<const uint16_t *>(ob + BLOCK_HDR + 24 * K)[j]));
        }
    }
    return K;
}

// ---- free-running stepped path (ccsp_advance): every slot at its own simulation of its own ply ---------------------------------
// One call = for every slot: take the answer (p, v) to the request it left last time (a leaf to expand, or a ply's root), then
// go on -- finish the ply when its simulations are done (pi, move, rules, log row), play opening plies, start the next ply, select --
// until the slot needs the evaluator again: planes out, request recorded, return.  What the lock-step kernels above do in five
// launches per ply and `sims` launches in between, with every slot waiting for the slowest, a slot does here at its own pace:
//   * a simulation that ends in a won leaf (MCTS.py:81-90) needs no evaluator: it is backed up and the next one starts at once;
//   * TREE REUSE: selfplay.make_move hands the chosen child back as a FRESH root (selfplay.py:130-133) and the next ply evaluates the
//     positions of its subtree all over again.  The evaluator is a function of the position alone (tests/test_model.py), so the
//     previous ply's tree (kept in the other pool) already holds the answer for every position the new search reaches inside the
//     played move's subtree: a block records the offset of the SAME position's block in the previous tree ("shadow", found through
//     its parent's: same path = same position, last two moves included), and a leaf whose shadow exists is expanded from that
//     block's P[K] and v -- the bits the net would return -- without leaving the kernel.  Trees, pi and games are unchanged
//     (tests: the free-running run against one lock-step slot per game); only the number of evaluator launches per ply falls.
// Two kernels per call, because what happens once per ply (Dirichlet noise, pi, sampling, the end-of-ply rules, opening plies: 80-130
// vector registers) must not set the register budget of what happens every simulation: the per-simulation kernel has to fit beside
// the evaluator's two waves per SIMD (352 of 512 registers).
//   boundary_kernel   slots at a ply boundary only (the others leave after reading 12 bytes): root expansion from the evaluator's answer,
//                     or: the finished ply's pi / move / rules / log row, opening plies of a new game, the next ply's root -- from the
//                     previous tree (reuse), else its planes go out as a request
//   advance_kernel    slots in a search: expansion + backup of the answered leaf, then selection -- and on through won leaves and
//                     reused positions -- until a leaf needs the evaluator (planes out) or the ply's simulations are done
// Slot word 15: bits 0-7 phase (0 = at a ply boundary, 1 = searching, 2 = simulations done), bit 8 = which pool holds the current tree,
// bits 32-63 = offset / 8, in the OTHER pool, of the block of the current ply's root position (0 = none).  Pending.kind: 0 no request,
// 1 a leaf, 3 the root.
// `budget`: simulations a slot may complete without the evaluator (won leaves, reused positions) in one launch; with the budget
// spent a slot selects once more and leaves its request if that leaf needs the evaluator -- otherwise it completes that one
// simulation too (the walk is done) and returns without a request (its row of the next evaluator launch is idle).  With `time_cap`
// and `deadline` (below) it bounds the launch's length.
// (bit 9: the start delay is spent; bits 10-11 `fin`: bit 11 = a free-running ply of this slot has been finished, bit 10 = the pool that
// holds ITS tree -- what ccsp_read_root / ccsp_debug_tree_digest read: with tree reuse it stays whole while the next ply is searched)
// bit 12: a root request of this slot is with the evaluator (the request buffer is the caller's: see boundary_kernel)
__device__ __forceinline__ void write_w15(const Params &P, int g, uint32_t phase, uint32_t half, uint32_t root_shadow, uint32_t fin, bool asked = false) {
    if (lane_id() == 0) P.slots[g].w[15] = (uint64_t)phase | ((uint64_t)half << 8) | (1ULL << 9) | ((uint64_t)fin << 10) | ((uint64_t)(asked ? 1 : 0) << 12) |
                                           ((uint64_t)root_shadow << 32);
}

#ifndef CCSP_ADVANCE_PRIO
#define CCSP_ADVANCE_PRIO 2
#endif
__global__ __launch_bounds__(64, 4) void boundary_kernel(Params P, const double *pk, const float *v, Pending *req, uint16_t *moves, uint8_t *model_sel, int flags, int stagger) {
    __shared__ Lds lds;
    const int g = blockIdx.x, lane = lane_id();
    const uint64_t w15 = uni64(P.slots[g].w[15]);
    uint32_t phase = (uint32_t)(w15 & 0xFF), half = (uint32_t)((w15 >> 8) & 1), root_shadow = (uint32_t)(w15 >> 32), fin = (uint32_t)((w15 >> 10) & 3);
    if (phase == 1) return;                               // in a search: advance_kernel's business
    // a root request of this slot is outstanding (bit 12 of word 15): only then does its record's kind mean anything -- ccsp_set_positions
    // starts the slots over without touching the records, and what a record holds from an earlier run is not to be taken up
    const bool asked = ((w15 >> 12) & 1) != 0;
    // THE HAND-OFF STATE IS THE ENGINE'S OWN (P.pend): kind, k, and in advance_kernel depth / link / the walk to resume are read from the
    // context's record, never from the caller's buffer -- `req` is an OUTPUT (position, kind, player, k: what an evaluator reads), so a
    // caller's buffer that was swapped, re-used, mis-sized or scribbled on between two calls cannot send a wave out of its tree pool
    const uint32_t kind = asked ? uni32(P.pend[g].kind) : 0u;
    // CCSP_ADVANCE_OVERLAPPED: this kernel runs BESIDE the evaluator launch that follows the call which wrote the request (the caller's
    // side stream), so that launch's answer is not to be trusted -- the next one's is: such a root request (kind 4) is two calls old
    // when its answer is taken.  In stream order (the default) the very next evaluator launch answers it (kind 3 at once).
    if (kind == 4) { if (lane == 0) { P.pend[g].kind = 3; req[g].kind = 3; } return; }
    // STAGGERED START: every slot's first game would begin in the same call and -- plies taking similar numbers of calls -- the slots
    // would end their plies in waves for dozens of plies: calls in which most slots are at the cheap middle of a search alternate with
    // calls in which most are at its expensive start (the reused top of the tree: simulation after simulation without the
    // evaluator).  A slot therefore sits out hash(id of its first game) mod `stagger` calls first (bits 16-31 of word 15 count them down; bit 9 = set
    // up).  A game's record does not depend on when it is played.
    if (phase == 0 && stagger > 1 && !((w15 >> 9) & 1)) {
        const uint32_t wait = (uint32_t)(ccsp_mix64(uni64(P.slots[g].w[4])) % (uint64_t)stagger);     // by GAME id: the same under any sharding
        if (lane == 0) P.slots[g].w[15] = (w15 & ~0xFFFF0000ULL) | (1ULL << 9) | ((uint64_t)wait << 16);
        if (wait) return;
    } else if (phase == 0 && ((w15 >> 16) & 0xFFFF)) {
        if (lane == 0) P.slots[g].w[15] = w15 - (1ULL << 16);
        return;
    }
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING) return;
    __builtin_amdgcn_s_setprio(CCSP_ADVANCE_PRIO);
    load_engine_lines(&lds.T, lane);
    __syncthreads();
    const bool reuse = (flags & CCSP_ADVANCE_REUSE) != 0;
    uint32_t *acc = P.stepacc + (size_t)g * 8;
    uint16_t *mv_row = moves + (size_t)g * REQ_MV;
    Tally tl; tally_zero(tl);
    uint32_t request = 0, req_k = 0;
    if (kind == 3) {                                      // the evaluator's answer for this ply's root (root_expand_kernel): selfplay.py:117-124
        const int K = (int)uni32(P.pend[g].k);
        uint8_t *pool = (half ? P.pool2 : P.pool) + (uint64_t)g * P.pool_stride;
        SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = 0; cx.player = sl.player; cx.pool_used = 0; cx.nsum_bias = 0;
        uint32_t off;
        wave_expand_answer<true>(&lds, cx, pool, sl.st, (int)sl.player, K, mv_row, pk + (size_t)g * REQ_MV, v[g], off);
        sl.root_k = (uint32_t)K; sl.pool_used = cx.pool_used; sl.sim = 0; sl.expansions += 1;
        if (K == 0) { sl.status = CCSP_ST_ERROR; tl.errors += 1; }             // assert, selfplay.py:118
        if (lane == 0) { acc[0] += 1u; acc[4] += (uint32_t)K; }
        phase = 1;
    } else {
        if (phase == 2) {                                 // the ply's search is done: pi, move, rules, log row (ply_end_kernel)
            // the sample log is shared by the slots and emptied by the host every few plies: a slot that could find it full waits
            // a call rather than lose its row (a game with a missing row ends in ERROR)
            if ((flags & CCSP_ADVANCE_LOG_GUARD) &&
                *reinterpret_cast<volatile unsigned long long *>(P.log_count) + (unsigned long long)P.n_slots > P.log_cap) {
                if ((flags & CCSP_ADVANCE_DEBUG) && lane == 0) atomicAdd(&P.counters[13], 1ULL);     // diagnostic: calls a slot waited for a log row
                return;
            }
            uint8_t *pool = (half ? P.pool2 : P.pool) + (uint64_t)g * P.pool_stride;
            const uint64_t game0 = sl.game;
            uint32_t cw = 0;
            wave_finish_ply(P, lds, sl, pool, tl, &cw);
            // the ply's tallies -> global counters
            tl.expansions += acc[0]; tl.terminal_sims += acc[1]; tl.sims += acc[2]; tl.sum_depth += acc[3];
            tl.sum_children += acc[4]; tl.select_edges += acc[5];
            if (lane == 0 && acc[6]) atomicAdd(&P.counters[CCSP_CNT_CACHE_HITS], (unsigned long long)acc[6]);
            __syncthreads();
            if (lane < 8) acc[lane] = 0u;
            __syncthreads();
            root_shadow = (reuse && sl.status == CCSP_ST_RUNNING && sl.game == game0 && cw != CHILD_LEAF && cw != CHILD_TERMINAL) ? (cw >> 7) : 0u;
            fin = 2u | half;                              // the finished ply's tree: in this pool
            if (reuse) half ^= 1u;                        // the tree just searched stays where it is; the next ply grows in the other pool
            phase = 0;
        }
        // ONE opening ply (selfplay.py:32-33) per call: what a slot does here between two evaluator launches stays short
        if (sl.status == CCSP_ST_RUNNING && sl.opening_left > 0) { wave_opening_ply(P, lds, sl, tl); root_shadow = 0; }
        if (sl.status == CCSP_ST_RUNNING && sl.opening_left == 0) {
            if (reuse && root_shadow != 0) {              // the root was a node of the previous ply's tree: its priors and value are there
                const uint8_t *ob = (half ? P.pool : P.pool2) + (uint64_t)g * P.pool_stride + ((uint64_t)root_shadow << 3);
                EvalCtx ev; ev.kind = CCSP_EVAL_CACHED; ev.p_row = nullptr; ev.p_edges = reinterpret_cast<const double *>(ob + BLOCK_HDR);
                ev.v_ext = reinterpret_cast<const float *>(ob + 32)[3]; ev.shadow = root_shadow;
                uint8_t *pool = (half ? P.pool2 : P.pool) + (uint64_t)g * P.pool_stride;
                SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = 0; cx.player = sl.player; cx.pool_used = 0; cx.nsum_bias = 0;
                uint32_t off;
                const int K = wave_expand(lds, cx, pool, sl.st, (int)sl.player, ev, 0, true, off);       // selfplay.py:117-124
                sl.root_k = (uint32_t)K; sl.pool_used = cx.pool_used; sl.sim = 0; sl.expansions += 1;
                if (K == 0) { sl.status = CCSP_ST_ERROR; tl.errors += 1; }
                if (lane == 0) { acc[0] += 1u; acc[4] += (uint32_t)K; acc[6] += 1u; }
                phase = 1;
            } else {                                      // ask the evaluator: the root's move list now, its block when the answer comes
                req_k = (uint32_t)wave_request_moves(lds, sl.st, (int)sl.player, mv_row);
                request = (flags & CCSP_ADVANCE_OVERLAPPED) ? 4u : 3u;
            }
        }
    }
    if (lane == 0) {
        ulonglong2 *q = reinterpret_cast<ulonglong2 *>(req + g), *own = reinterpret_cast<ulonglong2 *>(P.pend + g);
        if (request) {                                    // the caller's record: what an evaluator reads (position, kind, player, k)
            q[0] = make_ulonglong2(sl.st.occ0, sl.st.occ1);
            q[1] = make_ulonglong2(sl.st.a, sl.st.b);
        }
        q[2] = make_ulonglong2((uint64_t)request, (uint64_t)sl.player << 32);
        q[3] = make_ulonglong2((uint64_t)req_k, 0ULL);
        own[2] = make_ulonglong2((uint64_t)request, (uint64_t)~0u | ((uint64_t)sl.player << 32));      // the engine's own: kind, k (a root hangs nowhere,
        own[3] = make_ulonglong2((uint64_t)req_k, 0ULL);                                                 // no walk to resume; its position is the slot's)
        if (request && model_sel) model_sel[g] = (uint8_t)(sl.player == 2 ? 1 : 0);    // whose model answers (selfplay.py:30,36,59)
    }
    write_w15(P, g, phase, half, root_shadow, fin, request != 0);
    store_slot(P.slots + g, sl);
    tally_flush(P, tl);
}

// Occupancy hint = VGPR budget (512 / waves, granule 8).  Two tree waves per SIMD run beside the evaluator's two: while net_forward_kernel<8,8>
// took 174 (176) VGPRs that left 2 x 80 (hint 6: one 8-byte spill per simulation); at its 158 (160) of round 5 there are 2 x 96 -- hint 5: 90
// VGPRs, no scratch (A/B on one box: 22.91 -> 22.97-23.06 M node-expansions/s).  ccsp_net.hip asserts nothing about this: check
// tools/isa_resources.py when either kernel's registers change.
#ifndef CCSP_ADVANCE_WAVES
#define CCSP_ADVANCE_WAVES 5
#endif
struct AdvArgs {                  // advance_kernel's arguments (see the kernel's first lines)
    Params P;
    const double *pk; const float *v; Pending *req; uint16_t *moves; uint8_t *model_sel;
    int flags, budget, time_cap, deadline;
};
template <bool DBG>       // DBG: the per-phase cycle stamps (CCSP_ADVANCE_DEBUG) -- a build of their own: their ten 64-bit sums cost the plain kernel twenty scalar registers
__global__ __launch_bounds__(64, CCSP_ADVANCE_WAVES) void advance_kernel(AdvArgs args_in_kernarg) {
    // The arguments are read where they are used, through the kernel-argument segment (scalar loads), instead of by name: named, all of them
    // are loaded in the first block and what the selection loop does not need is parked in vector-register lanes for the whole call
    // (v_writelane / v_readlane: vector instructions, which is what a tree wave must not spend beside an evaluator launch).
    const AdvArgs &A = *(const AdvArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const Params &P = A.P;
    static_assert(offsetof(AdvArgs, P) == 0 && offsetof(Params, slots) == 0, "the parameter block comes first");
    const int flags = A.flags, budget = A.budget, time_cap = A.time_cap, deadline = A.deadline;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_LIGHT];      // the struct without its last member (pi / gam / rcp: never touched here)
    Lds &lds = *reinterpret_cast<Lds *>(lds_raw);
    const int g = blockIdx.x, lane = lane_id();
    uint64_t w15;
    Slot sl = load_slot_scalar(P.slots + g, w15);
    uint32_t phase = (uint32_t)(w15 & 0xFF);
    const uint32_t half = (uint32_t)((w15 >> 8) & 1), root_shadow = (uint32_t)(w15 >> 32), fin = (uint32_t)((w15 >> 10) & 3);
    if (phase != 1) return;                               // at a ply boundary: boundary_kernel's business
    if (sl.status != CCSP_ST_RUNNING) return;
    __builtin_amdgcn_s_setprio(CCSP_ADVANCE_PRIO);       // beside an evaluator launch: the short tree kernels go first
    const bool reuse = (flags & CCSP_ADVANCE_REUSE) != 0;
    uint8_t *pool = (half ? P.pool2 : P.pool) + (uint64_t)g * P.pool_stride;
    const uint8_t *old = (half ? P.pool : P.pool2) + (uint64_t)g * P.pool_stride;
    uint64_t *path = P.path + (uint64_t)g * P.path_stride;
        uint32_t a_exp = 0, a_term = 0, a_sims = 0, a_depth = 0, a_children = 0, a_edges = 0, a_hits = 0, errors = 0;
    // the hand-off state -- the leaf asked about, where its block hangs, the path length, a walk to resume -- is the ENGINE'S OWN record
    // (P.pend, written by this context's kernels in an earlier launch): nothing the caller's request buffer holds is used as an address
    Pending pd = load_pending_scalar(P.pend + g);
    const bool answered = pd.kind == 1;                   // the evaluator's answer for the leaf this slot asked about last time
    // a selection this slot gave up at the deadline of its last call: it goes on where it stopped (WalkFrom)
    WalkFrom walk; walk.c = pd.walk_c; walk.nsum = pd.walk_at >> 16; walk.level = answered ? 0 : (int)(pd.walk_at & 0xFFFFu);
    uint32_t walk_edges = walk.level != 0 ? pd.walk_edges : 0u;
    // `deadline` (10-ns ticks since the wave began; 0 = none): a wave that has done other work in this call -- the answered leaf's expansion,
    // evaluator-free simulations -- and is later than that gives its selection up (before it, or between two levels of it) and leaves no
    // request: one idle evaluator row instead of a launch that waits for its last wave.  A call that BEGINS with the selection never gives up,
    // so the slot moves on in its next call whatever the deadline.
    const bool began_with_work = answered;
    uint32_t request = 0, req_k = 0;
    uint32_t left_c = 0, left_at = 0, left_edges = 0;     // a walk given up in THIS call
    int spent = 0;
    // diagnostic (CCSP_ADVANCE_DEBUG): cycles of this wave per phase -- [0] set-up, [1] expansion, [2] backup, [3] selection + shadow,
    // [4] move list + hand-off, [5] whole call, [6] calls, [7] expansions, [8] selections -- summed over the waves into P.dbg
    constexpr bool dbg = DBG;
    unsigned long long t_exp = 0, t_bak = 0, t_sel = 0, t_enc = 0, n_exp = 0, n_sel = 0, tq = 0;
    const unsigned long long t_begin = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long r_begin = (dbg || time_cap > 0 || deadline > 0) ? __builtin_amdgcn_s_memrealtime() : 0;      // the constant 100 MHz clock (diagnostic: what a tick of s_memtime is worth)
    unsigned long long t_mark = t_begin;
#ifndef CCSP_ADVANCE_LAP_WAITS
#define CCSP_ADVANCE_LAP_WAITS 0      // 1: a phase ends when its memory operations have completed (serialises the wave: 2.4 x slower); 0: when its last instruction has issued
#endif
#define ADV_LAP(acc) do { if (dbg) { if (CCSP_ADVANCE_LAP_WAITS) __builtin_amdgcn_s_waitcnt(0); tq = __builtin_amdgcn_s_memtime(); acc += tq - t_mark; t_mark = tq; } } while (0)
    unsigned long long t_setup = 0;
    ADV_LAP(t_setup);
    // (1) the leaf this slot asked about last time: its block from the request's move row and the compact answer (no move generation, no
    // LDS: the list was made when the request was) + backup -- once per call and AHEAD of the loop
    if (answered) {
        const ccsp_sr leaf = pd.leaf; const int leaf_player = (int)pd.leaf_player, depth = (int)pd.depth; const uint32_t link_off = pd.link_off;
        const float val = A.v[g];
        const int lane1 = lane_id_here();                 // (this phase's lane-derived values die with it)
        const uint64_t mypath = (lane1 < depth) ? path[lane1] : 0;
        SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = sl.root_k; cx.player = sl.player; cx.pool_used = sl.pool_used; cx.nsum_bias = 0;
        uint32_t noff;
        const int k = wave_expand_answer<false, true>(nullptr, cx, pool, leaf, leaf_player, (int)pd.k, A.moves + (size_t)g * REQ_MV, A.pk + (size_t)g * REQ_MV, val, noff);
        sl.pool_used = cx.pool_used;
        if (k > 0 && lane1 == 0) *reinterpret_cast<uint32_t *>(pool + link_off) = ((noff >> 3) << 7) | (uint32_t)k;
        a_exp += 1; a_children += (uint32_t)k; sl.expansions += 1;
        __syncthreads();
        n_exp += 1;
        ADV_LAP(t_exp);
        wave_backup<true>(pool, path, mypath, 0.0, 0u, false, depth, false, val);
        sl.sim += 1;
        __syncthreads();                                  // this simulation's stores before the next one's loads
        ADV_LAP(t_bak);
    }
    // (2) selection -- and on through won leaves and reused positions -- until a leaf needs the evaluator or the call's budget is spent
    for (;;) {
        ccsp_sr leaf; int leaf_player, depth; uint32_t link_off;
        uint64_t mypath = 0; double myW = 0.0; uint32_t myN = 0;
        bool terminal = false, have_stats = false, stop_after = false;
        EvalCtx ev; ev.kind = CCSP_EVAL_CACHED; ev.p_row = nullptr; ev.v_ext = 0.0f; ev.p_edges = nullptr; ev.shadow = 0;
        if (sl.sim >= (uint32_t)P.sims) { phase = 2; break; }      // the search is done: boundary_kernel ends the ply in the next call
        // budget spent: one more selection -- its leaf's request goes out if it asks the evaluator; a won or reused leaf is still backed up
        // (the walk to it is the expensive part and is done), and the call ends there.  The budget is a number of simulations AND, past
        // the first one, a time (`time_cap`, 10-ns ticks since the wave began): a launch lasts as long as its slowest wave, and the
        // waves that go on through reused positions are the slowest -- a wave that has already been running for longer than the
        // usual one stops taking them up, a fast one may take up more.  Results do not depend on either.
        const bool last = spent >= budget ||
                          (time_cap > 0 && spent > 0 && (long long)(__builtin_amdgcn_s_memrealtime() - r_begin) > (long long)time_cap);
        // (the low 32 bits of the clock, compared through a signed difference; | 1: zero means "never")
        const uint32_t give_up_at = (CCSP_ADVANCE_DEADLINE_CODE && deadline > 0 && (began_with_work || spent > 0)) ? (((uint32_t)r_begin + (uint32_t)deadline) | 1u) : 0u;
        if (give_up_at != 0 && (int32_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - give_up_at) > 0) break;
        SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = sl.root_k; cx.player = sl.player; cx.pool_used = sl.pool_used; cx.nsum_bias = 0;
        uint32_t edges = 0;
#ifndef CCSP_ADVANCE_RCP
#define CCSP_ADVANCE_RCP 1
#endif
        // the two IEEE divisions per edge and level through the table of reciprocals (read from global memory here: 3.4 KB, cached --
        // the LDS copy the fused kernel keeps would cost this kernel four of its workgroups per CU): 28 vector instructions fewer
        // per edge; beside the evaluator every vector instruction of a tree wave waits for a gap between two MFMAs
        const Leaf lf = reuse ? wave_select<CCSP_ADVANCE_RCP != 0, false, true>(P.sqrt_tab, P.rcp_tab, cx, pool, path, sl.sim, mypath, myW, myN, edges, give_up_at, walk)
                              : wave_select<CCSP_ADVANCE_RCP != 0, false, false>(P.sqrt_tab, P.rcp_tab, cx, pool, path, sl.sim, mypath, myW, myN, edges, 0, walk);
        edges += walk_edges;                          // (a resumed walk: the edges its first part scanned)
        if (walk.level != 0 && lf.kind != 0) walk_rejoin(pool, path, walk.level, mypath, myW, myN);
        walk.level = 0; walk_edges = 0;
        have_stats = true;
        if (lf.kind == 0) {                           // given up between two levels: nothing was changed; the next call goes on from there
            left_c = lf.next_c; left_at = (uint32_t)lf.depth | (lf.next_nsum << 16); left_edges = edges;
            break;
        }
        a_sims += 1; a_depth += (uint32_t)lf.depth; a_edges += edges;
        leaf = lf.st; leaf_player = lf.player; depth = lf.depth; link_off = lf.link_off;
        n_sel += 1;
        if (lf.kind == 2) {                           // a won leaf: backed up at once (MCTS.py:81-90)
            stop_after = last;                        // budget spent: this simulation is still completed (its selection is done), nothing after it
            terminal = true; a_term += 1;
        } else {
            uint32_t shadow = 0;
            if (reuse && lf.parent_shadow != 0) {     // the leaf's position in the previous ply's tree, through its parent's block there
                const uint8_t *opb = old + ((uint64_t)lf.parent_shadow << 3);
                const uint32_t cw = uni32(*reinterpret_cast<const uint32_t *>(opb + BLOCK_HDR + 20 * lf.parent_k + 4 * lf.sel));
                if (cw != CHILD_LEAF && cw != CHILD_TERMINAL) shadow = cw >> 7;
            }
            if (shadow == 0) {                        // the evaluator is needed: the request is made behind the loop
                // THE REQUEST, made here and not behind the loop (the leaf's position would have to live in eight vector registers across the
                // loop's exit): the leaf's legal moves -- the line tables are the move generator's alone: loaded now, 21 cache lines, and only
                // by a call that asks -- then the record the evaluator reads.  Every lane-derived value of this block comes from
                // lane_id_here(): nothing of it can be hoisted above the loop.
                ADV_LAP(t_sel);
                load_engine_lines(&lds.T, lane_id_here());
                __syncthreads();
#ifdef CCSP_EXP_MOVEGEN_TWICE       // experiment (round 6): what the request's move list costs the pipeline -- generated twice, results unchanged
                (void)wave_request_moves<true>(lds, leaf, leaf_player, A.moves + (size_t)g * REQ_MV);
                __syncthreads();
#endif
                req_k = (uint32_t)wave_request_moves<true>(lds, leaf, leaf_player, A.moves + (size_t)g * REQ_MV);
                if (lane_id_here() == 0) {
                    ulonglong2 *q = reinterpret_cast<ulonglong2 *>(A.req + g), *own = reinterpret_cast<ulonglong2 *>(P.pend + g);
                    const ulonglong2 s0 = make_ulonglong2(leaf.occ0, leaf.occ1), s1 = make_ulonglong2(leaf.a, leaf.b);
                    // the caller's record: position, kind, player, k -- what an evaluator reads; the engine's own: the same + depth and link
                    q[0] = s0; q[1] = s1;
                    q[2] = make_ulonglong2(1ULL, (uint64_t)(uint32_t)leaf_player << 32);
                    q[3] = make_ulonglong2((uint64_t)req_k, 0ULL);
                    own[0] = s0; own[1] = s1;
                    own[2] = make_ulonglong2(1ULL | ((uint64_t)(uint32_t)depth << 32), (uint64_t)link_off | ((uint64_t)(uint32_t)leaf_player << 32));
                    own[3] = make_ulonglong2((uint64_t)req_k, 0ULL);
                }
                ADV_LAP(t_enc);
                request = 1;
                break;
            }
            stop_after = last;                        // (no request: this slot's row of the next evaluator launch is idle)
            ev.kind = CCSP_EVAL_CACHED; ev.v_ext = reinterpret_cast<const float *>(old + ((uint64_t)shadow << 3) + 32)[3]; ev.shadow = shadow;
            a_hits += 1;
        }
        spent += 1;
        ADV_LAP(t_sel);
        if (!terminal) {                                  // a reused position: its block of the previous ply's tree, re-created in this one
            SimCtx cx; cx.hgame = sl.hgame; cx.ply = sl.ply; cx.root_k = sl.root_k; cx.player = sl.player; cx.pool_used = sl.pool_used; cx.nsum_bias = 0;
            uint32_t noff;
            const int k = wave_copy_block(cx, pool, old + ((uint64_t)ev.shadow << 3), ev.shadow, noff);
            sl.pool_used = cx.pool_used;
            if (k > 0 && lane == 0) *reinterpret_cast<uint32_t *>(pool + link_off) = ((noff >> 3) << 7) | (uint32_t)k;
            a_exp += 1; a_children += (uint32_t)k; sl.expansions += 1;
            __syncthreads();
            n_exp += 1;
            ADV_LAP(t_exp);
        }
        wave_backup<true>(pool, path, mypath, myW, myN, have_stats, depth, terminal, ev.v_ext);
        sl.sim += 1;
        __syncthreads();                                  // this simulation's stores before the next one's loads
        ADV_LAP(t_bak);
        if (stop_after) break;
    }
    if (dbg && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        // per slot, plain read-modify-writes (ten atomics per wave on one cache line made the launch three times as long)
        unsigned long long *d = P.dbg + (size_t)g * CCSP_DBG_STRIDE;
        d[0] += t_setup; d[1] += t_exp; d[2] += t_bak; d[3] += t_sel; d[4] += t_enc; d[5] += t_end - t_begin; d[6] += 1ULL; d[7] += n_exp;
        d[8] += n_sel; if (t_end - t_begin > d[9]) d[9] = t_end - t_begin;
        d[10] += __builtin_amdgcn_s_memrealtime() - r_begin;
        const int sb = spent < 3 ? spent : 3;           // [11 .. 14] time of the calls that completed 0 / 1 / 2 / 3+ evaluator-free simulations, [15] their numbers (4 x 16 bits)
        d[11 + sb] += t_end - t_begin; d[15] += 1ULL << (16 * sb);
        const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();       // [16 .. 19]: this slot's LAST call -- begin, end (100 MHz clock, the same on every CU), simulations, levels
        d[16] = r_begin; d[17] = r_end; d[18] = (unsigned long long)spent | ((unsigned long long)request << 32); d[19] = a_depth;
    }
    if (lane == 0) {
        ulonglong2 *q = reinterpret_cast<ulonglong2 *>(A.req + g), *own = reinterpret_cast<ulonglong2 *>(P.pend + g);
        if (request != 1) {                               // nothing asked: the search is done, the budget is spent, or a walk waits to be resumed
            q[2] = make_ulonglong2(0ULL, 0ULL);           // (the caller's record: kind 0 = an idle evaluator row)
            own[2] = make_ulonglong2(0ULL, 0ULL);
            own[3] = make_ulonglong2((uint64_t)left_c << 32, (uint64_t)left_at | ((uint64_t)left_edges << 32));
        }
        if (request && A.model_sel) A.model_sel[g] = (uint8_t)(sl.player == 2 ? 1 : 0);    // whose model answers (selfplay.py:30,36,59)
        uint32_t *acc = P.stepacc + (size_t)g * 8;
        acc[0] += a_exp; acc[1] += a_term; acc[2] += a_sims; acc[3] += a_depth; acc[4] += a_children; acc[5] += a_edges; acc[6] += a_hits;
        if (errors) atomicAdd(&P.counters[CCSP_CNT_ERRORS], (unsigned long long)errors);
        if (flags & CCSP_ADVANCE_DEBUG) {                 // diagnostic tallies (tools/bench_free.py --debug): what the slots of this call ended on
            atomicAdd(&P.counters[12], (unsigned long long)(request == 1));                      // ... a request
            atomicAdd(&P.counters[14], (unsigned long long)(request == 0 && phase == 1));        // ... the budget / the deadline (idle evaluator row)
        }
    }
    write_w15(P, g, phase, half, root_shadow, fin);
    store_slot_search(P.slots + g, sl);
}

// stepped path, phase 5: pi, sampling, move, rules (or the random opening move)
__global__ __launch_bounds__(64) void ply_end_kernel(Params P) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING) return;
    load_engine_lines(&lds.T, lane_id());
    __syncthreads();
    uint8_t *pool = P.pool + (uint64_t)g * P.pool_stride;
    Tally tl; tally_zero(tl);
    const uint64_t game0 = sl.game;
    if (sl.opening_left > 0) wave_opening_ply(P, lds, sl, tl);
    else if (greedy_to_move(P, sl)) wave_greedy_ply(P, lds, sl, tl);
    else wave_finish_ply(P, lds, sl, pool, tl);
    fast_forward_opening(P, lds, sl, game0, tl);
    store_slot(P.slots + g, sl);
    {   // the ply's per-slot tallies of the stepped kernels -> global counters
        uint32_t *a = P.stepacc + (size_t)g * 8;
        tl.expansions += a[0]; tl.terminal_sims += a[1]; tl.sims += a[2]; tl.sum_depth += a[3];
        tl.sum_children += a[4]; tl.select_edges += a[5];
        __syncthreads();
        if (lane_id() < 8) a[lane_id()] = 0u;
    }
    tally_flush(P, tl);
}

// next-4: `n_plies` plies of GreedyDataGenerator.generate_play per slot in one launch (no search, no evaluator)
__global__ __launch_bounds__(64) void greedy_plies_kernel(Params P, int n_plies) {
    __shared__ Lds lds;
    const int g = blockIdx.x;
    Slot sl = load_slot(P.slots + g);
    if (sl.status != CCSP_ST_RUNNING) return;
    load_engine_lines(&lds.T, lane_id());
    __syncthreads();
    Tally tl; tally_zero(tl);
    for (int i = 0; i < n_plies && sl.status == CCSP_ST_RUNNING; i++) {
        if (sl.opening_left > 0) wave_opening_ply(P, lds, sl, tl);
        else wave_greedy_ply(P, lds, sl, tl);
    }
    store_slot(P.slots + g, sl);
    tally_flush(P, tl);
}

__global__ __launch_bounds__(64) void set_positions_kernel(Params P, const ccsp_state *states, const uint8_t *player,
                                                           const uint64_t *game, const uint32_t *ply, const uint8_t *det_tau) {
    const int g = blockIdx.x;
    Slot sl = empty_slot();
    sl.st = ccsp_load_sr(states + g);
    sl.game = game[g]; sl.hgame = ccsp_rng_game(P.seed, game[g]);
    sl.index = ~0ULL;                                   // not part of the result table
    sl.ply = ply[g]; sl.player = player[g]; sl.det_tau = det_tau[g];
    sl.n_hist = sl.ply >= CCSP_INITIAL_RANDOM_MOVES ? sl.ply - CCSP_INITIAL_RANDOM_MOVES : 0;
    sl.opening_left = 0;                                // positions set this way are always searched
    sl.player_turn = (uint32_t)(player[g] - 1);
    sl.status = CCSP_ST_RUNNING;
    store_slot(P.slots + g, sl);
    if (lane_id() == 0) { P.slots[g].w[14] = 0; P.slots[g].w[15] = 0; }
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------

// ccsp_advance's three limits (see ccsp_set_advance_limits below): the process-wide defaults a context starts with
static int g_advance_budget = 8;
#ifndef CCSP_ADVANCE_TIME_CAP
#define CCSP_ADVANCE_TIME_CAP 5000
#define CCSP_ADVANCE_DEADLINE 8000
#endif
static int g_advance_time_cap = CCSP_ADVANCE_TIME_CAP;    // 10-ns ticks; 0 = none
static int g_advance_deadline = CCSP_ADVANCE_DEADLINE;    // 10-ns ticks; 0 = none

struct ccsp_ctx {
    ccsp_config cfg;
    Params P;
    void *sqrt_tab, *pow_tab, *rcp_tab;
    uint64_t pool_bytes, path_bytes;
    int phase;                     // stepped path sequencing: 0 idle, 1 begun, 2 root expanded, 3 selected
    int stagger_span;              // CCSP_ADVANCE_STAGGER: ccsp_boundary calls over which the slots' first games begin (0 = cfg.sims)
    int adv_budget, adv_time_cap, adv_deadline;   // ccsp_advance's limits (ccsp_set_advance_limits; from the process defaults at ccsp_create)
    int opening_plies;             // fused plies played since ccsp_reset while EVERY slot is still in its random opening
                                   // (all games start together); -1 once that is no longer known
};

#define CTXCHK(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) { ccsp_set_hip_error(e_, #expr); if (err) *err = CCSP_EHIP; ccsp_destroy(ctx); return nullptr; } \
    } while (0)

#define CTXALLOC(ptr, bytes)                                                  \
    do {                                                                      \
        const int rc_ = ccsp_alloc_status(hipMalloc((void **)(ptr), (bytes)), "hipMalloc(" #ptr ")"); \
        if (rc_ != CCSP_OK) { if (err) *err = rc_; ccsp_destroy(ctx); return nullptr; } \
    } while (0)

// Every entry point runs on the context's own device, whatever the caller's current device is, and refuses a
// stream that belongs to another device (a launch there would fail with an invalid resource handle).
// The caller's current device is put back when the entry point returns (a process that holds several devices keeps its
// torch.cuda.current_device() across eng.counters() / close()).
struct ctx_scope {
    int prev = -1, rc = CCSP_OK;
    ctx_scope(const ccsp_ctx *ctx, void *stream) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != ctx->cfg.device && hipSetDevice(ctx->cfg.device) != hipSuccess) { rc = CCSP_EHIP; (void)hipGetLastError(); return; }
        if (stream) {
            hipDevice_t sd = -1;
            if (hipStreamGetDevice((hipStream_t)stream, &sd) != hipSuccess) { rc = CCSP_EINVAL; (void)hipGetLastError(); }
            else if ((int)sd != ctx->cfg.device) rc = CCSP_EINVAL;
        }
    }
    ctx_scope(const ctx_scope &) = delete;
    ctx_scope &operator=(const ctx_scope &) = delete;
    ~ctx_scope() { if (prev >= 0 && prev != -1) { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev); } }
};
#define CTX_ENTER(ctx, stream) ctx_scope scope_((ctx), (stream)); if (scope_.rc != CCSP_OK) return scope_.rc

extern "C" {

int ccsp_destroy(ccsp_ctx *ctx) {
    if (!ctx) return CCSP_OK;
    ctx_scope scope_(ctx, nullptr);
    Params &P = ctx->P;
    void *ptrs[] = {P.slots, P.pend, P.pool, P.pool2, P.dbg, P.path, ctx->sqrt_tab, ctx->pow_tab, ctx->rcp_tab, P.counters, P.stepacc, P.visit_hist,
                    P.log_state, P.log_meta, P.log_pi, P.log_count, P.results};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete ctx;
    return CCSP_OK;
}

static ccsp_ctx *create_on_device(const ccsp_config *cfg, int *err);

ccsp_ctx *ccsp_create(const ccsp_config *cfg, int *err) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    ccsp_ctx *ctx = create_on_device(cfg, err);
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);   // the caller's device stays current
    return ctx;
}

static ccsp_ctx *create_on_device(const ccsp_config *cfg, int *err) {
    if (err) *err = CCSP_OK;
    if (!cfg || cfg->n_slots <= 0 || cfg->sims <= 0 || cfg->sims > 4000 || cfg->game_stride == 0 || cfg->max_games == 0 ||
        cfg->mode < CCSP_MODE_SELFPLAY || cfg->mode > CCSP_MODE_GREEDY_DATA || cfg->greedy < 0 || cfg->greedy > 63 || cfg->stuck_limit < 0 ||
        (cfg->mode == CCSP_MODE_SELFPLAY && cfg->greedy != 0)) {
        if (err) *err = CCSP_EINVAL;
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
        if (err) *err = CCSP_ENODEVICE;                 // the product has no CPU path
        return nullptr;
    }
    ccsp_ctx *ctx = new ccsp_ctx();
    memset(&ctx->P, 0, sizeof(Params));
    ctx->sqrt_tab = ctx->pow_tab = ctx->rcp_tab = nullptr;
    ctx->cfg = *cfg;
    ctx->phase = 0;
    ctx->opening_plies = -1;
    ctx->stagger_span = 0;
    ctx->adv_budget = g_advance_budget; ctx->adv_time_cap = g_advance_time_cap; ctx->adv_deadline = g_advance_deadline;
    CTXCHK(hipSetDevice(cfg->device));
    Params &P = ctx->P;
    const uint64_t G = (uint64_t)cfg->n_slots;
    P.n_slots = cfg->n_slots; P.sims = cfg->sims; P.randomised = cfg->randomised; P.auto_restart = cfg->auto_restart;
    P.max_plies = cfg->max_plies > 0 ? cfg->max_plies : 1024;
    P.arena = cfg->mode == CCSP_MODE_ARENA; P.arena_det_tau = cfg->arena_det_tau != 0; P.enforce_move_limit = cfg->enforce_move_limit != 0;
    P.greedy = cfg->greedy; P.gen = cfg->mode == CCSP_MODE_GREEDY_DATA; P.stuck_limit = cfg->stuck_limit > 0 ? cfg->stuck_limit : 200;
    P.seed = cfg->seed; P.first_game = cfg->first_game; P.stride = cfg->game_stride; P.max_games = cfg->max_games;
    P.log_cap = cfg->log_capacity;
    // worst case: every one of the sims+1 expansions creates a full 126-edge block
    P.pool_stride = (uint64_t)(cfg->sims + 1) * MAX_BLOCK_BYTES;
    P.path_stride = (uint32_t)(cfg->sims + 2);
    ctx->pool_bytes = G * P.pool_stride;
    ctx->path_bytes = G * P.path_stride * sizeof(uint64_t);
    CTXALLOC(&P.slots, G * sizeof(SlotMem));
    CTXALLOC(&P.pend, G * sizeof(Pending));
    CTXALLOC(&P.pool, ctx->pool_bytes);
    CTXALLOC(&P.path, ctx->path_bytes);
    CTXALLOC(&P.counters, CCSP_CNT_COUNT * sizeof(unsigned long long));
    CTXALLOC(&P.dbg, (size_t)P.n_slots * CCSP_DBG_STRIDE * sizeof(unsigned long long));
    CTXCHK(hipMemset(P.dbg, 0, (size_t)P.n_slots * CCSP_DBG_STRIDE * sizeof(unsigned long long)));
    CTXALLOC(&P.stepacc, G * 8 * sizeof(uint32_t));
    CTXALLOC(&P.visit_hist, CCSP_NUM_ACTIONS * sizeof(unsigned long long));
    CTXALLOC(&P.log_count, sizeof(unsigned long long));
    const uint64_t cap = P.log_cap ? P.log_cap : 1;
    CTXALLOC(&P.log_state, cap * sizeof(ccsp_state));
    CTXALLOC(&P.log_meta, cap * sizeof(ccsp_sample_meta));
    CTXALLOC(&P.log_pi, cap * CCSP_NUM_ACTIONS * sizeof(double));
    CTXALLOC(&P.results, P.max_games * sizeof(ccsp_game_result));
    // integer-indexed tables from the HOST's libm: sqrt(n) (MCTS.py:62) and n**100 (MCTS.py:132, tau = 0.01)
    const int nt = cfg->sims + 2;
    std::vector<double> sq(nt), pw(nt);
    for (int i = 0; i < nt; i++) { sq[i] = sqrt((double)i); pw[i] = pow((double)i, 1. / 0.01); }
    CTXALLOC(&ctx->sqrt_tab, nt * sizeof(double));
    CTXALLOC(&ctx->pow_tab, nt * sizeof(double));
    CTXCHK(hipMemcpy(ctx->sqrt_tab, sq.data(), nt * sizeof(double), hipMemcpyHostToDevice));
    CTXCHK(hipMemcpy(ctx->pow_tab, pw.data(), nt * sizeof(double), hipMemcpyHostToDevice));
    P.sqrt_tab = (const double *)ctx->sqrt_tab; P.pow_tab = (const double *)ctx->pow_tab;
    const int nr = nt > RCP_N ? nt : RCP_N;                // reciprocals by IEEE division on the host (pick_edge)
    std::vector<double> rc(nr);
    for (int i = 0; i < nr; i++) rc[i] = i ? 1.0 / (double)i : 0.0;
    CTXALLOC(&ctx->rcp_tab, nr * sizeof(double));
    CTXCHK(hipMemcpy(ctx->rcp_tab, rc.data(), nr * sizeof(double), hipMemcpyHostToDevice));
    P.rcp_tab = (const double *)ctx->rcp_tab;
    if (ccsp_reset(ctx, nullptr) != CCSP_OK) { if (err) *err = CCSP_EHIP; ccsp_destroy(ctx); return nullptr; }
    CTXCHK(hipDeviceSynchronize());
    return ctx;
}

int ccsp_reset(ccsp_ctx *ctx, void *stream) {
    if (!ctx) return CCSP_EINVAL;
    Params &P = ctx->P;
    hipStream_t s = (hipStream_t)stream;
    CTX_ENTER(ctx, stream);
    CCSP_HIPCHK(hipMemsetAsync(P.counters, 0, CCSP_CNT_COUNT * sizeof(unsigned long long), s));
    CCSP_HIPCHK(hipMemsetAsync(P.stepacc, 0, (size_t)P.n_slots * 8 * sizeof(uint32_t), s));
    CCSP_HIPCHK(hipMemsetAsync(P.visit_hist, 0, CCSP_NUM_ACTIONS * sizeof(unsigned long long), s));
    CCSP_HIPCHK(hipMemsetAsync(P.log_count, 0, sizeof(unsigned long long), s));
    CCSP_HIPCHK(hipMemsetAsync(P.results, 0xFF, P.max_games * sizeof(ccsp_game_result), s));
    hipLaunchKernelGGL(reset_kernel, dim3(P.n_slots), dim3(64), 0, s, P);
    CCSP_HIPCHK(hipGetLastError());
    ctx->phase = 0;
    ctx->opening_plies = 0;
    return CCSP_OK;
}

int ccsp_set_positions(ccsp_ctx *ctx, const ccsp_state *states, const uint8_t *player, const uint64_t *game,
                       const uint32_t *ply, const uint8_t *det_tau, void *stream) {
    if (!ctx || !states || !player || !game || !ply || !det_tau) return CCSP_EINVAL;
    const Params &P = ctx->P;
    hipStream_t s = (hipStream_t)stream;
    const size_t G = (size_t)P.n_slots;
    CTX_ENTER(ctx, stream);
    // staging buffers are released on every way out (ccsp_devbuf), allocation failure is CCSP_ENOMEM
    ccsp_devbuf d_states, d_player, d_game, d_ply, d_tau;
    CCSP_ALLOCCHK(hipMalloc(&d_states.p, G * sizeof(ccsp_state)));
    CCSP_ALLOCCHK(hipMalloc(&d_player.p, G));
    CCSP_ALLOCCHK(hipMalloc(&d_game.p, G * 8));
    CCSP_ALLOCCHK(hipMalloc(&d_ply.p, G * 4));
    CCSP_ALLOCCHK(hipMalloc(&d_tau.p, G));
    CCSP_HIPCHK(hipMemcpy(d_states.p, states, G * sizeof(ccsp_state), hipMemcpyHostToDevice));
    CCSP_HIPCHK(hipMemcpy(d_player.p, player, G, hipMemcpyHostToDevice));
    CCSP_HIPCHK(hipMemcpy(d_game.p, game, G * 8, hipMemcpyHostToDevice));
    CCSP_HIPCHK(hipMemcpy(d_ply.p, ply, G * 4, hipMemcpyHostToDevice));
    CCSP_HIPCHK(hipMemcpy(d_tau.p, det_tau, G, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(set_positions_kernel, dim3(P.n_slots), dim3(64), 0, s, P, (const ccsp_state *)d_states.p,
                       (const uint8_t *)d_player.p, (const uint64_t *)d_game.p, (const uint32_t *)d_ply.p, (const uint8_t *)d_tau.p);
    CCSP_HIPCHK(hipGetLastError());
    CCSP_HIPCHK(hipStreamSynchronize(s));
    ctx->phase = 0;
    ctx->opening_plies = -1;           // these positions are searched at once
    return CCSP_OK;
}

static int g_plies_per_launch = 64;
int ccsp_debug_plies_per_launch(int n) { const int was = g_plies_per_launch; if (n >= 1) g_plies_per_launch = n; return was; }

int ccsp_play_plies(ccsp_ctx *ctx, int evaluator, int n_plies, void *stream) {
    if (!ctx || n_plies < 0 || evaluator < 0 || evaluator > CCSP_EVAL_ROLLOUT) return CCSP_EINVAL;
    if (n_plies == 0) return CCSP_OK;
    CTX_ENTER(ctx, stream);
    if (ctx->P.gen) {                                       // greedy data generator: nothing to search
        hipLaunchKernelGGL(greedy_plies_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, n_plies);
        CCSP_HIPCHK(hipGetLastError());
        return CCSP_OK;
    }
    if (g_plies_per_launch > 1 && n_plies > 1) {           // a wave carries its game through several plies per launch
        for (int done = 0; done < n_plies; done += g_plies_per_launch) {
            const int n = n_plies - done < g_plies_per_launch ? n_plies - done : g_plies_per_launch;
            hipLaunchKernelGGL(fused_plies_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, evaluator, n);
        }
        ctx->opening_plies = -1;
        CCSP_HIPCHK(hipGetLastError());
        return CCSP_OK;
    }
    for (int i = 0; i < n_plies; i++) {
        hipLaunchKernelGGL(fused_begin_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, evaluator);
        // the six random opening plies (selfplay.py:32-33) of the games started by ccsp_reset are played by the begin
        // kernel alone: no slot searches, so there is nothing for the other two kernels to do
        if (!ctx->P.arena && ctx->opening_plies >= 0 && ctx->opening_plies < CCSP_INITIAL_RANDOM_MOVES) { ctx->opening_plies++; continue; }
        ctx->opening_plies = -1;
        hipLaunchKernelGGL(fused_sims_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, evaluator);
        hipLaunchKernelGGL(fused_end_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P);
    }
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_ply_begin(ccsp_ctx *ctx, float *planes, void *stream) {
    if (!ctx || !planes) return CCSP_EINVAL;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(ply_begin_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, planes);
    CCSP_HIPCHK(hipGetLastError());
    ctx->phase = 1;
    return CCSP_OK;
}

int ccsp_root_expand(ccsp_ctx *ctx, const double *p, const float *v, void *stream) {
    if (!ctx || !p || !v) return CCSP_EINVAL;
    if (ctx->phase != 1) return CCSP_ESTATE;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(root_expand_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, p, v);
    CCSP_HIPCHK(hipGetLastError());
    ctx->phase = 2;
    return CCSP_OK;
}

int ccsp_select(ccsp_ctx *ctx, float *planes, void *stream) {
    if (!ctx || !planes) return CCSP_EINVAL;
    if (ctx->phase != 2) return CCSP_ESTATE;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(select_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, planes);
    CCSP_HIPCHK(hipGetLastError());
    ctx->phase = 3;
    return CCSP_OK;
}

int ccsp_expand_backup(ccsp_ctx *ctx, const double *p, const float *v, void *stream) {
    if (!ctx || !p || !v) return CCSP_EINVAL;
    if (ctx->phase != 3) return CCSP_ESTATE;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(expand_backup_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, p, v);
    CCSP_HIPCHK(hipGetLastError());
    ctx->phase = 2;
    return CCSP_OK;
}

int ccsp_expand_backup_select(ccsp_ctx *ctx, const double *p, const float *v, float *planes, void *stream) {
    if (!ctx || !p || !v || !planes) return CCSP_EINVAL;
    if (ctx->phase != 3) return CCSP_ESTATE;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(expand_backup_select_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, p, v, planes);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;                                        // phase stays 3: the next simulation's leaves are waiting for (p, v)
}

int ccsp_ply_end(ccsp_ctx *ctx, void *stream) {
    if (!ctx) return CCSP_EINVAL;
    if (ctx->phase != 2 && ctx->phase != 1) return CCSP_ESTATE;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(ply_end_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P);
    CCSP_HIPCHK(hipGetLastError());
    ctx->phase = 0;
    ctx->opening_plies = -1;           // a ply played through the stepped path: the fused path no longer knows the phase
    return CCSP_OK;
}

// ---- free-running stepped path ------------------------------------------------------------------------------------------

int ccsp_enable_tree_reuse(ccsp_ctx *ctx) {
    if (!ctx) return CCSP_EINVAL;
    if (ctx->P.pool2) return CCSP_OK;
    CTX_ENTER(ctx, nullptr);
    const int rc = ccsp_alloc_status(hipMalloc((void **)&ctx->P.pool2, ctx->pool_bytes), "hipMalloc(pool2)");
    if (rc != CCSP_OK) { ctx->P.pool2 = nullptr; return rc; }
    return CCSP_OK;
}

int ccsp_set_stagger_span(ccsp_ctx *ctx, int boundary_calls) {
    if (!ctx || boundary_calls < 0 || boundary_calls > 65535) return CCSP_EINVAL;      // (a slot's countdown is 16 bits of its word 15)
    ctx->stagger_span = boundary_calls;
    return CCSP_OK;
}

int ccsp_debug_read(ccsp_ctx *ctx, unsigned long long *out /* [64] */, int clear) {
    if (!ctx || !out) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    const size_t n = (size_t)ctx->P.n_slots * CCSP_DBG_STRIDE;
    std::vector<unsigned long long> all(n);
    CCSP_HIPCHK(hipMemcpy(all.data(), ctx->P.dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (clear) CCSP_HIPCHK(hipMemset(ctx->P.dbg, 0, n * sizeof(unsigned long long)));
    for (int i = 0; i < 64; i++) out[i] = 0;
    for (size_t g = 0; g < (size_t)ctx->P.n_slots; g++) {        // [0..8], [10..14] sums over the slots, [9] the longest call of any slot, [16..19] the numbers packed in [15]
        const unsigned long long *d = all.data() + g * CCSP_DBG_STRIDE;
        for (int i = 0; i < 16; i++) {
            if (i == 9) out[9] = d[9] > out[9] ? d[9] : out[9];
            else if (i == 15) { for (int k = 0; k < 4; k++) out[16 + k] += (d[15] >> (16 * k)) & 0xFFFF; }
            else out[i] += d[i];
        }
    }
    return CCSP_OK;
}

int ccsp_debug_read_slots(ccsp_ctx *ctx, unsigned long long *out /* [n_slots][CCSP_DEBUG_WORDS_PER_SLOT] */) {
    if (!ctx || !out) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    CCSP_HIPCHK(hipMemcpy(out, ctx->P.dbg, (size_t)ctx->P.n_slots * CCSP_DBG_STRIDE * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CCSP_OK;
}

// A launch lasts as long as its slowest wave, and the round waits for the launch: in one launch at 4096 x 400 the median wave takes 27 us, the
// 99th percentile 66, the last 76 (tools/bench_free.py --debug; a wave with three evaluator-free simulations: 49 us on average against 26
// without).  What trims that tail, measured in steady state on one MI355X with good_model.h5 (M node-expansions/s):
//   budget 3, no time limits                       18.21
//   budget 8, time cap 30 / 40 / 50 / 60 / 75 us   18.68 / 19.15 / 19.50 / 19.50 / 19.39    (budget 16: the same)
//   ... cap 50 + deadline 60 / 70 / 80 / 90 us     19.00 / 19.52 / 19.72 / 19.73            (the deadline's checks cost 2 % of what they win)
// Neither changes a game: a slot's record depends on the order of ITS simulations only.
// The two times are those of a batch of more than 1024 slots (an evaluator launch of 118 us beside the call); a smaller batch's evaluator
// launch is shorter (76 us up to 1024 positions, 47 us up to 512: ccsp_net_forward's workgroup shapes) and the times shrink with it
// (2048 slots in two half-batches: deadline 50 us 14.62 M, 80 us 14.37 M).
// Each context has its own three limits (ccsp_set_advance_limits); the process-wide values below are what ccsp_create starts a context
// with (the ccsp_debug_advance_* hooks change them for contexts created afterwards).
static int scaled_ticks(int ticks, int n_slots) {
    if (ticks <= 0) return 0;
    const int launch_us = n_slots <= 512 ? 47 : n_slots <= 1024 ? 76 : 120;
    const long long t = (long long)ticks * launch_us / 120;
    return (int)(t < 1 ? 1 : t);
}
int ccsp_debug_advance_deadline(int ticks) { const int was = g_advance_deadline; if (ticks >= 0) g_advance_deadline = ticks; return was; }
int ccsp_set_advance_limits(ccsp_ctx *ctx, int budget, int time_cap_ticks, int deadline_ticks) {
    if (!ctx) return CCSP_EINVAL;
    if (budget >= 1) ctx->adv_budget = budget;
    if (time_cap_ticks >= 0) ctx->adv_time_cap = time_cap_ticks;
    if (deadline_ticks >= 0) ctx->adv_deadline = deadline_ticks;
    return CCSP_OK;
}
int ccsp_debug_advance_budget(int n) { const int was = g_advance_budget; if (n >= 1) g_advance_budget = n; return was; }
int ccsp_debug_advance_time_cap(int ticks) { const int was = g_advance_time_cap; if (ticks >= 0) g_advance_time_cap = ticks; return was; }

int ccsp_advance(ccsp_ctx *ctx, const double *pk, const float *v, ccsp_request *req, uint16_t *moves, uint8_t *model_sel, int flags, void *stream) {
    if (!ctx || !pk || !v || !req || !moves || (flags & ~CCSP_ADVANCE_ALL_FLAGS)) return CCSP_EINVAL;
    if (ctx->cfg.mode != CCSP_MODE_SELFPLAY) return CCSP_EINVAL;          // Game.start's seats are served by the lock-step kernels
    if ((flags & CCSP_ADVANCE_REUSE) && !ctx->P.pool2) return CCSP_ESTATE;  // ccsp_enable_tree_reuse first (an allocation: not inside a captured graph)
    if (ctx->phase != 0) return CCSP_ESTATE;                                // not in the middle of a lock-step ply
    CTX_ENTER(ctx, stream);
    AdvArgs a;
    a.P = ctx->P; a.pk = pk; a.v = v; a.req = reinterpret_cast<Pending *>(req); a.moves = moves; a.model_sel = model_sel;
    a.flags = flags; a.budget = ctx->adv_budget;
    a.time_cap = scaled_ticks(ctx->adv_time_cap, ctx->P.n_slots); a.deadline = scaled_ticks(ctx->adv_deadline, ctx->P.n_slots);
    if (flags & CCSP_ADVANCE_DEBUG) hipLaunchKernelGGL(advance_kernel<true>, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(advance_kernel<false>, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, a);
    CCSP_HIPCHK(hipGetLastError());
    ctx->opening_plies = -1;
    return CCSP_OK;
}

int ccsp_boundary(ccsp_ctx *ctx, const double *pk, const float *v, ccsp_request *req, uint16_t *moves, uint8_t *model_sel, int flags, void *stream) {
    if (!ctx || !pk || !v || !req || !moves || (flags & ~CCSP_ADVANCE_ALL_FLAGS)) return CCSP_EINVAL;
    if (ctx->cfg.mode != CCSP_MODE_SELFPLAY) return CCSP_EINVAL;
    if ((flags & CCSP_ADVANCE_REUSE) && !ctx->P.pool2) return CCSP_ESTATE;
    if (ctx->phase != 0) return CCSP_ESTATE;
    CTX_ENTER(ctx, stream);
    hipLaunchKernelGGL(boundary_kernel, dim3(ctx->P.n_slots), dim3(64), 0, (hipStream_t)stream, ctx->P, pk, v, reinterpret_cast<Pending *>(req), moves, model_sel, flags,
                       (flags & CCSP_ADVANCE_STAGGER) ? (ctx->stagger_span > 0 ? ctx->stagger_span : ctx->cfg.sims) : 0);
    CCSP_HIPCHK(hipGetLastError());
    ctx->opening_plies = -1;
    return CCSP_OK;
}

// ---- read-back (synchronous; host buffers) ---------------------------------------------------------------

int ccsp_read_counters(ccsp_ctx *ctx, uint64_t *out /* [CCSP_CNT_COUNT] */) {
    if (!ctx || !out) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    CCSP_HIPCHK(hipMemcpy(out, ctx->P.counters, CCSP_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost));
    // plus what the stepped kernels have tallied per slot since the last ply_end
    std::vector<uint32_t> acc((size_t)ctx->P.n_slots * 8);
    CCSP_HIPCHK(hipMemcpy(acc.data(), ctx->P.stepacc, acc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    static const int map[7] = {CCSP_CNT_EXPANSIONS, CCSP_CNT_TERMINAL_SIMS, CCSP_CNT_SIMS, CCSP_CNT_SUM_DEPTH, CCSP_CNT_SUM_CHILDREN,
                               CCSP_CNT_SELECT_EDGES, CCSP_CNT_CACHE_HITS};
    for (int g = 0; g < ctx->P.n_slots; g++)
        for (int i = 0; i < 7; i++) out[map[i]] += acc[(size_t)g * 8 + i];
    return CCSP_OK;
}

int ccsp_read_visit_histogram(ccsp_ctx *ctx, uint64_t *out /* [294] */) {
    if (!ctx || !out) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    CCSP_HIPCHK(hipMemcpy(out, ctx->P.visit_hist, CCSP_NUM_ACTIONS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return CCSP_OK;
}

int ccsp_read_slots(ccsp_ctx *ctx, uint8_t *status, uint32_t *ply, uint64_t *game, ccsp_state *state, uint8_t *player) {
    if (!ctx) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    const int G = ctx->P.n_slots;
    std::vector<SlotMem> h((size_t)G);
    CCSP_HIPCHK(hipMemcpy(h.data(), ctx->P.slots, (size_t)G * sizeof(SlotMem), hipMemcpyDeviceToHost));
    for (int i = 0; i < G; i++) {
        if (status) status[i] = (uint8_t)((h[i].w[11] >> 8) & 0xFF);
        if (ply) ply[i] = (uint32_t)h[i].w[8];
        if (game) game[i] = h[i].w[4];
        if (state) memcpy(&state[i], &h[i].w[0], 32);
        if (player) player[i] = (uint8_t)(h[i].w[11] & 0xFF);
    }
    return CCSP_OK;
}

int ccsp_log_size(ccsp_ctx *ctx, uint64_t *n) {
    if (!ctx || !n) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    unsigned long long c = 0;
    CCSP_HIPCHK(hipMemcpy(&c, ctx->P.log_count, sizeof c, hipMemcpyDeviceToHost));
    *n = c < ctx->P.log_cap ? c : ctx->P.log_cap;
    return CCSP_OK;
}

// Rows read so far are given back: the log is empty again (stream-ordered).  A caller that reads the log every H plies and clears it
// never needs more than n_slots x H rows, however many games the context plays (a slot appends at most one row per ply).
int ccsp_log_clear(ccsp_ctx *ctx, void *stream) {
    if (!ctx) return CCSP_EINVAL;
    CTX_ENTER(ctx, stream);
    CCSP_HIPCHK(hipMemsetAsync(ctx->P.log_count, 0, sizeof(unsigned long long), (hipStream_t)stream));
    return CCSP_OK;
}

int ccsp_log_device_ptrs(ccsp_ctx *ctx, ccsp_state **state, ccsp_sample_meta **meta, double **pi) {
    if (!ctx) return CCSP_EINVAL;
    if (state) *state = ctx->P.log_state;
    if (meta) *meta = ctx->P.log_meta;
    if (pi) *pi = ctx->P.log_pi;
    return CCSP_OK;
}

int ccsp_read_log(ccsp_ctx *ctx, uint64_t first, uint64_t n, ccsp_state *state, ccsp_sample_meta *meta, double *pi) {
    if (!ctx) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    uint64_t have = 0;
    int rc = ccsp_log_size(ctx, &have);
    if (rc) return rc;
    if (first + n > have) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    if (state) CCSP_HIPCHK(hipMemcpy(state, ctx->P.log_state + first, n * sizeof(ccsp_state), hipMemcpyDeviceToHost));
    if (meta) CCSP_HIPCHK(hipMemcpy(meta, ctx->P.log_meta + first, n * sizeof(ccsp_sample_meta), hipMemcpyDeviceToHost));
    if (pi) CCSP_HIPCHK(hipMemcpy(pi, ctx->P.log_pi + first * CCSP_NUM_ACTIONS, n * CCSP_NUM_ACTIONS * sizeof(double), hipMemcpyDeviceToHost));
    return CCSP_OK;
}

int ccsp_read_results(ccsp_ctx *ctx, uint64_t first, uint64_t n, ccsp_game_result *out) {
    if (!ctx || !out || first + n > ctx->P.max_games) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    if (n) CCSP_HIPCHK(hipMemcpy(out, ctx->P.results + first, n * sizeof(ccsp_game_result), hipMemcpyDeviceToHost));
    return CCSP_OK;
}

// device address of the tree a slot FINISHED last: the first pool (fused and lock-step paths: there is no other), or -- on the free-running
// path, where with tree reuse consecutive plies alternate between two pools -- the pool boundary_kernel recorded in slot word 15 when it
// ended the ply (bits 10-11).  That record exists only once a free-running ply HAS ended: a context with a second pool that is driven
// through the lock-step entry points, or has not finished a ply yet, reads the pool of the search in progress (bit 8; 0 until then).
// With tree reuse the finished tree stays whole while the next ply is searched in the other pool; without, it lasts until the next
// ply's root is expanded (the answer to the root request).
static int tree_of_slot(ccsp_ctx *ctx, int slot, const uint8_t **out) {
    uint64_t w15 = 0;
    CCSP_HIPCHK(hipMemcpy(&w15, &ctx->P.slots[slot].w[15], sizeof w15, hipMemcpyDeviceToHost));
    uint32_t half = (uint32_t)((w15 >> 8) & 1);
    if ((w15 >> 11) & 1) half = (uint32_t)((w15 >> 10) & 1);
    if (!ctx->P.pool2) half = 0;
    *out = (half ? ctx->P.pool2 : ctx->P.pool) + (uint64_t)slot * ctx->P.pool_stride;
    return CCSP_OK;
}

// root edges of a slot's current tree (valid after a ply was searched, until the next one starts)
int ccsp_read_root(ccsp_ctx *ctx, int slot, int *k_out, uint32_t *N, double *W, double *Pr, uint16_t *mv) {
    if (!ctx || slot < 0 || slot >= ctx->P.n_slots || !k_out) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    std::vector<uint8_t> blk(MAX_BLOCK_BYTES);
    const uint8_t *tree = nullptr;
    { const int rc = tree_of_slot(ctx, slot, &tree); if (rc != CCSP_OK) return rc; }
    CCSP_HIPCHK(hipMemcpy(blk.data(), tree, MAX_BLOCK_BYTES, hipMemcpyDeviceToHost));
    const int K = (int)*reinterpret_cast<uint32_t *>(blk.data() + 32);
    if (K <= 0 || K > CCSP_MAX_MOVES) return CCSP_ESTATE;
    *k_out = K;
    if (Pr) memcpy(Pr, blk.data() + BLOCK_HDR, 8 * (size_t)K);
    if (W) memcpy(W, blk.data() + BLOCK_HDR + 8 * K, 8 * (size_t)K);
    if (N) memcpy(N, blk.data() + BLOCK_HDR + 16 * K, 4 * (size_t)K);
    if (mv) memcpy(mv, blk.data() + BLOCK_HDR + 24 * K, 2 * (size_t)K);
    return CCSP_OK;
}

// Debug/test: digest of a slot's whole tree in the reference's edge order (depth-first, each node's
// edges in list order): chain of mix64 over (N, bits(W), bits(P)) -- oracle/harness/gen_golden.py
// tree_digest() computes the same over the reference's Node/Edge objects.
static void digest_block(const uint8_t *pool, uint32_t off, uint64_t &h, uint64_t &nodes, uint64_t &edges) {
    const uint8_t *b = pool + off;
    const int K = (int)*reinterpret_cast<const uint32_t *>(b + 32);
    nodes++;
    const double *Pp = reinterpret_cast<const double *>(b + BLOCK_HDR);
    const double *Wp = reinterpret_cast<const double *>(b + BLOCK_HDR + 8 * K);
    const uint32_t *Np = reinterpret_cast<const uint32_t *>(b + BLOCK_HDR + 16 * K);
    const uint32_t *Cp = reinterpret_cast<const uint32_t *>(b + BLOCK_HDR + 20 * K);
    for (int j = 0; j < K; j++) {
        edges++;
        uint64_t wb, pb;
        memcpy(&wb, &Wp[j], 8); memcpy(&pb, &Pp[j], 8);
        h = ccsp_mix64(h ^ (uint64_t)Np[j]); h = ccsp_mix64(h ^ wb); h = ccsp_mix64(h ^ pb);
    }
    for (int j = 0; j < K; j++)
        if (Cp[j] != CHILD_LEAF && Cp[j] != CHILD_TERMINAL) digest_block(pool, (Cp[j] >> 7) << 3, h, nodes, edges);
}

int ccsp_debug_tree_digest(ccsp_ctx *ctx, int slot, uint64_t *digest, uint64_t *nodes, uint64_t *edges) {
    if (!ctx || slot < 0 || slot >= ctx->P.n_slots || !digest || !nodes || !edges) return CCSP_EINVAL;
    CTX_ENTER(ctx, nullptr);
    CCSP_HIPCHK(hipDeviceSynchronize());
    std::vector<uint8_t> pool(ctx->P.pool_stride);
    const uint8_t *tree = nullptr;
    { const int rc = tree_of_slot(ctx, slot, &tree); if (rc != CCSP_OK) return rc; }
    CCSP_HIPCHK(hipMemcpy(pool.data(), tree, ctx->P.pool_stride, hipMemcpyDeviceToHost));
    *digest = 0; *nodes = 0; *edges = 0;
    digest_block(pool.data(), 0, *digest, *nodes, *edges);
    return CCSP_OK;
}

}  // extern "C"
