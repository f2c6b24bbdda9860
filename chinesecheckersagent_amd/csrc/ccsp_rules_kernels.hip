// ccsp_rules_kernels.hip -- batched rules kernels for gfx950: move generation, step, plane
// encoding over arrays of 32-byte board records.
//
// Roofline: all three are streaming kernels (HBM-bound by design; DESIGN.md has the byte model):
//   movegen  reads 32 B + 1 B, writes 48 B masks + 1 B count + 2K B moves   (80 + 2K B/state)
//   step     reads 32 B + 3 B, writes 32 B + 1 B (+2 B progress)            (67 B/state)
//   encode   reads 32 B + 1 B, writes 1372 B                                (1404 B/state)
#include "ccsp_common.h"

namespace {

// diagnostic build (-DMG_STAMPS, never timed for throughput): cycles of wave 0's lane 0 of every workgroup per phase of movegen_kernel, summed
#ifdef MG_STAMPS
__device__ unsigned long long mg_stamps[8];
#define MG_LAP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&mg_stamps[i], now_ - lap_); lap_ = now_; } } while (0)
#else
#define MG_LAP(i) do { } while (0)
#endif

constexpr int MG_THREADS = 256;
constexpr int MG_WAVES = MG_THREADS / 64;
constexpr int MG_CHUNK = 32;                                // states per wave
constexpr int MG_STATES = MG_WAVES * MG_CHUNK;              // 128 states per workgroup
constexpr int MG_TASKS = MG_CHUNK * 6;                      // (state, checker) tasks per wave
constexpr int MG_STACK = 20;                                // largest value of the test hook ccsp_debug_movegen_stack_cap (the LDS stacks of
                                                            // rounds 2-5 held 20 entries; a lane's register stack holds MG_REGROOM)
constexpr int MG_BIGSTACK = 96;                             // >= 81 (<= 5 pending siblings per visited sub-lattice cell (16) + 1) + 6 tentative
constexpr int MG_SLOT = 24;                                 // bytes of LDS per checker list (<= 21 used)
#ifndef MG_LINES_STRIDE
#define MG_LINES_STRIDE 28
#endif
#ifndef MG_LIST_PAD
#define MG_LIST_PAD 0
#endif

constexpr int MG_LIST_STATE = 6 * MG_SLOT + MG_LIST_PAD;     // bytes of list staging per position
#define LST(L, s, c, i) (L).lists[(s) * MG_LIST_STATE + (c) * MG_SLOT + (i)]

constexpr int MG_REGROOM = 10;                              // entries of a lane's depth-first stack: 6 bits per cell in a 64-bit REGISTER (round 6).
                                                            // Real positions never hold more than 8 (tests/test_device_logic_on_host.py:
                                                            // 400 000 positions); a search that would need more is redone on the wave's one
                                                            // big stack in LDS (MG_BIGSTACK)

// preset (off-board) bits of the 13 diagonals' patterns, eight lines to a 64-bit word (columns and rows are seven cells long: nothing preset)
static constexpr uint64_t mg_diag_base(int first, int count) {
    const ccsp_line_tables t = ccsp_make_lines();
    uint64_t v = 0;
    for (int i = 0; i < count; i++) v |= (uint64_t)t.base[14 + first + i] << (8 * i);
    return v;
}
constexpr uint64_t MG_BASE_DA = mg_diag_base(0, 8), MG_BASE_DB = mg_diag_base(8, 5);

struct __attribute__((aligned(8))) MgWave {                      // per-wave LDS: 6.1 KB -> SIX 4-wave workgroups per CU (the per-lane LDS stacks of
                                                                 // rounds 2-5 made it 7.1 KB and five)
    uint8_t lines[MG_CHUNK][MG_LINES_STRIDE];               // 27 line patterns per state (+ pad)
    uint8_t lists[MG_CHUNK * MG_LIST_STATE];
    uint8_t cnt[MG_CHUNK][8];
    uint8_t wl[MG_TASKS];                                   // the tasks that enter the search loop: >= 2 first hops from the front, 1 from the back
    uint8_t big[MG_BIGSTACK];                               // (also the PACKED write-out's 32 x u16 list offsets, once the searches are done)
    uint32_t redo[MG_TASKS / 32];                           // tasks to redo on the big stack (bit per task)
};
static_assert(MG_BIGSTACK >= 2 * MG_CHUNK, "the packed write-out keeps 32 x u16 in `big`");


// B2-B4 over an array of positions.  The ordered hop search of one checker (board.py:166-211) is a serial
// depth-first walk whose length varies a lot between checkers, so lanes do not own a fixed checker: each wave
// takes a chunk of 32 positions = 192 (position, checker) tasks, every lane runs ONE flat state machine
// (one visited cell per iteration: six mirror-hop lookups HOP[line pattern][position][sense], ccsp_rules.h, and a
// stack in a 64-bit register) and pulls the next task
// of the chunk the moment its own is finished (ballot + rank) -- no lane waits for the longest walk of its wave.
// Round 6 (profiles/r6_movegen_ab.txt, 5.30 -> 6.40 G states/s): the ORIGIN's six hop look-ups are made in the walk phase -- every
// lane busy, no visited test, the three patterns already in registers -- and left as the task's initial stack image; only checkers that
// can hop at all enter the search loop (a compacted worklist), those with two or more first hops before those with one (a wave is done
// when its last lane is); the stack itself in a register (an LDS read and six byte-wide LDS writes off every visited cell's chain).
// Hop landings stay on the origin's sub-lattice (row and column keep their parity: <= 4 x 4 cells), so every cell a search
// sees has the parity of the origin's cell index (7 r + c = r + c mod 2) and cell >> 1 names it uniquely: the visited set is
// 25 bits indexed by cell >> 1 -- one shift per test, no 64-bit shifts on a cell mask, no (row / 2, column / 2) arithmetic
// (round 6; the 16-bit sub-lattice index cost two multiply-adds and a mask per hop direction).
// Per-checker lists are staged in LDS and written out in the reference's move order, a position at a time,
// neighbouring lanes writing neighbouring bytes.
// GREEDY (next-4): the same search, but what is written out is GreedyPlayer.decide_move(training=True)
// (player.py:72-118): of the position's moves only those of maximum forward distance that start on the row of the
// rear-most checker among them, into best[n][CCSP_GREEDY_MAX][2].
// PACKED (ccsp_movegen_packed): the lists of a wave's 32 positions go out back to back in position order instead of one
// 252-byte row each -- rows 252 bytes apart, ~67 bytes used, cost 1.27 x the algorithmic write bytes in partly written 32-byte
// sectors (profiles/counters.json); back to back they fill whole sectors.
template <bool GREEDY, bool PACKED = false>
__global__ __launch_bounds__(MG_THREADS) void movegen_kernel(const ccsp_state *__restrict__ states,
                                                             const uint8_t *__restrict__ player, int n,
                                                             uint8_t *__restrict__ moves, uint8_t *__restrict__ count,
                                                             uint64_t *__restrict__ dest_mask, int cap) {
    __shared__ __attribute__((aligned(16))) ccsp_line_tables T;
    __shared__ MgWave WV[MG_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    MgWave &L = WV[wave];
#ifdef MG_STAMPS
    unsigned long long lap_ = __builtin_amdgcn_s_memtime();
#endif
    // The order of a workgroup's start (round 6): the line tables' 559 dwords are REQUESTED first (three per thread, into registers), then the
    // wave's position records; the 27 patterns of every position are built from the record alone while the table is still in flight, and only
    // then does the table go into LDS -- one memory latency at a workgroup's start instead of two in a row (the stamps of the one-after-the-
    // other order: 29 % + 17 % of a workgroup's lifetime, profiles/r6_movegen_ab.txt).
    constexpr int TDW = (int)(sizeof(ccsp_line_tables) / 4), TPER = (TDW + MG_THREADS - 1) / MG_THREADS;
    uint32_t tab[TPER];
#pragma unroll
    for (int i = 0; i < TPER; i++) { const int j = tid + i * MG_THREADS; tab[i] = j < TDW ? reinterpret_cast<const uint32_t *>(&CCSP_LINES_DEV)[j] : 0u; }
    const long long base = ((long long)blockIdx.x * MG_WAVES + wave) * MG_CHUNK;     // first position of this wave
    const int here = (int)((n - base) < 0 ? 0 : ((n - base) < MG_CHUNK ? (n - base) : MG_CHUNK));
    if (lane < MG_TASKS / 32) L.redo[lane] = 0;
    MG_LAP(0);                                          // [0] requests issued

    // ---- line patterns: IN REGISTERS, half a wave per side -------------------------------------------------------
    // A checker at (r, c) sets bit 8 c + r of the columns' word, bit 8 r + c of the rows' word and bit 8 (r - c + 6) + min(r, c) of the
    // diagonals' two words -- shifts and ORs, no table, no chain of 36 dependent byte read-modify-writes in LDS; the lower half of the wave
    // takes player one's six checkers of its position, the upper half player two's; the halves meet through one cross-lane exchange and the
    // row goes to LDS as seven dwords: bytes 0-6 columns, 7-13 rows, 14-26 diagonals (ccsp_rules.h: LP / base).
    {
        static_assert(MG_LINES_STRIDE == 28 && MG_CHUNK == 32, "a position's patterns are seven dwords; half a wave per side");
        const int p = lane & 31, side = lane >> 5;
        uint64_t a0 = 0, a1 = 0, da = 0, db = 0;
        ccsp_sr s; s.occ0 = s.occ1 = s.a = s.b = 0;
        int my_player = 1;
        if (p < here) {
            s = ccsp_load_sr(states + base + p);
            my_player = player[base + p];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const int cell = ccsp_sr_pos(s, 6 * side + k);
                const int r = (int)(__umul24((unsigned)cell, 37u) >> 8), c = cell - 7 * r;
                a0 |= 1ULL << (8 * c + r);
                a1 |= 1ULL << (8 * r + c);
                const int sh = 8 * (r - c + 6) + (r < c ? r : c);              // 0 .. 102
                const uint64_t bit = 1ULL << (sh & 63);
                da |= sh < 64 ? bit : 0ULL;
                db |= sh < 64 ? 0ULL : bit;
            }
        }
        a0 |= __shfl_xor(a0, 32); a1 |= __shfl_xor(a1, 32); da |= __shfl_xor(da, 32); db |= __shfl_xor(db, 32);
        da |= MG_BASE_DA; db |= MG_BASE_DB;
        if (lane < here) {                                                      // (the lower half writes the rows ...)
            uint32_t *row = reinterpret_cast<uint32_t *>(&L.lines[lane][0]);
            const uint64_t w0 = a0 | (a1 << 56), w1 = (a1 >> 8) | (da << 48), w2 = (da >> 16) | (db << 48);
            row[0] = (uint32_t)w0; row[1] = (uint32_t)(w0 >> 32); row[2] = (uint32_t)w1; row[3] = (uint32_t)(w1 >> 32);
            row[4] = (uint32_t)w2; row[5] = (uint32_t)(w2 >> 32); row[6] = (uint32_t)(db >> 16);
        } else if (side == 1 && p < here) {                                     // ... the upper half what the tasks need: the origin cells of the
#pragma unroll                                                                  // side to move, in the last byte of each list slot
            for (int c = 0; c < 6; c++) LST(L, p, c, MG_SLOT - 1) = (uint8_t)ccsp_sr_pos(s, (my_player - 1) * 6 + c);
        }
    }
#pragma unroll
    for (int i = 0; i < TPER; i++) { const int j = tid + i * MG_THREADS; if (j < TDW) reinterpret_cast<uint32_t *>(&T)[j] = tab[i]; }
    __syncthreads();

    MG_LAP(1);                                          // [1] records in, line patterns, table into LDS
    const int ntasks = here * 6;
    // ---- walks: lane = task, three rounds.  Direction order N,E,SE,S,W,NW (board.py:149-155), read off the three line patterns
    // through the origin (bits beyond a line's end are preset; the byte is stored unconditionally, kept only if legal).  Done here,
    // with every lane busy, instead of inside the search loop, where the lanes reach their origins at different iterations and the
    // whole wave would step through these ~90 instructions every time one of them does.
    int n_two = 0, n_one = 0;                           // tasks with >= 2 / exactly 1 first hop (wave-uniform)
    for (int t0 = 0; t0 < ntasks; t0 += 64) {
        const int t = t0 + lane;
        int hops0 = 0;                                  // first hops of this lane's task
        if (t < ntasks) {
        const int s = (int)(__umul24((unsigned)t, 171u) >> 10), c = t - 6 * s;      // t / 6, exact for t < 515
        const int x = LST(L, s, c, MG_SLOT - 1);
        const int r = (int)(__umul24((unsigned)x, 37u) >> 8), col = x - 7 * r, m = r < col ? r : col;
        const uint8_t *pat = L.lines[s];
        const uint32_t p0 = pat[col], p1 = pat[7 + r], p2 = pat[20 + r - col];
        int k = 0;
#define MG_WALK(P, POS, D, STEP) { const int np = (POS) + (D); const bool ok = (np >= 0) & (np <= 6) & ((((P) >> (np & 7)) & 1u) == 0); \
                                   LST(L, s, c, k) = (uint8_t)(x + (STEP)); k += ok ? 1 : 0; }
        MG_WALK(p0, r, -1, -7) MG_WALK(p1, col, 1, 1) MG_WALK(p2, m, 1, 8) MG_WALK(p0, r, 1, 7) MG_WALK(p1, col, -1, -1) MG_WALK(p2, m, -1, -8)
#undef MG_WALK
        L.cnt[s][c] = (uint8_t)k;
        // the origin's own hops (the visit the search would begin with): the checker lifted off its three lines (board.py:158), both senses of
        // each line in one 16-bit read, the landings ranked NW, W, S, SE, E, N into the register image of the stack (N on top); nothing is
        // visited yet but the origin, which no hop lands on.  The image, the count, the origin and the walk count go into bytes 8..15 of the
        // task's list slot (free until the list has eight entries, by which time the task has long read them).
        {
            const uint32_t q0 = p0 & ~(1u << r), q1 = p1 & ~(1u << col), q2 = p2 & ~(1u << m);
            const uint32_t h0 = *reinterpret_cast<const uint16_t *>(&T.hop[q0][r][0]);
            const uint32_t h1 = *reinterpret_cast<const uint16_t *>(&T.hop[q1][col][0]);
            const uint32_t h2 = *reinterpret_cast<const uint16_t *>(&T.hop[q2][m][0]);
            const int b0 = x - 7 * r, b1 = x - col, b2 = x - 8 * m;
            uint32_t accA = 0, accB = 0; int nA = 0, nB = 0;
#define MG_ACC0(ACC, N, HP, BASE, STRIDE) { const int hp = (int)(HP); const bool ok = hp < 7; \
                                            ACC = ok ? ((ACC << 6) | (uint32_t)(hp * (STRIDE) + (BASE))) : ACC; N += ok ? 1 : 0; }
            MG_ACC0(accA, nA, h2 & 0xFF, b2, 8)      // NW
            MG_ACC0(accA, nA, h1 & 0xFF, b1, 1)      // W
            MG_ACC0(accA, nA, h0 >> 8, b0, 7)        // S
            MG_ACC0(accB, nB, h2 >> 8, b2, 8)        // SE
            MG_ACC0(accB, nB, h1 >> 8, b1, 1)        // E
            MG_ACC0(accB, nB, h0 & 0xFF, b0, 7)      // N
#undef MG_ACC0
            hops0 = nA + nB;
            const uint64_t image = (((uint64_t)accA << (6 * nB)) | (uint64_t)accB) | ((uint64_t)x << 36) | ((uint64_t)k << 44) | ((uint64_t)hops0 << 48);
            *reinterpret_cast<uint64_t *>(&LST(L, s, c, 8)) = image;
        }
        }
        // the worklist: tasks with two or more first hops from the front, with exactly one from the back -- the long searches start first
        // (a wave is done when its last lane is), checkers that cannot hop never enter the loop
        const uint64_t two = __ballot(hops0 >= 2), one = __ballot(hops0 == 1), below = (1ULL << lane) - 1;
        if (hops0 >= 2) L.wl[n_two + __popcll(two & below)] = (uint8_t)t;
        if (hops0 == 1) L.wl[MG_TASKS - 1 - (n_one + __popcll(one & below))] = (uint8_t)t;
        n_two += __popcll(two); n_one += __popcll(one);
    }
    __syncthreads();

    MG_LAP(2);                                          // [2] walks + the origins' hops + worklist
    // ---- task loop: lane = worker ---------------------------------------------------------------------------
    // The ordered hop search (board.py:166-211) with an explicit stack: popping a cell that is still unvisited
    // visits it (= the recursive call), looks up the mirror hop in all six directions (three line patterns, two
    // senses each: one 16-bit read per pattern) and pushes the legal unvisited landings, last direction first, so
    // that the first one is on top; a popped cell that was reached through another branch in the meantime is dropped
    // (= the `not in hops` test of the caller's loop).  One iteration per visited cell instead of one per hop test.
    int task = lane;                                    // current task (position-major: task = 6 * s + c)
    int next = 64;                                      // next unassigned task of the chunk (wave-uniform)
    int st_s = 0, st_c = 0, origin = 0, cnt_n = 0, sp = 0, orow = 0, oc = 0;
    uint32_t visited = 0;                               // cells seen: bit cell >> 1 (all of one parity, see above)
    const uint8_t *pat = L.lines[0];                    // the task's position: its 27 line patterns ...
    uint8_t *lst = &LST(L, 0, 0, 0);                    // ... and the task's list (both set once per task, not per visited cell)
    uint32_t lift0 = ~0u, lift1 = ~0u, lift2 = ~0u;     // the moving checker lifted off its three lines (board.py:158): pattern masks, per task
    int odiag = 0;                                      // orow - oc: names the origin's diagonal
    bool active = false;

    // a task starts with its origin on the stack, its list holding the walks
    auto start_task = [&](int t, uint8_t *stk) {
        st_s = (int)(__umul24((unsigned)t, 171u) >> 10);                   // t / 6, exact for t < 515 (tasks: < 192)
        st_c = t - 6 * st_s;
        pat = L.lines[st_s];
        lst = &LST(L, st_s, st_c, 0);
        origin = lst[MG_SLOT - 1];
        cnt_n = L.cnt[st_s][st_c]; visited = 0;
        orow = (int)(__umul24((unsigned)origin, 37u) >> 8); oc = origin - 7 * orow;
        lift0 = ~(1u << orow); lift1 = ~(1u << oc); lift2 = ~(1u << (orow < oc ? orow : oc)); odiag = orow - oc;
        if (stk != nullptr) stk[0] = (uint8_t)origin;
        sp = 1;
    };
    // one pop; returns false when the visit would not fit the stack (`room` entries): nothing is changed then
    auto step = [&](uint8_t *stk, int room) -> bool {
        const int x = stk[sp - 1];
        const int r = (int)(__umul24((unsigned)x, 37u) >> 8), c = x - 7 * r;
        const int xi = x >> 1;
        if ((visited >> xi) & 1u) { sp--; return true; }
        if (sp - 1 + 6 > room) return false;                           // six tentative pushes must fit
        sp--;
        visited |= 1u << xi;
        // the three lines through x; the moving checker is lifted off its own lines (board.py:158)
        uint32_t p0 = pat[c], p1 = pat[7 + r], p2 = pat[20 + r - c];
        const int m = r < c ? r : c;
        if (c == oc) p0 &= lift0;
        if (r == orow) p1 &= lift1;
        if (r - c == odiag) p2 &= lift2;
        lst[cnt_n] = (uint8_t)x;                                       // a hop landing (the byte stored for the origin itself, which is
        cnt_n += x != origin ? 1 : 0;                                   // not a move, is overwritten by the next landing)
        // both senses of a line in one 16-bit read: low byte = sense -, high byte = sense +
        const uint32_t h0 = *reinterpret_cast<const uint16_t *>(&T.hop[p0][r][0]);
        const uint32_t h1 = *reinterpret_cast<const uint16_t *>(&T.hop[p1][c][0]);
        const uint32_t h2 = *reinterpret_cast<const uint16_t *>(&T.hop[p2][m][0]);
        // directions N,E,SE,S,W,NW = (axis 0,-) (1,+) (2,+) (0,+) (1,-) (2,-); pushed in reverse order (the byte is
        // stored unconditionally and kept only if the landing is legal and unvisited: no branches).
        // landing of a hop to line position hp: cell x + (hp - pos) * stride = hp * stride + (x - pos * stride)
        const int b0 = x - 7 * r, b1 = x - c, b2 = x - 8 * m;
#define MG_PUSH(HP, BASE, STRIDE) { const int hp = (int)(HP); const int land = hp * (STRIDE) + (BASE); \
                                    stk[sp] = (uint8_t)land; sp += ((hp < 7) & (__builtin_amdgcn_ubfe(visited, (unsigned)land >> 1, 1u) == 0)) ? 1 : 0; }
        MG_PUSH(h2 & 0xFF, b2, 8)      // NW
        MG_PUSH(h1 & 0xFF, b1, 1)      // W
        MG_PUSH(h0 >> 8, b0, 7)        // S
        MG_PUSH(h2 >> 8, b2, 8)        // SE
        MG_PUSH(h1 >> 8, b1, 1)        // E
        MG_PUSH(h0 & 0xFF, b0, 7)      // N
#undef MG_PUSH
        return true;
    };
    auto finish_task = [&]() { L.cnt[st_s][st_c] = (uint8_t)cnt_n; };
    // The same pop with the lane's stack in a REGISTER: rs = cells of 6 bits, the top in the low bits.  The six landings are ranked into two
    // 18-bit accumulators (NW, W, S | SE, E, N: first pushed = highest) and shifted in behind the popped cell in one go; the visit is
    // refused -- nothing changed -- when the stack would exceed `room` (<= 10) entries: the big-stack redo path below, as before.
    // Takes one LDS read (the top) and six byte-wide LDS writes off every visited cell's dependent chain, and the stacks' 1280 bytes
    // off the wave's LDS (six workgroups per CU instead of five).
    uint64_t rs = 0;
    uint8_t *const stk = nullptr;                                    // (no LDS stack: start_task leaves the origin to `rs`)
    auto step_reg = [&](int room) -> bool {
        int x = (int)((uint32_t)rs & 63u);
        const int xi = x >> 1;
        if ((visited >> xi) & 1u) { rs >>= 6; sp--; return true; }
        const int r = (int)(__umul24((unsigned)x, 37u) >> 8), c = x - 7 * r;
        uint32_t p0 = pat[c], p1 = pat[7 + r], p2 = pat[20 + r - c];
        const int m = r < c ? r : c;
        if (c == oc) p0 &= lift0;
        if (r == orow) p1 &= lift1;
        if (r - c == odiag) p2 &= lift2;
        const uint32_t h0 = *reinterpret_cast<const uint16_t *>(&T.hop[p0][r][0]);
        const uint32_t h1 = *reinterpret_cast<const uint16_t *>(&T.hop[p1][c][0]);
        const uint32_t h2 = *reinterpret_cast<const uint16_t *>(&T.hop[p2][m][0]);
        const int b0 = x - 7 * r, b1 = x - c, b2 = x - 8 * m;
        const uint32_t seen = visited | (1u << xi);
        uint32_t accA = 0, accB = 0; int nA = 0, nB = 0;
#define MG_ACC(ACC, N, HP, BASE, STRIDE) { const int hp = (int)(HP); const int land = hp * (STRIDE) + (BASE); \
                                           const bool ok = (hp < 7) & (__builtin_amdgcn_ubfe(seen, (unsigned)land >> 1, 1u) == 0); \
                                           ACC = ok ? ((ACC << 6) | (uint32_t)land) : ACC; N += ok ? 1 : 0; }
        MG_ACC(accA, nA, h2 & 0xFF, b2, 8)      // NW
        MG_ACC(accA, nA, h1 & 0xFF, b1, 1)      // W
        MG_ACC(accA, nA, h0 >> 8, b0, 7)        // S
        MG_ACC(accB, nB, h2 >> 8, b2, 8)        // SE
        MG_ACC(accB, nB, h1 >> 8, b1, 1)        // E
        MG_ACC(accB, nB, h0 & 0xFF, b0, 7)      // N
#undef MG_ACC
        if (sp - 1 + nA + nB > room) return false;
        visited = seen;
        lst[cnt_n] = (uint8_t)x;
        cnt_n += x != origin ? 1 : 0;
        rs >>= 6;
        rs = (rs << (6 * nA)) | (uint64_t)accA;
        rs = (rs << (6 * nB)) | (uint64_t)accB;
        sp += nA + nB - 1;
        return true;
    };
    const int room = cap < MG_REGROOM ? cap : MG_REGROOM;
    const int nwork = n_two + n_one;                                 // entries of the worklist
    auto work_at = [&](int i) -> int { return (int)L.wl[i < n_two ? i : MG_TASKS - 1 - (i - n_two)]; };
    // a task starts behind its origin's visit: the stack image the walk phase left, the origin marked, the list holding the walks
    auto start_work = [&](int t) {
        st_s = (int)(__umul24((unsigned)t, 171u) >> 10);
        st_c = t - 6 * st_s;
        pat = L.lines[st_s];
        lst = &LST(L, st_s, st_c, 0);
        const uint64_t image = *reinterpret_cast<const uint64_t *>(lst + 8);
        rs = image & ((1ULL << 36) - 1);
        origin = (int)((image >> 36) & 63); cnt_n = (int)((image >> 44) & 15); sp = (int)((image >> 48) & 7);
        visited = 1u << (origin >> 1);
        orow = (int)(__umul24((unsigned)origin, 37u) >> 8); oc = origin - 7 * orow;
        lift0 = ~(1u << orow); lift1 = ~(1u << oc); lift2 = ~(1u << (orow < oc ? orow : oc)); odiag = orow - oc;
    };
    if (lane < nwork) { task = work_at(lane); start_work(task); active = true; }
#define MG_NTASKS_IN_LOOP nwork
#define MG_START(i) { task = work_at(i); start_work(task); }

    while (__any(active)) {
        if (active) {
            if (!step_reg(room)) {
                atomicOr(&L.redo[task >> 5], 1u << (task & 31));
                active = false;
            } else if (sp == 0) {
                finish_task();
                active = false;
            }
        }
        const uint64_t idle = __ballot(!active);
        if (next < MG_NTASKS_IN_LOOP && idle) {
            const int rank = __popcll(idle & ((1ULL << lane) - 1));
            const int i = next + rank;
            if (!active && i < MG_NTASKS_IN_LOOP) { MG_START(i) active = true; }
            next += __popcll(idle);
        }
    }
#undef MG_NTASKS_IN_LOOP
#undef MG_START
    MG_LAP(3);                                          // [3] the search loop
    // searches that did not fit a lane's stack: one at a time, lane 0, on the wave's big stack
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int w = 0; w < MG_TASKS / 32; w++) {
        uint32_t bits = L.redo[w];
        while (bits) {
            const int t = w * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            if (lane == 0) {
                start_task(t, L.big);
                while (sp > 0) (void)step(L.big, MG_BIGSTACK);
                finish_task();
            }
        }
    }
    __syncthreads();

    MG_LAP(4);                                          // [4] big-stack redo
    // ---- destination masks (optional output): lane = task, the cells of its finished list.  (Tried instead: the walk cells
    // stored here in the walk phase and the hop cells ORed in by an atomic when a task ends -- the extra instructions in the search
    // loop cost more; LDS atomics from the write-out lanes -- one word per checker serialises them.) ---------------------------
    if (dest_mask != nullptr) {
        for (int t = lane; t < ntasks; t += 64) {
            const int s = (int)(__umul24((unsigned)t, 171u) >> 10), c = t - 6 * s;
            const int k = L.cnt[s][c];
            uint64_t m = 0;
            for (int i = 0; i < k; i++) m |= 1ULL << LST(L, s, c, i);
            dest_mask[base * 6 + t] = m;
        }
        __syncthreads();                                // the counts are rewritten below
    }
    MG_LAP(5);                                          // [5] destination masks
    // ---- write out in the reference's move order: half a wave per position, lane = move slot ------------------
    int my_k = 0;                                       // this position's number of moves (lane = position)
    if (lane < here) {                                  // prefix sums of the six per-checker counts, packed one byte each
        uint64_t pre = 0; int acc = 0;
#pragma unroll
        for (int c = 0; c < 6; c++) { acc += L.cnt[lane][c]; pre |= (uint64_t)acc << (8 * (c + 1)); }
        *reinterpret_cast<uint64_t *>(L.cnt[lane]) = pre;           // cnt[s][c] now = moves of checkers < c; cnt[s][6] = K
        if (!GREEDY) count[base + lane] = (uint8_t)acc;
        my_k = acc;
    }
    if (PACKED) {                                       // where each position's list starts in the chunk's stream: prefix over positions
        const int k = my_k;
        int incl = k;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const int o = __shfl_up(incl, d); if ((lane & 31) >= d) incl += o; }
        if (lane < MG_CHUNK) reinterpret_cast<uint16_t *>(L.big)[lane] = (uint16_t)(incl - k);      // (the big stack is free by now)
    }
    __syncthreads();
    const int half = lane >> 5, hl = lane & 31;
    for (int s0 = 0; s0 < here; s0 += 2) {
        const int s = s0 + half;
        const bool on = s < here;
        const uint64_t pre = *reinterpret_cast<const uint64_t *>(L.cnt[on ? s : 0]);
        const int K = on ? (int)((pre >> 48) & 0xFF) : 0;
        const int p1 = (int)((pre >> 8) & 0xFF), p2 = (int)((pre >> 16) & 0xFF), p3 = (int)((pre >> 24) & 0xFF),
                  p4 = (int)((pre >> 32) & 0xFF), p5 = (int)((pre >> 40) & 0xFF);
        auto move_at = [&](int j, int &id, int &dest) {
            id = (j >= p1) + (j >= p2) + (j >= p3) + (j >= p4) + (j >= p5);
            dest = LST(L, on ? s : 0, id, j - (int)((pre >> (8 * id)) & 0xFF));
        };
        if (!GREEDY) {
            for (int j = hl; j < K; j += 32) {
                int id, dest;
                move_at(j, id, dest);
                const long long row = PACKED ? base * CCSP_MAX_MOVES + reinterpret_cast<const uint16_t *>(L.big)[on ? s : 0]
                                             : (base + s) * CCSP_MAX_MOVES;
                reinterpret_cast<uint16_t *>(moves)[row + j] = (uint16_t)id | ((uint16_t)dest << 8);
            }
        } else {
            // player.py:100-115 over the list in LDS: two maxima per position (half-wave reductions), then the survivors
            // in list order.  Both halves of the wave run the same number of 32-move chunks.
            const int mover = on ? (int)player[base + s] : 1;
            const int kother = __shfl_xor(K, 32);                          // (outside any lane-dependent branch)
            const int kmax = K > kother ? K : kother;
            auto human_row = [](int cell) { const int r = (int)(__umul24((unsigned)cell, 37u) >> 8); return 8 * r - cell + 7; };
            auto half_max = [](int v) {
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o > v ? o : v; }
                return v;
            };
            int dbest = 0;                                                  // distance + 32 (0 = no move)
            for (int j0 = 0; j0 < kmax; j0 += 32) {
                const int j = j0 + hl;
                int d = 0;
                if (j < K) {
                    int id, dest; move_at(j, id, dest);
                    const int sr = human_row(LST(L, s, id, MG_SLOT - 1)), er = human_row(dest);
                    d = (mover == 1 ? sr - er : er - sr) + 32;
                }
                dbest = dbest > d ? dbest : d;
            }
            dbest = half_max(dbest);
            int kbest = 0;                                                  // rear-most start row among the best, as a key
            for (int j0 = 0; j0 < kmax; j0 += 32) {
                const int j = j0 + hl;
                int k = 0;
                if (j < K) {
                    int id, dest; move_at(j, id, dest);
                    const int sr = human_row(LST(L, s, id, MG_SLOT - 1)), er = human_row(dest);
                    if ((mover == 1 ? sr - er : er - sr) + 32 == dbest) k = (mover == 1 ? sr : 14 - sr) + 1;
                }
                kbest = kbest > k ? kbest : k;
            }
            kbest = half_max(kbest);
            int nb = 0;
            uint8_t *dst = moves + (size_t)(base + (on ? s : 0)) * CCSP_GREEDY_MAX * 2;
            for (int j0 = 0; j0 < kmax; j0 += 32) {
                const int j = j0 + hl;
                bool keep = false; int id = 0, dest = 0;
                if (j < K) {
                    move_at(j, id, dest);
                    const int sr = human_row(LST(L, s, id, MG_SLOT - 1)), er = human_row(dest);
                    keep = (mover == 1 ? sr - er : er - sr) + 32 == dbest && (mover == 1 ? sr : 14 - sr) + 1 == kbest;
                }
                const uint32_t bits = (uint32_t)(__ballot(keep) >> (32 * half));
                const int rank = nb + __popc(bits & ((1u << hl) - 1u));
                if (keep && rank < CCSP_GREEDY_MAX) reinterpret_cast<uint16_t *>(dst)[rank] = (uint16_t)id | ((uint16_t)dest << 8);
                nb += __popc(bits);
            }
            if (on && hl == 0) count[base + s] = (uint8_t)(nb < CCSP_GREEDY_MAX ? nb : CCSP_GREEDY_MAX);
        }
    }
    MG_LAP(6);                                          // [6] write-out
}

// B5-B7: one lane per state; the record moves as two 16-byte accesses.
__global__ __launch_bounds__(256) void step_kernel(const ccsp_state *__restrict__ in, const uint8_t *__restrict__ player,
                                                   const uint8_t *__restrict__ mv, int n, ccsp_state *__restrict__ out,
                                                   uint8_t *__restrict__ winner, uint8_t *__restrict__ progress) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ccsp_sr a = ccsp_load_sr(in + i);
    const int pl = player[i];
    const uint16_t m = reinterpret_cast<const uint16_t *>(mv)[i];
    const ccsp_sr b = ccsp_place(a, pl, m & 0xFF, m >> 8);
    ccsp_store_sr(out + i, b);
    winner[i] = (uint8_t)ccsp_check_win(b.occ0, b.occ1);
    if (progress) {
        const uint16_t pr = (uint16_t)ccsp_progress(b, 1) | ((uint16_t)ccsp_progress(b, 2) << 8);
        reinterpret_cast<uint16_t *>(progress)[i] = pr;
    }
}

constexpr int ENC_THREADS = 256;
constexpr int ENC_STATES = 64;      // states per workgroup: 64 * 343 B image = 21952 B of LDS, 5488 float4 out

// C1: the 12 checkers of each state scatter their id+1 into a zeroed byte image in LDS (one lane per
// checker), then the workgroup converts bytes to float and streams 16-byte stores.
// REQ: the positions are the records of the free-running path's request buffer (ccsp_request: state at +0, kind at +32, player at +44);
// a record that asks for nothing (kind 0) gives an all-zero row
template <bool REQ>
__global__ __launch_bounds__(ENC_THREADS) void encode_kernel(const ccsp_state *__restrict__ states,
                                                             const uint8_t *__restrict__ player, int n,
                                                             float *__restrict__ planes) {
    __shared__ __attribute__((aligned(16))) uint8_t img[ENC_STATES * CCSP_PLANES];
    __shared__ uint8_t p2flag[ENC_STATES];
    const int tid = threadIdx.x;
    const long long base = (long long)blockIdx.x * ENC_STATES;
    const int here = (int)((n - base) < ENC_STATES ? (n - base) : ENC_STATES);
    uint32_t *img32 = reinterpret_cast<uint32_t *>(img);
    for (int i = tid; i < ENC_STATES * CCSP_PLANES / 4; i += ENC_THREADS) img32[i] = 0;
    __syncthreads();
    for (int t = tid; t < here * 12; t += ENC_THREADS) {
        const int sl = t / 12, k = t % 12;
        ccsp_sr a; int pl;
        if (REQ) {
            const ccsp_request *r = reinterpret_cast<const ccsp_request *>(states) + base + sl;
            if (r->kind == 0) { if (k == 0) p2flag[sl] = 0; continue; }
            a = ccsp_load_sr(&r->state); pl = (int)r->player;
        } else { a = ccsp_load_sr(states + base + sl); pl = player[base + sl]; }
        ccsp_scatter_checker(a, pl, k, &img[sl * CCSP_PLANES]);
        if (k == 0) p2flag[sl] = (pl == 2);
    }
    __syncthreads();
    const int total = here * CCSP_PLANES;                 // floats this workgroup writes
    float *dst = planes + base * CCSP_PLANES;             // base * 343 * 4 B: 16-byte aligned since ENC_STATES % 4 == 0
    for (int q = tid; q * 4 < total; q += ENC_THREADS) {
        const int e = q * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ee = e + j;
            const int sl = ee / CCSP_PLANES, rem = ee % CCSP_PLANES;
            v[j] = (ee < total) ? ((rem % 7 == 6) ? (float)p2flag[sl < here ? sl : 0] : (float)img[ee]) : 0.0f;
        }
        if (e + 3 < total) *reinterpret_cast<float4 *>(dst + e) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int j = 0; j < 4 && e + j < total; j++) dst[e + j] = v[j];
    }
}

// the compact answer of the free-running path from a full policy row: one workgroup per request, one lane per legal move
__global__ __launch_bounds__(CCSP_REQUEST_MOVES) void gather_priors_kernel(const ccsp_request *__restrict__ req, const uint16_t *__restrict__ moves,
                                                                           const double *__restrict__ p, double *__restrict__ pk) {
    const int i = blockIdx.x, j = threadIdx.x;
    if (req[i].kind == 0 || j >= (int)req[i].k || j >= CCSP_MAX_MOVES) return;
    const int a = moves[(size_t)i * CCSP_REQUEST_MOVES + j] & 0x1FF;         // (caller-owned rows: an entry that is no action index answers 0.0)
    pk[(size_t)i * CCSP_REQUEST_MOVES + j] = a < CCSP_NUM_ACTIONS ? p[(size_t)i * CCSP_NUM_ACTIONS + a] : 0.0;
}

// test hook: spec.hash_eval / forward_eval / the uniform stub (the fused path's built-in evaluators) answering requests
__global__ __launch_bounds__(CCSP_REQUEST_MOVES) void table_eval_kernel(int evaluator, const ccsp_request *__restrict__ req, const uint16_t *__restrict__ moves,
                                                                        double *__restrict__ pk, float *__restrict__ v) {
    const int i = blockIdx.x, j = threadIdx.x;
    if (req[i].kind == 0) return;
    const ccsp_sr st = ccsp_load_sr(&req[i].state);
    const int player = (int)req[i].player;
    const uint64_t key = evaluator == CCSP_EVAL_HASH ? ccsp_state_key(st, player) : 0;
    if (j == 0) v[i] = evaluator == CCSP_EVAL_HASH ? ccsp_hash_value(key) : (evaluator == CCSP_EVAL_FORWARD ? ccsp_forward_value(st, player) : 0.0f);
    if (j >= (int)req[i].k || j >= CCSP_MAX_MOVES) return;
    const int idx = moves[(size_t)i * CCSP_REQUEST_MOVES + j] & 0x1FF;
    if (idx >= CCSP_NUM_ACTIONS) { pk[(size_t)i * CCSP_REQUEST_MOVES + j] = 0.0; return; }
    pk[(size_t)i * CCSP_REQUEST_MOVES + j] = evaluator == CCSP_EVAL_HASH ? ccsp_hash_prior(key, idx)
                                            : (evaluator == CCSP_EVAL_FORWARD ? ccsp_forward_prior(st, player, idx / CCSP_NCELL, idx % CCSP_NCELL) : 1.0 / 294.0);
}

int g_cap = MG_STACK;                 // stack entries a lane may use (test hook below; MG_STACK in production)

}  // namespace

extern "C" {

// Test hook: limit the per-lane hop-search stack to `cap` entries (6 .. MG_STACK; anything else restores the default), so
// that searches overflow into the big-stack redo path, which no real position reaches.  Returns the value in force.
int ccsp_debug_movegen_stack_cap(int cap) {
    g_cap = (cap >= 6 && cap <= MG_STACK) ? cap : MG_STACK;
    return g_cap;
}

#ifdef MG_STAMPS
int ccsp_debug_movegen_stamps(unsigned long long *out, int clear) {
    CCSP_HIPCHK(hipDeviceSynchronize());
    CCSP_HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(mg_stamps), sizeof(unsigned long long) * 8));
    if (clear) { unsigned long long z[8] = {0}; CCSP_HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mg_stamps), z, sizeof(z))); }
    return CCSP_OK;
}
#endif

int ccsp_movegen(const ccsp_state *s, const uint8_t *player, int n, uint8_t *moves, uint8_t *count,
                 uint64_t *dest_mask, void *stream) {
    if (n < 0 || (n > 0 && (!s || !player || !moves || !count))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    const int grid = (n + MG_STATES - 1) / MG_STATES;
    hipLaunchKernelGGL(movegen_kernel<false>, dim3(grid), dim3(MG_THREADS), 0, (hipStream_t)stream, s, player, n, moves, count, dest_mask, g_cap);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_movegen_packed(const ccsp_state *s, const uint8_t *player, int n, uint8_t *moves, uint8_t *count,
                        uint64_t *dest_mask, void *stream) {
    if (n < 0 || (n > 0 && (!s || !player || !moves || !count))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    const int grid = (n + MG_STATES - 1) / MG_STATES;
    hipLaunchKernelGGL((movegen_kernel<false, true>), dim3(grid), dim3(MG_THREADS), 0, (hipStream_t)stream, s, player, n, moves, count, dest_mask, g_cap);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_greedy_best(const ccsp_state *s, const uint8_t *player, int n, uint8_t *best, uint8_t *count, void *stream) {
    if (n < 0 || (n > 0 && (!s || !player || !best || !count))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    const int grid = (n + MG_STATES - 1) / MG_STATES;
    hipLaunchKernelGGL(movegen_kernel<true>, dim3(grid), dim3(MG_THREADS), 0, (hipStream_t)stream, s, player, n, best, count,
                       (uint64_t *)nullptr, g_cap);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_step(const ccsp_state *in, const uint8_t *player, const uint8_t *mv, int n, ccsp_state *out,
              uint8_t *winner, uint8_t *progress, void *stream) {
    if (n < 0 || (n > 0 && (!in || !player || !mv || !out || !winner))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    hipLaunchKernelGGL(step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, player, mv, n, out, winner, progress);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_encode(const ccsp_state *s, const uint8_t *player, int n, float *planes, void *stream) {
    if (n < 0 || (n > 0 && (!s || !player || !planes))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    hipLaunchKernelGGL(encode_kernel<false>, dim3((n + ENC_STATES - 1) / ENC_STATES), dim3(ENC_THREADS), 0, (hipStream_t)stream, s, player, n, planes);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_encode_requests(const ccsp_request *req, int n, float *planes, void *stream) {
    if (n < 0 || (n > 0 && (!req || !planes))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    hipLaunchKernelGGL(encode_kernel<true>, dim3((n + ENC_STATES - 1) / ENC_STATES), dim3(ENC_THREADS), 0, (hipStream_t)stream,
                       reinterpret_cast<const ccsp_state *>(req), (const uint8_t *)nullptr, n, planes);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_gather_priors(const ccsp_request *req, const uint16_t *moves, const double *p, int n, double *pk, void *stream) {
    if (n < 0 || (n > 0 && (!req || !moves || !p || !pk))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    hipLaunchKernelGGL(gather_priors_kernel, dim3(n), dim3(CCSP_REQUEST_MOVES), 0, (hipStream_t)stream, req, moves, p, pk);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

int ccsp_debug_table_eval(int evaluator, const ccsp_request *req, const uint16_t *moves, int n, double *pk, float *v, void *stream) {
    if (n < 0 || evaluator < CCSP_EVAL_UNIFORM || evaluator > CCSP_EVAL_FORWARD || (n > 0 && (!req || !moves || !pk || !v))) return CCSP_EINVAL;
    if (n == 0) return CCSP_OK;
    hipLaunchKernelGGL(table_eval_kernel, dim3(n), dim3(CCSP_REQUEST_MOVES), 0, (hipStream_t)stream, evaluator, req, moves, pk, v);
    CCSP_HIPCHK(hipGetLastError());
    return CCSP_OK;
}

}  // extern "C"
