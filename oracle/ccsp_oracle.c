/*
 * ccsp_oracle.c -- CPU restatement of the reference's self-play path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (chinesecheckersagent_amd/) never does.
 *
 * Parity: PINNED.  Every function below is checked against vectors produced by running the
 * reference itself (oracle/harness/gen_golden.py imports /root/reference unmodified):
 * tests/golden/rules.npz + wins.npz (B1-B8, C1, S1), codec.npy (C2), tree.json (T1-T4, S2),
 * games.json (S3, O1), rng.json (the draw-substitution spec, oracle/harness/spec.py).
 * Exception: the evaluator arithmetic (row N1, Keras/TF) has no executable oracle here --
 * see oracle/net_oracle.py ("parity unpinned" at the Keras boundary).
 *
 * It follows the reference's data flow on purpose (cell array + id->position table, recursive
 * hop search, eager child states, running-max tie list) and shares no code with the HIP side,
 * which uses bitboards / stack-free traversal / closed-form tie sets.  All file:line citations
 * are relative to /root/reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  Floating point is
 * IEEE binary64 add/sub/mul/div/sqrt only, except pow() for pi (MCTS.py:132), which the
 * reference also takes from libm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define W7 7
#define NCELL 49
#define NCHK 6
#define NACT 294
#define MAXMV 126
#define NONE 255

/* config.py:3-40 */
#define TOTAL_HIST_MOVES 16
#define UNIQUE_DEST_LIMIT 3
#define DIRICHLET_ALPHA 0.03
#define DIR_NOISE_FACTOR 0.25
#define PROGRESS_MOVE_LIMIT 100
#define C_PUCT 3.5
#define EPSILON 1e-5
#define TOTAL_MOVES_TILL_TAU0 16
#define INITIAL_RANDOM_MOVES 6
#define BOARD_HIST_MOVES 3

/* board.py:33-40: N, E, SE, S, W, NW */
static const int DROW[6] = {-1, 0, 1, 1, 0, -1};
static const int DCOL[6] = {0, 1, 1, 0, -1, -1};

typedef struct {
    uint8_t cell[NCELL];      /* board[:, :, 0]: 0 / 1 / 2                       board.py:19-26 */
    uint8_t pos[2][NCHK];     /* checkers_pos[player][id] as r*7+c               board.py:42-46 */
    uint8_t last[4];          /* hist_moves[-1] (from,to), hist_moves[-2] (from,to); NONE while the
                                 matching history plane board[:, :, 1|2] is all-zero (utils.py:137) */
} board_t;

/* ------------------------------------------------------------------------------------------ */
/* draw-substitution spec: a second, independent copy of oracle/harness/spec.py                */

#define GOLD 0x9E3779B97F4A7C15ULL
enum { P_SELECT = 1, P_OPENING = 2, P_DIRICHLET = 3, P_SAMPLE = 4, P_INIT = 5, P_ROLLOUT = 6, P_GREEDY = 7 };

uint64_t orc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t orc_rng(uint64_t seed, uint64_t game, uint32_t ply, uint32_t sim, uint32_t level, uint32_t purpose) {
    uint64_t h = orc_mix64(seed + GOLD);
    h = orc_mix64(h + game + GOLD);
    h = orc_mix64(h + (((uint64_t)ply << 32) | sim));
    h = orc_mix64(h + (((uint64_t)level << 32) | purpose));
    return h;
}

uint32_t orc_choice(uint64_t u, uint32_t n) { return (uint32_t)(((unsigned __int128)u * n) >> 64); }

static double uniform_open(uint64_t u) { return ((double)(u >> 12) + 0.5) * 2.220446049250313e-16; }

static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

#define LN2_HI 6.93147180369123816490e-01
#define LN2_LO 1.90821492927058770002e-10
#define INV_LN2 1.44269504088896338700e+00

double orc_det_log(double x) {
    uint64_t b = to_bits(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    double m = from_bits((b & 0xFFFFFFFFFFFFFULL) | (1023ULL << 52));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 23.0;
    for (int k = 21; k >= 3; k -= 2) p = p * z + 1.0 / (double)k;
    p = p * z + 1.0;
    double r = (2.0 * s) * p;
    double fe = (double)e;
    return (fe * LN2_HI + r) + fe * LN2_LO;
}

double orc_det_exp(double x) {
    if (x < -708.0) return 0.0;
    double t = x * INV_LN2 + 0.5;
    double fk = floor(t);
    int k = (int)fk;
    double r = (x - fk * LN2_HI) - fk * LN2_LO;
    double c[14], f = 1.0;
    for (int i = 1; i <= 14; i++) { f = f * (double)i; c[i - 1] = 1.0 / f; }
    double p = c[13];
    for (int i = 12; i >= 0; i--) p = p * r + c[i];
    p = p * r + 1.0;
    return p * from_bits((uint64_t)(k + 1023) << 52);
}

double orc_gamma_small(uint64_t seed, uint64_t game, uint32_t ply, uint32_t edge, double alpha) {
    for (uint32_t t = 0;; t++) {
        double U = uniform_open(orc_rng(seed, game, ply, edge, 2 * t, P_DIRICHLET));
        double V = -orc_det_log(uniform_open(orc_rng(seed, game, ply, edge, 2 * t + 1, P_DIRICHLET)));
        if (U <= 1.0 - alpha) {
            double X = orc_det_exp(orc_det_log(U) / alpha);
            if (X <= V) return X;
        } else {
            double Y = -orc_det_log((1.0 - U) / alpha);
            double X = orc_det_exp(orc_det_log(1.0 - alpha + alpha * Y) / alpha);
            if (X <= V + Y) return X;
        }
    }
}

void orc_dirichlet(uint64_t seed, uint64_t game, uint32_t ply, int k, double alpha, double *out) {
    double s = 0.0;
    for (int i = 0; i < k; i++) { out[i] = orc_gamma_small(seed, game, ply, (uint32_t)i, alpha); s = s + out[i]; }
    for (int i = 0; i < k; i++) out[i] = (s == 0.0) ? 1.0 / (double)k : out[i] / s;
}

/* np.random.choice(294, p=pi) stand-in (MCTS.py:140): sequential cumsum / last, first cdf > u */
int orc_sample_index(uint64_t u64, const double *p, int n) {
    double u = (double)(u64 >> 11) * 1.1102230246251565e-16;
    double cdf[NACT];
    double s = 0.0;
    for (int i = 0; i < n; i++) { s = s + p[i]; cdf[i] = s; }
    for (int i = 0; i < n; i++) if (cdf[i] / cdf[n - 1] > u) return i;
    return n - 1;
}

void orc_pick_distinct(uint64_t seed, uint64_t game, int n, int k, int base, int *out) {
    int pool[NCELL];
    for (int i = 0; i < n; i++) pool[i] = i;
    for (int i = 0; i < k; i++) {
        int j = i + (int)orc_choice(orc_rng(seed, game, 0, (uint32_t)(base + i), 0, P_INIT), (uint32_t)(n - i));
        int t = pool[i]; pool[i] = pool[j]; pool[j] = t;
        out[i] = pool[i];
    }
}

/* table evaluators (spec.py: uniform_eval / hash_eval) */
#define HASH_SALT 0xC0FFEE1234567ULL
uint64_t orc_state_key(const uint8_t *pos12, int player) {
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 8; i++) a |= (uint64_t)pos12[i] << (8 * i);
    for (int i = 0; i < 4; i++) b |= (uint64_t)pos12[8 + i] << (8 * i);
    b |= (uint64_t)(player & 0xFF) << 32;
    uint64_t h = orc_mix64(HASH_SALT + GOLD);
    h = orc_mix64(h ^ a);
    h = orc_mix64(h + b);
    return h;
}

void orc_hash_eval(const uint8_t *pos12, int player, double *p, float *v) {
    uint64_t h = orc_state_key(pos12, player);
    for (int i = 0; i < NACT; i++) p[i] = (double)((orc_mix64(h + (uint64_t)i + 1) >> 40) + 1) * 1.862645149230957e-09;
    *v = (float)(((double)(orc_mix64(h + 1000) >> 48) - 32768.0) / 32768.0);
}

static int forward_score(int cell, int player) {
    int r = cell / 7, c = cell % 7;
    return player == 1 ? (6 - r) + c : r + (6 - c);
}

void orc_forward_eval(const uint8_t *pos12, int player, double *p, float *v) {
    for (int id = 0; id < 6; id++) {
        int o = forward_score(pos12[(player - 1) * 6 + id], player);
        for (int d = 0; d < 49; d++)
            p[id * 49 + d] = (double)(1 << (forward_score(d, player) - o + 12)) * 5.9604644775390625e-08;
    }
    int own = 0, opp = 0;
    for (int i = 0; i < 6; i++) {
        own += forward_score(pos12[(player - 1) * 6 + i], player);
        opp += forward_score(pos12[(2 - player) * 6 + i], 3 - player);
    }
    *v = (float)((double)(own - opp) / 4.0);
}

/* ------------------------------------------------------------------------------------------ */
/* B1: board construction                                                                      */

static void board_from_pos12(board_t *b, const uint8_t *pos12, const uint8_t *last4) {
    memset(b->cell, 0, NCELL);
    for (int pl = 0; pl < 2; pl++)
        for (int i = 0; i < NCHK; i++) {
            b->pos[pl][i] = pos12[pl * 6 + i];
            b->cell[pos12[pl * 6 + i]] = (uint8_t)(pl + 1);
        }
    for (int i = 0; i < 4; i++) b->last[i] = last4 ? last4[i] : NONE;
}

/* Board.__init__ (board.py:10-57) */
void orc_initial_pos12(uint8_t *pos12) {
    static const uint8_t init[12] = {42, 35, 43, 28, 36, 44, 6, 13, 5, 20, 12, 4};
    memcpy(pos12, init, 12);
}

/* Board.randomise_initial_state (board.py:61-85) with the substituted draw (spec.pick_distinct) */
void orc_randomised_pos12(uint64_t seed, uint64_t game, uint8_t *pos12) {
    int idx[12];
    orc_pick_distinct(seed, game, NCELL, 12, 0, idx);
    for (int i = 0; i < 12; i++) pos12[i] = (uint8_t)idx[i];
}

/* spec.near_win_position: crafted fixtures only */
void orc_near_win_pos12(uint64_t seed, uint64_t game, int who, uint8_t *pos12) {
    static const uint8_t T[2][6] = {{4, 5, 6, 12, 13, 20}, {28, 35, 36, 42, 43, 44}};
    const uint8_t *tgt = T[who - 1];
    int j = (int)orc_choice(orc_rng(seed, game, 0, 100, 0, P_INIT), 6);
    int pool[NCELL], np_ = 0;
    for (int c = 0; c < NCELL; c++) {
        int in = 0;
        for (int i = 0; i < 6; i++) if (tgt[i] == c) in = 1;
        if (!in) pool[np_++] = c;
    }
    int picks[7];
    orc_pick_distinct(seed, game, np_, 7, 200, picks);
    uint8_t mine[6], other[6];
    for (int i = 0; i < 6; i++) mine[i] = tgt[i];
    mine[j] = (uint8_t)pool[picks[0]];
    for (int i = 0; i < 6; i++) other[i] = (uint8_t)pool[picks[1 + i]];
    memcpy(pos12, who == 1 ? mine : other, 6);
    memcpy(pos12 + 6, who == 1 ? other : mine, 6);
}

/* board_utils.is_valid_pos (board_utils.py:15-16) */
static int is_valid_pos(int r, int c) { return r >= 0 && r < W7 && c >= 0 && c < W7; }

/* B6: Board.check_win (board.py:89-111) */
static int check_win(const board_t *b) {
    int one_win = 1, two_win = 1;
    for (int k = W7 - 3; k < W7; k++) {
        if (one_win)                                    /* diagonal(k): cells (i, i+k) */
            for (int i = 0; i + k < W7; i++) if (b->cell[i * W7 + i + k] != 1) one_win = 0;
        if (two_win)                                    /* diagonal(-k): cells (i+k, i) */
            for (int i = 0; i + k < W7; i++) if (b->cell[(i + k) * W7 + i] != 2) two_win = 0;
        if (!one_win && !two_win) return 0;
    }
    return one_win ? 1 : 2;
}

/* B7: Board.player_progress (board.py:254-266) */
static int player_progress(const board_t *b, int player) {
    int n = 0;
    for (int k = W7 - 3; k < W7; k++)
        for (int i = 0; i + k < W7; i++) {
            int cell = (player == 1) ? i * W7 + i + k : (i + k) * W7 + i;
            if (b->cell[cell] == player) n++;
        }
    return n;
}

/* B3: Board.valid_checker_jump_moves (board.py:166-211) -- recursive, pre-order */
static void jump_moves(board_t *b, uint8_t *out, int *n, uint8_t *check_map, int cur) {
    int curr_row = cur / W7, curr_col = cur % W7;
    for (int d = 0; d < 6; d++) {
        int step = 1;
        int row = curr_row + DROW[d], col = curr_col + DCOL[d];
        int valid = 1;
        for (;;) {                                           /* board.py:179-187 */
            if (!is_valid_pos(row, col)) { valid = 0; break; }
            if (b->cell[row * W7 + col] != 0) break;
            step++; row += DROW[d]; col += DCOL[d];
        }
        if (!valid) continue;
        for (int i = 0; i < step; i++) {                     /* board.py:193-198 */
            row += DROW[d]; col += DCOL[d];
            if (!is_valid_pos(row, col) || b->cell[row * W7 + col] != 0) { valid = 0; break; }
        }
        if (!valid) continue;
        if (check_map[row * W7 + col]) continue;             /* board.py:205 */
        out[(*n)++] = (uint8_t)(row * W7 + col);
        check_map[row * W7 + col] = 1;
        jump_moves(b, out, n, check_map, row * W7 + col);
    }
}

/* B2: Board.valid_checker_moves (board.py:139-162); returns destinations in reference order */
static int checker_moves(board_t *b, int player, int cpos, uint8_t *out) {
    uint8_t check_map[NCELL];
    int n = 0;
    memset(check_map, 0, NCELL);
    check_map[cpos] = 1;
    for (int d = 0; d < 6; d++) {
        int row = cpos / W7 + DROW[d], col = cpos % W7 + DCOL[d];
        if (!is_valid_pos(row, col)) continue;
        if (b->cell[row * W7 + col] == 0) { out[n++] = (uint8_t)(row * W7 + col); check_map[row * W7 + col] = 1; }
    }
    b->cell[cpos] = 0;                                       /* board.py:158 */
    jump_moves(b, out, &n, check_map, cpos);
    b->cell[cpos] = (uint8_t)player;                         /* board.py:160 */
    return n;
}

/* B4: Board.get_valid_moves (board.py:215-222): per checker id, then MCTS.py:97-99's flattening */
static int valid_moves(board_t *b, int player, uint8_t (*moves)[2], int *per_checker) {
    int n = 0;
    for (int id = 0; id < NCHK; id++) {
        uint8_t dest[32];
        int k = checker_moves(b, player, b->pos[player - 1][id], dest);
        if (per_checker) per_checker[id] = k;
        for (int i = 0; i < k; i++) { moves[n][0] = (uint8_t)id; moves[n][1] = dest[i]; n++; }
    }
    return n;
}

/* B5: Board.place (board.py:226-250); returns check_win() */
static int place(board_t *b, int player, int id, int dest) {
    int origin = b->pos[player - 1][id];
    uint8_t t = b->cell[origin]; b->cell[origin] = b->cell[dest]; b->cell[dest] = t;    /* 231-232 */
    b->pos[player - 1][id] = (uint8_t)dest;                                             /* 235-240 */
    b->last[2] = b->last[0]; b->last[3] = b->last[1];                                   /* 243-248 */
    b->last[0] = (uint8_t)origin; b->last[1] = (uint8_t)dest;
    return check_win(b);
}

/* C1: utils.to_model_input (utils.py:101-160); values are small integers, written as uint8 */
static void to_model_input(const board_t *b, int cur_player, uint8_t *out /* [7][7][7] */) {
    int op_player = 3 - cur_player;
    uint8_t cur_layer[NCELL], op_layer[NCELL];
    memset(cur_layer, 0, NCELL); memset(op_layer, 0, NCELL); memset(out, 0, 343);
    for (int i = 0; i < NCHK; i++) {
        cur_layer[b->pos[cur_player - 1][i]] = (uint8_t)(i + 1);
        op_layer[b->pos[op_player - 1][i]] = (uint8_t)(i + 1);
    }
    for (int c = 0; c < NCELL; c++) { out[c * 7 + 0] = cur_layer[c]; out[c * 7 + 1] = op_layer[c]; }
    int moved_player = op_player;
    for (int ch = 1; ch < BOARD_HIST_MOVES; ch++) {
        if (b->last[(ch - 1) * 2] == NONE) break;            /* utils.py:137 */
        int orig = b->last[(ch - 1) * 2], dest = b->last[(ch - 1) * 2 + 1];
        uint8_t *layer = (moved_player == cur_player) ? cur_layer : op_layer;
        uint8_t v = layer[dest]; layer[dest] = layer[orig]; layer[orig] = v;
        moved_player = 3 - moved_player;
        for (int c = 0; c < NCELL; c++) { out[c * 7 + ch * 2] = cur_layer[c]; out[c * 7 + ch * 2 + 1] = op_layer[c]; }
    }
    if (cur_player == 2) for (int c = 0; c < NCELL; c++) out[c * 7 + 6] = 1;
}

/* C2: utils.encode_checker_index / decode_checker_index (utils.py:164-183) */
int orc_encode_index(int id, int r, int c) { return id * W7 * W7 + r * W7 + c; }
void orc_decode_index(int idx, int *id, int *r, int *c) {
    *id = idx / (W7 * W7);
    int off = idx % (W7 * W7);
    *r = off / W7; *c = off % W7;
}

/* ---- flat entry points for the rules (ctypes) ---- */

int orc_movegen(const uint8_t *pos12, int player, uint8_t *moves /* [126][2] */) {
    board_t b; board_from_pos12(&b, pos12, NULL);
    return valid_moves(&b, player, (uint8_t (*)[2])moves, NULL);
}

int orc_step(const uint8_t *pos12, const uint8_t *last4, int player, int id, int dest,
             uint8_t *npos12, uint8_t *nlast4, uint8_t *nboard /* [7][7][3] or NULL */) {
    board_t b; board_from_pos12(&b, pos12, last4);
    board_t prev = b;
    int w = place(&b, player, id, dest);
    memcpy(npos12, b.pos, 12); memcpy(nlast4, b.last, 4);
    if (nboard) {            /* Board.board after place: plane 0 new, plane 1 = previous plane 0 */
        for (int c = 0; c < NCELL; c++) { nboard[c * 3] = b.cell[c]; nboard[c * 3 + 1] = prev.cell[c]; nboard[c * 3 + 2] = 0; }
    }
    return w;
}

int orc_check_win(const uint8_t *pos12) { board_t b; board_from_pos12(&b, pos12, NULL); return check_win(&b); }
int orc_progress(const uint8_t *pos12, int player) { board_t b; board_from_pos12(&b, pos12, NULL); return player_progress(&b, player); }
void orc_planes(const uint8_t *pos12, const uint8_t *last4, int player, uint8_t *out) {
    board_t b; board_from_pos12(&b, pos12, last4); to_model_input(&b, player, out);
}

/* S1: selfplay.make_random_move (selfplay.py:83-104) on the substituted stream.
 * draws: checker (retry while it has no move), then destination.  Returns the move. */
static int random_move(board_t *b, int player, uint64_t seed, uint64_t game, uint32_t ply, uint32_t purpose,
                       uint32_t sim, int level_is_counter, uint32_t *counter, int *id_out, int *dest_out) {
    uint8_t moves[MAXMV][2]; int per[NCHK];
    int n = valid_moves(b, player, moves, per);
    if (n == 0) return 0;                       /* the reference would spin forever here (selfplay.py:96) */
    int id;
    for (;;) {
        uint64_t u = level_is_counter ? orc_rng(seed, game, ply, sim, (*counter)++, purpose)
                                      : orc_rng(seed, game, ply, (*counter)++, 0, purpose);
        id = (int)orc_choice(u, NCHK);
        if (per[id] > 0) break;
    }
    uint64_t u = level_is_counter ? orc_rng(seed, game, ply, sim, (*counter)++, purpose)
                                  : orc_rng(seed, game, ply, (*counter)++, 0, purpose);
    int j = (int)orc_choice(u, (uint32_t)per[id]);
    int off = 0;
    for (int i = 0; i < id; i++) off += per[i];
    *id_out = id; *dest_out = moves[off + j][1];
    return 1;
}

int orc_random_move(const uint8_t *pos12, int player, uint64_t seed, uint64_t game, uint32_t ply, int *id, int *dest) {
    board_t b; board_from_pos12(&b, pos12, NULL);
    uint32_t counter = 0;
    return random_move(&b, player, seed, game, ply, P_OPENING, 0, 0, &counter, id, dest);
}

/* ------------------------------------------------------------------------------------------ */
/* T1: Node / Edge (MCTS.py:13-37)                                                             */

typedef struct {
    board_t state;
    int player;            /* currPlayer */
    int first_edge, n_edges;
} node_t;

typedef struct {
    int out_node;
    int mover;             /* Edge.currPlayer = inNode.currPlayer */
    uint8_t id, dest;
    int N;
    double W, Q, P;
} edge_t;

typedef void (*orc_eval_fn)(const uint8_t *planes343, const uint8_t *pos12, int player, double *p, float *v, void *user);

typedef struct {
    node_t *nodes; int n_nodes, cap_nodes;
    edge_t *edges; int n_edges, cap_edges;
    uint64_t seed, game; uint32_t ply;
    int evaluator;                  /* 0 uniform, 1 hash, 2 forward, 3 rollout (config 2b), 4 callback */
    orc_eval_fn fn; void *user;
    long evals, terminals;
    uint32_t sim;                   /* current simulation (P_SELECT key); rollout uses sim+1, root 0 */
    int max_depth; long sum_depth;
} tree_t;

static int new_node(tree_t *t, const board_t *st, int player) {
    if (t->n_nodes == t->cap_nodes) { t->cap_nodes *= 2; t->nodes = (node_t *)realloc(t->nodes, sizeof(node_t) * (size_t)t->cap_nodes); }
    node_t *n = &t->nodes[t->n_nodes];
    n->state = *st; n->player = player; n->first_edge = -1; n->n_edges = 0;
    return t->n_nodes++;
}

/* config 2b: random playout value (no reference counterpart; defined in DESIGN.md) */
static float rollout_value(tree_t *t, const board_t *leaf, int leaf_player, uint32_t sim_key) {
    board_t b = *leaf; int pl = leaf_player; uint32_t counter = 0;
    for (int step = 0; step < 64; step++) {
        int id, dest;
        if (!random_move(&b, pl, t->seed, t->game, t->ply, P_ROLLOUT, sim_key, 1, &counter, &id, &dest)) return 0.0f;
        int w = place(&b, pl, id, dest);
        if (w) return (w == leaf_player) ? 1.0f : -1.0f;
        pl = 3 - pl;
    }
    return 0.0f;
}

static void evaluate(tree_t *t, const board_t *st, int player, uint32_t sim_key, double *p, float *v) {
    t->evals++;
    if (t->evaluator == 0) { for (int i = 0; i < NACT; i++) p[i] = 1.0 / 294.0; *v = 0.0f; }
    else if (t->evaluator == 1) orc_hash_eval(&st->pos[0][0], player, p, v);
    else if (t->evaluator == 2) orc_forward_eval(&st->pos[0][0], player, p, v);
    else if (t->evaluator == 3) { for (int i = 0; i < NACT; i++) p[i] = 1.0 / 294.0; *v = rollout_value(t, st, player, sim_key); }
    else { uint8_t planes[343]; to_model_input(st, player, planes); t->fn(planes, &st->pos[0][0], player, p, v, t->user); }
}

/* T2: MCTS.moveToLeaf (MCTS.py:49-76) */
static int move_to_leaf(tree_t *t, int root, int *crumbs, int *n_crumbs) {
    int cur = root; *n_crumbs = 0;
    uint32_t level = 0;
    while (t->nodes[cur].n_edges != 0) {
        node_t *nd = &t->nodes[cur];
        double maxQU = -INFINITY;
        int chosen[MAXMV], n_chosen = 0;
        long N_sum = 0;
        for (int i = 0; i < nd->n_edges; i++) N_sum += t->edges[nd->first_edge + i].N;
        double sq = sqrt((double)N_sum);
        for (int i = 0; i < nd->n_edges; i++) {
            edge_t *e = &t->edges[nd->first_edge + i];
            double U = C_PUCT * e->P * sq / (1.0 + (double)e->N);      /* left to right, MCTS.py:62 */
            double QU = e->Q + U;
            if (QU > maxQU) { maxQU = QU; chosen[0] = nd->first_edge + i; n_chosen = 1; }
            else if (fabs(QU - maxQU) < EPSILON) chosen[n_chosen++] = nd->first_edge + i;
        }
        uint64_t u = orc_rng(t->seed, t->game, t->ply, t->sim, level++, P_SELECT);     /* MCTS.py:72 */
        int e = chosen[orc_choice(u, (uint32_t)n_chosen)];
        crumbs[(*n_crumbs)++] = e;
        cur = t->edges[e].out_node;
    }
    return cur;
}

/* T3: MCTS.expandAndBackUp (MCTS.py:79-118) */
static void expand_and_backup(tree_t *t, int leaf, const int *crumbs, int n_crumbs, uint32_t sim_key) {
    int leaf_player = t->nodes[leaf].player;
    int winner = check_win(&t->nodes[leaf].state);
    if (winner) {
        t->terminals++;
        for (int i = 0; i < n_crumbs; i++) {
            edge_t *e = &t->edges[crumbs[i]];
            int direction = (e->mover == leaf_player) ? -1 : 1;
            e->N += 1; e->W += 1 * direction; e->Q = e->W / (double)e->N;
        }
        return;
    }
    double p[NACT]; float v;
    evaluate(t, &t->nodes[leaf].state, leaf_player, sim_key, p, &v);
    board_t st = t->nodes[leaf].state;
    uint8_t moves[MAXMV][2];
    int n = valid_moves(&st, leaf_player, moves, NULL);
    if (t->n_edges + n > t->cap_edges) {
        while (t->n_edges + n > t->cap_edges) t->cap_edges *= 2;
        t->edges = (edge_t *)realloc(t->edges, sizeof(edge_t) * (size_t)t->cap_edges);
    }
    int first = t->n_edges;
    for (int i = 0; i < n; i++) {
        int id = moves[i][0], dest = moves[i][1];
        int prior_index = id * NCELL + dest;                         /* utils.encode_checker_index */
        board_t next = st;                                           /* copy.deepcopy, MCTS.py:104 */
        place(&next, leaf_player, id, dest);
        int child = new_node(t, &next, 3 - leaf_player);
        edge_t *e = &t->edges[t->n_edges++];
        e->out_node = child; e->mover = leaf_player; e->id = (uint8_t)id; e->dest = (uint8_t)dest;
        e->N = 0; e->W = 0.0; e->Q = 0.0; e->P = p[prior_index];
    }
    t->nodes[leaf].first_edge = first; t->nodes[leaf].n_edges = n;
    for (int i = 0; i < n_crumbs; i++) {
        edge_t *e = &t->edges[crumbs[i]];
        int direction = (e->mover == leaf_player) ? 1 : -1;
        e->N += 1; e->W += (double)v * direction; e->Q = e->W / (double)e->N;
    }
}

/* np.sum over 294 float64 = NumPy pairwise summation */
static double pairwise_sum(const double *a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

static uint64_t digest_node(tree_t *t, int node, uint64_t h, long *nodes, long *edges) {
    node_t *nd = &t->nodes[node];
    (*nodes)++;
    for (int i = 0; i < nd->n_edges; i++) {
        edge_t *e = &t->edges[nd->first_edge + i];
        (*edges)++;
        h = orc_mix64(h ^ (uint64_t)e->N); h = orc_mix64(h ^ to_bits(e->W)); h = orc_mix64(h ^ to_bits(e->P));
    }
    for (int i = 0; i < nd->n_edges; i++) {
        int c = t->edges[nd->first_edge + i].out_node;
        if (t->nodes[c].n_edges) h = digest_node(t, c, h, nodes, edges);
    }
    return h;
}

typedef struct {
    int n_root; int N[MAXMV]; double W[MAXMV], Q[MAXMV], P[MAXMV]; uint8_t id[MAXMV], dest[MAXMV];
    double pi[NACT]; int chosen_id, chosen_dest;
    long evals, terminals, nodes, edges; uint64_t digest; int max_depth; long sum_depth;
} orc_search_out;

/* S2 + T4: selfplay.make_move (selfplay.py:107-133) + MCTS.search (MCTS.py:121-153).
 * Returns the chosen child state in *next. */
/* arena != 0: AiPlayer.decide_move (player.py:139-166): MCTS(Node(board, player)).search() with NO root
 * pre-expansion and NO Dirichlet noise -- the first simulation finds the root as a leaf and expands it. */
static int make_move_ex(const board_t *root_state, int player, uint64_t seed, uint64_t game, uint32_t ply,
                        int sims, int det_tau, int evaluator, orc_eval_fn fn, void *user,
                        board_t *next, orc_search_out *out, double *pi_out, int arena);

static int make_move(const board_t *root_state, int player, uint64_t seed, uint64_t game, uint32_t ply,
                     int sims, int det_tau, int evaluator, orc_eval_fn fn, void *user,
                     board_t *next, orc_search_out *out, double *pi_out) {
    return make_move_ex(root_state, player, seed, game, ply, sims, det_tau, evaluator, fn, user, next, out, pi_out, 0);
}

static int make_move_ex(const board_t *root_state, int player, uint64_t seed, uint64_t game, uint32_t ply,
                        int sims, int det_tau, int evaluator, orc_eval_fn fn, void *user,
                        board_t *next, orc_search_out *out, double *pi_out, int arena) {
    tree_t t; memset(&t, 0, sizeof t);
    t.cap_nodes = 4096; t.nodes = (node_t *)malloc(sizeof(node_t) * (size_t)t.cap_nodes);
    t.cap_edges = 4096; t.edges = (edge_t *)malloc(sizeof(edge_t) * (size_t)t.cap_edges);
    t.seed = seed; t.game = game; t.ply = ply; t.evaluator = evaluator; t.fn = fn; t.user = user;
    int root = new_node(&t, root_state, player);
    int *crumbs = (int *)malloc(sizeof(int) * (size_t)(sims + 2));
    int k = 0;
    int rc = 0;
    if (!arena) {
    expand_and_backup(&t, root, crumbs, 0, 0);                                   /* selfplay.py:117 */
    k = t.nodes[root].n_edges;
    if (k == 0) { rc = -1; goto done; }                                          /* assert, selfplay.py:118 */
    }
    if (!arena) {
        double noise[MAXMV];
        orc_dirichlet(seed, game, ply, k, DIRICHLET_ALPHA, noise);               /* selfplay.py:121 */
        for (int i = 0; i < k; i++) {
            edge_t *e = &t.edges[t.nodes[root].first_edge + i];
            e->P *= (1. - DIR_NOISE_FACTOR);                                     /* selfplay.py:123 */
            e->P += DIR_NOISE_FACTOR * noise[i];                                 /* selfplay.py:124 */
        }
    }
    for (int i = 0; i < sims; i++) {                                             /* MCTS.py:123-125 */
        int n_crumbs;
        t.sim = (uint32_t)i;
        int leaf = move_to_leaf(&t, root, crumbs, &n_crumbs);
        if (n_crumbs > t.max_depth) t.max_depth = n_crumbs;
        t.sum_depth += n_crumbs;
        expand_and_backup(&t, leaf, crumbs, n_crumbs, (uint32_t)i + 1);
    }
    k = t.nodes[root].n_edges;
    if (k == 0) { rc = -1; goto done; }
    {
        double pi[NACT];
        memset(pi, 0, sizeof pi);
        double inv_tau = det_tau ? (1. / 0.01) : (1. / 1);                       /* MCTS.py:132 */
        for (int i = 0; i < k; i++) {
            edge_t *e = &t.edges[t.nodes[root].first_edge + i];
            pi[e->id * NCELL + e->dest] = pow((double)e->N, inv_tau);
            /* Python's pow(int, float) raises OverflowError where the result leaves float64 (N >= 1210 at tau = 0.01) */
            if (pi[e->id * NCELL + e->dest] > 1.7976931348623157e308) { rc = -3; goto done; }
        }
        double s = pairwise_sum(pi, NACT);                                       /* MCTS.py:137 */
        for (int i = 0; i < NACT; i++) pi[i] /= s;
        int idx = orc_sample_index(orc_rng(seed, game, ply, 0, 0, P_SAMPLE), pi, NACT);   /* MCTS.py:140 */
        int cid = idx / NCELL, cdest = idx % NCELL;
        int found = -1;
        for (int i = 0; i < k; i++) {
            edge_t *e = &t.edges[t.nodes[root].first_edge + i];
            if (e->id == cid && e->dest == cdest) { found = i; break; }
        }
        if (found < 0) { rc = -2; goto done; }                                   /* assert, MCTS.py:151 */
        *next = t.nodes[t.edges[t.nodes[root].first_edge + found].out_node].state;
        if (pi_out) memcpy(pi_out, pi, sizeof pi);
        if (out) {
            out->n_root = k;
            for (int i = 0; i < k; i++) {
                edge_t *e = &t.edges[t.nodes[root].first_edge + i];
                out->N[i] = e->N; out->W[i] = e->W; out->Q[i] = e->Q; out->P[i] = e->P; out->id[i] = e->id; out->dest[i] = e->dest;
            }
            memcpy(out->pi, pi, sizeof pi);
            out->chosen_id = cid; out->chosen_dest = cdest;
            out->nodes = 0; out->edges = 0;
            out->digest = digest_node(&t, root, 0, &out->nodes, &out->edges);
            out->max_depth = t.max_depth; out->sum_depth = t.sum_depth;
        }
    }
done:
    if (out) { out->evals = t.evals; out->terminals = t.terminals; }
    free(crumbs); free(t.nodes); free(t.edges);
    return rc;
}

int orc_search(const uint8_t *pos12, const uint8_t *last4, int player, uint64_t seed, uint64_t game, uint32_t ply,
               int sims, int det_tau, int evaluator, orc_eval_fn fn, void *user, orc_search_out *out) {
    board_t b, next; board_from_pos12(&b, pos12, last4);
    return make_move(&b, player, seed, game, ply, sims, det_tau, evaluator, fn, user, &next, out, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* S3: selfplay.selfplay (selfplay.py:11-80)                                                   */

enum { ST_WON_P1 = 1, ST_WON_P2 = 2, ST_DISCARD_REPETITION = 3, ST_DISCARD_NO_PROGRESS = 4, ST_ERROR = 5 };

typedef struct {
    int status; int reward;           /* reward for player one (utils.get_p1_winloss_reward) */
    int n_plies;                      /* all plies incl. the random opening */
    int n_hist;                       /* records returned in play_history */
    long evals, terminals;
    int n_searched;                   /* plies searched in all = rows the hist_* buffers hold when the game was NOT won (a discarded game
                                         returns (None, None), selfplay.py:45-47, 72-74, but its searches happened: tests compare them too) */
} orc_game_out;

/* buffers: ply_moves[max_plies][3] (kind 0 random / 1 tau=1 / 2 tau=0.01, id, dest);
 * hist_pos12[max_plies][12], hist_last[max_plies][4], hist_player[max_plies], hist_pi[max_plies][294] */
int orc_selfplay(uint64_t seed, uint64_t game, int sims, int evaluator, int randomised, int evaluator2 /* < 0: same as evaluator */,
                 orc_eval_fn fn, void *user, int max_plies,
                 uint8_t *ply_moves, uint8_t *hist_pos12, uint8_t *hist_last, uint8_t *hist_player, double *hist_pi,
                 orc_game_out *out) {
    board_t b; uint8_t pos12[12];
    if (randomised) orc_randomised_pos12(seed, game, pos12); else orc_initial_pos12(pos12);
    if (evaluator2 < 0) evaluator2 = evaluator;      /* model2 = model1 (selfplay.py:16-17) */
    board_from_pos12(&b, pos12, NULL);
    int player = 1;
    int player_progresses[2] = {0, 0};
    int player_turn = 0, num_useless_moves = 0, n_hist = 0, n_plies = 0;
    int det_tau = 0;
    uint8_t hist_moves[TOTAL_HIST_MOVES][2]; int n_hm = 0;       /* Board.hist_moves deque (board.py:246-248) */
    long evals = 0, terminals = 0;
    memset(out, 0, sizeof *out);
    for (;;) {
        if (n_plies >= max_plies) { out->status = ST_ERROR; break; }
        board_t next; int id, dest;
        if (n_hm < INITIAL_RANDOM_MOVES && n_plies < INITIAL_RANDOM_MOVES) {     /* selfplay.py:32 (len(hist_moves) < 6) */
            uint32_t counter = 0;
            if (!random_move(&b, player, seed, game, (uint32_t)n_plies, P_OPENING, 0, 0, &counter, &id, &dest)) { out->status = ST_ERROR; break; }
            next = b; place(&next, player, id, dest);
            ply_moves[n_plies * 3] = 0;
        } else {
            orc_search_out so;
            memcpy(hist_pos12 + n_hist * 12, b.pos, 12); memcpy(hist_last + n_hist * 4, b.last, 4);
            hist_player[n_hist] = (uint8_t)player;
            /* model1 searches for player one, model2 for player two (selfplay.py:30,36,59) */
            if (make_move(&b, player, seed, game, (uint32_t)n_plies, sims, det_tau, player == 1 ? evaluator : evaluator2, fn, user, &next, &so,
                          hist_pi + (size_t)n_hist * NACT)) { out->status = ST_ERROR; break; }
            evals += so.evals; terminals += so.terminals;
            id = so.chosen_id; dest = so.chosen_dest;
            ply_moves[n_plies * 3] = det_tau ? 2 : 1;
            n_hist++;
        }
        ply_moves[n_plies * 3 + 1] = (uint8_t)id; ply_moves[n_plies * 3 + 2] = (uint8_t)dest;
        if (n_hm == TOTAL_HIST_MOVES) { memmove(hist_moves, hist_moves + 1, (TOTAL_HIST_MOVES - 1) * 2); n_hm--; }
        hist_moves[n_hm][0] = b.pos[player - 1][id]; hist_moves[n_hm][1] = (uint8_t)dest; n_hm++;
        b = next; n_plies++;

        /* repetition rule, selfplay.py:40-47 */
        int n_cur = 0; uint8_t dests[TOTAL_HIST_MOVES]; int n_dests = 0;
        for (int i = n_hm - 1; i >= 0; i -= 2) {
            n_cur++;
            int seen = 0;
            for (int j = 0; j < n_dests; j++) if (dests[j] == hist_moves[i][1]) seen = 1;
            if (!seen) dests[n_dests++] = hist_moves[i][1];
        }
        if (n_cur * 2 >= TOTAL_HIST_MOVES && n_dests <= UNIQUE_DEST_LIMIT) { out->status = ST_DISCARD_REPETITION; break; }

        /* progress, selfplay.py:50-55 */
        int progress_evaluated = player_progress(&b, player_turn + 1);
        if (progress_evaluated > player_progresses[player_turn]) {
            num_useless_moves = (int)(num_useless_moves * (NCHK - 1) / NCHK);
            player_progresses[player_turn] = progress_evaluated;
        } else num_useless_moves += 1;

        player_turn = 1 - player_turn; player = 3 - player;                      /* selfplay.py:58-59 */
        if (n_hist + INITIAL_RANDOM_MOVES > TOTAL_MOVES_TILL_TAU0) det_tau = 1;  /* selfplay.py:62-65 */
        int w = check_win(&b);
        if (w) { out->status = (w == 1) ? ST_WON_P1 : ST_WON_P2; out->reward = (w == 1) ? 1 : -1; break; }   /* 67-69, utils.py:34-44 */
        if (num_useless_moves >= PROGRESS_MOVE_LIMIT) { out->status = ST_DISCARD_NO_PROGRESS; break; }        /* 72-74 */
    }
    out->n_plies = n_plies; out->evals = evals; out->terminals = terminals;
    if (out->status == ST_WON_P1 || out->status == ST_WON_P2) {
        int drop = randomised ? BOARD_HIST_MOVES : 0;                            /* selfplay.py:76-78 */
        if (drop > n_hist) drop = n_hist;
        if (drop) {
            memmove(hist_pos12, hist_pos12 + drop * 12, (size_t)(n_hist - drop) * 12);
            memmove(hist_last, hist_last + drop * 4, (size_t)(n_hist - drop) * 4);
            memmove(hist_player, hist_player + drop, (size_t)(n_hist - drop));
            memmove(hist_pi, hist_pi + (size_t)drop * NACT, (size_t)(n_hist - drop) * NACT * sizeof(double));
        }
        out->n_hist = n_hist - drop;
    } else out->n_hist = 0;
    out->n_searched = n_hist;
    return out->status;
}

/* fixed-work variant for the CPU baseline: `plies` MCTS plies from the position after the random
 * opening, no end-of-game rules (bench.py cpu_baseline leg).  Returns expansions performed. */
static long bench_plies_with(uint64_t seed, uint64_t game, int sims, int evaluator, int plies, orc_eval_fn fn, void *user);

long orc_bench_plies(uint64_t seed, uint64_t game, int sims, int evaluator, int plies) {
    return bench_plies_with(seed, game, sims, evaluator, plies, NULL, NULL);
}

/* the same with an external evaluator (evaluator code 4: `fn` is called once per expansion, batch of one, as MCTS.py:93
 * calls model.predict): the net-inclusive CPU baseline of bench.py */
long orc_bench_plies_fn(uint64_t seed, uint64_t game, int sims, int plies, orc_eval_fn fn, void *user) {
    return bench_plies_with(seed, game, sims, 4, plies, fn, user);
}

static long bench_plies_with(uint64_t seed, uint64_t game, int sims, int evaluator, int plies, orc_eval_fn fn, void *user) {
    board_t b; uint8_t pos12[12]; orc_initial_pos12(pos12); board_from_pos12(&b, pos12, NULL);
    int player = 1; long evals = 0; int n_plies = 0;
    for (; n_plies < INITIAL_RANDOM_MOVES; n_plies++) {
        uint32_t counter = 0; int id, dest;
        if (!random_move(&b, player, seed, game, (uint32_t)n_plies, P_OPENING, 0, 0, &counter, &id, &dest)) return evals;
        place(&b, player, id, dest); player = 3 - player;
    }
    for (int i = 0; i < plies; i++, n_plies++) {
        board_t next; orc_search_out so;
        if (check_win(&b)) break;
        if (make_move(&b, player, seed, game, (uint32_t)n_plies, sims, i + INITIAL_RANDOM_MOVES > TOTAL_MOVES_TILL_TAU0,
                      evaluator, fn, user, &next, &so, NULL)) break;
        evals += so.evals; b = next; player = 3 - player;
    }
    return evals;
}

/* ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------ */
/* next-4: GreedyPlayer (player.py:67-129, stochastic=False as Game builds it)    */

#define AVERAGE_TOTAL_MOVE 43
#define EV_GREEDY 100                 /* "evaluator" code of a GreedyPlayer seat in orc_arena_game */

static int human_row(int cell) { return cell / W7 - cell % W7 + W7; }                /* board_utils.py:3-7 */

/* decide_move(training=True): the moves of maximum forward distance (player.py:100-110), then only those
 * that start on the row of the LAST checker among them (112-115), in get_valid_moves order */
static int greedy_best(board_t *b, int player, uint8_t (*best)[2]) {
    uint8_t moves[MAXMV][2];
    int n = valid_moves(b, player, moves, NULL);
    int max_dist = -1000, nb = 0;
    int srow[MAXMV];
    for (int i = 0; i < n; i++) {
        int s = human_row(b->pos[player - 1][moves[i][0]]), e = human_row(moves[i][1]);
        int dist = e - s;                                                              /* player.py:103 */
        if (player == 1) dist = -dist;                                                 /* 104-105 */
        if (dist > max_dist) { max_dist = dist; nb = 0; }                              /* 106-108 */
        if (dist == max_dist) { best[nb][0] = moves[i][0]; best[nb][1] = moves[i][1]; srow[nb] = s; nb++; }   /* 109-110 */
    }
    if (nb == 0) return 0;
    int last = 0;                                                                      /* max(..., key) keeps the first maximum */
    for (int i = 1; i < nb; i++) {
        int ki = player == 1 ? srow[i] : -srow[i], kl = player == 1 ? srow[last] : -srow[last];
        if (ki > kl) last = i;
    }
    int k = 0;
    for (int i = 0; i < nb; i++)
        if (srow[i] == srow[last]) { best[k][0] = best[i][0]; best[k][1] = best[i][1]; k++; }   /* 115 */
    return k;
}

int orc_greedy_best(const uint8_t *pos12, int player, uint8_t *best /* [126][2] */) {
    board_t b; board_from_pos12(&b, pos12, NULL);
    return greedy_best(&b, player, (uint8_t (*)[2])best);
}

/* the greedy seat's move in Game.start (player.py:122): one draw among the filtered best moves */
static int greedy_move(board_t *b, int player, uint64_t seed, uint64_t game, uint32_t ply, int *id, int *dest) {
    uint8_t best[MAXMV][2];
    int k = greedy_best(b, player, best);
    if (k == 0) return 0;
    int j = (int)orc_choice(orc_rng(seed, game, ply, 0, 0, P_GREEDY), (uint32_t)k);
    *id = best[j][0]; *dest = best[j][1];
    return 1;
}

/* GreedyPlayer(stochastic=True).decide_move (player.py:77-97): the moves that go forward (dist > 0) are drawn with probability
 * proportional to their forward distance -- prior = dist / sum(dist) in float64, np.random.choice(len, p=prior) = the spec's
 * sample_index (player.py:94-96); with no forward move, a uniform draw among the others (= all moves), player.py:92.  One draw
 * per ply, the key of the deterministic player's. */
#define EV_GREEDY_STOCHASTIC 101
static int greedy_move_stochastic(board_t *b, int player, uint64_t seed, uint64_t game, uint32_t ply, int *id, int *dest) {
    uint8_t moves[MAXMV][2];
    int n = valid_moves(b, player, moves, NULL);
    if (n == 0) return 0;
    int fw[MAXMV], dist[MAXMV], bw[MAXMV], nf = 0, nb = 0;
    long sum = 0;
    for (int i = 0; i < n; i++) {
        int s = human_row(b->pos[player - 1][moves[i][0]]), e = human_row(moves[i][1]);
        int d = e - s;                                                                 /* player.py:83 */
        if (player == 1) d = -d;                                                       /* 84-85 */
        if (d > 0) { fw[nf] = i; dist[nf] = d; nf++; sum += d; }                       /* 86-88 */
        else bw[nb++] = i;                                                             /* 89-90 */
    }
    uint64_t u = orc_rng(seed, game, ply, 0, 0, P_GREEDY);
    int pick;
    if (nf == 0) pick = bw[orc_choice(u, (uint32_t)nb)];                               /* 92 */
    else {
        double p[MAXMV];
        for (int i = 0; i < nf; i++) p[i] = (double)dist[i] / (double)sum;             /* 94 */
        pick = fw[orc_sample_index(u, p, nf)];                                         /* 95-96 */
    }
    *id = moves[pick][0]; *dest = moves[pick][1];
    return 1;
}

int orc_greedy_stochastic_move(const uint8_t *pos12, int player, uint64_t seed, uint64_t game, uint32_t ply, int *id, int *dest) {
    board_t b; board_from_pos12(&b, pos12, NULL);
    return greedy_move_stochastic(&b, player, seed, game, ply, id, dest);
}

typedef struct { int status; int reward; int n_plies; int n_hist; int stuck; } orc_greedy_out;

/* GreedyDataGenerator.generate_play (data_generators.py:25-80).  `stuck_limit` replaces the wall-clock
 * STUCK_TIME_LIMIT (61): the game is given up right after ply stuck_limit + 1 (counted from the first greedy
 * ply's start... i.e. all plies incl. the random start) when nobody has won, returning the first
 * AVERAGE_TOTAL_MOVE records and reward 0.
 * buffers: ply_moves[max][2]; hist_pos12[max][12], hist_last[max][4], hist_player[max], hist_n[max] (moves sharing pi),
 * hist_idx[max][32] (action indices, ascending in list order) */
int orc_greedy_game(uint64_t seed, uint64_t game, int randomised, int random_start, int stuck_limit, int max_plies,
                    uint8_t *ply_moves, uint8_t *hist_pos12, uint8_t *hist_last, uint8_t *hist_player, int *hist_n, int *hist_idx,
                    orc_greedy_out *out) {
    board_t b; uint8_t pos12[12];
    if (randomised) orc_randomised_pos12(seed, game, pos12); else orc_initial_pos12(pos12);
    board_from_pos12(&b, pos12, NULL);
    int player = 1, n_plies = 0, n_hist = 0, checks = 0;
    memset(out, 0, sizeof *out);
    if (random_start)                                                                 /* data_generators.py:31-40 */
        for (int i = 0; i < INITIAL_RANDOM_MOVES; i++) {
            uint32_t counter = 0; int id, dest;
            if (!random_move(&b, player, seed, game, (uint32_t)n_plies, P_OPENING, 0, 0, &counter, &id, &dest)) { out->status = ST_ERROR; return ST_ERROR; }
            ply_moves[2 * n_plies] = (uint8_t)id; ply_moves[2 * n_plies + 1] = (uint8_t)dest;
            place(&b, player, id, dest);
            n_plies++;
            player = 3 - player;
        }
    int winner = 0;
    for (;;) {
        if (n_plies >= max_plies) { out->status = ST_ERROR; break; }
        uint8_t best[MAXMV][2];
        int k = greedy_best(&b, player, best);                                        /* 43 */
        if (k == 0 || k > 32) { out->status = ST_ERROR; break; }
        memcpy(hist_pos12 + 12 * n_hist, b.pos, 12);                                   /* 44-53: (deepcopy(board), pi) */
        memcpy(hist_last + 4 * n_hist, b.last, 4);
        hist_player[n_hist] = (uint8_t)player;
        hist_n[n_hist] = k;
        for (int i = 0; i < k && i < 32; i++) hist_idx[32 * n_hist + i] = best[i][0] * NCELL + best[i][1];
        n_hist++;
        int j = (int)orc_choice(orc_rng(seed, game, (uint32_t)n_plies, 0, 0, P_GREEDY), (uint32_t)k);   /* 55 */
        ply_moves[2 * n_plies] = best[j][0]; ply_moves[2 * n_plies + 1] = best[j][1];
        winner = place(&b, player, best[j][0], best[j][1]);                           /* 60 */
        n_plies++;
        if (winner) break;                                                             /* 61-63 */
        if (++checks > stuck_limit) { out->stuck = 1; break; }                        /* 66-67, in plies */
        player = 3 - player;                                                           /* 69 */
    }
    out->n_plies = n_plies;
    if (out->status == ST_ERROR) return ST_ERROR;
    if (out->stuck) {
        out->status = ST_DISCARD_NO_PROGRESS; out->reward = 0;
        out->n_hist = n_hist < AVERAGE_TOTAL_MOVE ? n_hist : AVERAGE_TOTAL_MOVE;       /* play_history[:AVERAGE_TOTAL_MOVE] */
        return out->status;
    }
    out->status = winner; out->reward = winner == 1 ? 1 : -1;                         /* utils.py:34-44 */
    out->n_hist = n_hist;                                                              /* caller drops the first BOARD_HIST_MOVES when randomised (77-78) */
    return out->status;
}

/* next-3: one arena game, Game.start (game.py:58-100) between two AiPlayers       */

typedef struct { int winner; int n_moves; long evals; int status; } orc_arena_out;

/* winner: 1 / 2, or 0 for "None" (repetition, or the move limit when enforce_move_limit); moves[n][2] = (id, dest) */
int orc_arena_game(uint64_t seed, uint64_t game, int sims, int evaluator1, int evaluator2, int det_tau_initial,
                   int enforce_move_limit, int max_moves, uint8_t *moves, orc_arena_out *out) {
    board_t b; uint8_t pos12[12]; orc_initial_pos12(pos12); board_from_pos12(&b, pos12, NULL);    /* Board() (game.py:34) */
    int player = 1, total_moves = 0, num_moves = 0;
    int tau_det[2] = {det_tau_initial, det_tau_initial};             /* each AiPlayer keeps its own tree_tau (player.py:137) */
    uint8_t history_dests[TOTAL_HIST_MOVES]; int n_hd = 0;
    long evals = 0;
    memset(out, 0, sizeof *out);
    for (;;) {
        if (total_moves >= max_moves) { out->status = ST_ERROR; break; }
        if (total_moves > TOTAL_MOVES_TILL_TAU0) tau_det[player - 1] = 1;                         /* player.py:152-155 */
        board_t next; orc_search_out so;
        const int seat = player == 1 ? evaluator1 : evaluator2;
        if (seat == EV_GREEDY || seat == EV_GREEDY_STOCHASTIC) {                                  /* GreedyPlayer.decide_move */
            int id, dest;
            if (!(seat == EV_GREEDY ? greedy_move(&b, player, seed, game, (uint32_t)total_moves, &id, &dest)
                                    : greedy_move_stochastic(&b, player, seed, game, (uint32_t)total_moves, &id, &dest))) { out->status = ST_ERROR; break; }
            so.chosen_id = id; so.chosen_dest = dest; so.evals = 0;
        } else if (make_move_ex(&b, player, seed, game, (uint32_t)total_moves, sims, tau_det[player - 1],
                                seat, NULL, NULL, &next, &so, NULL, 1)) { out->status = ST_ERROR; break; }
        evals += so.evals;
        moves[total_moves * 2] = (uint8_t)so.chosen_id; moves[total_moves * 2 + 1] = (uint8_t)so.chosen_dest;
        int winner = place(&b, player, so.chosen_id, so.chosen_dest);                             /* game.py:65 */
        total_moves++;
        if (winner) { out->winner = winner; out->status = winner; break; }                        /* game.py:70-71 */
        if (n_hd == TOTAL_HIST_MOVES) { memmove(history_dests, history_dests + 1, TOTAL_HIST_MOVES - 1); n_hd--; }
        history_dests[n_hd++] = (uint8_t)so.chosen_dest;                                          /* game.py:73-75 */
        uint64_t seen = 0;                                                                        /* game.py:78-82 */
        for (int i = n_hd - 1; i >= 0; i -= 2) seen |= 1ULL << history_dests[i];
        int distinct = 0; for (int i = 0; i < NCELL; i++) distinct += (int)((seen >> i) & 1);
        if (n_hd == TOTAL_HIST_MOVES && distinct <= UNIQUE_DEST_LIMIT) { out->status = ST_DISCARD_REPETITION; break; }
        num_moves++;
        if (enforce_move_limit && num_moves >= PROGRESS_MOVE_LIMIT) { out->status = ST_DISCARD_NO_PROGRESS; break; }   /* 86-89 */
        player = 3 - player;
    }
    out->n_moves = total_moves; out->evals = evals;
    return out->status;
}
