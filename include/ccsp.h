/*
 * ccsp.h -- C ABI of libccsp.so: the MI355X-native self-play data generator for
 * kenziyuliu/ChineseCheckersAgent (batched rules kernels + GPU-resident batched MCTS).
 *
 * The reference has no FFI layer: its boundary is the Python call contract of
 * selfplay.selfplay() (selfplay.py:11-80) as called from train.generate_self_play
 * (train.py:27-67).  This header is what a binding for that path binds instead; every entry
 * point names the reference code it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C, no torch / HIP types: device pointers are `void*`-compatible raw pointers,
 *     `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns 0 on success, a negative CCSP_E* code otherwise; nothing throws
 *     or aborts (the reference's `assert`s -- MCTS.py:80,151; selfplay.py:38,113,118 -- become
 *     per-game status CCSP_ST_ERROR);
 *   - all launches are asynchronous and stream-ordered; a context is not thread-safe; every entry point runs on the
 *     context's own device and puts the caller's current device back before it returns;
 *   - the caller owns every buffer it passes; the library keeps device memory only inside a
 *     ccsp_ctx between ccsp_create() and ccsp_destroy();
 *   - random draws are a pure function of (seed, global game id, ply, simulation, depth,
 *     purpose) -- oracle/harness/spec.py -- so results do not depend on batch size or sharding.
 */
#ifndef CCSP_H
#define CCSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCSP_NUM_ACTIONS 294   /* 6 checkers x 49 cells, utils.encode_checker_index (utils.py:164-171) */
#define CCSP_MAX_MOVES   126   /* <= 21 destinations per checker */
#define CCSP_PLANES      343   /* 7 x 7 x 7 model input, utils.to_model_input (utils.py:101-160) */
#define CCSP_NO_MOVE     255

/* B1. Board state (board.py:9-57) as a fixed 32-byte record.  Cell index = row*7 + col.
 * occ[p]  : bitboard of player p+1's checkers (bit = cell index)         <- Board.board[:, :, 0]
 * pos[p]  : cell of checker id 0..5 of player p+1                        <- Board.checkers_pos / checkers_id
 * last    : hist_moves[-1] (from,to) then hist_moves[-2] (from,to); CCSP_NO_MOVE while the
 *           matching history plane board[:, :, 1|2] is still empty       <- what to_model_input reads */
typedef struct ccsp_state {
    uint64_t occ[2];
    uint8_t  pos[2][6];
    uint8_t  last[4];
} ccsp_state;

enum {
    CCSP_OK = 0,
    CCSP_EINVAL = -1,      /* bad argument */
    CCSP_ENOMEM = -2,      /* device allocation failed */
    CCSP_EHIP = -3,        /* a HIP runtime call failed (see ccsp_last_hip_error) */
    CCSP_ENODEVICE = -4,   /* no gfx950 device / kernels not loadable: the product has no CPU path */
    CCSP_ESTATE = -5       /* call out of sequence (e.g. expand_backup without select) */
};

/* per-game status (selfplay.py:45-47, 67-74) */
enum {
    CCSP_ST_RUNNING = 0,
    CCSP_ST_WON_P1 = 1,
    CCSP_ST_WON_P2 = 2,
    CCSP_ST_DISCARD_REPETITION = 3,
    CCSP_ST_DISCARD_NO_PROGRESS = 4,
    CCSP_ST_ERROR = 5,
    CCSP_ST_IDLE = 6       /* slot has no game (game-id budget exhausted) */
};

/* built-in evaluators for the fused (no-net) path; CCSP_EVAL_EXTERNAL = caller supplies (p, v) */
enum {
    CCSP_EVAL_UNIFORM = 0,   /* p = 1/294, v = 0: what the reference computes with a stub model (config 2a) */
    CCSP_EVAL_HASH = 1,      /* spec.hash_eval: parity-test evaluator */
    CCSP_EVAL_FORWARD = 2,   /* spec.forward_eval: parity-test evaluator under which games end in wins */
    CCSP_EVAL_ROLLOUT = 3,   /* p = 1/294, v = random playout (config 2b; no reference counterpart) */
    CCSP_EVAL_EXTERNAL = 4
};

int ccsp_device_count(void);                 /* number of visible HIP devices, no GPU initialisation */
const char *ccsp_version(void);
const char *ccsp_last_hip_error(void);

/* ---- host-side helpers (no GPU) -------------------------------------------------------------- */

/* Build records from checker positions (pos12 = player-1 ids 0..5 then player-2 ids 0..5) and the
 * last two moves (last4 or NULL).  Host memory in, host memory out. */
int ccsp_pack_states(const uint8_t *pos12, const uint8_t *last4, int n, ccsp_state *out);

/* ---- batched rules kernels (device pointers) ------------------------------------------------- */

/* B2-B4  Board.get_valid_moves (board.py:139-222): per state the ordered legal-move list of
 * `player[i]` as (checker id, destination cell) pairs in the reference's order -- checker id
 * ascending; per checker the walks in direction order N,E,SE,S,W,NW, then the hop landings in the
 * reference's recursive depth-first pre-order.  moves: [n][126][2], count: [n], dest_mask: [n][6]
 * (bit = destination cell; may be NULL). */
int ccsp_movegen(const ccsp_state *s, const uint8_t *player, int n,
                 uint8_t *moves, uint8_t *count, uint64_t *dest_mask, void *stream);

/* The same lists in a PACKED layout: the move lists of the 32 positions of a chunk (positions 32c .. 32c + 31) stand back to
 * back in position order, position i's list starting at entry 32 * 126 * (i / 32) + sum(count[32 * (i / 32) .. i - 1]) of
 * `moves` (2 bytes per entry; the buffer has the same size as ccsp_movegen's, [n rounded up to 32][126][2]).  Sparse 252-byte
 * rows cost 1.27 x the algorithmic write bytes in partly written sectors; this layout writes whole ones. */
#define CCSP_MOVEGEN_CHUNK 32
int ccsp_movegen_packed(const ccsp_state *s, const uint8_t *player, int n,
                        uint8_t *moves, uint8_t *count, uint64_t *dest_mask, void *stream);

/* B5-B7  Board.place (board.py:226-250) + check_win (89-111) + player_progress (254-266):
 * mv: [n][2] = (checker id, destination).  winner: [n] in {0,1,2}; progress: [n][2] of the NEW
 * state (may be NULL). */
int ccsp_step(const ccsp_state *in, const uint8_t *player, const uint8_t *mv, int n,
              ccsp_state *out, uint8_t *winner, uint8_t *progress, void *stream);

/* C1  utils.to_model_input (utils.py:101-160): planes [n][7][7][7] (row, col, channel) float32. */
int ccsp_encode(const ccsp_state *s, const uint8_t *player, int n, float *planes, void *stream);

/* next-4  GreedyPlayer.decide_move(training=True) (player.py:72-118, stochastic=False): per state the moves of
 * maximum forward distance that start on the row of the last checker among them, as (checker id, destination)
 * pairs in get_valid_moves order.  best: [n][CCSP_GREEDY_MAX][2], count: [n] (0 = no legal move). */
#define CCSP_GREEDY_MAX 32
int ccsp_greedy_best(const ccsp_state *s, const uint8_t *player, int n, uint8_t *best, uint8_t *count, void *stream);

/* test hook: per-lane hop-search stack entries of ccsp_movegen / ccsp_greedy_best (6 .. 20; other values restore the
 * default 20).  Searches that need more are redone on a big stack -- no real position does; tests force it.  Returns the
 * value in force. */
int ccsp_debug_movegen_stack_cap(int cap);

/* ---- self-play engine ------------------------------------------------------------------------- */

typedef struct ccsp_ctx ccsp_ctx;

typedef struct ccsp_config {
    int32_t  n_slots;         /* concurrent games on this GPU (4096 in BASELINE.json) */
    int32_t  sims;            /* simulations per move, MCTS_SIMULATIONS (config.py:35; MCTS.py:41) */
    int32_t  randomised;      /* Board(randomised=True) starts (board.py:61-85) */
    int32_t  auto_restart;    /* 1: a finished slot starts its next game by itself: slot g plays game indices g, g + n_slots, ... */
    uint64_t seed;
    uint64_t first_game;      /* global id of this context's first game */
    uint64_t game_stride;     /* id step between consecutive games of this context (= world size) */
    uint64_t max_games;       /* game-id budget of this context (slots go idle when it is spent) */
    uint64_t log_capacity;    /* rows of the (state, pi) sample log */
    int32_t  device;          /* HIP device ordinal */
    int32_t  max_plies;       /* safety cap per game (status ERROR beyond); 0 = 1024 */
    int32_t  mode;            /* CCSP_MODE_*: 0 = self-play (selfplay.py), 1 = arena: Game.start (game.py:58-100) between
                                 AiPlayers (player.py:133-166: no random opening, no root pre-expansion, no Dirichlet
                                 noise) and/or GreedyPlayers, 2 = greedy data generator */
    int32_t  arena_det_tau;   /* arena: Game(tree_tau=DET_TREE_TAU) (1) or TREE_TAU until total_moves > 16 (0) */
    int32_t  enforce_move_limit;   /* arena: Game.start(enforce_move_limit=True): stop after 100 moves */
    int32_t  greedy;          /* next-4, CCSP_GREEDY_* bits: which seats of Game.start are GreedyPlayers (mode 1), random
                                 start of the data generator (mode 2) */
    int32_t  stuck_limit;     /* mode 2: plies without a winner after which generate_play gives the game up -- the ply
                                 form of the wall-clock STUCK_TIME_LIMIT (data_generators.py:66-67); 0 = 200 */
    int32_t  pad;
} ccsp_config;

/* ccsp_config.mode */
enum {
    CCSP_MODE_SELFPLAY = 0,      /* selfplay.selfplay (selfplay.py:11-80) */
    CCSP_MODE_ARENA = 1,         /* Game.start (game.py:58-100): AiPlayer and/or GreedyPlayer seats */
    CCSP_MODE_GREEDY_DATA = 2    /* GreedyDataGenerator.generate_play (data_generators.py:25-80): no search at all */
};
/* ccsp_config.greedy */
enum {
    CCSP_GREEDY_P1 = 1,          /* player one is a GreedyPlayer (game.py:19-20) */
    CCSP_GREEDY_P2 = 2,          /* player two is a GreedyPlayer (game.py:26-27) */
    CCSP_GREEDY_ALTERNATE = 4,   /* seats swap on odd game ids (ai_vs_greedy.py:47-48) */
    CCSP_GREEDY_RANDOM_START = 8,/* mode 2: GreedyDataGenerator(random_start=True) (data_generators.py:31-40) */
    CCSP_GREEDY_STOCHASTIC_P1 = 16,   /* mode 1: the GreedyPlayer of seat one is GreedyPlayer(stochastic=True) (player.py:68, 77-97; game.py:111) */
    CCSP_GREEDY_STOCHASTIC_P2 = 32    /* ... of seat two (the two bits swap with the seats under CCSP_GREEDY_ALTERNATE) */
};

/* one row of the sample log = one entry of selfplay()'s play_history (selfplay.py:128) */
typedef struct ccsp_sample_meta {
    uint64_t game;            /* global game id */
    uint32_t ply;             /* ply index in the game (the 6 random opening plies count) */
    uint8_t  player;          /* player to move in `state` */
    uint8_t  pad[3];
} ccsp_sample_meta;

/* per-game result, indexed by (game - first_game) / game_stride */
typedef struct ccsp_game_result {
    uint8_t  status;          /* CCSP_ST_* */
    int8_t   reward;          /* utils.get_p1_winloss_reward (utils.py:34-44) */
    uint16_t n_plies;         /* all plies played */
    uint32_t n_samples;       /* MCTS plies logged */
    uint64_t expansions;      /* evaluator calls spent on the game */
} ccsp_game_result;

/* running totals (device-side 64-bit counters; ccsp_read_counters copies them out) */
enum {
    CCSP_CNT_EXPANSIONS = 0,  /* non-terminal expandAndBackUp calls = evaluator calls (MCTS.py:93) */
    CCSP_CNT_TERMINAL_SIMS,   /* simulations that ended in a won leaf (MCTS.py:81-90) */
    CCSP_CNT_SIMS,            /* simulations run */
    CCSP_CNT_PLIES,           /* plies played (all kinds) */
    CCSP_CNT_MCTS_PLIES,
    CCSP_CNT_GAMES_WON,
    CCSP_CNT_GAMES_DISCARDED,
    CCSP_CNT_SUM_DEPTH,       /* sum over simulations of the selection depth D */
    CCSP_CNT_SUM_CHILDREN,    /* sum over expansions of the children created K */
    CCSP_CNT_SELECT_EDGES,    /* sum over selection levels of the edges scanned (for the byte model) */
    CCSP_CNT_SAMPLES,         /* rows appended to the sample log */
    CCSP_CNT_ERRORS,
    CCSP_CNT_CACHE_HITS = 15, /* expansions answered from the previous ply's tree instead of the evaluator (ccsp_advance with CCSP_ADVANCE_REUSE);
                                 they are expansions all the same: CCSP_CNT_EXPANSIONS counts them too */
    CCSP_CNT_COUNT = 16
};

ccsp_ctx *ccsp_create(const ccsp_config *cfg, int *err);
int ccsp_destroy(ccsp_ctx *ctx);

/* (re)start every slot on fresh games first_game, first_game+stride, ... and clear log + counters */
int ccsp_reset(ccsp_ctx *ctx, void *stream);

/* Test/arena hook: put slot i on the given position (host arrays of n_slots entries): game id,
 * ply index (keys the draw stream), player to move, tau flag (1 = DET_TREE_TAU).  Such slots have
 * no random opening plies left: the next ply is always a search.  Used to reproduce single
 * make_move() cases (selfplay.py:107-133). */
int ccsp_set_positions(ccsp_ctx *ctx, const ccsp_state *states, const uint8_t *player,
                       const uint64_t *game, const uint32_t *ply, const uint8_t *det_tau, void *stream);

/* Fused path: play `n_plies` plies on every running slot with a built-in evaluator, entirely on
 * the GPU: selfplay()'s loop body (selfplay.py:29-74) = make_random_move (83-104) or make_move
 * (107-133) = root expansion + Dirichlet noise + `sims` x {moveToLeaf, expandAndBackUp}
 * (MCTS.py:49-118) + pi + action sampling (MCTS.py:121-153) + the end-of-ply rules. */
int ccsp_play_plies(ccsp_ctx *ctx, int evaluator, int n_plies, void *stream);
/* Plies one launch of ccsp_play_plies carries each game through (default 64: one wave per game runs root expansion, simulations and
 * move for ply after ply without waiting for the slowest search of every ply; 1 = three launches per ply).  Results do not depend on
 * it.  Returns the previous value; n < 1 only reads it. */
int ccsp_debug_plies_per_launch(int n);

/* Stepped path (external evaluator, e.g. the policy/value net): one ply =
 *   ccsp_ply_begin            root planes out                       (selfplay.py:114-117, utils.py:101)
 *   [evaluate]  ccsp_root_expand(p, v)     root expansion + noise   (selfplay.py:117-124)
 *   sims x { ccsp_select      moveToLeaf + leaf planes out          (MCTS.py:49-76)
 *            [evaluate]  ccsp_expand_backup(p, v) }                 (MCTS.py:79-118)
 *   ccsp_ply_end              pi, sampling, Board.place, rules      (MCTS.py:127-153, selfplay.py:38-74)
 * planes: [n_slots][7][7][7] f32; p: [n_slots][294] f64 (already softmaxed, Model.predict's
 * contract, model.py:21-24); v: [n_slots] f32.  Slots in their random opening plies (or not
 * running) ignore p and v; ccsp_ply_end plays their random move. */
int ccsp_ply_begin(ccsp_ctx *ctx, float *planes, void *stream);
int ccsp_root_expand(ccsp_ctx *ctx, const double *p, const float *v, void *stream);
int ccsp_select(ccsp_ctx *ctx, float *planes, void *stream);
int ccsp_expand_backup(ccsp_ctx *ctx, const double *p, const float *v, void *stream);
/* ccsp_expand_backup followed by ccsp_select in ONE launch (inside a ply: [evaluate -> this] per simulation instead of
 * [evaluate -> expand_backup -> select]); same results */
int ccsp_expand_backup_select(ccsp_ctx *ctx, const double *p, const float *v, float *planes, void *stream);
int ccsp_ply_end(ccsp_ctx *ctx, void *stream);

/* Free-running stepped path (self-play mode): the same search, every slot at its own simulation of its own ply.  The caller's loop is
 *     for ever: [evaluate the requests -> (pk, v)]  ->  ccsp_advance(pk, v, req, moves)  ->  ccsp_boundary(pk, v, req, moves)
 * (the very first round has no answer to give: any pk, v).  Per slot, a call takes the answer to the request the slot left earlier and
 * goes on until it needs the evaluator again -- request record and move list out -- :
 *   ccsp_advance    slots in a search: expansion + backup of the answered leaf (MCTS.py:93-118), then selection (MCTS.py:49-76) --
 *                   and on through simulations that end in a won leaf (MCTS.py:81-90) and through reused positions -- until a leaf needs
 *                   the evaluator, or the ply's `sims` simulations are done;
 *   ccsp_boundary   slots between two searches: root expansion + Dirichlet noise from the answer (selfplay.py:117-124), or: pi, the move,
 *                   the end-of-ply rules and the log row of the finished ply (MCTS.py:127-153, selfplay.py:38-74), one opening ply
 *                   (selfplay.py:83-104), the next ply's root -- expanded from the previous tree (reuse) or asked for as a request.
 *                   In stream order (the default) a root request is answered by the very next evaluator launch.  With
 *                   CCSP_ADVANCE_OVERLAPPED it may run on ANOTHER stream beside the next evaluator launch, provided it starts after the
 *                   ccsp_advance of its round and ends before the ccsp_advance of the next (the few slots it serves are not in a search;
 *                   a root request is then answered by the evaluator launch AFTER the next one).
 *
 * THE HAND-OFF (what replaces the reference's call `model.predict(to_model_input(leaf))`, MCTS.py:93, for a whole batch):
 *   req    [n_slots] ccsp_request, device, CALLER-OWNED and zero-filled before the first call after ccsp_create / ccsp_reset /
 *          ccsp_set_positions; the same buffer in every call of a context (a slot's record is rewritten only when its state changes).  The
 *          engine only WRITES it: its own hand-off state is kept inside the context.
 *          A slot with kind != 0 asks for the evaluation of `state` with `player` to move: the 32-byte position record is all the
 *          evaluator needs (utils.to_model_input is a function of it: ccsp_net_forward_requests builds the 7 x 7 x 7 planes in its input
 *          phase; ccsp_encode_requests writes them out for evaluators that want planes) -- 64 bytes per request where float32 planes
 *          were 1372;
 *   moves  [n_slots][CCSP_REQUEST_MOVES] uint16, device, caller-owned, the same buffer in every call: the legal moves of `state` in
 *          Board.get_valid_moves' order (board.py:215-222), entry = action index (utils.encode_checker_index) | 0x8000 where the move wins;
 *          req.k entries.  The move list is generated when the request is made and READ BACK when the answer arrives (zero-fill it too: the
 *          evaluators answer 0.0 for an entry that is no action index);
 *   pk     [n_slots][CCSP_REQUEST_MOVES] float64, device: THE ANSWER, COMPACT -- pk[slot][j] = softmax(logits)[moves[slot][j] & 0x1FF], the
 *          prior of the j-th legal move (the reference reads p[encode_checker_index(...)] per legal move, MCTS.py:97-109, and nothing
 *          else of the 294 entries); v [n_slots] float32.  K x 8 bytes per answer where the full policy row was 2352.
 *   Rows of slots that ask for nothing (kind == 0: game over, between plies, a selection given up at the deadline) are ignored.
 *
 *   CCSP_ADVANCE_REUSE      selfplay.make_move returns the chosen child as a fresh root (selfplay.py:130-133) and the next ply evaluates the
 *                           positions of its subtree again; with this flag a position the previous ply's tree holds below the move that was
 *                           played is expanded from that tree's priors and value -- the evaluator is a function of the position alone, so
 *                           trees, pi and games are bit-identical, only the evaluator is asked less often (CCSP_CNT_CACHE_HITS).  Needs
 *                           ccsp_enable_tree_reuse (a second tree pool).  NOT for two-model games: the previous ply was searched with the
 *                           other player's model (selfplay.py:30,59).  The same flags go to both calls.
 *   CCSP_ADVANCE_LOG_GUARD  a slot whose finished ply might not find a free row in the sample log waits for the caller's next
 *                           ccsp_log_clear instead of ending its game in CCSP_ST_ERROR (for callers that harvest the log as they go).
 * model_sel (device, [n_slots], may be NULL): 1 where the request is to be answered by player two's model (selfplay.py:30,36,59).
 *   CCSP_ADVANCE_STAGGER    (ccsp_boundary) a slot starts its FIRST game hash(id of that game) mod `span` ccsp_boundary CALLS late (span:
 *                           ccsp_set_stagger_span, default `sims`; the countdown moves only in ccsp_boundary calls, so a caller that makes one
 *                           every k-th round spreads the starts over k x span rounds), so that the slots' plies end evenly spread over the
 *                           rounds from the start instead of in waves (every round then carries the same mix of cheap and expensive tree
 *                           work).  A game's record does not depend on when it is played.
 *   CCSP_ADVANCE_OVERLAPPED (ccsp_boundary) the call runs beside the next evaluator launch (see above).
 */
#define CCSP_REQUEST_MOVES 128
typedef struct ccsp_request {
    ccsp_state state;         /* the position to evaluate */
    uint32_t kind;            /* 0 = nothing asked; 1 = a leaf, 3 / 4 = a ply's root */
    uint32_t reserved0[2];    /* written as 0 */
    uint32_t player;          /* player to move in `state` (1 | 2) */
    uint32_t k;               /* legal moves of `state`: entries of moves[slot] and of pk[slot] */
    uint32_t reserved1[3];    /* written as 0 */
} ccsp_request;               /* OUTPUT ONLY: the engine never reads a record back -- where a new node hangs, the path length and a selection
                                 to be resumed live in the context's own memory, so nothing a caller's buffer holds is ever used as an address */
enum { CCSP_ADVANCE_REUSE = 1, CCSP_ADVANCE_LOG_GUARD = 2, CCSP_ADVANCE_STAGGER = 4, CCSP_ADVANCE_DEBUG = 8 /* diagnostic tallies in counters 12-14 */,
       CCSP_ADVANCE_OVERLAPPED = 16, CCSP_ADVANCE_ALL_FLAGS = 31 };
int ccsp_enable_tree_reuse(ccsp_ctx *ctx);
/* CCSP_ADVANCE_STAGGER: the number of ccsp_boundary calls over which the slots' first games begin (0 = the default, `sims`; at most 65535).
 * A span of a whole game's worth of calls puts a restarting run into its steady state -- games ending at an even rate -- as soon as the last
 * slot has started.  Call it before the first ccsp_boundary. */
int ccsp_set_stagger_span(ccsp_ctx *ctx, int boundary_calls);
int ccsp_advance(ccsp_ctx *ctx, const double *pk, const float *v, ccsp_request *req, uint16_t *moves, uint8_t *model_sel, int flags, void *stream);
int ccsp_boundary(ccsp_ctx *ctx, const double *pk, const float *v, ccsp_request *req, uint16_t *moves, uint8_t *model_sel, int flags, void *stream);
/* C1 for a batch of requests (utils.to_model_input, utils.py:101-160): planes [n][7][7][7] float32, all zero for rows that ask for nothing --
 * the input of evaluators that take planes (a PyTorch module, a reference-style model.predict). */
int ccsp_encode_requests(const ccsp_request *req, int n, float *planes, void *stream);
/* the compact answer from a FULL policy: pk[i][j] = p[i][moves[i][j] & 0x1FF] for j < req[i].k (p [n][294] float64 = Model.predict's first
 * result, model.py:21-24), for the same evaluators */
int ccsp_gather_priors(const ccsp_request *req, const uint16_t *moves, const double *p, int n, double *pk, void *stream);
/* test hook: the built-in table evaluators (CCSP_EVAL_UNIFORM / HASH / FORWARD) as an external evaluator of requests -> (pk, v) */
int ccsp_debug_table_eval(int evaluator, const ccsp_request *req, const uint16_t *moves, int n, double *pk, float *v, void *stream);
/* The three limits of ccsp_advance below, per context (a value < 1 / < 0 / < 0 leaves that one as it is).  A context starts with the
 * process-wide defaults, which the ccsp_debug_advance_* hooks change for contexts created AFTERWARDS.  The limits are passed BY VALUE with
 * every ccsp_advance launch: a caller that captures its rounds into a hipGraph must set them BEFORE the capture (a captured launch keeps the
 * values it was captured with) or capture again afterwards -- selfplay.BatchSelfPlay.set_advance_limits drops its graph for that reason. */
int ccsp_set_advance_limits(ccsp_ctx *ctx, int budget, int time_cap_ticks, int deadline_ticks);
/* evaluator-free simulations (won leaves, reused positions) a slot takes up in ONE ccsp_advance before the selection that ends the call
 * (default 8; if that selection too ends on such a leaf the simulation is completed and the call ends without a request: bounds the launch's
 * length; results do not depend on it; at least 1: a slot must be able to get past a won leaf).  Returns the previous value; n < 1 only reads it. */
int ccsp_debug_advance_budget(int n);
/* ... and, past a slot's first such simulation in a call, a time: the slot takes up no further one once `ticks` x 10 ns have passed since its
 * wave began (a launch lasts as long as its slowest wave).  Default 5000 (50 us); 0 = no cap.  Returns the previous value; ticks < 0 only reads it. */
int ccsp_debug_advance_time_cap(int ticks);
/* ... and a deadline for the selection that ends a call: a slot that has done other work in this call (its answered leaf's expansion,
 * evaluator-free simulations) and is later than `ticks` x 10 ns gives the selection up, before it or between two of its levels, and leaves
 * no request (one idle evaluator row); it selects again in the next call, which begins with the selection and never gives up.
 * Default 8000 (80 us); 0 = none.  Both times are scaled down for contexts of 1024 slots or fewer (a shorter evaluator launch runs beside the call). */
int ccsp_debug_advance_deadline(int ticks);
/* diagnostic: the cycle sums ccsp_advance keeps under CCSP_ADVANCE_DEBUG (64 words; see advance_kernel); clear != 0 zeroes them */
int ccsp_debug_read(ccsp_ctx *ctx, unsigned long long *out, int clear);
/* diagnostic: the same per slot, raw ([n_slots][CCSP_DEBUG_WORDS_PER_SLOT]; words 16-19 = the slot's LAST ccsp_advance: begin and end on
 * the 100 MHz clock every CU shares, simulations completed, levels walked) -- tools/bench_free.py --debug draws a launch's timeline from it */
#define CCSP_DEBUG_WORDS_PER_SLOT 20
int ccsp_debug_read_slots(ccsp_ctx *ctx, unsigned long long *out);

/* ---- evaluator: the policy/value network as one fused kernel (row N1; Model.predict, model.py:21-24) ---- */

/* Host.  plain: the network's parameters in Keras order with every BatchNormalization (inference form,
 * eps 1e-3) folded into the convolution in front of it -- ccsp_net_plain_size() floats, order documented at
 * ccsp_net_pack in csrc/ccsp_net.hip.  packed: ccsp_net_packed_size() floats in the kernel's own order. */
int ccsp_net_plain_size(void);
int ccsp_net_packed_size(void);
int ccsp_net_pack(const float *plain, float *packed);

/* Device.  planes [n][7][7][7] f32 (utils.to_model_input layout) -> logits [n][294] f32 (or NULL),
 * p [n][294] f64 = float64 softmax of the logits (utils.softmax, utils.py:187-192; or NULL), v [n] f32.
 * packed = device copy of ccsp_net_pack's output. */
int ccsp_net_forward(const float *packed, const float *planes, int n, float *logits, double *p, float *v, void *stream);

/* The same network on a batch of REQUESTS of the free-running path (ccsp_advance / ccsp_boundary above): the input phase builds the planes
 * of req[i].state itself (C1 inside the evaluator: 64 bytes read per position instead of 1372), the epilogue writes the float64 softmax of
 * the k legal moves only (pk [n][CCSP_REQUEST_MOVES]) and v [n].  Bit-identical to ccsp_net_forward on ccsp_encode_requests' planes followed
 * by ccsp_gather_priors.  Rows with kind == 0 are evaluated on an all-zero input and not written. */
int ccsp_net_forward_requests(const float *packed, const ccsp_request *req, const uint16_t *moves, int n, double *pk, float *v, void *stream);

/* test / measurement hook: workgroup shape of ccsp_net_forward / ccsp_net_forward_requests -- 8, 4, 2 or 1 positions per workgroup; any other
 * value restores the default: by batch size (batches that cannot fill the GPU run in the shape whose single workgroup is done sooner: 1 up
 * to 256 positions, 2 up to 512, 4 up to 1024, 8 beyond).  Bit-identical results in every shape.  Returns the value in force (0 = by batch size). */
int ccsp_debug_net_shape(int positions_per_workgroup);

/* ---- read-back (synchronous; host buffers unless said otherwise) ----------------------------------- */
int ccsp_read_counters(ccsp_ctx *ctx, uint64_t *out /* [CCSP_CNT_COUNT] */);
int ccsp_read_visit_histogram(ccsp_ctx *ctx, uint64_t *out /* [294]: sum of root visit counts per action */);
int ccsp_read_slots(ccsp_ctx *ctx, uint8_t *status, uint32_t *ply, uint64_t *game, ccsp_state *state, uint8_t *player);
int ccsp_log_size(ccsp_ctx *ctx, uint64_t *n);
/* forget the rows read so far (stream-ordered): a caller that harvests the log every H plies needs n_slots x H rows of
 * capacity whatever the number of games (train.generate_self_play's list of games grows on the host, train.py:61-64) */
int ccsp_log_clear(ccsp_ctx *ctx, void *stream);
int ccsp_log_device_ptrs(ccsp_ctx *ctx, ccsp_state **state, ccsp_sample_meta **meta, double **pi);   /* device pointers */
int ccsp_read_log(ccsp_ctx *ctx, uint64_t first, uint64_t n, ccsp_state *state, ccsp_sample_meta *meta, double *pi);
int ccsp_read_results(ccsp_ctx *ctx, uint64_t first, uint64_t n, ccsp_game_result *out);
/* root edges of the tree of the search a slot FINISHED last: Edge.stats N, W, P (MCTS.py:31-36) and the action index of each edge.  Fused and
 * lock-step paths: valid until the next ply's root is expanded.  Free-running path: with CCSP_ADVANCE_REUSE the finished tree stays whole
 * while the next ply is searched; without, until the answer to the next root request has been taken. */
int ccsp_read_root(ccsp_ctx *ctx, int slot, int *k, uint32_t *N, double *W, double *P, uint16_t *mv);
/* test hook: digest of the whole tree in the reference's edge order (see gen_golden.py tree_digest) */
int ccsp_debug_tree_digest(ccsp_ctx *ctx, int slot, uint64_t *digest, uint64_t *nodes, uint64_t *edges);

#ifdef __cplusplus
}
#endif
#endif
