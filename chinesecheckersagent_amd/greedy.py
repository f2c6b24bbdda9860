"""next-4: the greedy opponent and the greedy data generator on the GPU.

    GreedyDataGenerator(randomised, random_start).generate_play()      data_generators.py:14-80
    generate_greedy_games(n, ...)                                      the same, n games as one batch
    agent_greedy_match(model, num_games)                               ai_vs_greedy.py:26-59
    greedy_vs_greedy(num_games)                                        greedy_vs_greedy.py / game.py:108-116

GreedyPlayer.decide_move (player.py:67-129, stochastic=False as Game builds it) runs in the engine's kernels
(wave_greedy_best of csrc/ccsp_engine.hip); its one random choice per ply reads the counter-based stream.
The generator's wall-clock STUCK_TIME_LIMIT (data_generators.py:66-67) is a ply count here (`stuck_limit`,
200 by default): a wall clock has no meaning for a batch."""
import numpy as np

from . import _lib
from .board import BoardView
from .config import DET_TREE_TAU, MCTS_SIMULATIONS
from .engine import SelfPlayEngine
from .selfplay import _default_seed, _next_game

BOARD_HIST_MOVES = 3              # config.py:11
AVERAGE_TOTAL_MOVE = 43           # config.py:77


def generate_greedy_games(n_games, randomised=False, random_start=False, seed=None, first_game=0, game_stride=1,
                          stuck_limit=200, device=0):
    """n_games of GreedyDataGenerator.generate_play as one batch -> [(play_history, reward)] in game-id order;
    play_history = [(BoardView, pi)] with pi = 1/len(best_moves) on the greedy moves of that position"""
    e = SelfPlayEngine(n_slots=n_games, sims=1, seed=_default_seed[0] if seed is None else seed, first_game=first_game,
                       game_stride=game_stride, max_games=n_games, log_capacity=n_games * (stuck_limit + 8),
                       randomised=randomised, device=device, greedy_data=True, stuck_limit=stuck_limit,
                       greedy=_lib.GREEDY_RANDOM_START if random_start else 0)
    try:
        for _ in range(64):
            e.play_plies(0, 64)
            if (e.slots()['status'] != _lib.ST_RUNNING).all():
                break
        st, meta, pi = e.log()
        res = e.results()
    finally:
        e.close()
    order = np.lexsort((meta['ply'], meta['game']))
    by_game = {}
    for r in order:
        by_game.setdefault(int(meta['game'][r]), []).append(r)
    out = []
    for k in range(n_games):
        game = first_game + k * game_stride
        status = int(res['status'][k])
        if status == _lib.ST_ERROR or status == _lib.ST_RUNNING:
            raise _lib.CcspError('greedy game %d ended in status %d' % (game, status))
        rows = by_game.get(game, [])
        if status == _lib.ST_DISCARD_NO_PROGRESS:
            rows = rows[:AVERAGE_TOTAL_MOVE]                # data_generators.py:66-67: stuck -> draw
        elif randomised:
            rows = rows[BOARD_HIST_MOVES:]                  # data_generators.py:77-78
        out.append(([(BoardView(st[r]), pi[r].copy()) for r in rows], int(res['reward'][k])))
    return out


class GreedyDataGenerator(object):
    """data_generators.GreedyDataGenerator: generate_play() -> (play_history, reward); games are numbered from the
    module-wide game counter (selfplay.set_seed), so consecutive calls give consecutive games"""

    def __init__(self, randomised=False, random_start=False, stuck_limit=200):
        self.randomised, self.random_start, self.stuck_limit = randomised, random_start, stuck_limit

    def generate_play(self):
        game = _next_game[0]
        _next_game[0] += 1
        return generate_greedy_games(1, self.randomised, self.random_start, first_game=game, stuck_limit=self.stuck_limit)[0]

    def generate_plays(self, n):
        first = _next_game[0]
        _next_game[0] += n
        return generate_greedy_games(n, self.randomised, self.random_start, first_game=first, stuck_limit=self.stuck_limit)


def greedy_vs_greedy(num_games, enforce_move_limit=False, seed=None, first_game=0, stochastic=(False, False)):
    """Game(p1_type='greedy', p2_type='greedy').start() num_games times (greedy_vs_greedy.py): -> {1: wins, 2: wins, None: draws}.
    stochastic = (player one, player two): that seat is GreedyPlayer(stochastic=True) (player.py:77-97) -- game.py:105-119's
    deterministic-against-stochastic statistics are greedy_vs_greedy(10000, stochastic=(False, True))."""
    bits = _lib.GREEDY_P1 | _lib.GREEDY_P2 | (_lib.GREEDY_STOCHASTIC_P1 if stochastic[0] else 0) | (_lib.GREEDY_STOCHASTIC_P2 if stochastic[1] else 0)
    e = SelfPlayEngine(n_slots=num_games, sims=1, seed=_default_seed[0] if seed is None else seed, first_game=first_game,
                       max_games=num_games, log_capacity=1, arena=True, enforce_move_limit=enforce_move_limit,
                       greedy=bits)
    try:
        for _ in range(256):
            e.play_plies(0, 32)
            if (e.slots()['status'] != _lib.ST_RUNNING).all():
                break
        res = e.results()
    finally:
        e.close()
    count = {1: 0, 2: 0, None: 0}
    for k in range(num_games):
        st = int(res['status'][k])
        if st == _lib.ST_ERROR:
            raise _lib.CcspError('greedy game %d ended in ERROR status' % k)
        count[st if st in (1, 2) else None] += 1
    return count


def agent_greedy_match(model, num_games, verbose=False, tree_tau=DET_TREE_TAU, sims=MCTS_SIMULATIONS, seed=None, first_game=0,
                       enforce_move_limit=False):
    """ai_vs_greedy.agent_greedy_match (ai_vs_greedy.py:26-59): the model plays player one in even games and player two
    in odd ones against a GreedyPlayer; returns `model` (as passed in), 'greedy', or None on equal wins.
    first_game must be even (the seats alternate with the game id)."""
    from .arena import BatchArena, _load
    assert first_game % 2 == 0
    b = BatchArena(_load(model), None, num_games, sims=sims, seed=seed, first_game=first_game, tree_tau=tree_tau,
                   enforce_move_limit=enforce_move_limit, greedy=_lib.GREEDY_P2 | _lib.GREEDY_ALTERNATE)
    try:
        winners, _ = b.run()
    finally:
        b.close()
    win = {'ai': 0, 'greedy': 0}
    for i, w in enumerate(winners):
        if w is None:
            continue
        ai_seat = 1 if i % 2 == 0 else 2
        win['ai' if w == ai_seat else 'greedy'] += 1
    if win['ai'] > win['greedy']:
        return model
    if win['greedy'] > win['ai']:
        return 'greedy'
    return None


# ---- train_on_greedy.py: supervised bootstrap on greedy games -------------------------------------------------------
G_AVG_GAME_LEN = 21                     # config.py:61
G_DATA_RETENTION_RATE = 1. / G_AVG_GAME_LEN
G_ITER_PER_EPOCH = 100                  # config.py:64 (Keras `epochs` of one fit)
G_GAMES_PER_EPOCH = 15000               # config.py:65
G_VAL_SPLIT = 0.1
G_NORMAL_GAME_RATIO = 0.2
G_RAND_START_GAME_RATIO = 0.3
G_MODEL_PREFIX = 'greedy-model'
G_BATCH_SIZE = 32


def generate_greedy_training_games(num_self_play, seed=None, first_game=0, stuck_limit=200):
    """train_on_greedy.generate_self_play (train_on_greedy.py:15-41): 20 % normal starts, 30 % random starts, the rest
    randomised boards -- three batches on the GPU instead of a worker loop"""
    n_normal = int(G_NORMAL_GAME_RATIO * num_self_play)
    n_rs = int(G_RAND_START_GAME_RATIO * num_self_play)
    n_rand = num_self_play - n_normal - n_rs
    games = []
    g0 = first_game
    for n, randomised, random_start in ((n_normal, False, False), (n_rs, False, True), (n_rand, True, False)):
        if n > 0:
            games += generate_greedy_games(n, randomised, random_start, seed=seed, first_game=g0, stuck_limit=stuck_limit)
            g0 += n
    return games


def train_on_greedy(num_games, model_path, version, save_dir='saved-weights/', epochs=G_ITER_PER_EPOCH, seed=0, device=None):
    """train_on_greedy.train (train_on_greedy.py:73-120): greedy games -> convert / augment -> keep a random
    1/G_AVG_GAME_LEN of the samples -> fit (validation_split 0.1, batch 32, `epochs` epochs) -> save
    '{save_dir}/greedy-model{version:0>4}-weights.h5'.  `model_path` = weights to continue from, or None.  Returns the path."""
    from . import utils
    from .train import Trainer
    games = generate_greedy_training_games(num_games, seed=seed)
    board_x, pi_y, v_y = utils.convert_to_train_data(games)
    board_x, pi_y, v_y = utils.augment_train_data(board_x, pi_y, v_y)
    assert len(board_x) == len(pi_y) == len(v_y)
    n_train = int(G_DATA_RETENTION_RATE * len(board_x))
    idx = np.random.RandomState(seed).choice(len(board_x), n_train, replace=False)
    bx = np.array([board_x[i] for i in idx]); py = np.array([pi_y[i] for i in idx]); vy = np.array([v_y[i] for i in idx])
    t = Trainer(device=device)
    if model_path is not None:
        t.load_weights(model_path)
    t.fit(bx, py, vy, batch_size=G_BATCH_SIZE, epochs=epochs, validation_split=G_VAL_SPLIT, seed=seed)
    return t.save_weights(save_dir, G_MODEL_PREFIX, version)


if __name__ == '__main__':
    # ai_vs_greedy.py / greedy_vs_greedy.py from the command line:
    #   python -m chinesecheckersagent_amd.greedy <model.h5> [--games 20] [--sims 175]      model vs greedy, seats alternating
    #   python -m chinesecheckersagent_amd.greedy --games 50                                 greedy vs greedy
    import argparse
    ap = argparse.ArgumentParser(description='matches against the greedy player on the GPU')
    ap.add_argument('model', nargs='?')
    ap.add_argument('--games', type=int, default=20)
    ap.add_argument('--sims', type=int, default=MCTS_SIMULATIONS)
    ap.add_argument('--seed', type=int, default=None)
    a = ap.parse_args()
    if a.model:
        r = agent_greedy_match(a.model, a.games, sims=a.sims, seed=a.seed)
        print('winner over %d games: %s' % (a.games, r if r is not None else 'nobody (equal wins)'))
    else:
        c = greedy_vs_greedy(a.games, seed=a.seed)
        print('player one wins %d, player two wins %d, %d without a winner' % (c[1], c[2], c[None]))
