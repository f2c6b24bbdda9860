"""next-3: arena evaluation on the batched GPU tree -- the reference's
ai_vs_ai.agent_match (ai_vs_ai.py:28-52) and evaluate_models.evaluate (evaluate_models.py:10-53) /
train.evaluate (train.py:150-186), i.e. Game.start (game.py:58-100) between two AiPlayers
(player.py:133-166): no random opening, no root pre-expansion, no Dirichlet noise, tree_tau switched to
DET_TREE_TAU once total_moves > 16, the game's own repetition rule and optional 100-move limit.
All games of a match are played as ONE batch."""
import numpy as np

from . import _lib
from .config import DET_TREE_TAU, MCTS_SIMULATIONS
from .engine import SelfPlayEngine
from .selfplay import _batched, _default_seed, _strict, _warn


class BatchArena(object):
    def __init__(self, model1, model2, n_games, sims=MCTS_SIMULATIONS, seed=None, first_game=0, tree_tau=DET_TREE_TAU,
                 enforce_move_limit=False, alternate=False, device=0, greedy=0, use_graph=True, game_stride=1, index0=0):
        """game k of this batch is game index0 + k * game_stride of the match (id first_game + k * game_stride): a rank of an
        N-GPU match passes first_game = base + rank, game_stride = N, index0 = rank"""
        import torch
        self.torch = torch
        self.m1, self.m2 = _batched(model1), _batched(model2 if model2 is not None else model1)
        self.eng = SelfPlayEngine(n_slots=n_games, sims=sims, seed=_default_seed[0] if seed is None else seed,
                                  first_game=first_game, game_stride=game_stride, max_games=n_games, log_capacity=n_games * 1024,
                                  device=device, arena=True, arena_det_tau=(tree_tau == DET_TREE_TAU),
                                  enforce_move_limit=enforce_move_limit,
                                  greedy=greedy)         # next-4: GreedyPlayer seats (_lib.GREEDY_*) move without a search
        dev = torch.device('cuda', device)
        self.planes = torch.zeros((n_games, 7, 7, 7), dtype=torch.float32, device=dev)
        # evaluate_models.py:37-41: odd games of the match swap colours (model2 plays player one)
        swap = ((index0 + np.arange(n_games) * game_stride) % 2 == 1) if alternate else np.zeros(n_games, dtype=bool)
        self.swap = torch.from_numpy(swap).to(dev)
        self.swap_host = swap
        self.n_games, self.sims = n_games, sims
        self._root_is_p2 = torch.zeros(n_games, dtype=torch.bool, device=dev)
        self.use_graph, self._graph = bool(use_graph), None

    def _evaluate(self, root_is_p2):
        p1, v1 = self.m1.evaluate_batch(self.planes)
        if self.m2 is self.m1:                            # one model on both sides: one forward, nothing to choose
            return p1, v1
        p2, v2 = self.m2.evaluate_batch(self.planes)
        use2 = root_is_p2 ^ self.swap                     # AiPlayer(player_num=2, model=model2) (game.py:24-30)
        return self.torch.where(use2[:, None], p2, p1).contiguous(), self.torch.where(use2, v2, v1).contiguous()

    def _capture(self, root_is_p2):
        """the simulation steps of a move (select kernel -> forward(s) -> expand/backup kernel) captured once into a
        hipGraph, several steps per graph, as selfplay.BatchSelfPlay does; falls back to plain launches if capture fails"""
        torch, e = self.torch, self.eng
        n = self.sims - 1
        self._unroll = max(k for k in (24, 20, 16, 12, 10, 8, 6, 5, 4, 3, 2, 1) if n % k == 0) if n > 0 else 0
        if not (self.use_graph and self._unroll and hasattr(self.m1, 'model') and hasattr(self.m2, 'model')):
            self.use_graph = False
            return
        selected = False
        try:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):                   # warm-up on a side stream (allocator)
                for _ in range(2):
                    self._evaluate(root_is_p2)
            torch.cuda.current_stream().wait_stream(st)
            g = torch.cuda.CUDAGraph()
            keep = []
            with torch.cuda.graph(g):
                e.select(self.planes)
                selected = True
                for i in range(self._unroll):             # [forward(s) -> expand/backup + the next selection] in one tree launch
                    gp, gv = self._evaluate(root_is_p2)
                    if i + 1 < self._unroll:
                        e.expand_backup_select(gp, gv, self.planes)
                    else:
                        e.expand_backup(gp, gv)
                        selected = False
                    keep.append((gp, gv))
            self._graph, self._graph_out = g, keep
        except Exception as ex:
            self.use_graph, self._graph = False, None
            if selected:                                  # close the half-captured step on the host side
                p, v = self._evaluate(root_is_p2)
                e.expand_backup(p, v)
            if _strict():
                raise
            _warn('hipGraph capture of the arena\'s simulation steps failed (%r): this match runs on plain launches' % (ex,))

    def play_move(self):
        e = self.eng
        e.ply_begin(self.planes)
        self._root_is_p2.copy_(self.planes[:, 0, 0, 6] == 1)
        root_is_p2 = self._root_is_p2
        p, v = self._evaluate(root_is_p2)
        e.root_expand(p, v)                               # = the search's first simulation (MCTS.py:123-125 on a leaf root)
        if self.use_graph and self._graph is None:
            self._capture(root_is_p2)
        if self._graph is not None:
            for _ in range((self.sims - 1) // self._unroll):
                self._graph.replay()
        elif self.sims > 1:
            e.select(self.planes)
            for i in range(self.sims - 1):
                p, v = self._evaluate(root_is_p2)
                if i + 2 < self.sims:
                    e.expand_backup_select(p, v, self.planes)
                else:
                    e.expand_backup(p, v)
        e.ply_end()

    def run(self, max_moves=4096):
        for i in range(max_moves):
            self.play_move()
            if i % 8 == 7 and (self.eng.slots()['status'] != _lib.ST_RUNNING).all():
                break
        res = self.eng.results()
        winners = []
        for k in range(self.n_games):
            st = int(res['status'][k])
            if st == _lib.ST_ERROR:
                raise _lib.CcspError('arena game %d ended in ERROR status' % k)
            w = st if st in (_lib.ST_WON_P1, _lib.ST_WON_P2) else None       # Game.start returns None on repetition / limit
            winners.append(w)
        return winners, res

    def close(self):
        self.eng.close()


def agent_match(model1, model2, num_games, verbose=False, tree_tau=DET_TREE_TAU, enforce_move_limit=False,
                sims=MCTS_SIMULATIONS, seed=None, first_game=0):
    """ai_vs_ai.agent_match (ai_vs_ai.py:28-52): model1 plays player one in every game; returns model1 or model2
    (whatever was passed in: a path or a model object) if it wins more than int(0.55 * num_games) games, else None."""
    m1, m2 = _load(model1), _load(model2)
    b = BatchArena(m1, m2, num_games, sims=sims, seed=seed, first_game=first_game, tree_tau=tree_tau,
                   enforce_move_limit=enforce_move_limit)
    try:
        winners, _ = b.run()
    finally:
        b.close()
    win_count = {1: sum(1 for w in winners if w == 1), 2: sum(1 for w in winners if w == 2)}
    if win_count[1] > int(0.55 * num_games):
        return model1
    if win_count[2] > int(0.55 * num_games):
        return model2
    return None


def evaluate(model1, model2, num_games, enforce_move_limit=False, sims=MCTS_SIMULATIONS, seed=None, first_game=0,
             tree_tau=DET_TREE_TAU, dist=None, device=0):
    """evaluate_models.evaluate (evaluate_models.py:10-53) / train.evaluate (train.py:150-186, which passes
    enforce_move_limit=True): colours alternate with the game index; returns (model1 wins, model2 wins, draws).
    With a process group (`dist`: evaluate_in_parallel, evaluate_models.py:57-102, one rank per GPU) rank r plays the games
    r, r + R, ... of the match and the three counts meet in one all-reduce: every rank returns the match's totals."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    mine = len(range(rank, num_games, world))
    w1 = w2 = d = 0
    if mine > 0:
        m1, m2 = _load(model1, device), _load(model2, device)
        b = BatchArena(m1, m2, mine, sims=sims, seed=seed, first_game=first_game + rank, tree_tau=tree_tau,
                       enforce_move_limit=enforce_move_limit, alternate=True, device=device, game_stride=world, index0=rank)
        try:
            winners, _ = b.run()
        finally:
            b.close()
        for i, w in enumerate(winners):
            if w is None:
                d += 1
            elif (w == 1) != bool(b.swap_host[i]):
                w1 += 1
            else:
                w2 += 1
    if dist is not None:
        import torch
        from .launch import coll_device
        t = torch.tensor([w1, w2, d], dtype=torch.int64, device=coll_device(dist))
        dist.all_reduce(t)
        w1, w2, d = (int(x) for x in t.cpu())
    return w1, w2, d


def _load(model, device=None):
    if isinstance(model, str):                             # load_agent (ai_vs_ai.py:15-25)
        from .model import ResidualCNN
        m = ResidualCNN(device=None if device is None else 'cuda:%d' % device)
        m.load_weights(model)
        return m
    return model


if __name__ == '__main__':
    # evaluate_models.py / ai_vs_ai.py from the command line:
    #   python -m chinesecheckersagent_amd.arena <model1.h5> <model2.h5> [--games 24] [--sims 175] [--no-limit] [--seed S]
    import argparse
    ap = argparse.ArgumentParser(description='model-vs-model match on the GPU, colours alternating (evaluate_models.py)')
    ap.add_argument('model1')
    ap.add_argument('model2')
    ap.add_argument('--games', type=int, default=24)
    ap.add_argument('--sims', type=int, default=MCTS_SIMULATIONS)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--no-limit', action='store_true', help='do not enforce the 100-move limit')
    a = ap.parse_args()
    w1, w2, d = evaluate(a.model1, a.model2, a.games, enforce_move_limit=not a.no_limit, sims=a.sims, seed=a.seed)
    print('%s wins %d, %s wins %d, %d draws of %d games' % (a.model1, w1, a.model2, w2, d, a.games))
