"""Host-side mirror of the reference helpers the self-play path's callers use (utils.py): the action
index codec (C2), the reward, and the (state, pi, z) -> board_x / pi_y / v_y output format (O1).
Plane encoding for whole batches runs on the GPU (rules.encode); the per-board to_model_input here is
the reference-shaped convenience for single boards."""
import os

import numpy as np

from .config import (BOARD_HEIGHT, BOARD_HIST_MOVES, BOARD_WIDTH, NUM_CHECKERS, PLAYER_ONE, PLAYER_TWO, REWARD, SAVE_TRAIN_DATA_DIR,
                     SAVE_TRAIN_DATA_PREF)

NO_MOVE = 255          # ccsp_state.last: no move recorded (the matching history plane is all-zero)


def encode_checker_index(checker_id, coord):
    """utils.py:164-171"""
    return checker_id * BOARD_WIDTH * BOARD_HEIGHT + coord[0] * BOARD_WIDTH + coord[1]


def decode_checker_index(model_output_index):
    """utils.py:175-183"""
    checker_id = model_output_index // (BOARD_WIDTH * BOARD_HEIGHT)
    offset = model_output_index % (BOARD_WIDTH * BOARD_HEIGHT)
    return checker_id, (offset // BOARD_WIDTH, offset % BOARD_WIDTH)


def softmax(x):
    """utils.py:187-192"""
    x = np.copy(x).astype('float64')
    x -= np.max(x, axis=-1, keepdims=True)
    e = np.exp(x)
    return e / np.sum(e, axis=-1, keepdims=True)


def get_p1_winloss_reward(board, winner=None):
    """utils.py:34-44"""
    winner = winner or board.check_win()
    if winner == PLAYER_ONE:
        return REWARD['win']
    if winner == PLAYER_TWO:
        return REWARD['lose']
    return REWARD['draw']


def to_model_input(board, cur_player):
    """utils.py:101-160 for one Board-like object (BoardView or anything with .board, .checkers_pos,
    .hist_moves): 7x7x7 float64."""
    out = np.zeros((BOARD_WIDTH, BOARD_HEIGHT, BOARD_HIST_MOVES * 2 + 1))
    op_player = PLAYER_ONE + PLAYER_TWO - cur_player
    cur_layer = np.zeros((BOARD_WIDTH, BOARD_HEIGHT))
    op_layer = np.zeros((BOARD_WIDTH, BOARD_HEIGHT))
    for cid, rc in board.checkers_pos[cur_player].items():
        cur_layer[rc] = cid + 1
    for cid, rc in board.checkers_pos[op_player].items():
        op_layer[rc] = cid + 1
    out[:, :, 0], out[:, :, 1] = cur_layer, op_layer
    hist = list(board.hist_moves)
    moved, idx = op_player, len(hist) - 1
    for ch in range(1, BOARD_HIST_MOVES):
        if not np.any(board.board[:, :, ch]):
            break
        orig, dest = hist[idx]
        layer = cur_layer if moved == cur_player else op_layer
        layer[dest], layer[orig] = layer[orig], layer[dest]
        idx -= 1
        moved = PLAYER_ONE + PLAYER_TWO - moved
        out[:, :, ch * 2], out[:, :, ch * 2 + 1] = cur_layer, op_layer
    if cur_player == PLAYER_TWO:
        out[:, :, BOARD_HIST_MOVES * 2] = 1.0
    return out


def states_to_model_input(states, players):
    """to_model_input for MANY 32-byte records at once (numpy, no Python loop over positions):
    states = structured array with 'pos' [2][6] and 'last' [4] (ccsp_state), players = player to move per record.
    -> float64 [N, 7, 7, 7], the same values as to_model_input(BoardView(record), player) row by row.
    Built by SCATTERING the twelve checker ids of each of the six board planes straight into the zeroed float64 result (6 x 6 + 1
    stores per position instead of several passes over 343 values: the conversion of a GPU's sample rows runs beside the GPU on a
    fraction of a host core).  Undoing a move on a layer (utils.py:135-155) swaps two cells: on the list of checker positions that
    is "a checker on one of the two cells stands on the other"."""
    states = np.asarray(states)
    n = len(states)
    pos = np.asarray(states['pos'], dtype=np.int64).reshape(n, 12)
    last = np.asarray(states['last'], dtype=np.int64).reshape(n, 4)
    pl = np.asarray(players, dtype=np.int64).reshape(n)
    nc = BOARD_WIDTH * BOARD_HEIGHT
    nch = BOARD_HIST_MOVES * 2 + 1
    out = np.zeros((n, nc, nch), dtype=np.float64)
    if n == 0:
        return out.reshape(n, BOARD_WIDTH, BOARD_HEIGHT, nch)
    flat = out.reshape(-1)
    ids = np.arange(1, NUM_CHECKERS + 1, dtype=np.float64)[None, :]
    one = (pl == PLAYER_ONE)[:, None]
    mine = np.where(one, pos[:, :6], pos[:, 6:])
    theirs = np.where(one, pos[:, 6:], pos[:, :6])
    base = (np.arange(n, dtype=np.int64) * (nc * nch))[:, None]

    def put(rows, cells, ch):
        """plane `ch` of the positions `rows` (None = all): id i + 1 on the cell of checker i"""
        if rows is None:
            flat[(base + cells * nch + ch).reshape(-1)] = np.broadcast_to(ids, cells.shape).reshape(-1)
        elif len(rows):
            flat[(base[rows] + cells[rows] * nch + ch).reshape(-1)] = np.broadcast_to(ids, (len(rows), NUM_CHECKERS)).reshape(-1)

    def swapped(cells, f, t):
        f, t = f[:, None], t[:, None]
        return np.where(cells == f, t, np.where(cells == t, f, cells))
    put(None, mine, 0)
    put(None, theirs, 1)
    # one ply back: the opponent's last move undone on the opponent's layer (utils.py:135-155); NO_MOVE ends the history
    m1 = last[:, 0] != NO_MOVE
    r1 = np.nonzero(m1)[0]
    theirs1 = swapped(theirs, last[:, 0], last[:, 1])
    put(r1, mine, 2)
    put(r1, theirs1, 3)
    # two plies back: the mover's own previous move undone on its layer
    m2 = m1 & (last[:, 2] != NO_MOVE)
    r2 = np.nonzero(m2)[0]
    mine2 = swapped(mine, last[:, 2], last[:, 3])
    put(r2, mine2, 4)
    put(r2, theirs1, 5)
    out[pl == PLAYER_TWO, :, nch - 1] = 1.0
    return out.reshape(n, BOARD_WIDTH, BOARD_HEIGHT, nch)


def log_to_train_data(states, meta, pi, results, first_game=0, game_stride=1, randomised=False, return_games=False):
    """The engine's sample log + result table -> (board_x [N,7,7,7] f64, pi_y [N,294] f64, v_y [N] int64) without a
    Python object per position: convert_to_train_data(collect()) for large batches.  Same rows in the same order
    (games by id, plies in order; won games only; the first BOARD_HIST_MOVES rows of a randomised game dropped,
    selfplay.py:76-78) and the same labelling: the reference labels the first RETURNED row of a game as player one's
    and alternates from there (utils.py:64-72), whatever the row's real player to move -- kept as it is."""
    from . import _lib
    meta = np.asarray(meta)
    order = np.lexsort((meta['ply'], meta['game']))
    g = np.asarray(meta['game'], dtype=np.int64)[order]
    k = (g - first_game) // game_stride                              # row of the result table
    status = np.asarray(results['status'], dtype=np.int64)
    won = (status[k] == _lib.ST_WON_P1) | (status[k] == _lib.ST_WON_P2)
    # index of the row within its game (rows are sorted by game, ply)
    start = np.r_[0, np.nonzero(np.diff(g))[0] + 1]
    idx_in_game = np.arange(len(g)) - np.repeat(start, np.diff(np.r_[start, len(g)]))
    drop = BOARD_HIST_MOVES if randomised else 0
    keep = won & (idx_in_game >= drop)
    sel = order[keep]
    j = idx_in_game[keep] - drop                                     # index in the returned play_history
    label_player = np.where(j % 2 == 0, PLAYER_ONE, PLAYER_TWO)
    board_x = states_to_model_input(np.asarray(states)[sel], label_player)
    reward = np.asarray(results['reward'], dtype=np.int64)[k[keep]]
    v_y = np.where(j % 2 == 0, reward, -reward).astype(np.int64)
    if return_games:                                                 # + the game id of every row (rows of a game are contiguous, in ply order)
        return board_x, np.asarray(pi, dtype=np.float64)[sel], v_y, g[keep]
    return board_x, np.asarray(pi, dtype=np.float64)[sel], v_y


def _records_of(boards):
    """BoardView objects -> their 32-byte records as one structured array (None if some object is not a BoardView)"""
    from ._lib import STATE_DTYPE
    if not all(hasattr(b, 'pos12') and hasattr(b, 'last4') for b in boards):
        return None
    rec = np.zeros(len(boards), dtype=STATE_DTYPE)
    rec['pos'] = np.array([b.pos12 for b in boards], dtype=np.uint8).reshape(-1, 2, NUM_CHECKERS)
    rec['last'] = np.array([b.last4 for b in boards], dtype=np.uint8).reshape(-1, 4)
    return rec


def convert_to_train_data(self_play_games):
    """[(play_history, p1_reward)] -> (board_x, pi_y, v_y) as python lists, the output contract of utils.py:60-73: row j
    of a game is encoded for player one if j is even, player two otherwise, and carries the reward with alternating sign.
    The planes of all rows come from ONE call of the vectorised encoder (states_to_model_input)."""
    boards, pi_y, labels, v_y = [], [], [], []
    for history, reward in self_play_games:
        n = len(history)
        boards += [b for b, _ in history]
        pi_y += [pi for _, pi in history]
        labels += [PLAYER_ONE if j % 2 == 0 else PLAYER_TWO for j in range(n)]
        v_y += [reward if j % 2 == 0 else -reward for j in range(n)]
    rec = _records_of(boards)
    if rec is not None:
        board_x = list(states_to_model_input(rec, np.array(labels, dtype=np.int64))) if len(boards) else []
    else:                                              # foreign Board-like objects (e.g. the reference's own Board)
        board_x = [to_model_input(b, p) for b, p in zip(boards, labels)]
    return board_x, pi_y, v_y


def augment_train_data(board_x, pi_y, v_y):
    """The symmetry augmentation of utils.py:77-97 on whole arrays: every sample is appended once more with its planes
    mirrored along the anti-diagonal (what fliplr(rot90(.)) does to each 7x7 channel: out[r, c] = in[6 - c, 6 - r]).
    As in the reference pi is copied UN-mirrored (a reference quirk kept for drop-in equality) and the
    same list objects are returned, extended in place."""
    n = len(board_x)
    if n:
        x = np.asarray(board_x)
        mirrored = x[:, ::-1, ::-1, :].transpose(0, 2, 1, 3)
        board_x += list(np.ascontiguousarray(mirrored))
        pi_y += [np.copy(p) for p in pi_y[:n]]
        v_y += list(v_y[:n])
    return board_x, pi_y, v_y


def save_train_data(board_x, pi_y, v_y, version, directory=SAVE_TRAIN_DATA_DIR):
    """utils.py:48-56: generated-training-data/data-for-iter-{version}.h5 with datasets board_x
    [N,7,7,7] f64, pi_y [N,294] f64, v_y [N] int64 -- written by h5lite (readable by h5py / the
    reference's train.py and combine_data.py)."""
    from .h5lite import write_datasets
    if not os.path.exists(directory):
        os.makedirs(directory)
    path = '{}/{}{}.h5'.format(directory, SAVE_TRAIN_DATA_PREF, version)
    write_datasets(path, [('board_x', np.asarray(board_x, dtype=np.float64)),
                          ('pi_y', np.asarray(pi_y, dtype=np.float64)),
                          ('v_y', np.asarray(v_y, dtype=np.int64))])
    return path
