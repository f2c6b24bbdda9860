"""Batched rules on the GPU (ccsp_movegen / ccsp_step / ccsp_encode of include/ccsp.h): the
array-at-a-time counterparts of Board.get_valid_moves (board.py:215-222), Board.place
(board.py:226-250) and utils.to_model_input (utils.py:101-160).  Inputs are torch CUDA tensors
(uint8 views of the 32-byte records) or numpy arrays, which are copied to the device first."""
import numpy as np

from . import _lib
from ._lib import MAX_MOVES, PLANES, STATE_DTYPE, check
from .engine import _stream_ptr


def to_device_states(states):
    """numpy structured array (STATE_DTYPE) -> torch uint8 CUDA tensor [n, 32]"""
    import torch
    states = np.ascontiguousarray(states, dtype=STATE_DTYPE)
    return torch.from_numpy(states.view(np.uint8).reshape(-1, 32).copy()).cuda()


def _dev_u8(x):
    import torch
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint8)).cuda()
    assert x.is_cuda and x.dtype == torch.uint8 and x.is_contiguous()
    return x


def movegen(states_dev, player_dev, want_masks=True):
    """-> (moves uint8 [n,126,2] (id, dest), count uint8 [n], dest_mask int64 [n,6] or None), CUDA tensors"""
    import torch
    _lib.require_gpu()
    states_dev, player_dev = _dev_u8(states_dev), _dev_u8(player_dev)
    n = states_dev.shape[0]
    moves = torch.zeros((n, MAX_MOVES, 2), dtype=torch.uint8, device='cuda')
    count = torch.zeros(n, dtype=torch.uint8, device='cuda')
    masks = torch.zeros((n, 6), dtype=torch.int64, device='cuda') if want_masks else None
    check(_lib.lib().ccsp_movegen(states_dev.data_ptr(), player_dev.data_ptr(), n, moves.data_ptr(), count.data_ptr(),
                                  masks.data_ptr() if want_masks else None, _stream_ptr()), 'ccsp_movegen')
    return moves, count, masks


def movegen_packed(states_dev, player_dev, want_masks=True):
    """ccsp_movegen_packed: the same lists, those of each chunk of 32 positions back to back.
    -> (moves uint8 [n * 126, 2] (id, dest) in the packed layout, count uint8 [n], dest_mask or None); list_starts() gives the
    entry at which every position's list begins"""
    import torch
    _lib.require_gpu()
    states_dev, player_dev = _dev_u8(states_dev), _dev_u8(player_dev)
    n = states_dev.shape[0]
    moves = torch.zeros((n * MAX_MOVES, 2), dtype=torch.uint8, device='cuda')
    count = torch.zeros(n, dtype=torch.uint8, device='cuda')
    masks = torch.zeros((n, 6), dtype=torch.int64, device='cuda') if want_masks else None
    check(_lib.lib().ccsp_movegen_packed(states_dev.data_ptr(), player_dev.data_ptr(), n, moves.data_ptr(), count.data_ptr(),
                                         masks.data_ptr() if want_masks else None, _stream_ptr()), 'ccsp_movegen_packed')
    return moves, count, masks


def list_starts(count, chunk=32):
    """entry of the packed layout at which the list of every position starts: 126 * chunk * (i // chunk) + the counts of the
    positions before i in its chunk (count: torch tensor or array of n counts) -> int64 tensor / array [n]"""
    import torch
    if isinstance(count, np.ndarray):
        c = count.astype(np.int64)
        n = len(c)
        pad = np.zeros((n + chunk - 1) // chunk * chunk, dtype=np.int64)
        pad[:n] = c
        pad = pad.reshape(-1, chunk)
        excl = np.cumsum(pad, axis=1) - pad
        return (excl + np.arange(pad.shape[0])[:, None] * (MAX_MOVES * chunk)).reshape(-1)[:n]
    c = count.long()
    n = c.shape[0]
    pad = torch.zeros((n + chunk - 1) // chunk * chunk, dtype=torch.int64, device=c.device)
    pad[:n] = c
    pad = pad.view(-1, chunk)
    excl = torch.cumsum(pad, dim=1) - pad
    return (excl + torch.arange(pad.shape[0], device=c.device)[:, None] * (MAX_MOVES * chunk)).reshape(-1)[:n]


def step(states_dev, player_dev, mv_dev):
    """mv uint8 [n,2] (id, dest) -> (next states uint8 [n,32], winner uint8 [n], progress uint8 [n,2])"""
    import torch
    _lib.require_gpu()
    states_dev, player_dev, mv_dev = _dev_u8(states_dev), _dev_u8(player_dev), _dev_u8(mv_dev)
    n = states_dev.shape[0]
    out = torch.empty_like(states_dev)
    winner = torch.zeros(n, dtype=torch.uint8, device='cuda')
    progress = torch.zeros((n, 2), dtype=torch.uint8, device='cuda')
    check(_lib.lib().ccsp_step(states_dev.data_ptr(), player_dev.data_ptr(), mv_dev.data_ptr(), n, out.data_ptr(),
                               winner.data_ptr(), progress.data_ptr(), _stream_ptr()), 'ccsp_step')
    return out, winner, progress


def encode(states_dev, player_dev, out=None):
    """-> float32 [n,7,7,7] (row, col, channel), the model input of utils.to_model_input"""
    import torch
    _lib.require_gpu()
    states_dev, player_dev = _dev_u8(states_dev), _dev_u8(player_dev)
    n = states_dev.shape[0]
    if out is None:
        out = torch.empty((n, 7, 7, 7), dtype=torch.float32, device='cuda')
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n * PLANES
    check(_lib.lib().ccsp_encode(states_dev.data_ptr(), player_dev.data_ptr(), n, out.data_ptr(), _stream_ptr()), 'ccsp_encode')
    return out


def greedy_best(states_dev, player_dev):
    """GreedyPlayer.decide_move(training=True) (player.py:72-118) for every position:
    -> (best uint8 [n,32,2] (id, dest) in get_valid_moves order, count uint8 [n]), CUDA tensors"""
    import torch
    _lib.require_gpu()
    states_dev, player_dev = _dev_u8(states_dev), _dev_u8(player_dev)
    n = states_dev.shape[0]
    best = torch.zeros((n, _lib.GREEDY_MAX, 2), dtype=torch.uint8, device='cuda')
    count = torch.zeros(n, dtype=torch.uint8, device='cuda')
    check(_lib.lib().ccsp_greedy_best(states_dev.data_ptr(), player_dev.data_ptr(), n, best.data_ptr(), count.data_ptr(),
                                      _stream_ptr()), 'ccsp_greedy_best')
    return best, count
