"""The product's lane-local device functions (chinesecheckersagent_amd/csrc/ccsp_rules.h)
compiled for the host by tests/host_check and checked against the golden vectors and the
oracle -- so rule/ordering/arithmetic bugs are caught here, without a GPU.  (The GPU parity
tests in test_gpu_*.py remain the real gate: this only checks the scalar logic.)"""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_ffi as orc
from test_oracle_rules import replay_all

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def hc():
    src = os.path.join(HERE, 'host_check', 'host_check.cpp')
    so = os.path.join(HERE, 'host_check', 'libhost_check.so')
    hdr = os.path.join(HERE, '..', 'chinesecheckersagent_amd', 'csrc', 'ccsp_rules.h')
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-o', so, src])
    L = C.CDLL(so)
    u8p = C.POINTER(C.c_uint8)
    L.hc_movegen.restype = C.c_int; L.hc_movegen.argtypes = [u8p, C.c_int, u8p, C.POINTER(C.c_uint64)]
    L.hc_movegen_lines.restype = C.c_int; L.hc_movegen_lines.argtypes = [u8p, C.c_int, u8p]
    L.hc_movegen_stack.restype = C.c_int; L.hc_movegen_stack.argtypes = [u8p, C.c_int, u8p]
    L.hc_stack_depth.restype = C.c_int; L.hc_stack_depth.argtypes = [u8p, C.c_int]
    L.hc_step.restype = C.c_int; L.hc_step.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, u8p, u8p]
    L.hc_progress.restype = C.c_int; L.hc_progress.argtypes = [u8p, C.c_int]
    L.hc_planes.argtypes = [u8p, u8p, C.c_int, u8p]
    L.hc_planes_scatter.argtypes = [u8p, u8p, C.c_int, u8p]
    L.hc_rng.restype = C.c_uint64
    L.hc_rng.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.hc_choice.restype = C.c_uint32; L.hc_choice.argtypes = [C.c_uint64, C.c_uint32]
    L.hc_det_log.restype = C.c_double; L.hc_det_log.argtypes = [C.c_double]
    L.hc_det_exp.restype = C.c_double; L.hc_det_exp.argtypes = [C.c_double]
    L.hc_gamma.restype = C.c_double; L.hc_gamma.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double]
    L.hc_hash_eval.argtypes = [u8p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.hc_forward_eval.argtypes = L.hc_hash_eval.argtypes
    L.hc_div_sweep.restype = C.c_long; L.hc_div_sweep.argtypes = [C.c_int, C.c_int]
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def test_rules_digests_match_reference(hc, golden_dir):
    rules = np.load(golden_dir + '/rules.npz')

    def movegen(pos12, player):
        a = np.ascontiguousarray(pos12, dtype=np.uint8)
        out = np.zeros((126, 2), dtype=np.uint8)
        masks = (C.c_uint64 * 6)()
        n = hc.hc_movegen(_p(a), int(player), _p(out), masks)
        mv = out[:n]
        out2 = np.zeros((126, 2), dtype=np.uint8)                  # the line-table form the tree kernels use
        assert hc.hc_movegen_lines(_p(a), int(player), _p(out2)) == n and (out2[:n] == mv).all()
        out3 = np.zeros((126, 2), dtype=np.uint8)                  # the explicit-stack form both GPU generators run
        assert hc.hc_movegen_stack(_p(a), int(player), _p(out3)) == n and (out3[:n] == mv).all()
        for cid in range(6):                       # dest_mask = exactly the destinations of that checker
            want = 0
            for d in mv[mv[:, 0] == cid][:, 1]:
                want |= 1 << int(d)
            assert masks[cid] == want
        return mv.copy()

    def planes(pos12, last, player):
        a = np.ascontiguousarray(pos12, dtype=np.uint8)
        l = np.ascontiguousarray(last, dtype=np.uint8)
        out = np.zeros(343, dtype=np.uint8)
        hc.hc_planes(_p(a), _p(l), int(player), _p(out))
        out2 = np.full(343, 9, dtype=np.uint8)
        hc.hc_planes_scatter(_p(a), _p(l), int(player), _p(out2))      # the form the encode kernel uses
        assert (out == out2).all()
        return out

    def step(pos12, last, player, cid, dest):
        a = np.ascontiguousarray(pos12, dtype=np.uint8)
        l = np.ascontiguousarray(last, dtype=np.uint8)
        npos = np.zeros(12, dtype=np.uint8)
        nlast = np.zeros(4, dtype=np.uint8)
        w = hc.hc_step(_p(a), _p(l), int(player), int(cid), int(dest), _p(npos), _p(nlast))
        return npos, nlast, w

    def progress(pos12, player):
        a = np.ascontiguousarray(pos12, dtype=np.uint8)
        return hc.hc_progress(_p(a), int(player))

    n, hm, hs, hp = replay_all(rules, movegen, planes, step, progress)
    assert n == int(rules['n_records'])
    assert hm == rules['sha_moves'].tobytes()
    assert hs == rules['sha_step'].tobytes()
    assert hp == rules['sha_planes'].tobytes()


def test_wins(hc, golden_dir):
    z = np.load(golden_dir + '/wins.npz')
    npos = np.zeros(12, dtype=np.uint8)
    nlast = np.zeros(4, dtype=np.uint8)
    for r in z['moves']:
        a = np.ascontiguousarray(r[:12])
        assert hc.hc_step(_p(a), _p(orc.NO_LAST), int(r[12]), int(r[13]), int(r[14]), _p(npos), _p(nlast)) == int(r[15])


def test_draw_spec(hc, golden_dir):
    doc = json.load(open(golden_dir + '/rng.json'))
    ka, seed = doc['known'], doc['seed']
    bits = lambda x: struct.unpack('<Q', struct.pack('<d', float(x)))[0]
    frombits = lambda b: struct.unpack('<d', struct.pack('<Q', b))[0]
    for k, y in ka['rng']:
        assert hc.hc_rng(*k) == y
    for u, n, y in ka['choice']:
        assert hc.hc_choice(u, n) == y
    for x, y in ka['det_log']:
        assert bits(hc.hc_det_log(frombits(x))) == y
    for x, y in ka['det_exp']:
        assert bits(hc.hc_det_exp(frombits(x))) == y
    for g, ply, e, y in ka['gamma']:
        assert bits(hc.hc_gamma(seed, g, ply, e, 0.03)) == y
    # many more gammas against the oracle
    L = orc.lib()
    for g in range(300):
        assert bits(hc.hc_gamma(seed, g, g % 90, g % 70, 0.03)) == bits(L.orc_gamma_small(seed, g, g % 90, g % 70, 0.03))
    for pos12, player, key, ps, v in ka['hash_eval']:
        a = np.array(pos12, dtype=np.uint8)
        p = (C.c_double * 294)()
        vv = C.c_float()
        hc.hc_hash_eval(_p(a), player, p, C.byref(vv))
        assert [bits(p[i]) for i in range(8)] + [bits(p[293])] == ps and bits(vv.value) == v
    for pos12, player, ps, v in ka['forward_eval']:
        a = np.array(pos12, dtype=np.uint8)
        p = (C.c_double * 294)()
        vv = C.c_float()
        hc.hc_forward_eval(_p(a), player, p, C.byref(vv))
        assert [bits(p[i]) for i in range(8)] + [bits(p[293])] == ps and bits(vv.value) == v


def test_table_division_is_correctly_rounded(hc):
    """PUCT's two divisions (MCTS.py:62, 89/118) go through a reciprocal table in the fused kernel"""
    assert hc.hc_div_sweep(4200, 2500) == 0
    assert hc.hc_div_sweep(70000, 60) == 0


def test_explicit_stack_depth_stays_far_below_the_lds_allotment(hc, golden_dir):
    """the kernels give each hop-search stack 96 (tree) / 92 (batched generator) bytes of LDS; the provable bound is
    81 entries (16 sub-lattice cells, one pop and at most six pushes each); what positions actually reach is far less"""
    g = np.load(golden_dir + '/rules.npz')
    deepest = 0
    for p12, pl in zip(g['pos12'], g['player']):
        deepest = max(deepest, hc.hc_stack_depth(_p(np.ascontiguousarray(p12)), int(pl)))
    rng = np.random.RandomState(12)
    for _ in range(60000):                                   # random placements, incl. dense hop lattices
        cells = np.ascontiguousarray(rng.permutation(49)[:12].astype(np.uint8))
        deepest = max(deepest, hc.hc_stack_depth(_p(cells), 1 + int(rng.randint(2))))
    # adversarial: 12 checkers on one sub-lattice's neighbours maximise hop connectivity
    for shift in range(49):
        cells = np.ascontiguousarray(((np.arange(12) * 2 + shift) % 49).astype(np.uint8))
        if len(set(cells.tolist())) == 12:
            deepest = max(deepest, hc.hc_stack_depth(_p(cells), 1), hc.hc_stack_depth(_p(cells), 2))
    assert 4 <= deepest <= 40, deepest


def test_sanitizers_find_nothing(tmp_path):
    """AddressSanitizer + UndefinedBehaviorSanitizer (CPU build: GPU sanitizers are not available on the pool) over the host build
    of the device functions and the C oracle: 1500 openings played 24 plies deep through every formulation of the move generator,
    step, planes, evaluator tables, random stream and samplers -- no out-of-bounds table index, no undefined shift, and the
    formulations agree on every position"""
    here = os.path.join(HERE, 'host_check')
    root = os.path.join(HERE, '..')
    flags = ['-O1', '-g', '-ffp-contract=off', '-fsanitize=address,undefined', '-fno-sanitize-recover=all']
    obj, exe = str(tmp_path / 'orc_san.o'), str(tmp_path / 'sanitize_main')
    subprocess.check_call(['gcc'] + flags + ['-w', '-c', '-o', obj, os.path.join(root, 'oracle', 'ccsp_oracle.c')])
    subprocess.check_call(['g++', '-std=c++17'] + flags + ['-w', '-o', exe, os.path.join(here, 'sanitize_main.cpp'), obj, '-lm'])
    r = subprocess.run([exe, '1500'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert r.returncode == 0 and 'mismatches 0' in r.stdout, r.stdout[-2000:]
