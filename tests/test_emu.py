"""Host-logic check of the expansion move: the SAME per-site bodies and orchestration that the HIP kernels run
(progressive-x_amd/csrc/maxflow_body.hip.h + maxflow_driver.inl), executed sequentially on the CPU in natural and in
shuffled "thread" order, must reproduce the oracle's Dinic min-cut labels bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import random_sym_graph, realistic_labeling_problem

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("PGX_EMU_SO") or os.path.join(HERE, "emu", "libmf_emu.so")   # (scripts/sanitize.sh: the sanitizer build)


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "mf_emu.cpp")
    deps = [src] + [os.path.join(HERE, "..", "progressive-x_amd", "csrc", f)
                    for f in ("maxflow_body.hip.h", "maxflow_driver.inl")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, src])
    return C.CDLL(SO)


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def emu_expand(emu, Dq, graph, lq, hq, alpha, labels, seed=0, sweeps_per_relabel=0):
    n, L = Dq.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32).copy()
    Dq = np.ascontiguousarray(Dq, dtype=np.int64)
    off, idx, mult = (np.ascontiguousarray(g, dtype=np.int32) for g in graph)
    if idx.size == 0:
        idx, mult = np.zeros(1, np.int32), np.ones(1, np.int32)
    ch = C.c_int64()
    st = np.zeros(8, np.int64)
    r = emu.emu_expand_alpha(C.c_int64(n), C.c_int(L), _p(Dq, C.c_int64), _p(off, C.c_int32), _p(idx, C.c_int32),
                             _p(mult, C.c_int32), C.c_int64(lq), C.c_int64(hq), C.c_int(alpha), _p(lab, C.c_int32),
                             C.c_uint64(seed), C.c_int(sweeps_per_relabel), C.byref(ch), _p(st, C.c_int64))
    assert r == 0, f"emulated push-relabel returned {r}"
    return lab, ch.value, st


def test_emulated_move_matches_oracle_random(emu, oracle):
    rng = np.random.default_rng(1)
    for trial in range(600):
        n, L = int(rng.integers(2, 40)), int(rng.integers(2, 6))
        Dq = rng.integers(0, 20, (n, L)).astype(np.int64)
        graph = random_sym_graph(rng, n, float(rng.choice([0.0, 0.1, 0.3])))
        lq, hq = int(rng.integers(0, 5)) * 2, int(rng.choice([0, 3, 10, 40]))
        lab = rng.integers(0, L, n).astype(np.int32)
        if rng.random() < 0.3:
            lab[:] = rng.integers(0, L)  # unused labels -> alpha hub in play
        alpha = int(rng.integers(0, L))
        ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, lab)
        for seed, spr in ((0, 0), (trial + 1, 0), (trial + 7, 1)):
            got, ch, _ = emu_expand(emu, Dq, graph, lq, hq, alpha, lab, seed=seed, sweeps_per_relabel=spr)
            assert np.array_equal(got, ref), (trial, seed)
            assert ch == ref_changed


def test_emulated_cycle_on_radius_graph(emu, oracle):
    Dq, graph = realistic_labeling_problem(3000, L=6, lam=0.3, seed=3)
    lq, hq = oracle.quantize_lambda(0.3), oracle.quantize(10.0)
    lab = np.zeros(3000, np.int32)
    for alpha in range(6):
        ref, _, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, lab)
        got, _, st = emu_expand(emu, Dq, graph, lq, hq, alpha, lab, seed=alpha)
        assert np.array_equal(got, ref)
        assert st[1] < 2000 and st[2] < 100  # sweeps / global relabels stay small on realistic energies
        lab = ref


def test_closed_form_lambda0_move_matches_oracle_with_ties(emu, oracle):
    """lambda = 0: the product solves a move in closed form (maxflow_l0.hip.h).  Tiny integer costs make ties between
    'switch all', 'switch individually' and 'nobody switches' frequent; every case must equal the oracle's min-cut."""
    rng = np.random.default_rng(7)
    for trial in range(4000):
        n, L = int(rng.integers(1, 12)), int(rng.integers(2, 6))
        Dq = rng.integers(0, 4, (n, L)).astype(np.int64)
        graph = (np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        hq = int(rng.integers(0, 7))
        lab = rng.integers(0, L, n).astype(np.int32)
        if rng.random() < 0.4:
            lab[:] = rng.integers(0, L)
        alpha = int(rng.integers(0, L))
        ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, 0, hq, alpha, lab)
        got, ch, _ = emu_expand(emu, Dq, graph, 0, hq, alpha, lab)
        assert np.array_equal(got, ref) and ch == ref_changed, (trial, n, L, hq, alpha)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_one_launch_schedules_leave_the_cut_unchanged(emu, oracle, monkeypatch, mode):
    """Round 6 (maxflow_xcd.hip.h): the later rounds of a move without hubs in one call (bit 0) and a whole global relabel in one
    call while the hubs are passive (bit 1), emulated sequentially with the product's own per-site step.  Rounds are cut short
    (one sweep per round) so that moves end their sweeps with work left and the new paths are the ones that run; the labels must
    be the oracle's on problems full of ties, with and without label costs, in natural and shuffled order."""
    monkeypatch.setenv("MF_EMU_XCD", str(mode))
    rng = np.random.default_rng(60 + mode)
    taken = 0
    for trial in range(300):
        n, L = int(rng.integers(2, 60)), int(rng.integers(2, 6))
        Dq = rng.integers(0, 20, (n, L)).astype(np.int64)
        graph = random_sym_graph(rng, n, float(rng.choice([0.05, 0.1, 0.3])))
        lq, hq = int(rng.integers(1, 5)) * 2, int(rng.choice([0, 3, 10, 40]))
        lab = rng.integers(0, L, n).astype(np.int32)
        if rng.random() < 0.3:
            lab[:] = rng.integers(0, L)
        alpha = int(rng.integers(0, L))
        ref, ref_changed, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, lab)
        for seed, spr in ((0, 1), (trial + 1, 1), (trial + 7, 2)):
            got, ch, st = emu_expand(emu, Dq, graph, lq, hq, alpha, lab, seed=seed, sweeps_per_relabel=spr)
            assert np.array_equal(got, ref), (trial, seed)
            assert ch == ref_changed
            taken += int(st[7])
    assert taken % 1000000 > 50 if mode & 1 else True     # round launches
    assert taken // 1000000 > 50 if mode & 2 else True    # one-launch searches
    # a whole cycle on a realistic energy
    Dq, graph = realistic_labeling_problem(3000, L=6, lam=0.3, seed=3)
    lq, hq = oracle.quantize_lambda(0.3), oracle.quantize(10.0)
    lab = np.zeros(3000, np.int32)
    for alpha in range(6):
        ref, _, _ = oracle.expand_alpha(Dq, graph, lq, hq, alpha, lab)
        got, _, st = emu_expand(emu, Dq, graph, lq, hq, alpha, lab, seed=alpha, sweeps_per_relabel=2)
        assert np.array_equal(got, ref)
        lab = ref
