"""
CPU tier (iii): host logic of the swarm driver -- the numpy mirror of the device kernels
(nmrfit_amd.pso.HostSwarm) with the oracle injected as the evaluator (test infrastructure;
the product never does this), the counter RNG, the pyswarm stopping rule and the sharding.
"""
import numpy as np
import pytest

from nmrfit_amd import pso, synth
from oracle import c_oracle
from oracle import nmrfit_oracle as onp


def _problem(N=512, P=2, seed=5):
    sp = synth.make_spectrum(N, P, seed=seed)

    def evaluate(X):
        return c_oracle.objective_batch(X, sp["w"], sp["u"], sp["v"], sp["weights"], threads=4)
    return sp, evaluate


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, expect in kat:
        out = pso.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], key[0], key[1])
        assert tuple(int(o[0]) for o in out) == expect


def test_uniform2_properties():
    a, b = pso.uniform2(1234, 3, 100, 22, offset=0)
    assert a.shape == b.shape == (100, 22)
    assert (a >= 0).all() and (a < 1).all() and (b >= 0).all() and (b < 1).all()
    assert abs(a.mean() - 0.5) < 0.03 and abs(b.mean() - 0.5) < 0.03
    # the stream of a particle depends on its GLOBAL index only
    a2, b2 = pso.uniform2(1234, 3, 40, 22, offset=60)
    np.testing.assert_array_equal(a[60:], a2)
    np.testing.assert_array_equal(b[60:], b2)
    a3, _ = pso.uniform2(1234, 4, 100, 22, offset=0)
    assert not np.array_equal(a, a3)


def test_shard_covers_swarm():
    for S in (1, 7, 204, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [pso.shard(S, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(n for _, n in spans) == S
            for (o1, n1), (o2, _) in zip(spans, spans[1:]):
                assert o1 + n1 == o2


def test_host_swarm_converges_and_matches_pyswarm_restatement_statistically():
    sp, evaluate = _problem()
    f_true = evaluate(sp["x_true"][None, :])[0]
    sw = pso.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=64, seed=11)
    x, f = pso.run_sharded(sw, pso.LocalExchange(), maxiter=150)
    # pyswarm's minfunc rule stops at the first improvement smaller than 1e-8, typically after
    # 50-150 generations here, at 1.0-1.5x the objective of the generating parameters
    assert sw.stop == 1 and f <= 1.6 * f_true
    assert (x >= sp["lower"]).all() and (x <= sp["upper"]).all()
    assert f == pytest.approx(evaluate(x[None, :])[0], rel=1e-15)
    # the oracle's restatement of pyswarm (different RNG, same rule) behaves the same
    # statistically: medians over 5 seeds of the objective each run stops at
    ours, theirs = [], []
    for seed in range(5):
        s = pso.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=64, seed=seed)
        ours.append(pso.run_sharded(s, pso.LocalExchange(), maxiter=300)[1])
        theirs.append(onp.pso(lambda x_, *a: onp.objective(x_, *a), sp["lower"], sp["upper"],
                              args=(sp["w"], sp["u"], sp["v"], sp["weights"]), swarmsize=64, maxiter=300,
                              omega=pso.DEFAULTS["omega"], phip=pso.DEFAULTS["phip"], phig=pso.DEFAULTS["phig"],
                              rng=np.random.default_rng(seed))[1])
    m_ours, m_theirs = np.median(ours), np.median(theirs)
    assert m_ours <= 1.5 * f_true and m_theirs <= 1.5 * f_true
    assert 0.7 <= m_ours / m_theirs <= 1.4


def test_stop_rules_follow_pyswarm():
    D = 7
    lb, ub = -np.ones(D), np.ones(D)
    calls = {"n": 0}

    def flat(X):            # improvement of 1e-9 per generation -> |fg - fc| <= minfunc on the first better one
        calls["n"] += 1
        return np.full(X.shape[0], 1.0 - 1e-9 * calls["n"]) + 1e-12 * np.arange(X.shape[0])

    sw = pso.HostSwarm(flat, lb, ub, swarmsize=10, seed=1)
    x, f = pso.run_sharded(sw, pso.LocalExchange(), maxiter=50)
    assert sw.stop == 1 and sw.iteration == 1       # stopped by the first generation after init
    assert f < sw.fg                                 # it returns (p_min, fp[i_min]), not (g, fg)
    np.testing.assert_array_equal(x, sw.best_x)

    def bowl(X):
        return np.sum(X * X, axis=1)

    sw = pso.HostSwarm(bowl, lb, ub, swarmsize=30, seed=2, minfunc=0.0, minstep=1e-3)
    pso.run_sharded(sw, pso.LocalExchange(), maxiter=500)
    assert sw.stop == 2                              # position change below minstep
    sw = pso.HostSwarm(bowl, lb, ub, swarmsize=30, seed=2, minfunc=-1.0, minstep=-1.0)
    pso.run_sharded(sw, pso.LocalExchange(), maxiter=25)
    assert sw.stop == 0 and sw.iteration == 25       # never stops: maximum iterations reached
    with pytest.raises(AssertionError):
        pso.HostSwarm(bowl, ub, lb, swarmsize=4)     # pyswarm: assert np.all(ub > lb)


def test_two_shards_in_process_equal_one_swarm_bitwise():
    """SURVEY 8(e) determinism: sharding must not change the trajectory."""
    sp, evaluate = _problem(N=256, P=1, seed=8)
    one = pso.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=21, seed=99)
    one.init()
    one.apply_global(one.candidate()[None, :])
    shards = []
    for r in range(3):
        off, n = pso.shard(21, r, 3)
        s = pso.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=21, offset=off, S_local=n, seed=99)
        s.init()
        shards.append(s)
    cands = np.stack([s.candidate() for s in shards])
    for s in shards:
        s.apply_global(cands)
    for _ in range(12):
        one.step_local()
        one.apply_global(one.candidate()[None, :])
        for s in shards:
            s.step_local()
        cands = np.stack([s.candidate() for s in shards])
        for s in shards:
            s.apply_global(cands)
        for s in shards:
            np.testing.assert_array_equal(s.g, one.g)
            assert s.fg == one.fg and s.stop == one.stop
    np.testing.assert_array_equal(np.concatenate([s.x for s in shards]), one.x)
    np.testing.assert_array_equal(np.concatenate([s.fp for s in shards]), one.fp)


def test_empty_shard_is_harmless():
    sp, evaluate = _problem(N=128, P=1, seed=8)
    s = pso.HostSwarm(evaluate, sp["lower"], sp["upper"], swarmsize=4, offset=4, S_local=0, seed=1)
    s.init()
    assert s.candidate()[0] == np.inf
