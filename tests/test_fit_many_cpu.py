"""
CPU tier: the host logic of nmrfit_amd.fit_many that needs no GPU -- which jobs may share a device batch, that jobs
carrying a communicator run one after another (a communicator serves one swarm at a time: ADVICE r4), the
spectra-parallel multi-GPU mode (jobs divided over ranks, results gathered over the rendezvous channel, real
processes, world 2 and 3), and the warning for swarm shards that are smaller than the exchange they add.
The reference's counterpart is the per-spectrum loop over nmrfit.fit (nmrfit/core.py:64) and its process pool over
particles (nmrfit/utils.py:182).
"""
import json
import os
import subprocess
import sys
import threading
import warnings

import numpy as np
import pytest

from nmrfit_amd import _cabi, core, synth, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _job(N, P, seed, **extra):
    sp = synth.make_spectrum(N, P, seed=seed)
    job = dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]), upper=list(sp["upper"]))
    job.update(extra)
    return job


def test_batch_key_groups_equal_shapes_only():
    def key(job, **kw):
        args = dict(job, **kw)
        f = utils.FitUtility(args.pop("data"), args.pop("lower"), args.pop("upper"), **args)
        return f._batch_key(f._plan())
    a, b, c = _job(4096, 6, 1), _job(4096, 4, 2), _job(8192, 6, 3)
    big = _job(32768, 12, 4)
    ka, kb, kc = key(a), key(b), key(c)
    assert ka is not None and ka == kb == kc                        # peak counts and (round 6) grid lengths may differ
    assert key(a, options={"swarmsize": 100}) == ka                  # (round 6: swarm sizes may differ too)
    assert key(a, options={"maxiter": 100}) != ka and key(a, options={"check_every": 8}) != ka
    assert key(a, fit_im=True) not in (None, ka) and key(a, fit_im="sum") not in (None, ka, key(a, fit_im=True))
    assert key(big, fit_im="sum")[3] == _cabi.VARIANT_FARFIELD         # (every peak's imaginary line on a large grid: the far-field
                                                                       # kernel, batched too since round 6)
    assert key(_job(8192, 6, 5), fit_im="sum")[3] == _cabi.VARIANT_DEFAULT
    assert key(a, options={"polish": True}) == ka                     # (round 6: the swarm of a polished fit runs in the batch)
    assert key(a, options={"exchange": object()}) is None
    assert key(a, options={"variant": "norec"}) is None
    assert key(big)[3] == _cabi.VARIANT_FARFIELD and key(a)[3] == _cabi.VARIANT_DEFAULT


def test_jobs_with_a_communicator_run_one_after_another(monkeypatch):
    """fit_many(jobs, threads>1, options={'exchange': ex}): one communicator would otherwise serve several swarms at
    once from several host threads.  The fits run serially, in job order, on the calling thread."""
    calls = []

    def fake_fit(self, plan=None):
        calls.append((threading.get_ident(), self.options.get("tag")))
        self.params, self.error = np.zeros(len(self.lower)), 0.0
    monkeypatch.setattr(utils.FitUtility, "fit", fake_fit)
    ex = object()
    jobs = [_job(1024, 2, 10 + k, options={"exchange": ex, "tag": k}) for k in range(5)]
    out = core.fit_many(jobs, threads=4)
    assert [t for _, t in calls] == [0, 1, 2, 3, 4]
    assert {tid for tid, _ in calls} == {threading.get_ident()}
    assert len(out) == 5
    # without a communicator the same call may use the pool (order of completion is free, results keep job order)
    calls.clear()
    jobs = [_job(1024, 2, 10 + k, options={"tag": k, "variant": "norec"}) for k in range(5)]      # (NOREC: not batchable)
    out = core.fit_many(jobs, threads=4)
    assert sorted(t for _, t in calls) == [0, 1, 2, 3, 4] and [f.options["tag"] for f in out] == [0, 1, 2, 3, 4]


class _FakeBatch:
    """Stands in for batch.FitBatch where no device is: records which thread ran it."""

    def __init__(self, log, tags):
        self.log, self.tags, self.closed = log, tags, False

    def run(self, maxiter, check_every):
        self.log.append((threading.get_ident(), self.tags))

    def close(self):
        self.closed = True


def _fake_pipeline(monkeypatch, refuse=()):
    """core's device calls replaced: returns the logs (made, ran, collected, lone, batches)."""
    made, ran, collected, lone, batches = [], [], [], [], []

    def fake_create(fits, plans, key):
        tags = [f.options["tag"] for f in fits]
        if any(t in refuse for t in tags):
            raise _cabi.NmrfitError(_cabi.E_UNSUPPORTED, "too many peaks for the kernel's LDS records in a batched launch")
        made.append((threading.get_ident(), tags))
        batches.append(_FakeBatch(ran, tags))
        return batches[-1], fits, plans, key

    def fake_read(fb, fits, scale=False):
        fb.close()
        return [fb.tags, threading.get_ident()], None, scale

    def fake_store(fits, plans, key, status, best, results, scale=False, threads=1):
        collected.append((threading.get_ident(), status[0], scale))
        assert threads >= 1 and results == scale
        for f in fits:
            f.params, f.error = np.zeros(len(f.lower)), 0.0

    def fake_collect(fb, fits, plans, key, scale=False, threads=1):
        fake_store(fits, plans, key, *fake_read(fb, fits, scale), scale, threads)

    def fake_fit(self, plan=None):
        lone.append((self.options["tag"], plan is not None))
        self.params, self.error = np.zeros(len(self.lower)), 0.0

    def fake_generate(self, scale=1):
        self.generated = scale
    monkeypatch.setattr(core, "_batch_create", fake_create)
    monkeypatch.setattr(core, "_batch_collect", fake_collect)
    monkeypatch.setattr(core, "_batch_read", fake_read)
    monkeypatch.setattr(core, "_batch_store", fake_store)
    monkeypatch.setattr(utils.FitUtility, "fit", fake_fit)
    monkeypatch.setattr(utils.FitUtility, "generate_result", fake_generate)
    return made, ran, collected, lone, batches


def test_long_job_lists_go_through_spans_and_leftovers_find_partners(monkeypatch):
    """The host side of the pipeline (no device: the batch's creation, run and read-back are replaced): a list longer
    than BATCH_JOBS is cut into spans of equal size, every span's groups of equal key become batches made on the second
    thread, run two at a time on runner threads and read back, in job order, on another; a job without a partner inside
    its span is batched with the leftovers of the other spans, and what stays alone goes through fit() with its plan."""
    monkeypatch.setattr(core, "BATCH_JOBS", 4)
    made, ran, collected, lone, batches = _fake_pipeline(monkeypatch)
    # 10 jobs -> 3 spans of 4, 4, 2.  maxiter (a batch's fits share one): span 0 = [A A A B], span 1 = [A A B C], span 2 = [A A]
    sizes = [100, 100, 100, 50, 100, 100, 50, 25, 100, 100]
    jobs = [_job(1024, 2, 20 + k, options={"tag": k, "maxiter": n}) for k, n in enumerate(sizes)]
    out = core.fit_many(jobs, threads=1)
    assert [f.options["tag"] for f in out] == list(range(10)) and all(f.error == 0.0 for f in out)
    me = threading.get_ident()
    order = [[0, 1, 2], [4, 5], [8, 9], [3, 6]]                                  # spans in order, then the leftovers' batch
    assert sorted(tags for _, tags in ran) == sorted(order)
    assert all(tid != me for tid, _ in ran) and len({tid for tid, _ in ran}) <= core.RUN_AT_ONCE   # runner threads drive the device
    assert [tags for _, tags in made[:3]] == order[:3] and all(tid != me for tid, _ in made[:3])
    assert [tags for _, tags, _ in collected] == order                           # results stored in job order ...
    assert len({tid for tid, _, _ in collected}) == 1 and collected[0][0] not in (me, made[0][0])   # ... on a thread of its own
    assert all(scale is False for _, _, scale in collected) and all(b.closed for b in batches)
    assert lone == [(7, True)]                                                   # the only maxiter = 25 job: its plan is reused
    assert not hasattr(out[7], "generated")


def test_generate_reaches_batches_and_lone_fits(monkeypatch):
    """generate=True / a scale: the batches' read-back gets the scale, fits that ran alone get generate_result(scale)."""
    monkeypatch.setattr(core, "BATCH_JOBS", 4)
    made, ran, collected, lone, _ = _fake_pipeline(monkeypatch)
    jobs = [_job(1024, 2, 40 + k, options={"tag": k, "maxiter": n}) for k, n in enumerate([100, 100, 50])]
    out = core.fit_many(jobs, threads=1, generate=True)
    assert [(tags, scale) for _, tags, scale in collected] == [([0, 1], 1)] and out[2].generated == 1
    collected.clear()
    out = core.fit_many(jobs, threads=2, generate=2.5)
    assert [(tags, scale) for _, tags, scale in collected] == [([0, 1], 2.5)] and out[2].generated == 2.5


def test_a_batch_the_device_refuses_runs_as_lone_fits(monkeypatch):
    """NMRFIT_E_UNSUPPORTED (or an allocation failure) from the batch's creation must not abort fit_many: the
    group's fits run one by one, the other groups stay batched."""
    monkeypatch.setattr(core, "BATCH_JOBS", 8)
    made, ran, collected, lone, batches = _fake_pipeline(monkeypatch, refuse={2})
    sizes = [100, 100, 50, 50, 50]
    jobs = [_job(1024, 2, 60 + k, options={"tag": k, "maxiter": n}) for k, n in enumerate(sizes)]
    out = core.fit_many(jobs, threads=1)
    assert [tags for _, tags in ran] == [[0, 1]]
    assert sorted(t for t, _ in lone) == [2, 3, 4] and all(had_plan for _, had_plan in lone)
    assert all(f.error == 0.0 for f in out) and all(b.closed for b in batches)


def test_an_error_in_a_run_closes_every_batch(monkeypatch):
    monkeypatch.setattr(core, "BATCH_JOBS", 2)
    made, ran, collected, lone, batches = _fake_pipeline(monkeypatch)

    def bad_run(self, maxiter, check_every):
        if self.tags == [2, 3]:
            raise _cabi.NmrfitError(_cabi.E_HIP, "injected")
        ran.append((threading.get_ident(), self.tags))
    monkeypatch.setattr(_FakeBatch, "run", bad_run)
    jobs = [_job(1024, 2, 80 + k, options={"tag": k}) for k in range(6)]
    with pytest.raises(_cabi.NmrfitError):
        core.fit_many(jobs, threads=1)
    assert [0, 1] in [tags for _, tags in ran] and [2, 3] not in [tags for _, tags in ran]
    assert all(b.closed for b in batches) and len(batches) >= 2


def test_the_rank_device_reaches_jobs_that_bring_their_own_options(monkeypatch):
    """a job's own `options` replaces the shared dict when the two are merged, so the device of a
    sharded (or devices=[...]) call must be stamped into every job's options -- a seed per job is exactly what callers
    pass."""
    seen = []

    def fake_local(jobs, threads, batch, kwargs, generate=False):
        seen.append([dict(j.get("options") or {}) for j in jobs])
        out = []
        for j in jobs:
            f = utils.FitUtility(j["data"], j["lower"], j["upper"], summary=False, options=j.get("options", {}))
            f.params, f.error, f.seed = np.zeros(len(j["lower"])), 0.0, 1
            out.append(f)
        return out
    monkeypatch.setattr(core, "_fit_many_local", fake_local)
    monkeypatch.setattr(core, "_cabi_device_count", lambda: 8)
    jobs = [_job(512, 2, 90 + k, options={"seed": k}) for k in range(4)] + [_job(512, 2, 99)]

    class Chan:
        def all_gather(self, blob):
            return [blob]

        def close(self):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    core._fit_many_sharded(jobs, 1, True, dict(summary=False, options={"maxiter": 5}), 0, 1, channel=Chan())
    assert [o["device"] for o in seen[-1]] == [3] * 5
    assert [o.get("seed") for o in seen[-1]] == [0, 1, 2, 3, None] and all(o["maxiter"] == 5 for o in seen[-1])
    # a device the job names itself wins in the sharded mode ...
    core._fit_many_sharded([_job(512, 2, 1, options={"device": 6})], 1, True, dict(summary=False), 0, 1, channel=Chan())
    assert seen[-1][0]["device"] == 6
    # ... and devices=[...] assigns -- without a TypeError when the job's options already hold a device
    seen.clear()
    core.fit_many([_job(512, 2, 2, options={"device": 0, "seed": 1}), _job(512, 2, 3)], devices=[4, 5])
    got = sorted((o["device"], o.get("seed")) for share in seen for o in share)
    assert got == [(4, 1), (5, None)]


def test_batch_size_rule():
    """Jobs per device batch: a quarter of the list, between 40 and 200 (profiles/r06/fit_many_span_sweep.txt); a number in
    core.BATCH_JOBS overrides."""
    assert core.BATCH_JOBS is None
    assert [core._batch_jobs(n) for n in (2, 40, 100, 200, 400, 1000, 5000)] == [40, 40, 40, 50, 100, 200, 200]
    try:
        core.BATCH_JOBS = 64
        assert core._batch_jobs(1000) == 64
    finally:
        core.BATCH_JOBS = None


def test_small_shard_warning():
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert utils.small_shard_warning(26, 4096, 6, world=8) is True          # the reference's default fit over 8 GPUs
        assert utils.small_shard_warning(4096, 65536, 24, world=8) is False      # C4's shard: worth sharding
        assert utils.small_shard_warning(204, 4096, 6, world=1) is False
        assert utils.small_shard_warning(26, 4096, 6, world=8, rank=3) is True   # (silent on the other ranks)
    assert len(rec) == 1 and "fit_many(jobs, shard=True)" in str(rec[0].message)


_RANK = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from nmrfit_amd import core, rendezvous, synth, utils

rank, _, world = rendezvous.env_rank_world()
jobs = []
for k in range(7):
    sp = synth.make_spectrum(512, 2, seed=20 + k)
    jobs.append(dict(data=synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), lower=list(sp["lower"]), upper=list(sp["upper"])))

def local(my_jobs):          # stands in for the GPU: a result that names the job and the rank that "fitted" it
    out = []
    for job in my_jobs:
        f = utils.FitUtility(job["data"], job["lower"], job["upper"], summary=False)
        f.params = np.asarray(job["lower"]) + rank
        f.error = float(np.sum(job["upper"]))
        f.seed = 1000 + rank
        out.append(f)
    return out

res = core._fit_many_sharded(jobs, 1, True, dict(summary=False), rank, world, local=local)
print(json.dumps(dict(rank=rank, owners=[int(round(r.params[0] - j["lower"][0])) for r, j in zip(res, jobs)],
                      errors=[r.error for r in res], seeds=[r.seed for r in res])))
"""


@pytest.mark.parametrize("world", [2, 3])
def test_jobs_shard_over_ranks_and_every_rank_gets_every_result(world, tmp_path):
    """Spectra-parallel mode, real processes: rank r fits jobs r, r + world, ...; the records travel over the
    standard-library rendezvous channel; every rank ends up with all seven results in job order."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK % dict(root=ROOT))
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", NMRFIT_RDZV_TOKEN="t%d" % os.getpid())
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    want_owner = [k % world for k in range(7)]
    for o in outs:
        assert o["owners"] == want_owner, o
        assert o["seeds"] == [1000 + r for r in want_owner]
        assert o["errors"] == outs[0]["errors"]
