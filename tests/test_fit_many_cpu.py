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
    assert ka is not None and ka == kb and kc != ka                 # peak counts may differ, grid lengths may not
    assert key(a, options={"swarmsize": 100}) != ka
    assert key(a, fit_im=True) not in (None, ka) and key(a, fit_im="sum") not in (None, ka, key(a, fit_im=True))
    assert key(big, fit_im="sum")[3] == _cabi.VARIANT_DEFAULT          # (every peak's imaginary line: the direct kernel)
    assert key(a, options={"polish": True}) is None
    assert key(a, options={"exchange": object()}) is None
    assert key(a, options={"variant": "norec"}) is None
    assert key(big)[3] == _cabi.VARIANT_FARFIELD and key(a)[3] == _cabi.VARIANT_DEFAULT


def test_jobs_with_a_communicator_run_one_after_another(monkeypatch):
    """fit_many(jobs, threads>1, options={'exchange': ex}): one communicator would otherwise serve several swarms at
    once from several host threads.  The fits run serially, in job order, on the calling thread."""
    calls = []

    def fake_fit(self):
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
    jobs = [_job(1024, 2, 10 + k, options={"tag": k, "polish": True}) for k in range(5)]      # (polish: not batchable)
    out = core.fit_many(jobs, threads=4)
    assert sorted(t for _, t in calls) == [0, 1, 2, 3, 4] and [f.options["tag"] for f in out] == [0, 1, 2, 3, 4]


def test_long_job_lists_go_through_spans_and_leftovers_find_partners(monkeypatch):
    """The host side of the pipeline (no device: the batch's creation and run are replaced): a list longer than
    BATCH_JOBS is cut into spans of equal size, every span's groups of equal key become batches made on the second
    thread and run on the calling one in span order, a job without a partner inside its span is batched with the
    leftovers of the other spans, and what stays alone goes through fit()."""
    monkeypatch.setattr(core, "BATCH_JOBS", 4)
    made, ran, lone = [], [], []

    class FakeBatch:
        def close(self):
            pass

    def fake_create(fits, plans, key):
        made.append((threading.get_ident(), [f.options["tag"] for f in fits]))
        return FakeBatch(), fits, plans, key

    def fake_finish(fb, fits, plans, key):
        ran.append((threading.get_ident(), [f.options["tag"] for f in fits]))
        for f in fits:
            f.params, f.error = np.zeros(len(f.lower)), 0.0

    def fake_fit(self):
        lone.append(self.options["tag"])
        self.params, self.error = np.zeros(len(self.lower)), 0.0
    monkeypatch.setattr(core, "_batch_create", fake_create)
    monkeypatch.setattr(core, "_batch_finish", fake_finish)
    monkeypatch.setattr(utils.FitUtility, "fit", fake_fit)
    # 10 jobs -> 3 spans of 4, 4, 2.  Grid lengths: span 0 = [A A A B], span 1 = [A A B C], span 2 = [A A]
    lengths = [1024, 1024, 1024, 2048, 1024, 1024, 2048, 4096, 1024, 1024]
    jobs = [_job(n, 2, 20 + k, options={"tag": k}) for k, n in enumerate(lengths)]
    out = core.fit_many(jobs, threads=1)
    assert [f.options["tag"] for f in out] == list(range(10)) and all(f.error == 0.0 for f in out)
    me = threading.get_ident()
    assert [tags for _, tags in ran] == [[0, 1, 2], [4, 5], [8, 9], [3, 6]]      # spans in order, then the leftovers' batch
    assert all(tid == me for tid, _ in ran)                                      # the device is driven from the calling thread
    assert [tags for _, tags in made[:3]] == [[0, 1, 2], [4, 5], [8, 9]] and all(tid != me for tid, _ in made[:3])
    assert lone == [7]                                                           # the only 4096-point job


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
