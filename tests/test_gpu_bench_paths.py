"""
GPU tier (iv), end to end through bench.py: the one-rank path, the RCCL path (a communicator
created through the C-ABI, ncclAllGather inside nmrfit_pso_step) forced with a single rank, the
self-launching form (`python bench.py --gpus N`, no launcher) and the torchrun form, both with
several ranks sharing the one GPU of the test box (the candidate record staged through the host
over sockets: RCCL itself refuses two ranks on one device).  All must report the same swarm best
bit for bit: sharding and the exchange path do not change the trajectory.  The 8-GPU RCCL run
itself belongs to the driver; this covers its code path as far as one GPU can (a GPU box allows
at most 6 processes on its card, so the C4 rehearsal runs 4 ranks x 4096 particles and the
8-shard identity is tested in one process, tests/test_gpu_pso.py).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra, timeout=900, expect_rc=0, stderr_has=()):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=timeout)
    assert out.returncode == expect_rc, (out.returncode, out.stderr[-2000:])
    for needle in stderr_has:
        assert needle in out.stderr, (needle, out.stderr[-2000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    # the driver reads ONE JSON line from stdout: nothing else may be printed there
    assert [l for l in out.stdout.splitlines() if l.strip()] == lines, out.stdout[:2000]
    return json.loads(lines[0])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_paths_agree():
    common = ["--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--workload", "C2", "--preheat-seconds", "0.1"]
    plain = _run([sys.executable, "bench.py", "--swarm-per-gpu", "512"] + common, {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_streaming", "step_ms", "kernel_ms",
                "scaling_model"):
        assert key in plain, key
    assert plain["dtype"] == "f64" and plain["vs_baseline"] is None and plain["n_gpus"] == 1
    # `roofline` names the resource that binds (fp64 VALU issue); the metric's "HBM GB/s vs peak" under the
    # streaming-operand model is `roofline_streaming`, an effective rate
    assert plain["roofline"]["bound"] == "fp64_valu_issue"
    r = plain["roofline_streaming"]
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    # the kernel is timed inside the steps that contain it
    assert plain["kernel_ms"]["n"] == 5 and plain["step_ms"]["n"] == 5
    assert plain["kernel_ms"]["mean"] <= plain["ms_per_step"]
    assert plain["kernel_ms"]["max"] <= plain["step_ms"]["max"] * 1.001
    assert 500.0 < plain["roofline"]["clock_mhz_in_run"] < 3000.0
    sm = plain["scaling_model"]
    assert sm["expected_ms_per_step"] == [plain["ms_per_step"]] * 2 and sm["measured_over_expected"] is None
    assert "error" not in plain
    # both peaks of SURVEY 8(d)(i), the kind of rate, the spectrum the headline is measured on
    # and the dense-spectrum figure beside it
    assert r["rate_kind"] == "effective" and r["peak_measured_copy"] == 6290.0
    assert r["frac_vs_measured_copy_peak"] == pytest.approx(r["achieved"] / 6290.0)
    assert plain["config"]["spectrum"].startswith("sparse lines")
    ds = plain["dense_spectrum"]
    assert ds["kernel_ms"] > 0 and ds["units_per_s"] == pytest.approx(512 * 4096 * 6 / (ds["kernel_ms"] * 1e-3))
    assert ds["farfield"]["kernel_ms"] > 0 and "broad overlapping" in ds["spectrum"]
    ic = plain["imaginary_channel"]      # round 6: the imaginary-channel kernels ride in the line
    assert all(ic[v][k] > plain["kernel_ms"]["mean"] * 0.3 for v in ("default", "farfield") for k in ("fit_im_true_ms", "fit_im_sum_ms"))
    # socket power and shader clock from the device's hwmon files (null where they are not readable): this run's own
    # evidence that the part sits at its power cap under the fp64 load and the clock is what is left
    assert "power" in plain
    if plain["power"] is not None:
        pw = plain["power"]
        assert pw["samples"] >= 1 and 50.0 < pw["socket_power_w"] <= 1.1 * (pw["power_cap_w"] or 2000.0)
        assert 90.0 < pw["sclk_mhz"] < 3000.0 and "/hwmon" in pw["source"]
    rccl = _run([sys.executable, "bench.py", "--swarm-per-gpu", "512"] + common, {"NMRFIT_BENCH_FORCE_DIST": "1"})
    assert rccl["config"]["swarm_best_f"] == plain["config"]["swarm_best_f"]
    assert "ncclAllGather" in rccl["config"]["exchange"]
    # how many ranks RCCL itself saw is a top-level field
    assert rccl["rccl"]["nranks"] == 1 and rccl["rccl"]["ranks_counted_by_all_reduce"] == 1
    assert rccl["rccl"]["version"] > 0 and "PCI" in rccl["rccl"]["rank0"] and "error" not in rccl
    # which GPU every rank ended up on rides in the `rccl` object
    pl = rccl["rccl"]["placement"]
    assert len(pl) == 1 and pl[0]["rank"] == 0 and pl[0]["device"] == 0 and pl[0]["visible_devices"] >= 1
    assert ":" in pl[0]["pci"] and rccl["rccl"]["distinct_pci_ids"] == 1
    assert "rccl" not in plain
    # self-launching form: no launcher, no WORLD_SIZE in the environment
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--swarm-per-gpu", "256"] + common,
               {"NMRFIT_BENCH_BACKEND": "host", "NMRFIT_BENCH_REPLICA_JOBS": "24"})
    assert two["n_gpus"] == 2 and two["config"]["swarm_total"] == 512
    _check_replicas(two, 2, 24)
    assert two["rccl"] is None and "error" not in two          # the host-staged exchange was asked for explicitly
    assert two["config"]["swarm_best_f"] == plain["config"]["swarm_best_f"]
    assert two["config"]["generations_done"] == plain["config"]["generations_done"] == 7
    # the driver's form: torch.distributed.run as the launcher (torch is not imported by the ranks)
    three = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "3",
                  "--swarm-per-gpu", "256"] + common, {"NMRFIT_BENCH_BACKEND": "host", "NMRFIT_BENCH_REPLICA_JOBS": "12"})
    assert three["n_gpus"] == 3 and three["config"]["swarm_total"] == 768 and "error" not in three
    _check_replicas(three, 3, 12)


def _check_replicas(line, world, jobs):
    """the N > 1 line carries BOTH multi-GPU modes -- the swarm-sharded `value` (with `scaling_model`)
    and the spectra-parallel replicas, every rank its own spectra through fit_many / fit_many(shard=True)."""
    rp = line["replicas"]
    assert "error" not in rp and rp["ranks"] == world and rp["jobs_per_rank"] == jobs
    for key in ("stopping_rule_on", "stopping_rule_off"):
        q = rp[key]
        assert len(q["per_rank_fits_per_s"]) == world and min(q["per_rank_fits_per_s"]) > 0
        assert q["aggregate_fits_per_s"] == pytest.approx(world * jobs / (jobs / q["per_rank_spread"]["min"]))
        assert q["expected_aggregate_fits_per_s"] == pytest.approx(sum(q["per_rank_fits_per_s"]))
        assert 0.0 < q["aggregate_over_expected"] <= 1.0 + 1e-9
        sh = q["sharded_call"]
        assert sh["results_gathered_on_every_rank"] and sh["own_share_equals_local_run"] and sh["aggregate_fits_per_s"] > 0
    sm = line["scaling_model"]["replicas"]
    assert sm["measured_aggregate_fits_per_s"] == rp["stopping_rule_on"]["aggregate_fits_per_s"]


def test_bench_says_so_when_rccl_is_unavailable():
    """RCCL missing on every rank (forced with NMRFIT_RCCL_LIB): no rank hangs in the rendezvous or in
    ncclCommInitRank (the ranks compare notes first), the run still measures with the host-staged
    exchange -- but N > 1 without RCCL is NOT a scaling number: the line carries an "error" field,
    `rccl` is null and the exit code is 5, under the self-launcher and under torch.distributed.run."""
    common = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--workload", "C2", "--preheat-seconds", "0.1",
              "--no-extras", "--swarm-per-gpu", "256"]
    bad = {"NMRFIT_RCCL_LIB": "/nonexistent/librccl.so", "NMRFIT_BENCH_SHARE_GPU": "1"}
    d = _run([sys.executable, "bench.py", "--gpus", "2"] + common, bad, expect_rc=5,
             stderr_has=("RCCL exchange unavailable", "creating the RCCL communicator on HIP device"))
    assert d["n_gpus"] == 2 and "RCCL FAILED" in d["config"]["exchange"]
    assert d["rccl"] is None and "did not run over RCCL" in d["error"] and d["value"] > 0
    t = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + common,
             bad, expect_rc=1)
    assert t["rccl"] is None and "did not run over RCCL" in t["error"]
    # only ONE rank without RCCL: every rank hears about it before anyone enters the collective
    # (a second rank on the same device would make RCCL itself refuse; the point is that nobody hangs)
    one_bad = {"NMRFIT_BENCH_SHARE_GPU": "1", "NMRFIT_BENCH_TEST_RCCL_MISSING_ON": "1"}
    d1 = _run([sys.executable, "bench.py", "--gpus", "2", "--launch-timeout", "120"] + common, one_bad, expect_rc=5,
              stderr_has=("RCCL is not available on rank(s) [1]",))
    assert d1["rccl"] is None and "error" in d1


def test_c4_rehearsal_four_ranks_on_one_gpu():
    """C4's per-GPU shape (4096 particles x 65536 points x 24 peaks per rank) with 4 ranks on one
    device through the self-launcher, against the one-rank 16384-particle run: same swarm best."""
    common = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-extras", "--preheat-seconds", "0.1"]
    four = _run([sys.executable, "bench.py", "--gpus", "4", "--swarm-per-gpu", "4096"] + common,
                {"NMRFIT_BENCH_BACKEND": "host"})
    one = _run([sys.executable, "bench.py", "--swarm-per-gpu", "16384"] + common, {})
    assert four["n_gpus"] == 4 and four["config"]["swarm_total"] == 16384 == one["config"]["swarm_total"]
    assert four["config"]["swarm_best_f"] == one["config"]["swarm_best_f"]
    assert four["config"]["generations_done"] == one["config"]["generations_done"] == 4
    # BASELINE configs 2 and 5 ride in the default single-GPU line
    oc = one["other_configs"]
    assert oc["C1"]["shape"] == {"rows": 50, "grid": 4096, "peaks": 6} and oc["C1"]["kernel_ms"] > 0
    assert oc["C2"]["shape"] == {"rows": 1024, "grid": 4096, "peaks": 6} and oc["C2"]["kernel_ms"] > 0
    assert oc["C5"]["shape"] == {"rows": 41, "grid": 16384, "peaks": 12} and "residual" in oc["C5"]["kind"]
    for k in ("C1", "C2", "C5"):
        # these launches are latency-bound; no machine-readable fraction above 1, the streaming-operand
        # rate (an effective rate served from L2) stands beside it without one
        r = oc[k]["roofline"]
        assert r["bound"] == "latency" and 0.0 < r["frac"] < 1.0 and r["frac_kind"] == "estimate"
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
        assert r["streaming_operand_GBps_effective"] == pytest.approx(r["bytes_per_launch"] / (oc[k]["kernel_ms"] * 1e-3) / 1e9)
    assert oc["C2"]["roofline"]["bytes_per_launch"] == 1024 * (4 * 4096 * 8) + 1024 * 22 * 8 + 1024 * 8
    # ... and so does the reference's default fit, end to end (204 particles, all 2000 generations)
    rf = one["reference_default_fit"]
    assert rf["shape"] == {"swarm": 204, "grid": 4096, "peaks": 6, "generations": 2000}
    assert 5.0 < rf["wall_ms"] < 500.0 and 0.0 < rf["error"] < 0.05
    # ... and the same default fits device-batched (round 5): 40 of them in one batch, one launch per generation
    bf = one["reference_default_fit_batched"]
    assert bf["shape"]["fits"] == 40 and bf["stopping_rule_off"]["generations"] == {"min": 2000, "max": 2000, "mean": 2000.0}
    assert bf["stopping_rule_off"]["fits_per_s"] > 2.0 * 1e3 / rf["wall_ms"]      # at least twice a loop of lone fits
    assert bf["stopping_rule_on"]["generations"]["max"] <= 2000 and bf["stopping_rule_on"]["fits_per_s"] > bf["stopping_rule_off"]["fits_per_s"]
    assert bf["stopping_rule_off"]["geometry"]["mode"] == "wave"
    rg = bf["ragged_lengths"]            # round 6: spectra of different lengths in one batch
    assert 3000 <= rg["lengths"]["min"] < rg["lengths"]["max"] <= 6000 and rg["ragged_over_equal_fits_per_s"] > 0.8
    e2e = bf["fit_many_end_to_end"]      # the user-level call on the same spectra (its own weights: FitUtility._compute_weights)
    assert 0.0 < e2e["stopping_rule_off"]["error_fit0"] < 0.05
    assert e2e["stopping_rule_on"]["fits_per_s"] > e2e["stopping_rule_off"]["fits_per_s"] > 2.0 * 1e3 / rf["wall_ms"]
    # ... and the rest of the README script (round 6): fit -> generate_result -> area fractions, batched against the loop
    rp = one["readme_pipeline"]
    assert rp["shape"]["jobs"] == 200 and "error" not in rp
    for key in ("stopping_rule_off", "stopping_rule_on"):
        assert rp[key]["batched_equals_plain_loop_bit_for_bit"] is True
        assert rp[key]["pipeline_fits_per_s"] > 2.0 * rp[key]["plain_loop_fits_per_s"]
        assert rp[key]["result_bytes_per_fit"] == (2 * 6 + 6) * 4096 * 8
    assert rp["stopping_rule_off"]["pipeline_over_fit_only"] > 0.8
    assert "replicas" not in one and "replicas" not in four          # (--no-extras, N = 1)
    # the sharded run carries the model its step time is to be judged against
    sm = four["scaling_model"]
    assert len(sm["expected_ms_per_step"]) == 2 and sm["max_rank_kernel_ms"] > 0 and sm["measured_over_expected"] is not None


def test_bench_line_carries_live_pmc_counters():
    """the PMC-derived fields of the default single-GPU line (physical HBM traffic,
    VALU instructions per unit, VALU busy) are measured by THIS run -- three short child passes of the
    same script under `rocprofv3 --kernel-trace --pmc`, started before the parent touches the GPU --
    not echoed from a committed file; so is the far-field kernel's entry in `fit_default`."""
    d = _run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--cpu-seconds", "0", "--no-other-configs",
              "--preheat-seconds", "0.2"], {})
    r, fd = d["roofline"], d["fit_default"]
    v = r
    if "pmc_live_errors" in d:
        # a profiler hiccup on this box must not cost the run its line: the fields then fall back to the
        # committed passes and SAY so -- check that, and report the reason instead of failing the suite
        assert r["traffic_from_committed_profile"] is True or r["traffic"] is None or v.get("from_committed_profile")
        pytest.skip("rocprofv3 child passes failed here: %s" % d["pmc_live_errors"])
    assert r["traffic_from_committed_profile"] is False and r["traffic_source"].startswith("live")
    # the four grid arrays are 2 MiB and L2-resident: physical traffic is tens of MB per launch against
    # 8.6 GB of streamed operands in the byte model
    assert 2e6 < r["traffic"] < 3e8 and r["l2_hit_rate"] > 0.98
    assert v["from_committed_profile"] is False and v["pmc_source"].startswith("live")
    assert 5.0 < v["valu_instructions_per_unit"] < 7.0 and 0.8 < v["valu_busy_frac"] <= 1.0
    assert 3.9 < v["valu_cycles_per_instruction"] < 4.6
    # the line's roofline is the issued-VALU fraction, reproducible from the counters it carries
    assert r["bound"] == "fp64_valu_issue" and r["peak"] == pytest.approx(1024 * 2.4e9 / 4)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.6 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(v["valu_instructions_per_unit"] * r["units_per_launch"] / 64.0 / (r["kernel_ms"] * 1e-3))
    assert d["roofline_streaming"]["rate_kind"] == "effective" and d["roofline_streaming"]["physical_hbm_frac_of_peak"] < 0.05
    # round 5: every unit evaluated (NOSKIP: an A/B kernel, timed and counted by children that load libnmrfit_amd_ab.so)
    eu = d["every_unit"]
    assert eu["kernel_ms"] > 2.0 * d["kernel_ms"]["mean"] and 15.0 < eu["roofline"]["valu_instructions_per_unit"] < 35.0
    assert 0.5 < eu["roofline"]["valu_busy_frac"] <= 1.0
    assert fd["variant"] == "farfield" and fd["from_committed_profile"] is False
    assert 1.5 < fd["valu_instructions_per_unit"] < 4.0 and fd["kernel_ms"] < d["kernel_ms"]["mean"]


def test_real_rccl_two_ranks_on_one_device_is_refused_loudly():
    """As close to a real multi-rank RCCL start as a one-GPU box gets: two ranks, the real librccl, one
    device.  The ranks' unique-id hand-over and RCCL's own bootstrap between the two processes go through;
    RCCL then refuses the duplicate device with an error from ncclCommInitRank on BOTH ranks within seconds.
    bench.py must not hang and must not pass the run off as an RCCL number: line printed, `rccl` null,
    "error" field, exit code 5."""
    import time
    common = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--workload", "C2", "--preheat-seconds", "0.1",
              "--no-extras", "--swarm-per-gpu", "256", "--launch-timeout", "60"]
    t0 = time.time()
    d = _run([sys.executable, "bench.py", "--gpus", "2"] + common, {"NMRFIT_BENCH_SHARE_GPU": "1"}, expect_rc=5,
             stderr_has=("ncclCommInitRank", "RCCL exchange unavailable"))
    assert time.time() - t0 < 120
    assert d["rccl"] is None and "did not run over RCCL" in d["error"] and "RCCL FAILED" in d["config"]["exchange"]


def test_rank_isolated_by_visible_devices_lands_on_device_zero():
    """a launcher that isolates each rank with HIP_VISIBLE_DEVICES shows every rank ONE
    device, number 0, whatever its LOCAL_RANK -- `device = LOCAL_RANK` would die with "device index out of
    range" on every rank but 0.  Here: LOCAL_RANK=3 with one visible device, through bench.py's RCCL branch
    (a one-rank communicator) and through fit(exchange="rccl")."""
    common = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--workload", "C2", "--preheat-seconds", "0.1",
              "--no-extras", "--swarm-per-gpu", "256"]
    iso = {"HIP_VISIBLE_DEVICES": "0", "RANK": "0", "LOCAL_RANK": "3", "WORLD_SIZE": "1", "LOCAL_WORLD_SIZE": "8",
           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "NMRFIT_BENCH_FORCE_DIST": "1"}
    d = _run([sys.executable, "bench.py"] + common, iso, stderr_has=("this rank is isolated, using device 0",))
    pl = d["rccl"]["placement"][0]
    assert pl["device"] == 0 and pl["local_rank"] == 3 and pl["visible_devices"] == 1
    assert "HIP_VISIBLE_DEVICES=0" in pl["env"] and "error" not in d      # (the box may export ROCR_/CUDA_ forms too)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import nmrfit_amd\nfrom nmrfit_amd import synth\n"
            "sp = synth.make_spectrum(4096, 6, seed=21)\n"
            "data = synth.SynthData(sp['w'], sp['u'], sp['v'], sp['peaks'])\n"
            "res = nmrfit_amd.fit(data, list(sp['lower']), list(sp['upper']), summary=False,\n"
            "                     options={'swarmsize': 64, 'maxiter': 5, 'exchange': 'rccl', 'seed': 3})\n"
            "assert res._device() == 0 and res.error > 0\nres.generate_result()\nprint('ok')\n" % ROOT)
    env = dict(os.environ, **iso)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    assert "this rank is isolated, using device 0" in out.stderr
