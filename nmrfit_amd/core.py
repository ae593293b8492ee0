"""
``fit`` -- the reference's entry point (nmrfit/core.py:64-95), same signature and return
value, executed on the MI355X.  ``load`` (instrument file parsing through nmrglue,
core.py:9-61) is outside the hot-path scope: build a ``Data``-like object (attributes
w, u, v, peaks) with the reference package or ``nmrfit_amd.synth`` and pass it in.
"""
import numpy as np

from . import utils


def fit(data, lower, upper, expon=0.5, dynamic_weighting=True, fit_im=False, processes=1, summary=True,
        options={}):
    """Perform a fit of NMR spectroscopy data (reference: nmrfit/core.py:64).

    data : object with ``w, u, v`` (ndarrays) and ``peaks`` (each with ``bounds``, ``height``)
    lower, upper : parameter box, 4 + 3P floats (nmrfit/containers.py:193-217)
    expon, dynamic_weighting : error weighting (nmrfit/utils.py:191-224)
    fit_im : False (real part only, the default); True = the reference's imaginary term exactly as
             nmrfit/equations.py:197-209 computes it (last peak's line only); "sum" = all peaks.
             The Kramers-Kronig partner is evaluated in closed form on the GPU
    processes : accepted for compatibility; the batched GPU launch replaces the process pool
    summary : print the fit summary table
    options : swarmsize, maxiter, omega, phip, phig (+ minstep, minfunc, seed, device,
              check_every, polish, variant, exchange="rccl" for one process per GPU)

    Returns the FitUtility holding ``params``, ``error``, ``weights``.
    """
    f = utils.FitUtility(data, lower, upper, expon, dynamic_weighting, fit_im, processes, summary, options)
    f.fit()
    return f


def fit_many(jobs, threads=4, batch=True, shard=False, devices=None, generate=False, channel=None, **kwargs):
    """Fit several spectra: ``jobs`` is a sequence of ``(data, lower, upper)`` triples (or dicts of ``fit``'s
    arguments); every job is fitted as ``fit`` would fit it with the same keyword arguments, and the list of
    FitUtility objects comes back in the order of ``jobs``.  Not in the reference (its users loop over
    ``nmrfit.fit``, nmrfit/core.py:64; its only parallel mode spreads ONE fit's particles over processes,
    nmrfit/utils.py:182).  ``summary`` defaults to False here.

    How the jobs run:

    * ``batch=True`` (default): jobs of equal kernel variant, ``fit_im``, ``maxiter`` and ``check_every`` -- grid lengths,
      peak counts and swarm sizes may differ -- are fitted as ONE device batch -- one kernel launch per swarm generation for all of them (nmrfit_amd.batch.FitBatch,
      csrc/batch.hip).  A 204-particle swarm fills a fraction of an MI355X; a batch fills it.  Each fit's ``params`` and
      ``error`` are bit-identical to what ``fit`` returns for it alone with the same ``options['seed']``.  Job lists go
      through batches of a quarter of the list, between 40 and 200 jobs (``nmrfit_amd.core.BATCH_JOBS`` overrides),
      three stages in flight: a second host thread prepares the next batch (error weights, plans, device state) while
      the device runs two batches side by side (the tail of one -- its last swarms to stop -- overlaps with the other)
      and another thread reads back the one before.
    * ``generate=True`` (or a number: the ``scale`` of ``FitUtility.generate_result``): the rest of the reference's
      per-spectrum script, README.md:64-72 -- every returned fit has had ``generate_result(scale)`` called on it
      (nmrfit/utils.py:226-295: ``u, v, V, I, w, real_contribs, imag_contribs``; ``calculate_area_fraction()`` then
      needs nothing more).  For the fits of a device batch that is ONE launch over the batch's resident spectra and best
      positions (nmrfit_batch_contributions, csrc/result.hip) instead of a context, four uploads and a launch per fit;
      the values are bit-identical to the lone call's.  The data objects get ``p0, p1, V, I`` set to what
      ``data.shift_phase(method='manual', p0, p1)`` computes (nmrfit/containers.py:68-78) from the same launch; the
      method itself is called only on the lone path.
    * ``options['polish']``: the swarm runs in the batch, the least-squares refinement that follows it per fit on
      ``threads`` host threads (scipy's trust-region iterations, a context per fit) -- the answers are the lone call's.
    * whatever cannot be batched (a lone shape, more than 132 peaks, a batch the device refuses) runs
      through ``fit`` on ``threads`` host threads, each fit with its own context and HIP stream -- serially when
      ``options['exchange']`` is given: a communicator serves one swarm at a time.
    * ``shard=True`` in a multi-GPU launch (one process per GPU, RANK / WORLD_SIZE / LOCAL_RANK set by the launcher):
      the JOBS are divided over the ranks -- rank r takes jobs r, r + world, ... on its own GPU -- and the results are
      gathered so that every rank returns the full list (``params``, ``error``, ``seed``; the arrays of ``generate`` stay
      on the rank that made them).  Replicas: no collective touches the fits themselves.  This is the multi-GPU mode
      for many small fits (sharding one 204-particle swarm over GPUs is slower than one GPU).  ``channel``: an open
      ``nmrfit_amd.rendezvous.Channel`` to gather over (a caller that already has one); else one is made.
    * ``devices=[0, 1, ...]`` (or ``"all"``): the same replicas WITHOUT a launcher -- this one process drives several
      GPUs, a host thread per device, job k on ``devices[k % len(devices)]``; each device runs its share as device
      batches of its own.  Nothing crosses devices.  (``shard`` and ``devices`` exclude each other.)"""
    kwargs.setdefault("summary", False)
    jobs = [dict(job) if isinstance(job, dict) else dict(zip(("data", "lower", "upper"), job)) for job in jobs]
    if devices is not None:
        if shard:
            raise ValueError("fit_many: shard=True divides the jobs over PROCESSES, devices=[...] over the GPUs of this "
                             "process: use one of them")
        if isinstance(devices, str):
            if devices != "all":
                raise ValueError("fit_many: devices must be a list of device indices or \"all\"")
            devices = list(range(_cabi_device_count()))
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("fit_many: no devices")
        return _fit_many_devices(jobs, threads, batch, kwargs, devices, generate)
    if shard:
        from . import rendezvous
        rank, _, world = rendezvous.env_rank_world()
        if world > 1:
            return _fit_many_sharded(jobs, threads, batch, kwargs, rank, world, channel=channel, generate=generate)
    return _fit_many_local(jobs, threads, batch, kwargs, generate)


def _result_record(f):
    """What travels between ranks for one finished fit."""
    return dict(params=list(map(float, f.params)), error=float(f.error), seed=getattr(f, "seed", None))


def _with_device(job, shared_options, device, force=False):
    """The job with ``device`` in ITS OWN options: a job's ``options`` dict replaces the shared one when the two are
    merged (dict(kwargs, **job)), so a device that only sits in the shared options is lost for every job that brings a
    seed or a maxiter.  ``force``: the caller assigns the device (devices=[...]); otherwise a device the job or the
    shared options name wins."""
    opts = dict(shared_options)
    opts.update(job.get("options") or {})
    if force or opts.get("device") is None:
        opts["device"] = device
    return dict(job, options=opts)


def _fit_many_sharded(jobs, threads, batch, kwargs, rank, world, channel=None, local=None, generate=False):
    """Jobs r, r + world, ... on this rank's GPU; every rank returns every result (the other ranks' as FitUtility
    objects holding ``params`` / ``error`` / ``seed``; their ``weights`` are recomputed on demand only by ``fit``)."""
    import json
    from . import rendezvous
    mine = list(range(rank, len(jobs), world))
    if local is None:
        device, note = rendezvous.pick_device(_cabi_device_count())
        if note:
            import sys
            sys.stderr.write("nmrfit: %s\n" % note)

        def local(my_jobs):
            return _fit_many_local(my_jobs, threads, batch, kwargs, generate)
    else:
        device = None
    # the ranks meet BEFORE they fit: a mis-launched world shows at once, and the channel's connect deadline does not
    # have to cover the slowest rank's share of the work
    own = channel is None
    if own:
        channel = rendezvous.Channel()
    try:
        shared = kwargs.get("options") or {}
        done = local([_with_device(jobs[i], shared, device) if device is not None else jobs[i] for i in mine])
        # (JSON, not pickle: what arrives from another rank is data, never code; repr round-trips a float64 exactly)
        parts = channel.all_gather(json.dumps([[i, _result_record(f)] for i, f in zip(mine, done)]).encode())
    finally:
        if own:
            channel.close()
    out = [None] * len(jobs)
    for i, f in zip(mine, done):
        out[i] = f
    for r, blob in enumerate(parts):
        if r == rank:
            continue
        for i, rec in json.loads(blob.decode()):
            job = jobs[i]
            args = {k: v for k, v in dict(kwargs, **job).items() if k not in ("data", "lower", "upper")}
            f = utils.FitUtility(job["data"], job["lower"], job["upper"], **args)
            f.params, f.error, f.seed = np.array(rec["params"]), rec["error"], rec["seed"]
            out[i] = f
    return out


def _fit_many_devices(jobs, threads, batch, kwargs, devices, generate=False):
    """Job k on devices[k % len(devices)], one host thread per device (the library releases the GIL inside its calls;
    every call binds its own device), results back in job order."""
    from concurrent.futures import ThreadPoolExecutor
    shares = [list(range(i, len(jobs), len(devices))) for i in range(len(devices))]
    shared = kwargs.get("options") or {}

    def one(i):
        mine = [_with_device(jobs[k], shared, devices[i], force=True) for k in shares[i]]
        return _fit_many_local(mine, threads, batch, kwargs, generate)
    out = [None] * len(jobs)
    with ThreadPoolExecutor(max_workers=len(devices)) as pool:
        for idx, res in zip(shares, pool.map(one, range(len(devices)))):
            for k, f in zip(idx, res):
                out[k] = f
    return out


def _cabi_device_count():
    from . import _cabi
    return _cabi.device_count()


# Jobs per device batch in fit_many.  From ~40 default-size fits on a batch holds the MI355X's issue rate (DESIGN.md
# 4.5), and a long list wants large batches (1000 default jobs with pyswarm's rule: 2050 fits/s in batches of 64, 2270 in
# batches of 200) while a short one wants at least four of them, so that the three stages overlap -- the host prepares
# batch c + 1 (error weights, plans, device state) on a second thread while the device runs batch c and a third thread
# reads back batch c - 1 (200 jobs: 1970 fits/s in batches of 50, 1740 as one batch; profiles/r06/fit_many_span_sweep.txt).
# BATCH_JOBS = None: that rule (a quarter of the list, between 40 and 200); a number: batches of about that many.
BATCH_JOBS = None
BATCH_JOBS_MIN, BATCH_JOBS_MAX, PIPELINE_BATCHES = 40, 200, 4
# Device batches driven at the same time.  With pyswarm's rule the swarms of a batch stop at different generations (112
# ... 411 for default fits) and the batch's last generations hold a few swarms each, at a launch's latency; a second
# batch running beside it fills the device meanwhile: 4 batches of 50 default fits 93.7 -> 82.4 ms (2134 -> 2427 fits/s
# on the device), three at a time 79.8 (tools/concurrent_batches.py, profiles/r06/concurrent_batches.txt).
RUN_AT_ONCE = 2
# End to end (tools/pipeline_ab.py, profiles/r06/fit_many_pipeline_ab.txt; one box, best of three): 200 default jobs with
# pyswarm's rule 1924 -> 2038 fits/s, 1000 jobs 2349 -> 2405 -- less than on the device alone, the rest of the call being
# the first batch's preparation and the last one's read-back; with the reconstruction 1707 -> 1728 / 2146 -> 2120 (noise).
# (measured and not kept: the first batch of a list cut in two halves to start the device sooner -- no difference.)
READ_ON_RUNNER = False    # A/B knob: the read-back on the thread that ran the batch instead of the storing thread (same rates)


def _batch_jobs(n):
    if BATCH_JOBS is not None:
        return max(1, int(BATCH_JOBS))
    return min(BATCH_JOBS_MAX, max(BATCH_JOBS_MIN, -(-n // PIPELINE_BATCHES)))


def _fit_many_local(jobs, threads, batch, kwargs, generate=False):
    from concurrent.futures import ThreadPoolExecutor
    from ._cabi import NmrfitError
    scale = 1 if generate is True else generate      # (False: no reconstruction)
    fits = []
    for job in jobs:
        args = dict(kwargs, **job)
        fits.append(utils.FitUtility(args.pop("data"), args.pop("lower"), args.pop("upper"), **args))
    plans = {}
    alone = list(range(len(fits)))
    if batch and len(fits) > 1:
        n = len(fits)
        nspans = -(-n // _batch_jobs(n))
        size = -(-n // nspans)
        spans = [range(a, min(a + size, n)) for a in range(0, n, size)]

        def group(members):
            """Device batches of the members that have a partner, ready to run; the others as (index, key)."""
            groups = {}
            for i, key in members:
                groups.setdefault(key, []).append(i)
            ready, single = [], []
            for key, idx in groups.items():
                if key is not None and len(idx) > 1:
                    try:
                        ready.append(_batch_create([fits[i] for i in idx], [plans[i] for i in idx], key) + (idx,))
                        continue
                    except NmrfitError:
                        # the device refused this batch (LDS budget of its peak counts, memory): its fits run one by one
                        key = None
                single.extend((i, key) for i in idx)
            return ready, single

        def prepare(span):
            for i in span:
                plans[i] = fits[i]._plan()
            return group([(i, fits[i]._batch_key(plans[i])) for i in span])

        leftover = []
        batched = set()
        made = []          # every batch created and not yet closed (closed by _batch_collect, or below on an error)
        import collections
        with ThreadPoolExecutor(max_workers=1) as host, ThreadPoolExecutor(max_workers=1) as post, \
                ThreadPoolExecutor(max_workers=RUN_AT_ONCE) as runner:
            pending = host.submit(prepare, spans[0])
            posted = []
            inflight = collections.deque()      # (future of a batch's run, the batch), in job order
            try:
                def drain(limit):
                    """Wait for the oldest runs until at most ``limit`` are in flight; what each read back goes to the
                    thread that stores results, in job order (pyswarm's closing lines print in job order)."""
                    while len(inflight) > limit:
                        fut, (fb, bfits, bplans, key, _) = inflight.popleft()
                        got = fut.result()
                        if got is None:
                            posted.append(post.submit(_batch_collect, fb, bfits, bplans, key, scale, threads))
                        else:
                            posted.append(post.submit(_batch_store, bfits, bplans, key, got[0], got[1], got[2], scale, threads))

                def run_and_read(fb, bfits, key):
                    fb.run(key[4], key[5])                      # (maxiter, check_every)
                    if not READ_ON_RUNNER:
                        return None
                    return _batch_read(fb, bfits, scale)        # status, best rows, reconstruction; closes the batch

                def run(ready, last=False):
                    made.extend(r[0] for r in ready)
                    for n, b in enumerate(ready):
                        fb, bfits, bplans, key, idx = b
                        batched.update(idx)
                        if last and n == len(ready) - 1 and not posted and not inflight:
                            # the only batch of the call: nothing to overlap it with -- run and read back here, without
                            # a thread's first HIP call in the way (a 40-job list is 30 ms in all)
                            fb.run(key[4], key[5])
                            _batch_collect(fb, bfits, bplans, key, scale, threads)
                            continue
                        # RUN_AT_ONCE batches are driven at the same time (threads of their own: the C calls release
                        # the GIL): while the last swarms of one batch finish -- a generation of three swarms costs a
                        # launch of 8-20 us whatever it holds -- or its results travel to the host, the other batch
                        # fills the device
                        drain(RUN_AT_ONCE - 1)
                        inflight.append((runner.submit(run_and_read, fb, bfits, key), b))
                for c in range(len(spans)):
                    ready, single = pending.result()
                    pending = host.submit(prepare, spans[c + 1]) if c + 1 < len(spans) else None
                    leftover.extend(single)
                    run(ready, last=(c + 1 == len(spans) and not any(key is not None for _, key in leftover)))
                # what found no partner inside its span may have one in another
                ready, _ = group([(i, key) for i, key in leftover if key is not None])
                run(ready, last=True)
                drain(0)
                for p in posted:
                    p.result()
            except BaseException:
                if pending is not None:      # (batches made for a span that will not run)
                    try:
                        made.extend(r[0] for r in pending.result()[0])
                    except Exception:
                        pass
                for fut, _ in inflight:      # (runs in flight: let them end before their batches are closed)
                    try:
                        fut.result()
                    except Exception:
                        pass
                for p in posted:
                    try:
                        p.result()
                    except Exception:
                        pass
                for fb in made:
                    fb.close()
                raise
        alone = [i for i in range(len(fits)) if i not in batched]
    if not alone:
        return fits

    def lone(i):
        fits[i].fit(plan=plans.get(i))
        if scale is not False:
            fits[i].generate_result(scale)
    with_exchange = any(fits[i].options.get("exchange") is not None for i in alone)
    if threads <= 1 or len(alone) <= 1 or with_exchange:
        for i in alone:
            lone(i)
        return fits
    with ThreadPoolExecutor(max_workers=int(threads)) as pool:
        list(pool.map(lone, alone))
    return fits


def _batch_create(fits, plans, key):
    """The device state of one batch (spectra, weights, boxes, swarms) -- everything up to the first launch."""
    from .batch import FitBatch
    device, _, _, variant, _, _, fit_im = key
    swarmsize = [int(p['swarmsize']) for p in plans]
    spectra = [(f.data.w, f.data.u, f.data.v, f.weights) for f in fits]
    kw = {name: [p['kw'][name] for p in plans] for name in ("omega", "phip", "phig", "minstep", "minfunc")}
    fb = FitBatch(spectra, [f.lower for f in fits], [f.upper for f in fits], swarmsize=swarmsize,
                  seeds=[p['seed'] for p in plans], variant=variant, fit_im=fit_im, device=device, **kw)
    return fb, fits, plans, key


def _fit_batch(fits, plans, key, generate=False):
    """One device batch: FitBatch over the fits' spectra, run to the common maxiter, results into the FitUtility objects
    (what FitUtility.fit does for one, utils.py:164-189; with ``generate`` also what generate_result does)."""
    fb, fits, plans, key = _batch_create(fits, plans, key)
    try:
        fb.run(key[4], key[5])
    except BaseException:
        fb.close()
        raise
    _batch_collect(fb, fits, plans, key, 1 if generate is True else generate)


def _batch_read(fb, fits, scale=False):
    """What is read from the device after a batch's generations: stop codes, best positions and -- ``scale`` not False --
    the reconstruction of every fit in one launch (FitBatch.generate).  Closes the batch."""
    polished = sum(1 for f in fits if f.options.get('polish', False))
    with fb:
        status = fb.status()
        best = fb.best()
        results = fb.generate(scale) if scale is not False and polished < len(fits) else None
    return status, best, results


def _batch_collect(fb, fits, plans, key, scale=False, threads=1):
    """Read back (``_batch_read``) and store (``_batch_store``) in one go."""
    status, best, results = _batch_read(fb, fits, scale)
    _batch_store(fits, plans, key, status, best, results, scale, threads)


def _batch_store(fits, plans, key, status, best, results, scale=False, threads=1):
    """A batch's results into the FitUtility objects (what FitUtility.fit and generate_result leave behind).  Fits with
    options['polish'] are refined first (FitUtility._polish, ``threads`` at a time) and reconstructed from the refined
    parameters."""
    from .pso import STOP_MESSAGES
    maxiter = key[4]
    polished = [k for k, f in enumerate(fits) if f.options.get('polish', False)]
    if polished:
        def refine(k):
            return fits[k]._polish(best[k][0], best[k][1], plans[k])
        if threads > 1 and len(polished) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=int(threads)) as pool:
                refined = list(pool.map(refine, polished))
        else:
            refined = [refine(k) for k in polished]
        for k, xf in zip(polished, refined):
            best[k] = xf
    for k, (f, p, st, (x, fx)) in enumerate(zip(fits, plans, status, best)):
        # (pyswarm's closing line, once per fit like the plain loop prints it)
        if st["stop"]:
            print(STOP_MESSAGES[st["stop"]].format(minfunc=p['kw']['minfunc'], minstep=p['kw']['minstep']))
        else:
            print('Stopping search: maximum iterations reached --> {:}'.format(maxiter))
        f._finish(x, fx)
        if k in polished:
            if scale is not False:
                f.generate_result(scale)      # (from the refined parameters: the batch's launch used the swarm's)
        elif results is not None:
            r = results[k]
            f._store_result(f.data.w if r["w"] is None else r["w"], r["real"], r["imag"],
                            (r["V"], r["I"], r["u"], r["v"]), (r["data_V"], r["data_I"]), call_shift_phase=False)
