"""
Host-side mirror of the reference's fit driver (nmrfit/utils.py:96-339 ``FitUtility``):
same constructor, same attributes, same ``options`` keys -- the swarm loop and the objective
run on the MI355X instead of going through pyswarm one particle at a time.

Kept verbatim from the reference interface:
  FitUtility(data, lower, upper, expon=0.5, dynamic_weighting=True, fit_im=False,
             processes=1, summary=True, options={})
  attributes data, lower, upper, expon, dynamic_weighting, fit_im, summary, processes, options;
  after fit(): weights, params, error.
  options: swarmsize (204), maxiter (2000), omega (-0.2134), phip (-0.3344), phig (2.3259)
           (utils.py:177-181).  Extra opt-in keys: minstep, minfunc (pyswarm's 1e-8 defaults,
           which the reference does not forward), seed, device, check_every, polish,
           variant (kernel variant by name or number; default: "farfield" when grid x peaks
           >= 1e5, else "default" -- see default_variant; "farfield32" is the opt-in mixed-precision
           form of the far-field kernel, f within 5e-12 of it, 5 % faster at C3 size), exchange ("rccl" for a
           multi-GPU fit, one process per GPU: the swarm axis is sharded and the global best is
           exchanged by one RCCL all-gather per generation inside libnmrfit_amd.so).

``processes`` is accepted and ignored: the reference's only use of it is to spread the
per-particle objective calls (and the Kramers-Kronig quadratures of generate_result) over a
multiprocessing.Pool (utils.py:182, 259-261), which the batched launches replace.
``fit_im=True`` follows the reference to the letter (equations.py:197-209: the imaginary
model is the last peak's line only); ``fit_im="sum"`` fits the imaginary part of all peaks.
"""
import numpy as np

from . import _cabi, equations, proc_autophase, pso
from .peaks import (AutoPeakSelector, BoundsSelector, Peak, Peaks, find_peak, rnd_data,  # noqa: F401  (reference names:
                    sample_noise)                                                          # nmrfit.utils.<name>)


def compute_weights(w, peaks, expon=0.5):
    """The ``weights`` array of a fit: what FitUtility._compute_weights produces (nmrfit/utils.py:191-224),
    computed once per fit on the host as the reference does.

    Every peak claims the grid points between the two points nearest to its ``bounds``, at weight
    ``(tallest |height| / its |height|) ** expon``; where regions overlap the LAST peak in ``peaks`` wins (the
    reference assigns the slices in list order); unclaimed points weigh 1.  Ten Laplacian sweeps
    (``equations.laplace1d``) then round the steps off.  Here the claim is resolved for all peaks at once on a
    (peak, point) membership mask -- which peak owns a point is the highest peak index whose region holds it --
    and the weights are gathered through that owner index, instead of one slice assignment per peak."""
    grid = np.asarray(w, dtype=float)
    if len(peaks) == 0:
        return equations.laplace1d(np.ones(grid.size))
    edges = np.array([[pk.bounds[0], pk.bounds[1]] for pk in peaks], dtype=float)           # (P, 2)
    heights = np.abs(np.array([pk.height for pk in peaks], dtype=float))
    # nearest grid point to each region edge (first one on ties, like argmin), edges sorted per region
    nearest = np.abs(grid[None, None, :] - edges[:, :, None]).argmin(axis=2)
    first, last = nearest.min(axis=1), nearest.max(axis=1)
    # per-peak level: scalar power, one value per peak (numpy's array power may take a vectorised libm whose last
    # bit differs from the scalar one the reference's per-peak call goes through)
    top = heights.max()
    level = np.array([np.power(top / h, expon) for h in heights])
    j = np.arange(grid.size)
    member = (j[None, :] >= first[:, None]) & (j[None, :] <= last[:, None])                # (P, N)
    owner = len(peaks) - 1 - member[::-1].argmax(axis=0)                                   # highest claiming peak index
    weights = np.where(member.any(axis=0), level[owner], 1.0)
    return equations.laplace1d(weights)


def default_variant(N, P, fit_im=False):
    """Kernel variant ``fit`` uses when options['variant'] is absent: the far-field form
    (distant peaks' Lorentzian tails through one shared expansion per 512-point chunk, values
    within 1e-14 of the direct kernel's) once there is enough grid x peaks for it to pay, the
    direct kernel below that.  The threshold was re-measured in round 4 with the final kernels at 204, 1024
    and 4096 particles (tools/archive/variant_threshold.py, per-generation time of the swarm loop, far-field /
    direct; profiles/r04/variant_threshold.txt):

        grid x peaks      204      1024     4096 particles
        4096 x 6          1.06     1.11     1.25
        4096 x 24         1.09     1.11     1.14      (short grid: few chunks, every peak near)
        8192 x 12         1.02     1.01     1.05
        16384 x 6         1.01     1.05     1.08
        16384 x 12        0.93     0.89     0.88
        8192 x 24         0.99     0.96     0.96
        32768 x 12        0.84     0.77     0.74
        65536 x 24        0.57     0.50     0.47

    i.e. the direct kernel wins everywhere below grid x peaks = 1e5 and the far-field kernel everywhere from 2e5
    on, for every swarm size: the rule stands.  The whole GPU test suite passes with either as the default of every
    context (NMRFIT_DEFAULT_VARIANT), and tests/test_gpu_parity.py::test_farfield_adversarial_spectra
    covers the spectra where no peak is far.  bench.py's headline is always measured on the direct
    kernel; its `fit_default` entry reports this one.

    ``fit_im="sum"`` (every peak's imaginary line): the same rule.  Up to round 5 the far-field kernel needed 190 VGPRs with
    the imaginary sums (two waves per SIMD: 3.0 ms at 4096 x 65536 x 24 against the direct kernel's 2.6) and was never
    selected; round 6 gave it the pair expansions with the imaginary model -- for the real and for the imaginary series --
    and a quarter-interval Dawson table (162 VGPRs, three waves, no scratch): 1.63 ms against 2.38 there, and per swarm
    generation far-field / direct (tools/variant_threshold_im.py, profiles/r06/variant_threshold_im.txt) 1.00-1.03 below
    grid x peaks = 1e5, 0.95-1.00 at 1e5, 0.94-0.99 at 2e5, 0.85-0.94 at 3.9e5, 0.70-0.74 at 65536 x 24."""
    return "farfield" if int(N) * int(P) >= 100000 else "default"


def generate_solution_bounds(peaks, p0=0.0, p1=0.0, force_p0=False, force_p1=False):
    """The parameter box of a fit, as Data.generate_solution_bounds lays it out (nmrfit/containers.py:175-217),
    in the parameter order of the objective (section a4: p0, p1, r, yoff, then width, loc, area per peak):

        p0, p1   [-pi, pi], or the estimate +- 0.001 when ``force_p0`` / ``force_p1`` is True
        r        [0, 1]            yoff  [-0.01, 0.01]
        width    [0.5, 1.5] x the peak's width
        loc      a tenth of the way from the peak's centre to either region bound
        area     [0.5, 1.5] x the peak's area

    Built as two (4 + 3P) arrays -- the per-peak block is one (P, 3) table per side -- and returned as the plain
    lists ``(lower, upper)`` the reference returns."""
    def phase_box(estimate, forced):
        return (estimate - 0.001, estimate + 0.001) if forced is True else (-np.pi, np.pi)
    head = np.array([phase_box(p0, force_p0), phase_box(p1, force_p1), (0.0, 1.0), (-0.01, 0.01)])   # (4, 2)
    if len(peaks) == 0:
        return head[:, 0].tolist(), head[:, 1].tolist()
    width = np.array([pk.width for pk in peaks], dtype=float)
    loc = np.array([pk.loc for pk in peaks], dtype=float)
    area = np.array([pk.area for pk in peaks], dtype=float)
    edge = np.array([[pk.bounds[0], pk.bounds[1]] for pk in peaks], dtype=float)           # (P, 2)
    sides = []
    for side, scale in ((0, 0.5), (1, 1.5)):
        block = np.stack([width * scale, loc - 0.1 * (loc - edge[:, side]), area * scale], axis=1)   # (P, 3)
        sides.append(np.concatenate([head[:, side], block.ravel()]).tolist())
    return sides[0], sides[1]


# A swarm generation costs at least one objective launch plus, sharded, one all-gather and one fold launch (~15-25 us of
# latency on xGMI whatever the payload): below this much work per rank and generation the exchange costs more than
# the shard saves, and the fit is FASTER on one GPU (the reference's default 204 x 4096 x 6 is 5e6 units, 11.7 us).
SMALL_SHARD_UNITS = 1.0e8


def small_shard_warning(S_local, N, P, world, rank=0):
    """Warn (rank 0, once per call) when options['exchange'] shards a swarm whose per-rank generation is shorter than
    the exchange it adds: independent fits scale across GPUs as REPLICAS -- ``nmrfit_amd.fit_many(jobs, shard=True)``
    gives each rank its own spectra and needs no collective (nmrfit/utils.py:182 is the reference's counterpart: a
    process pool over particles).  Returns True when the warning applies."""
    units = float(S_local) * float(N) * float(max(P, 1))
    if world <= 1 or units >= SMALL_SHARD_UNITS:
        return False
    if rank == 0:
        import warnings
        warnings.warn("nmrfit: sharding this swarm over %d GPUs leaves %.2g particle x point x peak units per rank and "
                      "generation (< %.0e): the per-generation exchange costs more than the shard saves, one GPU is "
                      "faster.  For many spectra use nmrfit_amd.fit_many(jobs, shard=True): the ranks fit different "
                      "spectra and exchange nothing." % (world, units, SMALL_SHARD_UNITS), RuntimeWarning, stacklevel=3)
    return True


class FitUtility:
    """Interface used to perform a fit of the data (reference: nmrfit/utils.py:96)."""

    def __init__(self, data, lower, upper, expon=0.5, dynamic_weighting=True, fit_im=False, processes=1,
                 summary=True, options={}):
        self.data = data
        self.lower = lower
        self.upper = upper
        self.expon = expon
        self.dynamic_weighting = dynamic_weighting
        self.fit_im = fit_im
        self.summary = summary
        self.processes = processes
        self.options = options

    def _compute_weights(self):
        return compute_weights(self.data.w, self.data.peaks, self.expon)

    def _device(self):
        """The GPU this process works on: options['device'] if given; else, with a ready
        pso.RcclExchange, the device its communicator lives on; else, in a multi-rank fit
        (options['exchange'] = "rccl" or another exchange object: one process per GPU), the launcher's
        LOCAL_RANK -- or device 0 when the launcher shows this rank one device only
        (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES isolation; rendezvous.pick_device, which
        raises when the two disagree any other way); else 0.  fit() and generate_result() use the
        same one."""
        device = self.options.get('device')
        exchange = self.options.get('exchange')
        if device is None and isinstance(exchange, pso.RcclExchange):
            device = exchange.ev.device
        if device is None and exchange is not None:
            import sys
            from . import rendezvous
            device, note = rendezvous.pick_device(_cabi.device_count())
            if note and not getattr(self, '_device_note_shown', False):
                sys.stderr.write("nmrfit: %s\n" % note)
                self._device_note_shown = True
        return 0 if device is None else int(device)

    def _plan(self):
        """What fit() decides before anything touches the GPU (utils.py:164-181): the weights, the swarm's constants and
        size, the seed, the kernel variant.  Shared by fit() and by core.fit_many's device-batched path."""
        self.weights = self._compute_weights()
        if self.dynamic_weighting is False:
            self.weights = np.ones_like(self.weights)
        opt = self.options
        plan = dict(kw=dict(omega=opt.get('omega', pso.DEFAULTS['omega']), phip=opt.get('phip', pso.DEFAULTS['phip']),
                            phig=opt.get('phig', pso.DEFAULTS['phig']), minstep=opt.get('minstep', pso.DEFAULTS['minstep']),
                            minfunc=opt.get('minfunc', pso.DEFAULTS['minfunc'])),
                    swarmsize=opt.get('swarmsize', pso.DEFAULTS['swarmsize']),
                    maxiter=opt.get('maxiter', pso.DEFAULTS['maxiter']), check_every=opt.get('check_every', 64))
        seed = opt.get('seed')
        if seed is None:     # pyswarm draws from numpy's unseeded global RNG: do the equivalent
            seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
        self.seed = seed     # (extra attribute: the seed this fit ran with -- rank 0's in a multi-rank fit)
        plan['seed'] = seed
        # kernel variant: by name or number, default by problem size (default_variant above)
        n_peaks = (len(self.lower) - 4) // 3
        plan['n_peaks'] = n_peaks
        plan['variant'] = _cabi.variant_id(opt.get('variant', default_variant(len(self.data.w), n_peaks, self.fit_im)))
        return plan

    def _batch_key(self, plan):
        """The key under which core.fit_many may put this fit into a device batch with others (csrc/batch.hip: equal
        kernel variant, imaginary-channel mode, maxiter and check_every; grid lengths, peak counts and swarm sizes may
        differ) -- or None when it must run on its own."""
        opt = self.options
        mode = equations.fit_im_mode(self.fit_im)
        if opt.get('exchange') is not None:
            return None
        if plan['variant'] not in (_cabi.VARIANT_DEFAULT, _cabi.VARIANT_FARFIELD) or len(self.lower) > 400:
            return None
        return (self._device(), None, None, plan['variant'], int(plan['maxiter']), int(plan['check_every']), mode)

    def _finish(self, xopt, fopt):
        self.params = xopt
        self.error = fopt
        if self.summary is True:
            self._print_summary()

    def fit(self, plan=None):
        """utils.py:164-189: weights, minimise, store params/error, optional summary.  (``plan``: what ``_plan()``
        returned for this fit, when the caller -- core.fit_many -- has made it already.)"""
        if plan is None:
            plan = self._plan()
        opt = self.options
        kw, swarmsize, maxiter, seed = plan['kw'], plan['swarmsize'], plan['maxiter'], plan['seed']
        # Multi-GPU fits (one process per GPU, every rank makes the same fit() call):
        # options['exchange'] = "rccl" builds the RCCL communicator from the launcher's
        # environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; nmrfit_amd.rendezvous), or pass a
        # ready pso.RcclExchange (used as it is, fit after fit: a communicator serves any context of its
        # device) / an exchange object with the same interface (tests).
        exchange = opt.get('exchange')
        own_exchange = False
        if isinstance(exchange, str) and exchange.lower() != "rccl":
            raise ValueError("options['exchange'] must be \"rccl\" or an exchange object")
        device = self._device()

        ev = equations.Evaluator(self.data.w, self.data.u, self.data.v, self.weights, device=device)
        try:
            if isinstance(exchange, str):
                exchange = pso.RcclExchange(ev)
                own_exchange = True
            elif isinstance(exchange, pso.RcclExchange) and exchange.ev is not ev:
                # A ready communicator, made on another context of the same device (self._device() took the
                # exchange's): it serves this fit's swarm as it is -- the all-gather runs on the stream of the
                # swarm's own context (csrc/comm.hip), so ONE ncclCommInitRank covers fit after fit.  Collective,
                # like the fit itself: every rank passes its exchange object.
                if exchange.channel is None or not exchange.handle.value:
                    raise ValueError("options['exchange']: this RcclExchange has been closed")
            ev.set_fit_im(self.fit_im)     # True: the reference's imaginary term (equations.py:197-209)
            ev.set_variant(plan['variant'])
            if exchange is None or (exchange.world == 1 and not isinstance(exchange, pso.RcclExchange)):
                xopt, fopt = pso.pso(ev, self.lower, self.upper, swarmsize=swarmsize, maxiter=maxiter, seed=seed,
                                     check_every=plan['check_every'], verbose=True, **kw)
            else:
                # every rank must run the same swarm: rank 0's seed wins (an unseeded fit would
                # otherwise draw a different seed on every rank)
                seed = exchange.broadcast_seed(seed)
                self.seed = seed
                off, n = pso.shard(swarmsize, exchange.rank, exchange.world)
                small_shard_warning(n, len(self.data.w), plan['n_peaks'], exchange.world, rank=exchange.rank)
                sw = pso.DeviceSwarm(ev, self.lower, self.upper, swarmsize, offset=off, S_local=n, seed=seed, **kw)
                try:
                    xopt, fopt = pso.run_sharded(sw, exchange, maxiter, check_every=plan['check_every'],
                                                 verbose=True)
                finally:
                    sw.close()
            if opt.get('polish', False):
                # opt-in extension (no reference counterpart): trust-region least squares on the
                # batched residual kernel, started from the swarm's answer
                from . import lsq
                xopt, fopt, _ = lsq.polish(ev, xopt, self.lower, self.upper, fit_im=self.fit_im)
        finally:
            if own_exchange:
                exchange.close()
            ev.close()
        self._finish(xopt, fopt)

    def _polish(self, xopt, fopt, plan):
        """options['polish'] for a fit whose swarm ran in a device batch (core.fit_many): the same least-squares
        refinement fit() applies, on a context made like fit()'s (weights, kernel variant, imaginary-channel mode), so
        the polished answer is the lone call's bit for bit.  The trust-region iterations are scipy's, one fit at a time
        on the host (each Jacobian is one launch of D + 1 residual rows): they do not run in lock step across fits."""
        from . import lsq
        ev = equations.Evaluator(self.data.w, self.data.u, self.data.v, self.weights, device=self._device())
        try:
            ev.set_fit_im(self.fit_im)
            ev.set_variant(plan['variant'])
            xopt, fopt, _ = lsq.polish(ev, xopt, self.lower, self.upper, fit_im=self.fit_im)
        finally:
            ev.close()
        return xopt, fopt

    def generate_result(self, scale=1):
        """utils.py:226-295: per-peak real and imaginary contributions of the fitted parameters
        on the data grid (or a grid upsampled by ``scale``), their sums V, I, and the fit
        rotated back to the (u, v) frame -- one launch of the reconstruction kernel (csrc/result.hip).
        The imaginary contributions are the Kramers-Kronig partners in closed form (the reference
        integrates each point numerically).  ``core.fit_many(jobs, generate=True)`` does this for all
        the fits of a device batch in one launch."""
        if scale == 1.0:
            w = self.data.w
        else:
            w = np.linspace(self.data.w.min(), self.data.w.max(), int(scale * self.data.w.shape[0]))
        ev = equations.Evaluator(self.data.w, self.data.u, self.data.v, np.ones(len(self.data.w)),
                                 device=self._device())
        try:
            real, imag, fit, rotated = ev.generate_result(self.params, None if scale == 1.0 else w)
        finally:
            ev.close()
        self._store_result(w, real, imag, fit, rotated)

    def _store_result(self, w, real, imag, fit, rotated, call_shift_phase=True):
        """The attributes generate_result leaves behind (utils.py:251, 289-295), from the arrays the reconstruction
        kernel returns (csrc/result.hip) -- for one fit (generate_result) or as views of a batch's arrays
        (core.fit_many(generate=...))."""
        p0, p1 = self.params[0], self.params[1]
        # phase shift data by fit theta (utils.py:251; containers.py:68-78 with method='manual').  A data object that
        # brings the reference's method gets it called; one that does not gets the same attributes set from the kernel's
        # rotation of the spectrum.
        if call_shift_phase and hasattr(self.data, 'shift_phase'):
            self.data.shift_phase(method='manual', p0=p0, p1=p1)
        else:
            self.data.p0, self.data.p1 = p0, p1
            self.data.V, self.data.I = rotated[0], rotated[1]
        # (V_fit, I_fit: the contributions added peak after peak from zero, utils.py:276-277; u_fit, v_fit: the fit
        # rotated back, utils.py:284 -- all four evaluated next to the contributions on the GPU)
        self.u = fit[2]
        self.v = fit[3]
        self.V = fit[0]
        self.I = fit[1]
        self.w = w
        self.real_contribs = [real[k] for k in range(real.shape[0])]
        self.imag_contribs = [imag[k] for k in range(imag.shape[0])]

    def get_areas(self):
        """utils.py:312-322."""
        return np.array([self.params[i] for i in range(6, len(self.params), 3)])

    def calculate_area_fraction(self):
        """utils.py:297-310: satellites (areas below the mean) over total."""
        areas = self.get_areas()
        m = np.mean(areas)
        peaks = areas[areas >= m].sum()
        sats = areas[areas < m].sum()
        return sats / (peaks + sats)

    def _print_summary(self):
        """What FitUtility._print_summary reports (nmrfit/utils.py:324-339): the four global parameters, one row per
        peak, the error -- through pandas' DataFrame.to_string(index=False) like the reference, so that whoever parses
        the reference's summary parses this one; a plain-text table of the same columns only where pandas is missing."""
        values = np.asarray(self.params, dtype=float)
        head, peak = ['p0', 'p1', 'r', 'y-off'], ['width', 'location', 'area']
        try:
            import pandas as pd
        except ImportError:
            pd = None

        def table(header, rows):
            if pd is not None:
                return pd.DataFrame(np.asarray(rows, dtype=float).reshape(-1, len(header)), columns=header).to_string(index=False)
            cells = [["%.6g" % x for x in row] for row in rows]
            widths = [max(len(h), *(len(c[k]) for c in cells)) for k, h in enumerate(header)]
            lines = [" ".join(h.rjust(n) for h, n in zip(header, widths))]
            lines += [" ".join(c.rjust(n) for c, n in zip(row, widths)) for row in cells]
            return "\n".join(lines)
        print("\nFit Summary:")
        print("------------")
        print("Global parameters")
        print(table(head, [values[:4]]))
        print("\nPeak parameters")
        print(table(peak, values[4:].reshape(-1, 3)))
        print("Error:\t", self.error)
