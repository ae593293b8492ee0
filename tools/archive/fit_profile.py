"""cProfile of repeated nmrfit_amd.fit() calls with the reference's defaults and pyswarm's stopping rule armed (the fit
ends after a couple of hundred generations): where the host time of a short fit goes."""
import cProfile, os, pstats, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nmrfit_amd
from nmrfit_amd import synth
sp = synth.make_spectrum(4096, 6, seed=1)
data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
args = (data, list(sp["lower"]), list(sp["upper"]))
with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit(*args, summary=False, options={"seed": 7})
    t0 = time.perf_counter()
    for _ in range(50):
        r = nmrfit_amd.fit(*args, summary=False, options={"seed": 7})
    dt = (time.perf_counter() - t0) / 50
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        nmrfit_amd.fit(*args, summary=False, options={"seed": 7})
    pr.disable()
print("%.3f ms per fit (error %.6g)" % (dt * 1e3, r.error))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
