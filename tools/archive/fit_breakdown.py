import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from nmrfit_amd import synth, pso, utils, equations, _cabi
sp = synth.make_spectrum(4096, 6, seed=1)
data = synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"])
_cabi.lib(); _cabi.device_count()
for rep in range(3):
    t0=time.perf_counter(); w = utils.compute_weights(data.w, data.peaks, 0.5); t1=time.perf_counter()
    ev = equations.Evaluator(data.w, data.u, data.v, w); t2=time.perf_counter()
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], 204, seed=7, minfunc=-1.0, minstep=-1.0); t3=time.perf_counter()
    sw.run(2000, 64); t4=time.perf_counter()
    x,f = sw.best(); st=sw.status(); t5=time.perf_counter()
    sw.close(); ev.close(); t6=time.perf_counter()
    print("weights %.2f ms, ctx %.2f, swarm create %.2f, run %.2f, best+status %.2f, close %.2f" % tuple(1e3*v for v in (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5)))
