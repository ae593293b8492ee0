import contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import nmrfit_amd
from nmrfit_amd import synth, utils, core
from nmrfit_amd.batch import FitBatch
K=200
jobs=[]
for k in range(K):
    sp = synth.make_spectrum(4096, 6, seed=100 + k % 8)
    jobs.append((synth.SynthData(sp["w"], sp["u"], sp["v"], sp["peaks"]), list(sp["lower"]), list(sp["upper"])))
with contextlib.redirect_stdout(io.StringIO()):
    nmrfit_amd.fit(*jobs[0], summary=False, options={"seed": 1, "maxiter": 5})
    nmrfit_amd.fit_many(jobs[:8], options={"seed": 7})
for rep in range(2):
    t0=time.perf_counter()
    fits=[utils.FitUtility(d,l,u,summary=False,options={"seed":7}) for d,l,u in jobs]
    t1=time.perf_counter()
    plans=[f._plan() for f in fits]
    t2=time.perf_counter()
    key=fits[0]._batch_key(plans[0])
    device,_,swarmsize,variant,maxiter,check_every,fit_im=key
    spectra=[(f.data.w,f.data.u,f.data.v,f.weights) for f in fits]
    kw={name:[p['kw'][name] for p in plans] for name in ("omega","phip","phig","minstep","minfunc")}
    t3=time.perf_counter()
    fb=FitBatch(spectra,[f.lower for f in fits],[f.upper for f in fits],swarmsize=swarmsize,seeds=[p['seed'] for p in plans],variant=variant,fit_im=fit_im,device=device,**kw)
    t4=time.perf_counter()
    fb.run(maxiter,check_every)
    t5=time.perf_counter()
    st=fb.status(); best=fb.best(); fb.close()
    t6=time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        for f,p,s,(x,fx) in zip(fits,plans,st,best): f._finish(x,fx)
    t7=time.perf_counter()
    print("objects %.1f ms, plans %.1f, gather %.1f, create %.1f, run %.1f, read+close %.1f, finish %.1f; total %.1f ms; generations max %d mean %.0f; check_every %d"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(t6-t5)*1e3,(t7-t6)*1e3,(t7-t0)*1e3,max(s['iteration'] for s in st),sum(s['iteration'] for s in st)/K,check_every))
