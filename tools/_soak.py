import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import fluidsolver_amd as fs, oracle
for cfg in (dict(), dict(flags=fs.FS_FLAG_VORT_CONFINE, pressure_solver=fs.FS_PRESSURE_MGPCG)):
    n=(64,64,64); kw=dict(jacobi_iters=20)
    okw=dict(kw)
    if cfg: okw.update(vort_confine=1, pressure_solver=oracle.PRESSURE_MGPCG); kw.update(jacobi_iters=30); okw.update(jacobi_iters=30)
    g=fs.Solver(fs.default_params(*n, **kw, **cfg)).Initialize(); o=oracle.Oracle(oracle.default_params(*n, **okw))
    t=time.time(); steps=1000 if not cfg else 300
    for s in range(steps):
        g.step(); o.step()
        if (s+1) % 100 == 0:
            bad=[f for f in range(6) if not np.array_equal(g.field(f), o.get(f))]
            err=max(float(np.abs(g.field(f)-o.get(f)).max())/max(float(np.abs(o.get(f)).max()),1e-30) for f in range(6))
            print("%s step %d: fields differing %r, max rel err %.1e, %.0f s" % ("shipped" if cfg else "jacobi", s+1, bad, err, time.time()-t), flush=True)
    g.close(); o.close()
