import sys, os, time
sys.path.insert(0, os.getcwd())
import fluidsolver_amd as fs
for pad in ("0", "1"):
    os.environ["FS_MG_LDS_PAD"] = pad
    for dims in ((64, 128, 64), (64, 64, 64), (128, 128, 128)):
        s = fs.Solver(fs.default_params(*dims, jacobi_iters=30, flags=fs.FS_FLAG_VORT_CONFINE, pressure_solver=fs.FS_PRESSURE_MGPCG)).Initialize()
        s.step(50); s.sync()
        t = time.perf_counter(); s.step(300); s.sync(); dt = (time.perf_counter() - t) / 300
        print("FS_MG_LDS_PAD=%s %s shipped configuration: %.3f ms/step, %d CG iterations in the last step" % (pad, dims, dt * 1e3, s.cg_info()[0]), flush=True)
        s.close()
