import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
import bench
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
def run(n, label, eig=None, **kw):
    t = time.time()
    a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1, **kw), ps, eig=eig, collect=False)
    print(label, "R", a["stream_factor"], "sec", round(a["seconds_total"], 2), "wall", round(time.time() - t, 1), "Gp/s",
          round(n**3 / a["seconds_total"] / 1e9, 2), {k: round(v) for k, v in a["kernel_ms"].items()},
          "var", repr(a["density_variance"]), "maxd", list(a["max_disp"]), flush=True)
    return a
a = run(4096, "4096 ZA")
b = run(8192, "8192 kc2", k_cutoff=2.0)
print("variance ratio (expect 8):", b["density_variance"] / a["density_variance"])
eig = bench.synthetic_eigenmodes(128)
run(4096, "4096 PLT", eig=eig, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
