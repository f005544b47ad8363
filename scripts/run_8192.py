"""PPD=8192 ZD_k_cutoff=2 (BASELINE C5 workload) on one GPU through zd_generate with the NULL sink: wall time, stream factor"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zeldovich_plt_amd.api as zd
WMAP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wmap1new.pow")
ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kc = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
version = int(sys.argv[4]) if len(sys.argv) > 4 else 2      # 1: legacy mt19937 streams, PPD / NumBlock of them
numblock = int(sys.argv[5]) if len(sys.argv) > 5 else 2
plt = int(sys.argv[6]) if len(sys.argv) > 6 else 0           # 1: ZD_qPLT + rescale on a synthetic 128^3 eigenmode table
eig, kw = None, {}
if plt:
    from oracle import zdo
    eig = zdo.synthetic_eigenmodes(128)
    kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
p = zd.make_params(n, k_cutoff=kc, icformat="RVZel", profile=1, store_mode=mode, version=version, numblock=numblock, **kw)
t0 = time.time()
out = zd.generate(p, ps, eig=eig, collect=False)
print("PPD=%d k_cutoff=%g store=%s version=%d NumBlock=%d PLT=%d: R=%d  %.2f s (library), wall %.1f s; kernel ms %s; dens var %.17g max_disp %s" % (
    n, kc, mode, version, numblock, plt, out["stream_factor"], out["seconds_total"], time.time() - t0, {k: round(v) for k, v in out["kernel_ms"].items()},
    out["density_variance"], out["max_disp"]), flush=True)
