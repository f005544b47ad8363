"""Timed runs of big / composite PPDs: python scripts/np2_times.py 3456 6912:2 6400:2 6912:1:plt 3456:1:dens   (PPD[:k_cutoff[:plt|dens|dens2|pltdens]];
dens = ZD_qdensity = 1, density planes produced in HBM and dropped like the records)"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
for spec in sys.argv[1:]:
    f = spec.split(":")
    n, kc, plt = int(f[0]), float(f[1]) if len(f) > 1 and f[1] else 1.0, len(f) > 2 and f[2] == "plt"
    kw, eig = {}, None
    if len(f) > 2 and f[2] == "dens":
        kw = dict(qdensity=1)
    if len(f) > 2 and f[2] == "dens2":
        kw = dict(qdensity=2)
    pltdens = len(f) > 2 and f[2] == "pltdens"  # PLT + ZD_qdensity = 1 (composite grids: a density-only pass in front of every PLT pass)
    if plt or pltdens:
        import bench
        eig = bench.synthetic_eigenmodes(128)
        kw = dict(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
        if pltdens:
            kw.update(qdensity=1)
            plt = True
    a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1, k_cutoff=kc, **kw), ps, eig=eig, collect=False)
    print(n, ("PLT+density" if kw.get("qdensity") else "PLT") if plt else ("ZA+density" if kw.get("qdensity") == 1 else ("density only" if kw.get("qdensity") == 2 else "ZA")), "k_cutoff", kc, "R", a["stream_factor"], "sec", round(a["seconds_total"], 2),
          {k: round(v) for k, v in a["kernel_ms"].items()}, "var", repr(a["density_variance"]), flush=True)
