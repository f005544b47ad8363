import os, sys, numpy as np
sys.path.insert(0, ".")
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
for fmt in ("ZelSimple", "RVdoubleZel"):
    for R in (1, 2):
        a = zd.generate(zd.make_params(64, icformat=fmt, stream_factor=R), ps)
        d = a["records"]["d"]
        print(fmt, R, "nan count", int(np.isnan(d).sum()), "var", a["density_variance"], "maxd", a["max_disp"])
