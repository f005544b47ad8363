import os, sys, numpy as np
sys.path.insert(0, ".")
import zeldovich_plt_amd.api as zd
import bench
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
eig = bench.synthetic_eigenmodes(16)
for resc in (0, 1):
    kw = dict(icformat="RVdoubleZel", stream_factor=1, qPLT=1, qPLTrescale=resc, PLT_target_z=5.0, z_initial=49.0)
    a = zd.generate(zd.make_params(64, **kw), ps, eig=eig)
    np.save("/tmp/plt_%s_%d.npy" % (os.environ.get("ZD_GEN_GENERAL", "0"), resc), a["records"])
    print("resc", resc, "var", a["density_variance"], a["max_disp"])
