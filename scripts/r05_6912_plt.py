"""PPD=6912 PLT + rescale (R = 48, z lines of 144): one run, kernel spans (tuning library knobs ZD_ZQ_NC144, serial_z through SERIAL=1)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
import bench
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
eig = bench.synthetic_eigenmodes(128)
a = zd.generate(zd.make_params(6912, icformat="RVZel", profile=1, qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0,
                               serial_z=int(os.environ.get("SERIAL", "0"))), ps, eig=eig, collect=False)
print("6912 PLT R", a["stream_factor"], "ZD_ZQ_NC144", os.environ.get("ZD_ZQ_NC144"), "serial_z", os.environ.get("SERIAL", "0"), "sec", round(a["seconds_total"], 2),
      {k: round(v) for k, v in a["kernel_ms"].items()}, flush=True)
