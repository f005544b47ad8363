# round-1 re-entry probe: GPU tests, then kernel breakdown of PPD=4096 ZA (R=8) and gen ablations at R=8
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/p1_tests.log 2>&1 || { tail -20 gpurun_out/p1_tests.log; exit 1; }
tail -2 gpurun_out/p1_tests.log
cat > /tmp/probe.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
n = int(sys.argv[1]); R = int(sys.argv[2])
a = zd.generate(zd.make_params(n, icformat="RVZel", profile=1, stream_factor=R), ps, collect=False)
print("PPD", n, "R", a["stream_factor"], "ABL", os.environ.get("ZD_ABLATE"), "NOOVL", os.environ.get("ZD_NO_OVERLAP"),
      "sec", round(a["seconds_total"], 3), "Gp/s", round(n**3 / a["seconds_total"] / 1e9, 2),
      {k: round(v, 1) for k, v in a["kernel_ms"].items()}, flush=True)
PY
python /tmp/probe.py 4096 0 >> gpurun_out/p1_probe.log 2>&1
ZD_NO_OVERLAP=1 python /tmp/probe.py 4096 0 >> gpurun_out/p1_probe.log 2>&1
for abl in 0 1 2 4 6 7; do
ZD_NO_OVERLAP=1 ZD_ABLATE=$abl python /tmp/probe.py 2048 8 >> gpurun_out/p1_probe.log 2>&1
done
ZD_NO_OVERLAP=1 python /tmp/probe.py 2048 1 >> gpurun_out/p1_probe.log 2>&1
python /tmp/probe.py 2048 1 >> gpurun_out/p1_probe.log 2>&1
cat gpurun_out/p1_probe.log
