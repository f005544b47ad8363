# GPU tests + timing probe (kernel breakdown) for a list of "ppd R plt" cases
set -e
mkdir -p gpurun_out
TAG=${TAG:-p2}
if [ -z "$SKIPTESTS" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }; tail -2 gpurun_out/${TAG}_tests.log; fi
cat > /tmp/probe.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import zeldovich_plt_amd.api as zd
import bench
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
n = int(sys.argv[1]); R = int(sys.argv[2]); plt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kw = dict(icformat="RVZel", profile=1, stream_factor=R)
eig = None
if plt:
    kw.update(qPLT=1, qPLTrescale=1, PLT_target_z=5.0, z_initial=49.0)
    eig = bench.synthetic_eigenmodes(128)
a = zd.generate(zd.make_params(n, **kw), ps, eig=eig, collect=False)
print("PPD", n, "PLT", plt, "R", a["stream_factor"], "env", {k: v for k, v in os.environ.items() if k.startswith("ZD_")},
      "sec", round(a["seconds_total"], 3), "Gp/s", round(n**3 / a["seconds_total"] / 1e9, 2),
      {k: round(v, 1) for k, v in a["kernel_ms"].items()}, "var", repr(a["density_variance"]), flush=True)
PY
rm -f gpurun_out/${TAG}_probe.log
while read -r line; do
  [ -z "$line" ] && continue
  env $line >> gpurun_out/${TAG}_probe.log 2>&1 || { tail -5 gpurun_out/${TAG}_probe.log; exit 1; }
done <<CASES
$CASES
CASES
grep "^PPD" gpurun_out/${TAG}_probe.log
