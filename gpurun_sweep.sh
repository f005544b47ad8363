python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft_lines" 2>&1 | tail -2
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
t=time.time(); a = zd.generate(zd.make_params(4096, icformat="RVZel", profile=1), ps, collect=False); ta=time.time()-t
print("4096 ZA: R", a["stream_factor"], "sec", round(a["seconds_total"],2), "wall", round(ta,1), "Gp/s", round(4096**3/a["seconds_total"]/1e9,2), {k:round(v) for k,v in a["kernel_ms"].items()})
t=time.time(); b = zd.generate(zd.make_params(8192, k_cutoff=2.0, icformat="RVZel", profile=1), ps, collect=False); tb=time.time()-t
print("8192 kc2: R", b["stream_factor"], "sec", round(b["seconds_total"],2), "wall", round(tb,1), "Gp/s", round(8192**3/b["seconds_total"]/1e9,2), {k:round(v) for k,v in b["kernel_ms"].items()})
print("variance ratio (expect 8):", b["density_variance"]/a["density_variance"], "max_disp", a["max_disp"], b["max_disp"])
PY
