"""In-kernel phase stamps of the field-store y stage (`make stamps` library, zdk_set_stamps): where a workgroup's time goes.
   ZD_LIB_PATH=.../libzeldovich_hip_stamps.so python scripts/yf_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zeldovich_plt_amd.api as zd
WMAP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wmap1new.pow")
N = int(os.environ.get("N", "4096"))
W = int(os.environ.get("W", "4"))
ps = zd.PowerSpectrum.from_file(WMAP, 720.0)
p = zd.make_params(N, icformat="RVZel", profile=1, stream_factor=int(os.environ.get("R", "8")), numblock=64)
nplanes = int(os.environ.get("PLANES", "16"))
L = zd.load_library()
plan = zd.Plan(p, ps)
store = torch.empty(plan.exchange_bytes, dtype=torch.uint8, device="cuda")
plan.stage_z(0, store.data_ptr())
torch.cuda.synchronize()
out = torch.zeros(nplanes * N * N * 32, dtype=torch.uint8, device="cuda")
plan.stage_x(0, store.data_ptr(), 0, nplanes, out.data_ptr())
torch.cuda.synchronize()
ring_planes = 7  # field_ring_planes(4096)
nunits = 3 * (N // W) * ring_planes
buf = torch.zeros(nunits * 8, dtype=torch.int64, device="cuda")
L.zdk_set_stamps.argtypes = [C.c_void_p]
assert L.zdk_set_stamps(buf.data_ptr()) == 0
plan.stats()
plan.stage_x(0, store.data_ptr(), 0, 2 * ring_planes, out.data_ptr())   # ONE y launch of ring_planes store planes
torch.cuda.synchronize()
st = plan.stats()
assert L.zdk_set_stamps(None) == 0
print("yfft %.3f ms for %d store planes -> %.0f ms/step" % (st["kernel_ms"]["k_yfft"], ring_planes, st["kernel_ms"]["k_yfft"] * (N // 2) / ring_planes))
s = buf.cpu().numpy().reshape(-1, 8).astype(np.int64)
ok = s[:, 0] != 0
s = s[ok]
d = np.diff(s[:, :6], axis=1).astype(np.float64)
names = ["row records", "potential loads", "fft", "store issue", "store drain"]
print("units stamped: %d" % len(s))
for i, n in enumerate(names):
    print("%-16s mean %8.0f  median %8.0f  p90 %8.0f shader cycles" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
tot = (s[:, 5] - s[:, 0]).astype(np.float64)
print("%-16s mean %8.0f  median %8.0f" % ("start..drained", tot.mean(), np.median(tot)))
# per CU: gaps between consecutive workgroups (100 MHz real-time clock) and the clock
hw = s[:, 7]
cu = ((hw >> 32) & 0xf) * 1024 + ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 0x3) + 64 * ((hw >> 13) & 0x7)   # xcc, cu_id, sh_id, se_id
gaps, durs = [], []
for c in np.unique(cu):
    m = cu == c
    t0 = np.sort(s[m, 6])
    if len(t0) > 2:
        gaps.append(np.diff(t0))
print("CUs seen: %d; start-to-start interval per CU: median %.2f us, mean %.2f us" % (len(np.unique(cu)), np.median(np.concatenate(gaps)) / 100.0, np.mean(np.concatenate(gaps)) / 100.0))
span = (s[:, 6].max() - s[:, 6].min()) / 100.0
print("launch span %.1f us; shader clock est: total cycles/us = %.0f MHz" % (span, np.median(tot) / (np.median(np.concatenate(gaps)) / 100.0)))
a = (np.arange(len(ok))[ok] % (3 * (N // W)))
