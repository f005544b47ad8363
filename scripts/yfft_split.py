import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = zd.make_params(n, icformat="RVZel", stream_factor=16 if n == 4096 else 2)
plan = zd.Plan(p, ps)
buf = torch.empty(plan.exchange_bytes // 8, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
plan.stage_z(0, buf.data_ptr(), st)
torch.cuda.synchronize()
for i in range(3):
    t = time.perf_counter(); plan.stage_y(buf.data_ptr(), st); torch.cuda.synchronize()
    print("YSPLIT", n, os.environ.get("ZD_PRUNE"), round((time.perf_counter() - t) * 1e3, 1), "ms per pass", flush=True)
