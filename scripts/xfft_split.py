# how much of the x pass is the record stores: time stage_x with and without a record buffer (PPD=4096 ZA, one pass)
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
plan.stage_y(buf.data_ptr(), st)
chunk = 32
ring = torch.empty(chunk * n * n * 32, dtype=torch.uint8, device="cuda")
def run(rec):
    torch.cuda.synchronize(); t = time.perf_counter()
    for p0 in range(0, plan.local_planes, chunk):
        plan.stage_x(0, buf.data_ptr(), p0, chunk, rec, None, st)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
for rec, name in ((ring.data_ptr(), "with records"), (None, "reductions only"), (ring.data_ptr(), "with records")):
    print("XSPLIT", n, name, round(run(rec), 1), "ms per pass", flush=True)
