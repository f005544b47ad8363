# Z-stage time of each rank's share (emulated on one GPU): checks the cyclic row assignment's load balance
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import zeldovich_plt_amd.api as zd
ps = zd.PowerSpectrum.from_file('tests/golden/wmap1new.pow', 720.0)
n, world = 2048, 4
p = zd.make_params(n, icformat="RVZel", stream_factor=4)
for rank in range(world):
    plan = zd.Plan(p, ps, rank=rank, nranks=world)
    buf = torch.empty(plan.exchange_bytes // 8, dtype=torch.float64, device="cuda")
    plan.stage_z(0, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for ps_ in range(plan.passes):
        plan.stage_z(ps_, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("RANK", rank, "of", world, "passes", plan.passes, "z-stage ms", round((time.perf_counter() - t) * 1e3, 1), flush=True)
    plan.close()
    del buf
