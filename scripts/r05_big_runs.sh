# round 5: every row of DESIGN §4's table on the round's final build (one GPU; bench.py for the BASELINE workloads, zd_generate with the
# NULL sink for the large / composite ones)      bash scripts/r05_big_runs.sh > gpurun_out/r05_big_runs.log 2>&1
cd $GRAFT_REPO_ROOT
F='WARNING|spectrum|extrap|make sure|that goes|it\) to|sigma|amdgpu.ids|^$|Loading'
b() {
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench $*:', d['config']['workload'], 'R', d['config']['stream_factor'], 'passes', d['config']['passes'], 'step ms', round(d['ms_per_step'],1), 'particles/s %.3e' % d['value'], 'path', round(d['roofline_path_frac'],3), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items()})" || exit 1
}
b
b --ppd 2048
b --ppd 1024
b --ppd 2048 --plt 1
b --ppd 2048 --plt 1 --store-mode packed
b --ppd 1024 --plt 1
b --ppd 4096 --plt 1
timeout -k 10 200 python scripts/run_8192.py 8192 2.0 2>&1 | grep -Ev "$F" || exit 1
timeout -k 10 200 python scripts/run_8192.py 8192 2.0 auto 2 2 1 2>&1 | grep -Ev "$F" || exit 1
timeout -k 10 300 python scripts/run_8192.py 16384 4.0 2>&1 | grep -Ev "$F" || exit 1
timeout -k 10 600 python scripts/np2_times.py 3456 3456:1:plt 3456:1:dens 6912:2 6912:2:plt 6912 6912:1:plt 4000 1000 2000 2>&1 | grep -Ev "$F" || exit 1
