cd $GRAFT_REPO_ROOT
F='WARNING|spectrum|extrap|make sure|that goes|it\) to|sigma|amdgpu.ids|^$'
timeout -k 10 200 python scripts/run_8192.py 8192 2.0 auto 2 2 1 2>&1 | grep -Ev "$F" || exit 1
timeout -k 10 300 python bench.py --ppd 4096 --plt 1 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PPD=4096 PLT', 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()})" || exit 1
timeout -k 10 300 python scripts/np2_times.py 6912:1:plt 2>&1 | grep -Ev "$F"
