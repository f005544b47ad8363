cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/r05_fz_check.py 1024 2048 2>&1 | grep "worst\|OK\|Error\|error" || exit 1
for i in 1 2; do
timeout -k 10 300 python bench.py --ppd 2048 --plt 1 --steps 3 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('product step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items()})" || exit 1
done
timeout -k 10 300 python bench.py --ppd 1024 --plt 1 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PPD 1024 PLT step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items()})" || exit 1
timeout -k 10 300 python bench.py --ppd 1024 --plt 1 --steps 5 --warmup 1 --no-cpu-baseline --no-isolated --store-mode packed 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PPD 1024 PLT two kernels step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items()})" || exit 1
