# PLT generator capped at 168 VGPRs (three workgroups per CU) — times for PPD = 2048 / 4096 PLT + rescale
cd $GRAFT_REPO_ROOT
for args in "--ppd 2048 --plt 1" "--ppd 4096 --plt 1"; do
timeout -k 10 300 python bench.py $args --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', 'R', d['config']['stream_factor'], 'step', round(d['ms_per_step']), {k: round(v['ms_per_step']) for k,v in d['kernels'].items()}, 'isolated', {k: round(v['ms_per_step']) for k,v in d['kernels_isolated'].items()})" || exit 1
done
