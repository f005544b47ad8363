# round 5, GPU probe 2: fused PLT Z stage — parity against the two-kernel stage, then C3 timing both ways
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python scripts/r05_fz_check.py 1024 2048 || exit 1
for sm in "" "2"; do
  ZD_BENCH_STORE_MODE=$sm timeout -k 10 300 python bench.py --ppd 2048 --plt 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('store_mode=$sm step', round(d['ms_per_step'],1), {k: round(v['ms_per_step'],1) for k,v in d['kernels'].items()}, 'isolated', {k: round(v['ms_per_step'],1) for k,v in (d['kernels_isolated'] or {}).items()})" || exit 1
done
