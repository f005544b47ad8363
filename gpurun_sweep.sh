python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for cfg in "1024 0" "1024 1" "2048 1"; do
  set -- $cfg
  ZD_NO_OVERLAP=1 timeout 600 python bench.py --ppd $1 --plt $2 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('noovl',d['config']['workload'][:22], 'ms',round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
done
python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('default', 'ms',round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
