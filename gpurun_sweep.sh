python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for cfg in "1024 0" "1024 1" "2048 0" "2048 1"; do
  set -- $cfg
  timeout 600 python bench.py --ppd $1 --plt $2 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:22], 'R',d['config']['stream_factor'],'ms',round(d['ms_per_step'],1),'Gp/s',round(d['value']/1e9,2),'frac',round(d['roofline_path_frac'],3), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})"
done
