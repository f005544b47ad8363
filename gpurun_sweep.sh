for cfg in "2048 0" "2048 1" "4096 0"; do
  set -- $cfg
  timeout 600 python bench.py --ppd $1 --plt $2 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
done
