export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final_bench.log 2>&1
tail -1 $R/gpurun_out/prof_final_bench.log | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc2_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ['GRAFT_REPO_ROOT']
res=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/pmc2_{c}/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(lambda:[0,0.0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name")!=c: continue
            k=row["Kernel_Name"].split("<")[0].replace("void ","").strip()
            agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    for k,(n,v) in agg.items():
        res[k][c]={"launches":n,"total_KB":v}
out={}
for k,v in res.items():
    if not k.startswith("k_"): continue
    n=v["FETCH_SIZE"]["launches"]
    # MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of the bytes of wide coalesced reads on gfx950 -> x2; WRITE_SIZE exact; unit KB
    fetch=2*v["FETCH_SIZE"]["total_KB"]*1024; write=v["WRITE_SIZE"]["total_KB"]*1024
    out[k]=(fetch+write)/n
    print(k,"launches",n,"fetch GB (x2 corrected)",fetch/1e9,"write GB",write/1e9,"per launch MB",out[k]/1e6)
json.dump(out,open(f"{R}/gpurun_out/traffic_latest.json","w"),indent=1)
PY
cat $(find $R/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1) | head -8
timeout 900 python bench.py --ppd 4096 --plt 0 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-200 gpurun_out/bench_default.json; python3 -c "
import json;d=json.load(open('gpurun_out/bench_default.json'));print(d['cpu_baseline']);print(d['roofline'])"
