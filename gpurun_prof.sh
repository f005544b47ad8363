export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final2_bench.log 2>&1
tail -1 $R/gpurun_out/prof_final2_bench.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc3_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
R=os.environ['GRAFT_REPO_ROOT']
res=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/pmc3_{c}/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(lambda:[0,0.0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name")!=c: continue
            k=row["Kernel_Name"].split("<")[0].replace("void ","").strip()
            agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    with open(f"{R}/gpurun_out/pmc3_{c}_summary.csv","w") as o:
        o.write("kernel,launches,sum_%s_KB\n"%c)
        for k,(n,v) in agg.items():
            res[k][c]={"launches":n,"total_KB":v}; o.write('"%s",%d,%.1f\n'%(k,n,v))
out={}
for k,v in res.items():
    if not k.startswith("k_"): continue
    n=v["FETCH_SIZE"]["launches"]
    fetch=2*v["FETCH_SIZE"]["total_KB"]*1024; write=v["WRITE_SIZE"]["total_KB"]*1024
    out[k]=(fetch+write)/n
    print(k,"launches",n,"fetch GB (x2)",round(fetch/1e9,1),"write GB",round(write/1e9,1),"per launch MB",round(out[k]/1e6,1))
json.dump(out,open(f"{R}/gpurun_out/traffic_latest.json","w"),indent=1)
PY
cat $(find $R/gpurun_out/prof_final2 -name "*kernel_stats.csv" | head -1) | head -6 | cut -c1-200
cp gpurun_out/traffic_latest.json profiles/traffic_latest.json
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_default2.json; python3 -c "
import json;d=json.load(open('gpurun_out/bench_default2.json'));print(d['value'],d['ms_per_step'],d['roofline_path_frac']);print(d['cpu_baseline']);print(d['roofline']);print(d['kernels'])"
python -m pytest tests -m gpu -q 2>&1 | tail -1
