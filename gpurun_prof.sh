export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --ppd 1024 --plt 1 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
tail -1 $R/gpurun_out/pmc_$c.log | cut -c1-200
done
ls -R $R/gpurun_out/pmc_FETCH_SIZE | head
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no file", c); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    with open(f[0]) as fh:
        rd=csv.DictReader(fh)
        for row in rd:
            if row.get("Counter_Name")!=c: continue
            k=row["Kernel_Name"].split("(")[0][:40]
            agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    for k,(n,v) in agg.items(): print(c,k,n,"total KB",v, "GB", v*1024/1e9 if v>1e6 else v/1e9)
PY
