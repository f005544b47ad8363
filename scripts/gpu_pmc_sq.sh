# One rocprofv3 --pmc pass with SQ issue counters for the default bench workload (one step): where the waves' cycles go per kernel.
#   gpurun -- 'TAG=r03d bash scripts/gpu_pmc_sq.sh'   -> gpurun_out/pmc_<TAG>_SQ_summary.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03d}
ARGS=${ARGS:-}
cd /tmp
PMC=${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU}
SUF=${SUF:-SQ}
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_${SUF} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated $ARGS > $R/gpurun_out/pmc_${TAG}_${SUF}.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_${SUF}.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']; TAG="$TAG"; SUF="$SUF"; PMC="$PMC".split()
f=glob.glob(f"{R}/gpurun_out/pmc_{TAG}_{SUF}/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for row in csv.DictReader(open(f)):
    k=row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ","").strip()
    agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
    if row["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
names=PMC
with open(f"{R}/gpurun_out/pmc_{TAG}_{SUF}_summary.csv","w") as o:
    o.write("kernel,launches,"+",".join(names)+","+",".join(c+"/SQ_WAVE_CYCLES" for c in names[1:])+"\n")
    for k,v in agg.items():
        if not k.startswith("k_"): continue
        wc=v["SQ_WAVE_CYCLES"] or 1
        line='"%s",%d,'%(k,n[k])+",".join("%.4g"%v[c] for c in names)+","+",".join("%.3f"%(v[c]/wc) for c in names[1:])
        o.write(line+"\n"); print(line)
PY
