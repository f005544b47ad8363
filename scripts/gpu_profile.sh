# rocprofv3 evidence for the default bench workload (run through gpurun from the repo root):
#   1. --kernel-trace --stats of `bench.py --steps 2 --warmup 1`      -> gpurun_out/prof_<TAG>/…kernel_stats.csv
#   2. separate --pmc FETCH_SIZE / WRITE_SIZE passes (one step each)    -> per-kernel HBM bytes per launch
#   3. a plain `python bench.py` (no profiler)                          -> gpurun_out/bench_<TAG>.json
# Summaries are copied to profiles/ by hand afterwards (profiles/README.md).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02a}
ARGS=${ARGS:-}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated $ARGS > $R/gpurun_out/prof_${TAG}_bench.log 2>&1 || { tail -5 $R/gpurun_out/prof_${TAG}_bench.log; exit 1; }
grep '^{"metric"' $R/gpurun_out/prof_${TAG}_bench.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json
cut -c1-300 $R/gpurun_out/${TAG}_bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated $ARGS > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_$c.log; exit 1; }
done
# 2b. one more pass with the SQ issue counters (where the waves' cycles go: bench.py reports them as roofline.valu_issue when the
#     dominant unit is the vector-bound Z stage)
SQ="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_SQ -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-isolated $ARGS > $R/gpurun_out/pmc_${TAG}_SQ.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_SQ.log; exit 1; }
cd $R
python3 - <<PY
import csv, glob, collections, os, json
R=os.environ['GRAFT_REPO_ROOT']; TAG="$TAG"
res=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{R}/gpurun_out/pmc_{TAG}_{c}/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(lambda:[0,0.0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name")!=c: continue
            k=row["Kernel_Name"].split("<")[0].replace("void ","").strip()
            agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    with open(f"{R}/gpurun_out/pmc_{TAG}_{c}_summary.csv","w") as o:
        o.write("kernel,launches,sum_%s_KB\n"%c)
        for k,(n,v) in agg.items():
            res[k][c]={"launches":n,"total_KB":v}; o.write('"%s",%d,%.1f\n'%(k,n,v))
line=json.loads([l for l in open(f"{R}/gpurun_out/pmc_{TAG}_FETCH_SIZE.log").read().splitlines() if l.startswith('{"metric"')][-1])
wl=line["config"]["workload"]
out={"workload": "PPD=%s plt=%d" % (wl.split()[0].split("=")[1], 1 if "qPLT=1" in wl else 0),
     "source_sha": line.get("source_sha"),  # the native sources this profile was taken on (bench.py drops a stale figure)
     "store_arrays": 3 if ("fields" in line["config"]["store"] or line["config"]["store"].startswith("3")) else int(line["config"]["store"].split()[0]),
     "passes": line["config"]["passes"], "bytes_per_launch": {}, "bytes_per_step": {},
     "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024/launches: gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md, HBM)"}
NAMES={"k_genf":"k_gen","k_genz_plt":"k_gen","k_zfft_f":"k_zfft","k_yfft_f":"k_yfft","k_xfft_seq":"k_xfft","k_xfft_seq_plt":"k_xfft","k_xfft_q2_plt":"k_xfft"}
# SQ counters per kernel: fractions of the waves' cycles (summed over waves) spent issuing / issuing VALU / waiting to issue
sqf=glob.glob(f"{R}/gpurun_out/pmc_{TAG}_SQ/**/*counter_collection.csv", recursive=True)
if sqf:
    sq=collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(sqf[0])):
        k=row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ","").strip()
        sq[NAMES.get(k,k)][row["Counter_Name"]]+=float(row["Counter_Value"])
    out["sq"]={k:{"valu_issue_per_wave_cycle": v["SQ_ACTIVE_INST_VALU"]/v["SQ_WAVE_CYCLES"], "any_issue_per_wave_cycle": v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"],
                  "wait_to_issue_per_wave_cycle": v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"]} for k,v in sq.items() if k.startswith("k_") and v["SQ_WAVE_CYCLES"]>0}
for k,v in res.items():
    if not k.startswith("k_") or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v: continue
    n=v["FETCH_SIZE"]["launches"]
    fetch=2*v["FETCH_SIZE"]["total_KB"]*1024; write=v["WRITE_SIZE"]["total_KB"]*1024
    k=k.split("(")[0]
    name=NAMES.get(k,k)
    out["bytes_per_launch"][name]=out["bytes_per_launch"].get(name,0)+(fetch+write)/n
    out["bytes_per_step"][name]=out["bytes_per_step"].get(name,0)+(fetch+write)
    print(k,"launches",n,"fetch GB (x2)",round(fetch/1e9,1),"write GB",round(write/1e9,1),"per launch MB",round((fetch+write)/n/1e6,1))
# the Z stage as one unit (bench.py's z_stage): generator + z FFT bytes per pass
zs=sum(out["bytes_per_step"].get(k,0) for k in ("k_gen","k_zfft"))
if zs: out["bytes_per_launch"]["z_stage"]=zs/line["config"]["passes"]; out["bytes_per_step"]["z_stage"]=zs
json.dump(out,open(f"{R}/gpurun_out/traffic_{TAG}.json","w"),indent=1)
PY
cat $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) | head -8 | cut -c1-200
cp gpurun_out/traffic_$TAG.json profiles/traffic_latest.json
timeout 900 python bench.py $ARGS 2>&1 | tail -1 > gpurun_out/bench_$TAG.json; python3 -c "
import json;d=json.load(open('gpurun_out/bench_$TAG.json'));print(d['value'],d['ms_per_step'],d['roofline_path_frac']);print(d['cpu_baseline']);print(d['roofline']);print(d['kernels_isolated'])"
# (everything stays under gpurun_out/, which gpurun merges back; the summaries are copied into profiles/ by hand afterwards)
i=0; for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | sort); do cp $f gpurun_out/${TAG}_kernel_stats$([ $i -gt 0 ] && echo _$i).csv; i=$((i+1)); done
