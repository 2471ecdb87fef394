mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --sc-db 200 --no-overlap --prof-every 0 > $R/gpurun_out/pmc_fetch.log 2>&1
echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --sc-db 200 --no-overlap --prof-every 0 > $R/gpurun_out/pmc_write.log 2>&1
echo write rc=$?
ls $R/gpurun_out/pmc_fetch/* | head; 
# keep only the per-kernel aggregates (the raw counter CSVs are large)
python3 - <<'PY'
import csv, glob, json, os, collections
R=os.environ['GRAFT_REPO_ROOT']
out={}
for tag in ('fetch','write'):
    fs=glob.glob(f'{R}/gpurun_out/pmc_{tag}/*/*counter_collection.csv')
    agg=collections.defaultdict(lambda:[0.0,0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][0]+=float(r['Counter_Value']); agg[r['Kernel_Name']][1]+=1
    out[tag]={k:{'sum':v[0],'dispatches':v[1]} for k,v in agg.items()}
json.dump(out, open(f'{R}/gpurun_out/pmc_summary.json','w'), indent=1)
print({k:len(v) for k,v in out.items()})
PY
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
