mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcm_fetch $R/gpurun_out/pmcm_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_fetch -- python3 $R/tools/bench_mapmerge.py --steps 3 --warmup 1 --cpu-frames 2 > $R/gpurun_out/pmcm_fetch.log 2>&1
echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_write -- python3 $R/tools/bench_mapmerge.py --steps 3 --warmup 1 --cpu-frames 2 > $R/gpurun_out/pmcm_write.log 2>&1
echo write rc=$?
python3 - <<'PY'
import csv, glob, json, os, collections
R=os.environ['GRAFT_REPO_ROOT']
out={}
for tag in ('fetch','write'):
    agg=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob(f'{R}/gpurun_out/pmcm_{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_mm_' in r['Kernel_Name']:
                agg[r['Kernel_Name'].split('(')[0]][0]+=float(r['Counter_Value']); agg[r['Kernel_Name'].split('(')[0]][1]+=1
    out[tag]={k:{'kb_per_dispatch':v[0]/v[1],'dispatches':v[1]} for k,v in agg.items()}
json.dump(out, open(f'{R}/gpurun_out/pmc_mapmerge.json','w'), indent=1)
print(json.dumps(out))
PY
rm -rf $R/gpurun_out/pmcm_fetch $R/gpurun_out/pmcm_write
