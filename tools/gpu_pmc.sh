# HBM traffic per dispatch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, kernel trace only) of the BENCHMARKED
# configuration of bench.py (scal_pipeline schedule, 5,000-keyframe database), aggregated per kernel -> gpurun_out/pmc_summary.json
# (copy to profiles/ after checking)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG="--steps 20 --warmup 5 --reps 1 --min-timed-s 0 --h2d 0 --cpu-sample 0 --cpp-sample 0 --seqs 0 --prof-every 0"
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py $CFG > $R/gpurun_out/pmc_fetch.log 2>&1
echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py $CFG > $R/gpurun_out/pmc_write.log 2>&1
echo write rc=$?
python3 - <<PY
import csv, glob, json, os, collections, re
R=os.environ['GRAFT_REPO_ROOT']
def short(n):
    m = re.search(r'(k_[a-z0-9_]+)', n)
    s = m.group(1) if m else n
    if s == 'k_lm_solve':
        s = 'k_lm_solve_map' if 'MapPoseDone' in n else 'k_lm_solve_odom'
    return s
agg={}
for tag in ('fetch','write'):
    a=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob(f'{R}/gpurun_out/pmc_{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            a[short(r['Kernel_Name'])][0]+=float(r['Counter_Value']); a[short(r['Kernel_Name'])][1]+=1
    agg[tag]=a
out={'config':'python3 bench.py $CFG (the benchmarked schedule and database; counter collection serialises the dispatches), FETCH_SIZE / WRITE_SIZE in KB per dispatch as rocprofv3 reports them; '
     'FETCH_SIZE raw (the guide\\'s x2 correction applies to 16-B-per-lane streaming reads; these kernels read 4-8 B per lane)','kernels':{}}
for k in sorted(set(agg['fetch'])|set(agg['write'])):
    f,w=agg['fetch'].get(k,[0,0]),agg['write'].get(k,[0,0])
    out['kernels'][k]={'dispatches':max(f[1],w[1]),'fetch_kb_per_dispatch':f[0]/max(1,f[1]),'write_kb_per_dispatch':w[0]/max(1,w[1])}
json.dump(out, open(f'{R}/gpurun_out/pmc_summary.json','w'), indent=1)
print(len(out['kernels']),'kernels')
PY
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
