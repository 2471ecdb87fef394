#!/bin/bash
# PMC passes for the dense ScanContext matrix (k_sc_gram): matrix-core busy cycles + chip-active cycles, then HBM fetch and write
# sizes, each in its own rocprofv3 run (counters never combined with trace domains other than --kernel-trace).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in mfma fetch write; do
  case $tag in
    mfma) ctr="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    fetch) ctr="FETCH_SIZE";;
    write) ctr="WRITE_SIZE";;
  esac
  rm -rf $R/gpurun_out/pmcs_$tag
  timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$tag -- python3 $R/tools/bench_sc_matrix.py --steps 2 --warmup 1 --cpu-pairs 100 --slice 16 > $R/gpurun_out/pmcs_$tag.log 2>&1
  rc=$?
  echo $tag rc=$rc
  [ $rc -eq 0 ] || { tail -5 $R/gpurun_out/pmcs_$tag.log; exit $rc; }
done
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ['GRAFT_REPO_ROOT']
out = {}
for tag in ('mfma', 'fetch', 'write'):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f'{R}/gpurun_out/pmcs_{tag}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_sc_gram(' in r['Kernel_Name']:
                a = agg[r['Counter_Name']]
                a[0] += float(r['Counter_Value']); a[1] += 1
    for k, v in agg.items():
        out[k] = {'per_dispatch': v[0] / max(1, v[1]), 'dispatches': v[1]}
json.dump(out, open(f'{R}/gpurun_out/pmc_sc_matrix.json', 'w'), indent=1)
print(json.dumps(out))
PY
rm -rf $R/gpurun_out/pmcs_mfma $R/gpurun_out/pmcs_fetch $R/gpurun_out/pmcs_write
