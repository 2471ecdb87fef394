#!/bin/bash
# A/B of a HIP runtime environment variable on the default bench (same box, alternating runs)
mkdir -p gpurun_out
for i in 1 2; do
  python bench.py --cpu-sample 0 > gpurun_out/env_base_$i.log 2>&1 || exit 1
  env "$1" python bench.py --cpu-sample 0 > gpurun_out/env_try_$i.log 2>&1 || exit 1
done
for f in gpurun_out/env_base_1.log gpurun_out/env_try_1.log gpurun_out/env_base_2.log gpurun_out/env_try_2.log; do
  echo $f $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['final_map_pose']['t'][0])")
done
