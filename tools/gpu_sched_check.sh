# the stage-pipelined and the serial schedule must end at the same pose (same scans, same arithmetic)
mkdir -p gpurun_out
python bench.py --steps 150 --warmup 5 --cpu-sample 0 --cpp-sample 0 --seqs 0 --prof-every 0 > gpurun_out/s1.log 2>&1 || exit 1
python bench.py --steps 150 --warmup 5 --cpu-sample 0 --cpp-sample 0 --seqs 0 --prof-every 0 --no-overlap > gpurun_out/s2.log 2>&1 || exit 1
python - <<'PY'
import json
a=json.loads(open('gpurun_out/s1.log').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/s2.log').read().strip().splitlines()[-1])
print(a['value'], b['value'])
print(a['final_map_pose']); print(b['final_map_pose'])
print('identical:', a['final_map_pose']==b['final_map_pose'])
PY
