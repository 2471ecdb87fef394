# throughput against the number of scans pushed ahead of the awaited pose and the stage-C steps queued on the device
mkdir -p gpurun_out
for cfg in "--ahead 4 --depth 2" "--ahead 6 --depth 2" "--ahead 8 --depth 2" "--ahead 8 --depth 3" "--ahead 3 --depth 2"; do
  python bench.py --steps 100 --warmup 30 --reps 4 --min-timed-s 0 --h2d 0 --cpu-sample 0 --cpp-sample 0 --seqs 0 $cfg > gpurun_out/ah.json 2> gpurun_out/ah.err || { tail -3 gpurun_out/ah.err; exit 1; }
  python -c "
import json; r=json.load(open('gpurun_out/ah.json')); print('$cfg', round(r['value'],1), [round(x,3) for x in r['rep_ms_per_step']])"
done
