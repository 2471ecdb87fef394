mkdir -p gpurun_out
for args in "--prof-every 8" "--prof-every 0" "--prof-every 8 --ahead 6" "--prof-every 0 --ahead 6"; do
  timeout -k 10 600 python3 bench.py --h2d 0 --cpu-sample 0 --cpp-sample 0 --seqs 0 $args > gpurun_out/try.json 2> gpurun_out/try.err || { echo failed; tail -5 gpurun_out/try.err; exit 1; }
  python3 -c "
import json; r=json.load(open('gpurun_out/try.json')); print('$args:', round(r['value'],1), sorted(round(x,3) for x in r['rep_ms_per_step']))"
done
