mkdir -p gpurun_out
for flags in "--no-overlap" "--side-thread 0" "--ring 2" ""; do
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-sample 0 $flags > gpurun_out/m.log 2>&1 || { echo "FAILED: $flags"; tail -5 gpurun_out/m.log; exit 1; }
tail -1 gpurun_out/m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$flags', round(d['value'],1), round(d['ms_per_step'],4), d['config']['schedule'][:20])"
done
