mkdir -p gpurun_out
for flags in "--side-thread 1"; do
timeout -k 10 400 python bench.py --steps 100 --warmup 10 --cpu-sample 0 $flags --host-timing > gpurun_out/b2.log 2>&1 || { tail -5 gpurun_out/b2.log; exit 1; }
tail -1 gpurun_out/b2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$flags', d['value'], d['ms_per_step'], d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac'])); print(d['host_us_per_step']); print(d['kernel_ms_per_step'])"
done
