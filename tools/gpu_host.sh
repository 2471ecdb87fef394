mkdir -p gpurun_out
python bench.py --steps 60 --warmup 10 --cpu-sample 0 --prof-every 0 --host-timing > gpurun_out/host.log 2>&1 || exit 1
tail -1 gpurun_out/host.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['host_us_per_step'])"
