mkdir -p gpurun_out
for pe in 0 1 8; do
python bench.py --steps 60 --warmup 10 --cpu-sample 0 --prof-every $pe > gpurun_out/ab_$pe.log 2>&1 || exit 1
tail -1 gpurun_out/ab_$pe.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$pe', d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['kernel'])"
done
