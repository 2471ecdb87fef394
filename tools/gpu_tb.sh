# GPU tests, then the driver-style and default bench lines (short summary)
bash tools/gpu_test.sh "" 900 || exit 1
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b20.json 2> gpurun_out/b20.err || { echo "bench 20 failed"; tail -20 gpurun_out/b20.err; exit 1; }
timeout -k 10 600 python3 bench.py --gpus 1 --cpu-sample 0 > gpurun_out/b100.json 2> gpurun_out/b100.err || { echo "bench 100 failed"; tail -20 gpurun_out/b100.err; exit 1; }
python3 - <<'PY'
import json
for f in ('b20','b100'):
    r=json.load(open(f'gpurun_out/{f}.json'))
    print(f, round(r['value'],1), 'scans/s', round(r['ms_per_step'],4), 'ms; reps', r['repetitions'], 'best', round(min(r['rep_ms_per_step']),3), 'loops', r['loops_detected'], 'h2d', r['h2d_inclusive'] and round(r['h2d_inclusive']['value'],1))
    print('   cpp', {k:(v.get('scans_per_s'), v.get('error')) for k,v in (r['cpp_host'] or {}).items()})
    print('   pose', r['final_map_pose']['t'])
PY
