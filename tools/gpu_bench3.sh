# round 3: driver-style bench (20 steps), the serial schedule for the pose check, then the GPU tests
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/b20.json 2> gpurun_out/b20.err || { echo "bench 20 failed"; tail -20 gpurun_out/b20.err; exit 1; }
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/b20.json'))
print('NEW 20-step:', round(r['value'],1), 'scans/s', round(r['ms_per_step'],4), 'ms; reps', r['repetitions'], 'timed total', round(r['timed_total_s'],3), 's; loops', r['loops_detected'])
print('  reps ms:', sorted(round(x,3) for x in r['rep_ms_per_step']))
print('  h2d', r['h2d_inclusive'] and round(r['h2d_inclusive']['value'],1), 'as_integrated', r['as_integrated'] and (round(r['as_integrated']['value'],1), r['as_integrated']['latency_ms_p50']))
print('  cpp', {k:(v.get('scans_per_s'), v.get('error')) for k,v in (r['cpp_host'] or {}).items()})
print('  cpu', r['cpu_baseline'] and round(r['cpu_baseline']['value'],2), 'gen', round(r['input_gen_s'],1), 'db', round(r['database_gen_s'],1))
print('  stages', {k:(round(v['kernel_ms_per_scan'],3), round(v['frac'],5)) for k,v in (r['roofline']['stages'] or {}).items()})
print('  roofline', r['roofline']['frac'], r['roofline']['avg_launch_us'])
print('  final pose', r['final_map_pose'])
PY
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-overlap --cpu-sample 0 --cpp-sample 0 --h2d 0 > gpurun_out/b20_serial.json 2> gpurun_out/b20_serial.err || { echo "serial failed"; tail -5 gpurun_out/b20_serial.err; }
python3 -c "
import json
r=json.load(open('gpurun_out/b20_serial.json')); print('SERIAL 20-step:', round(r['value'],1), r['final_map_pose'])"
timeout -k 10 600 python3 bench.py --gpus 1 > gpurun_out/b100.json 2> gpurun_out/b100.err || { echo "bench 100 failed"; tail -20 gpurun_out/b100.err; exit 1; }
python3 -c "
import json
r=json.load(open('gpurun_out/b100.json')); print('DEFAULT (100 steps):', round(r['value'],1), sorted(round(x,3) for x in r['rep_ms_per_step']), 'loops', r['loops_detected'])"
