# the pipelined schedule twice and the serial schedule once over the same 330 scans: final map pose and loop count must be identical
mkdir -p gpurun_out
C="--steps 300 --warmup 30 --reps 1 --min-timed-s 0 --h2d 0 --cpu-sample 0 --cpp-sample 0 --seqs 0"
python bench.py $C > gpurun_out/det_a.json 2> gpurun_out/det_a.err || { tail -3 gpurun_out/det_a.err; exit 1; }
python bench.py $C > gpurun_out/det_b.json 2> gpurun_out/det_b.err || { tail -3 gpurun_out/det_b.err; exit 1; }
python bench.py $C --no-overlap > gpurun_out/det_c.json 2> gpurun_out/det_c.err || { tail -3 gpurun_out/det_c.err; exit 1; }
python - <<'PY'
import json
r=[json.load(open(f'gpurun_out/det_{x}.json')) for x in 'abc']
for x in r: print(round(x['value'],1), x['loops_detected'], x['final_map_pose'])
ok = all(x['final_map_pose']==r[0]['final_map_pose'] and x['loops_detected']==r[0]['loops_detected'] for x in r)
print('identical:', ok)
raise SystemExit(0 if ok else 1)
PY
