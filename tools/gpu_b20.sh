# driver-style bench line, short summary
mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > gpurun_out/b20.json 2> gpurun_out/b20.err || { echo "bench 20 failed"; tail -20 gpurun_out/b20.err; exit 1; }
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/b20.json'))
print('b20', round(r['value'],1), 'scans/s', round(r['ms_per_step'],4), 'ms; reps', r['repetitions'], 'best', round(min(r['rep_ms_per_step']),3), 'loops', r['loops_detected'], 'h2d', r['h2d_inclusive'] and round(r['h2d_inclusive']['value'],1))
print('   cpp', {k:(v.get('scans_per_s'), v.get('error')) for k,v in (r['cpp_host'] or {}).items()})
b=r.get('batched')
if b: print('   batched', b['seqs'], round(b['scans_per_s'],1), 'scans/s speedup', round(b['speedup_vs_single_sequence'],2), [round(x,3) for x in b['rep_ms_per_step']], b.get('roofline',{}).get('frac'), b.get('roofline',{}).get('avg_launch_us'))
print('   roofline', r['roofline']['frac'], r['roofline']['avg_launch_us'])
PY
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/b20.json'))
b=r.get('batched') or {}
t=b.get('kernel_ms_per_step_all_seqs') or {}
s=r['kernel_ms_per_step']
tot=0
for k,v in sorted(t.items(), key=lambda x:-x[1])[:40]:
    tot+=v
    print(f"   {k:24s}{v*1000:8.1f} us batched   single {s.get(k,0)*1000:7.1f}")
print('   total batched per super-step', round(sum(t.values()),3), 'ms; single per step', round(sum(s.values()),3))
PY
