# round 3: every artefact that goes to profiles/, from ONE box: default and driver-style bench lines, rocprofv3 kernel statistics
# of the default command, chain latencies, PMC traffic of the benchmarked configuration, batched scaling, config 3
mkdir -p gpurun_out/final
F=gpurun_out/final
timeout -k 10 600 python3 bench.py > $F/bench_default.json 2> $F/bench_default.err || { echo "default bench failed"; tail -5 $F/bench_default.err; exit 1; }
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $F/bench_20.json 2> $F/bench_20.err || { echo "bench 20 failed"; exit 1; }
for s in 2 3; do
  timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --reps 3 --min-timed-s 0 --h2d 0 --cpu-sample 0 --cpp-sample 0 --seqs $s > $F/bench_seqs$s.json 2> $F/bench_seqs$s.err || { echo "seqs $s failed"; exit 1; }
done
timeout -k 10 600 python3 bench.py --config 3 --steps 200 --warmup 30 --cpu-sample 30 > $F/bench_config3.json 2> $F/bench_config3.err || { echo "config 3 failed"; tail -5 $F/bench_config3.err; exit 1; }
python3 - <<'PY'
import json
F='gpurun_out/final/'
for f in ('bench_default','bench_20'):
    r=json.load(open(F+f+'.json'))
    b=r['batched']
    print(f, round(r['value'],1), 'scans/s', round(r['ms_per_step'],4), 'ms; best', round(min(r['rep_ms_per_step']),3), 'reps', r['repetitions'], 'loops', r['loops_detected'],
          '| h2d', round(r['h2d_inclusive']['value'],1), '| as_integrated', round(r['as_integrated']['value'],1), r['as_integrated']['latency_ms_p50'],
          '| cpu', round(r['cpu_baseline']['value'],2), round(r['cpu_baseline']['pipelined_scans_per_s'],2),
          '| batched4', round(b['scans_per_s'],1), round(b['speedup_vs_single_sequence'],2), '| frac', r['roofline']['frac'], b['roofline']['frac'])
    print('   cpp', {k:v.get('scans_per_s') for k,v in r['cpp_host'].items()})
    print('   stages', {k:(round(v['kernel_ms_per_scan'],3), round(v['frac'],5)) for k,v in r['roofline']['stages'].items()})
for s in (2,3):
    r=json.load(open(F+f'bench_seqs{s}.json')); b=r['batched']
    print('seqs', s, round(b['scans_per_s'],1), round(b['speedup_vs_single_sequence'],2), 'single', round(r['value'],1))
r=json.load(open(F+'bench_config3.json'))
print('config3', round(r['value'],1), r['roofline'] and r['roofline']['frac'], r['cpu_baseline'] and round(r['cpu_baseline']['value'],2), r['keyframes'], r['loops_detected'])
PY
bash tools/gpu_chains.sh --seqs 0 > $F/chains.txt 2>&1; tail -9 $F/chains.txt
bash tools/gpu_pmc.sh > $F/pmc.log 2>&1; cp gpurun_out/pmc_summary.json $F/pmc_fetch_write.json; tail -2 $F/pmc.log
bash tools/gpu_prof_default.sh --cpu-sample 0 --cpp-sample 0 > $F/prof.log 2>&1; cp $(ls -S gpurun_out/prof_default/*/*kernel_stats.csv | head -1) $F/bench_kernel_stats.csv; grep "^{\"metric\"" gpurun_out/prof_default.log > $F/bench_under_rocprof.json; tail -3 $F/prof.log
