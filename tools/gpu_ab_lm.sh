# A/B on one box: bench value and the event-timed average of the two LM solve kernels
for i in 1 2; do
  for L in ${@:-sc-a-loam_amd/lib/alt/A.so sc-a-loam_amd/lib/libscaloam_hip.so}; do
    SCALOAM_LIB=$PWD/$L python bench.py --steps 100 --warmup 30 --reps 2 --h2d 0 --cpu-sample 0 --cpp-sample 0 --seqs 0 --prof-every 4 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', round(d['value']), 'lm_map us', round(r['avg_launch_us'],1), 'lm_odom us', round(r['stage_b']['avg_launch_us'],1), {k: round(v*1000) for k,v in d['kernel_ms_per_step'].items() if v>0.02})"
  done
done
