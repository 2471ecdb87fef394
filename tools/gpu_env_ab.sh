# A/B of one environment switch on ONE box (alternating runs): bash tools/gpu_env_ab.sh NAME=VALUE [runs]
mkdir -p gpurun_out
for i in $(seq 1 ${2:-3}); do
  for E in "SCALOAM_NONE=1" "$1"; do
    v=$(env $E python bench.py --steps 100 --warmup 30 --cpu-sample 0 --cpp-sample 0 --seqs 0 --prof-every 0 --h2d 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), [round(x,3) for x in d['rep_ms_per_step']])")
    echo "$E $v"
  done
done
