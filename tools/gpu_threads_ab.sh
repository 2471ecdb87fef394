# one or two side host threads, alternating runs on ONE box
mkdir -p gpurun_out
for i in 1 2 3; do
 for D in ${DELAYS:-0 3}; do
  for T in ${THREADS:-2 3}; do
    v=$(SCALOAM_HOST_DELAY_US=$D python bench.py --steps 100 --warmup 30 --cpu-sample 0 --prof-every 0 --h2d 0 --side-thread $T 2>gpurun_out/thr.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), [round(x,3) for x in d['rep_ms_per_step']], d['final_map_pose']['t'][0])")
    echo "delay=$D side-thread=$T $v"
  done
 done
done
