# smoke() and a two-rank rehearsal of bench.py's N > 1 path (gloo, both ranks share the one GPU): per-scan exchange, the batched
# exchange (--sc-exchange-every 4) - same loop answers - and the serial schedule
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; echo smoke rc=$rc; tail -2 gpurun_out/smoke.log
grep -q "Memory access fault" gpurun_out/smoke.log && exit 1
[ $rc -eq 0 ] || exit $rc
run() {
  tag=$1; shift
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 12 --warmup 3 --cpu-sample 0 --sc-db 300 --backend gloo --reps 2 --min-timed-s 0 "$@" > gpurun_out/n2_$tag.log 2>&1; rc=$?
  echo "n2 $tag rc=$rc"
  grep -q "Memory access fault" gpurun_out/n2_$tag.log && exit 1
  [ $rc -eq 0 ] || { tail -12 gpurun_out/n2_$tag.log; exit $rc; }
  tail -1 gpurun_out/n2_$tag.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('  ', round(r['value'],1), 'scans/s (2 ranks on one GPU), loops', r['loops_detected'], r['config']['parallelism'], r['final_map_pose']['t'])"
}
run q1 --sc-exchange-every 1
run q4 --sc-exchange-every 4
run serial --no-overlap
