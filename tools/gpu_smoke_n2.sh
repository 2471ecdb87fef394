mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; echo smoke rc=$rc; tail -2 gpurun_out/smoke.log
grep -q "Memory access fault" gpurun_out/smoke.log && exit 1
[ $rc -eq 0 ] || exit $rc
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 12 --warmup 3 --cpu-sample 0 --sc-db 300 --backend gloo --reps 2 > gpurun_out/n2.log 2>&1; rc=$?
echo n2 rc=$rc; tail -3 gpurun_out/n2.log | cut -c1-600
grep -q "Memory access fault" gpurun_out/n2.log && exit 1
exit $rc
