mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 8 --warmup 2 --cpu-sample 0 --sc-db 300 --backend gloo --no-overlap > gpurun_out/n2s.log 2>&1; rc=$?
echo n2 serial rc=$rc; tail -1 gpurun_out/n2s.log | cut -c1-200
exit $rc
