#!/bin/bash
# rehearsal of the N=2 sharded VoxelGrid with gloo ranks sharing the one GPU (the RCCL path needs a multi-GPU node)
mkdir -p gpurun_out
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29651 tools/bench_sharded_downsample.py --frames 64 --backend gloo > gpurun_out/shard_ds.log 2>&1
rc=$?
echo rc=$rc; tail -3 gpurun_out/shard_ds.log | cut -c1-900
exit $rc
