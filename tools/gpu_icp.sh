#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voxel_sc_gpu.py -m gpu -x -q -p no:hypothesis -k "icp or mfma or voxel or reject" > gpurun_out/icp_t.log 2>&1
rc=$?
echo tests rc=$rc; tail -15 gpurun_out/icp_t.log
if grep -q "Memory access fault" gpurun_out/icp_t.log; then exit 1; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/bench_icp.py > gpurun_out/icp_bench.log 2>&1
rc=$?
echo bench rc=$rc; tail -1 gpurun_out/icp_bench.log | cut -c1-2500
exit $rc
