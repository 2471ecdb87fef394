#!/bin/bash
# dense ScanContext matrix on the matrix cores: parity test, issue-rate probe, bench line, rocprofv3 kernel statistics
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_voxel_sc_gpu.py -m gpu -x -q -p no:hypothesis -k "sc_" > gpurun_out/scm_t.log 2>&1
rc=$?
echo tests rc=$rc; tail -5 gpurun_out/scm_t.log
if grep -q "Memory access fault" gpurun_out/scm_t.log; then exit 1; fi
[ $rc -eq 0 ] || exit $rc
hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak && timeout -k 10 120 gpurun_out/mfma_f64_peak
timeout -k 10 900 python tools/bench_sc_matrix.py --db-size ${SCM_N:-5000} --probe gpurun_out/mfma_f64_peak > gpurun_out/scm_bench.log 2>&1
rc=$?
echo bench rc=$rc; tail -1 gpurun_out/scm_bench.log | cut -c1-1800
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/scm_prof
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/scm_prof -- python3 $GRAFT_REPO_ROOT/tools/bench_sc_matrix.py --db-size ${SCM_N:-5000} --steps 3 --cpu-pairs 200 > $GRAFT_REPO_ROOT/gpurun_out/scm_prof.log 2>&1
echo prof rc=$?
f=$(find $GRAFT_REPO_ROOT/gpurun_out/scm_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/scm_kernel_stats.csv && head -6 $f
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 tools/bench_sc_matrix.py --gpus 2 --db-size 2000 --steps 3 --backend gloo > gpurun_out/scm_n2.log 2>&1
echo n2 rc=$?; tail -1 gpurun_out/scm_n2.log | cut -c1-700
exit 0
