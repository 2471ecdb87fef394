mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_mapping_gpu.py -m gpu -x -q -s -p no:hypothesis > gpurun_out/map.log 2>&1
rc=$?
echo rc=$rc
head -80 gpurun_out/map.log
if grep -q "Memory access fault" gpurun_out/map.log; then exit 1; fi
exit $rc
