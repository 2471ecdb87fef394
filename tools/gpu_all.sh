mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:hypothesis > gpurun_out/t.log 2>&1
rc=$?
echo tests rc=$rc; tail -6 gpurun_out/t.log
if grep -q "Memory access fault" gpurun_out/t.log; then exit 1; fi
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 60 --warmup 10 --cpu-sample 0 > gpurun_out/bench.log 2>&1
rc=$?
echo bench rc=$rc; tail -1 gpurun_out/bench.log | cut -c1-1500
if grep -q "Memory access fault" gpurun_out/bench.log; then exit 1; fi
exit $rc
