# GPU tests with captured stdout (-s), optional -k expression; log in gpurun_out/ts.log
mkdir -p gpurun_out
timeout -k 10 ${2:-900} python -m pytest tests -m gpu -x -q -s -p no:hypothesis ${1:+-k "$1"} > gpurun_out/ts.log 2>&1
rc=$?
echo tests rc=$rc; grep -E "differ|worst|passed|failed|Error|error" gpurun_out/ts.log | tail -40
if grep -q "Memory access fault" gpurun_out/ts.log; then exit 1; fi
exit $rc
