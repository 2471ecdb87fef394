# GPU tests only (optionally a -k expression), log in gpurun_out/t.log
mkdir -p gpurun_out
timeout -k 10 ${2:-900} python -m pytest tests -m gpu -x -q -p no:hypothesis ${1:+-k "$1"} > gpurun_out/t.log 2>&1
rc=$?
echo tests rc=$rc; tail -25 gpurun_out/t.log
if grep -q "Memory access fault" gpurun_out/t.log; then exit 1; fi
exit $rc
