# usage: bash tools/gpu_t.sh <pytest args...>   (log -> gpurun_out/t.log)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest "$@" -m gpu -x -q -s -p no:hypothesis > gpurun_out/t.log 2>&1
rc=$?
echo rc=$rc
head -100 gpurun_out/t.log
if grep -q "Memory access fault" gpurun_out/t.log; then exit 1; fi
exit $rc
