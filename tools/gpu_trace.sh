# kernel trace (per-dispatch start/end timestamps, queue ids) of a short pipelined bench run, for timeline analysis
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python3 $R/bench.py --steps ${1:-40} --warmup 10 --cpu-sample 0 --cpp-sample 0 --seqs 0 --sc-db 200 --prof-every 0 ${2:-} > $R/gpurun_out/trace.log 2>&1
echo trace rc=$?
tail -1 $R/gpurun_out/trace.log | cut -c1-300
ls -la $R/gpurun_out/trace/*/
