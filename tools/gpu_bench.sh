mkdir -p gpurun_out
python bench.py --steps 30 --warmup 5 --cpu-sample 20 > gpurun_out/bench.log 2>&1
rc=$?
echo rc=$rc; tail -5 gpurun_out/bench.log
if grep -q "Memory access fault" gpurun_out/bench.log; then exit 1; fi
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --cpu-sample 0 --cpp-sample 0 --seqs 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo prof rc=$?
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -20
