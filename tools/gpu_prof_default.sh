# rocprofv3 kernel statistics of the DEFAULT bench command (the one the driver runs at N=1), for profiles/
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py ${@} > $GRAFT_REPO_ROOT/gpurun_out/prof_default.log 2>&1
echo prof rc=$?
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_default.log | cut -c1-200
ls $GRAFT_REPO_ROOT/gpurun_out/prof_default/*/
