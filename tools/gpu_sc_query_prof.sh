# config #4 on one GPU: the full 5,000-keyframe run (JSON) and rocprofv3 kernel statistics of the same command
mkdir -p gpurun_out
bash tools/gpu_sc_query.sh || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sc_query
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sc_query -- python3 $GRAFT_REPO_ROOT/tools/bench_sc_query.py > $GRAFT_REPO_ROOT/gpurun_out/prof_sc_query.log 2>&1
echo prof rc=$?
ls $GRAFT_REPO_ROOT/gpurun_out/prof_sc_query/*/ | head
