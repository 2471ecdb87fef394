mkdir -p gpurun_out
python tools/bench_mapmerge.py > gpurun_out/mapmerge.json 2> gpurun_out/mapmerge.err || { tail -5 gpurun_out/mapmerge.err; exit 1; }
tail -1 gpurun_out/mapmerge.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mm
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mm -- python3 $GRAFT_REPO_ROOT/tools/bench_mapmerge.py --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_mm.log 2>&1
echo prof rc=$?
