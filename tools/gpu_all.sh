mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:hypothesis > gpurun_out/t.log 2>&1
rc=$?
echo tests rc=$rc; tail -6 gpurun_out/t.log
if grep -q "Memory access fault" gpurun_out/t.log; then exit 1; fi
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 60 --warmup 10 --cpu-sample 0 --cpp-sample 0 --seqs 0 > gpurun_out/bench.log 2>&1
rc=$?
echo bench rc=$rc; tail -1 gpurun_out/bench.log | cut -c1-300; tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step'], d['roofline'])"
if grep -q "Memory access fault" gpurun_out/bench.log; then exit 1; fi
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --cpu-sample 0 --cpp-sample 0 --seqs 0 --sc-db 500 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
echo prof rc=$?
