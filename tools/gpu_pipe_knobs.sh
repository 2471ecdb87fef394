# C++ host, pipeline mode, 260 scans resident: development knobs
mkdir -p gpurun_out
python3 - <<'PY'
import sys, os
sys.path[:0]=['sc-a-loam_amd/python','tools/synth']
import scansynth
from scaloam import formats
w=scansynth.World(scansynth.HDL64,205,threads=16)
formats.write_scan_stream('/tmp/s260.bin', [w.scan(k) for k in range(260)])
PY
run() { a=$1; shift; env "$@" ./sc-a-loam_amd/bin/replay_main --scans /tmp/s260.bin --mode pipeline --resident 1 --warmup 30 --sc-db 5000 --ahead $a 2> gpurun_out/knob.err | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('ahead $a $*:', round(r['scans_per_s'],1), 'scans/s', r['latency_ms'])"; }
run 4 X=1; run 4 X=1
run 4 HIP_FORCE_DEV_KERNARG=1; run 4 HIP_FORCE_DEV_KERNARG=1
run 4 HIP_FORCE_DEV_KERNARG=0; run 4 HIP_FORCE_DEV_KERNARG=0
