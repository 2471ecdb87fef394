#!/bin/bash
# Several independent sequences in flight on ONE GPU ("report batched/streamed numbers separately"): P copies of
# bench.py started together, each its own process, context and streams.  The aggregate is only meaningful where the timed
# regions overlap, so each copy prints its wall-clock window and the script reports the sum of the per-copy rates.
P=${1:-2}
mkdir -p gpurun_out
SYNC=$(mktemp -d)
pids=""
for i in $(seq 1 $P); do
  timeout -k 10 500 python bench.py --steps 300 --warmup 30 --cpu-sample 0 --sync-dir $SYNC --sync-n $P > gpurun_out/multi_$i.log 2>&1 &
  pids="$pids $!"
done
rc=0
for p in $pids; do wait $p || rc=$?; done
echo rc=$rc
python3 - $P <<'PY'
import json, sys
P = int(sys.argv[1])
rows = []
for i in range(1, P + 1):
    line = open(f"gpurun_out/multi_{i}.log").read().strip().splitlines()[-1]
    d = json.loads(line)
    rows.append({"value": d["value"], "ms_per_step": d["ms_per_step"], "window": d.get("timed_window_unix")})
t0 = max(r["window"][0] for r in rows); t1 = min(r["window"][1] for r in rows)
span = max(r["window"][1] for r in rows) - min(r["window"][0] for r in rows)
out = {"processes": P, "per_process": rows, "sum_scans_per_s": sum(r["value"] for r in rows),
       "overlap_fraction_of_span": max(0.0, t1 - t0) / span if span > 0 else None}
print(json.dumps(out))
json.dump(out, open(f"gpurun_out/multi_seq_{P}.json", "w"), indent=1)
PY
exit $rc
