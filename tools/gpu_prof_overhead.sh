# does the timing inside the timed region (the two LM solve kernels on every step) cost throughput?  alternating runs on ONE box
for i in 1 2 3; do
  for PE in 0 8; do
    v=$(python bench.py --cpu-sample 0 --cpp-sample 0 --seqs 0 --h2d 0 --prof-every $PE 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), [round(x,3) for x in d['rep_ms_per_step']])")
    echo "prof-every=$PE $v"
  done
done
