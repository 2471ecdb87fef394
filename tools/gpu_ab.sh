# A/B of several builds of the library on ONE box (alternating runs): paths relative to the repo root
mkdir -p gpurun_out
LIBS=${@:-sc-a-loam_amd/lib/alt/A.so sc-a-loam_amd/lib/libscaloam_hip.so}
for i in 1 2 3; do
  for L in $LIBS; do
    v=$(SCALOAM_LIB=$PWD/$L python bench.py --steps 100 --warmup 30 --cpu-sample 0 --cpp-sample 0 --seqs 0 --prof-every 0 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
    echo "$L $v"
  done
done
