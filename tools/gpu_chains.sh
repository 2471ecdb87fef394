# chain latencies of the four streams (HIP events on the first / last kernel of every chain, unprofiled pipelined run) and the host
# time the pipeline's three threads spend inside each stage call
mkdir -p gpurun_out
K=k_pre,k_compact,k_map_gather,k_vox_small.C,k_vox_small_reduce.C,k_vox_keys.C+D,k_vox_reduce.C+D,k_sc_bin,k_sc_detect,k_odom_gather,k_odom_cellfill,k_assoc_knn.0,k_map_end
SCALOAM_PIPE_TIMING=1 python bench.py --steps 100 --warmup 30 --reps 1 --min-timed-s 0 --h2d 0 --cpu-sample 0 --cpp-sample 0 --timeline gpurun_out/tl.csv --timeline-kernels $K "$@" > gpurun_out/tl.json 2> gpurun_out/tl.err
grep scal_pipeline gpurun_out/tl.err
python tools/chain_latency.py gpurun_out/tl.csv A:k_pre:k_compact A+corner:k_pre:k_vox_small_reduce.C filters:k_vox_keys.C+D:k_vox_reduce.C+D sc_search:k_sc_bin:k_sc_detect side:k_vox_keys.C+D:k_sc_detect B:k_odom_gather:k_odom_cellfill C:k_assoc_knn.0:k_map_end
python -c "import json; d=json.loads(open('gpurun_out/tl.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['rep_ms_per_step'])"
