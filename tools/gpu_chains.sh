# chain latencies of the four streams (HIP events on the first / last kernel of every chain, unprofiled pipelined run)
mkdir -p gpurun_out
K=k_pre,k_vox_reduce,k_vox_small,k_sc_detect,k_odom_gather,k_odom_handover,k_map_begin,k_map_end,k_sc_bin,k_map_gather,k_compact
python bench.py --steps 100 --warmup 30 --reps 1 --h2d 0 --cpu-sample 0 --timeline gpurun_out/tl.csv --timeline-kernels $K > gpurun_out/tl.json 2> gpurun_out/tl.err
python tools/chain_latency.py gpurun_out/tl.csv A:k_pre:k_compact Asurf:k_pre:k_vox_reduce gatherfilt:k_map_gather:k_vox_reduce side_corner:k_vox_small:k_vox_small side_sc:k_sc_bin:k_sc_detect side_all:k_vox_small:k_sc_detect B:k_odom_gather:k_odom_handover C:k_map_begin:k_map_end
python -c "import json; d=json.loads(open('gpurun_out/tl.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
