mkdir -p gpurun_out
python -m pytest tests/test_features_gpu.py tests/test_voxel_sc_gpu.py -m gpu -x -q -p no:hypothesis > gpurun_out/first.log 2>&1
echo rc=$?
head -60 gpurun_out/first.log
