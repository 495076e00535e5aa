mkdir -p gpurun_out
( timeout 500 python -m pytest tests/test_gpu_training.py tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -12 ) | tee gpurun_out/tr_tests.txt
( timeout 200 python tools/bench_train.py 32 16000 --no-torch 2>&1 | grep -v amdgpu | tail -3 ) | tee gpurun_out/tr_bench.txt
bash tools/profile_train.sh dfg16 > /dev/null
