#!/bin/bash
# GPU session script (one parametrised script instead of a file per run).   tools/gpu_session.sh <tag> <what...>
#   what: new   -- the tests round 6 added / touched        full  -- the whole GPU suite + smoke + the default bench line
#         train -- tools/bench_train.py (bf16 + fp32 step)  bench -- the default bench line only     k=<expr> -- pytest -k <expr>
TAG=$1; shift
mkdir -p gpurun_out
cd /root/repo
for what in "$@"; do
  case $what in
    new)
      ( time timeout 1500 python -m pytest tests/test_gpu_train_cfg5.py tests/test_gpu_convergence.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 ) 2>&1 | tee gpurun_out/pytest_new_$TAG.txt
      ( time timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee -a gpurun_out/pytest_new_$TAG.txt
      ( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "busy or resident or taken" 2>&1 | tail -12 ) 2>&1 | tee -a gpurun_out/pytest_new_$TAG.txt ;;
    full) bash tools/gpu_validate.sh $TAG ;;
    train) ( timeout 600 python tools/bench_train.py 32 16000 --no-torch --reps=5 2>&1 | tail -6 ) | tee gpurun_out/train_$TAG.txt ;;
    traindet) ( WN_DETERMINISTIC=1 timeout 600 python tools/bench_train.py 32 16000 --no-torch --reps=5 2>&1 | tail -6 ) | tee gpurun_out/traindet_$TAG.txt ;;
    cfg5) ( time timeout 1500 python -m pytest tests/test_gpu_train_cfg5.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 ) 2>&1 | tee gpurun_out/pytest_cfg5_$TAG.txt ;;
    bench) ( time timeout 900 python bench.py 2> gpurun_out/bench_$TAG.err | tail -1 > gpurun_out/bench_$TAG.json ) 2>&1 | tail -3; tail -3 gpurun_out/bench_$TAG.err ;;
    k=*) ( time timeout 1500 python -m pytest tests -m gpu -q -x -s -k "${what#k=}" 2>&1 | tail -30 ) 2>&1 | tee gpurun_out/pytest_k_$TAG.txt ;;
    *) echo "unknown: $what" ;;
  esac
done
