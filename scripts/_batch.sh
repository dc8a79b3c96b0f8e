cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/b12
timeout 1200 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/b12/pytest_train.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or remus_helpers or remus_model" 2>&1 | tail -5 > gpurun_out/b12/pytest_c5.log
timeout -k 10 900 python scripts/bench_train.py --steps 10 --cpu-steps 0 --phases 2> gpurun_out/b12/train_stderr.log | tail -1 > gpurun_out/b12/train_bench_100k.json
timeout 900 python bench.py > gpurun_out/b12/bench_stdout.log 2> gpurun_out/b12/bench_stderr.log; tail -1 gpurun_out/b12/bench_stdout.log > gpurun_out/b12/bench_n1.json
