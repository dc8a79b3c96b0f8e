timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bx6i or distributed or overlapped or headline" 2>&1 | tail -4
run() { env "$@" timeout 300 python bench.py $WL --no-cpu-baseline --no-roofline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['outputs_finite'])"; }
for WL in "--workload headline" "--nodes 50000" "--nodes 25000"; do
  echo "=== $WL"
  for rep in 1 2; do
    echo -n "default (bx6i from 100k rows): "; run A=1
    echo -n "bx6i from 400k rows: "; run G4C_BX6I_MIN_ROWS=400000
    echo -n "bx6i from 50k rows: "; run G4C_BX6I_MIN_ROWS=50000
  done
done
for n in 2 4; do
  echo "--- headline, $n ranks on one GPU (functional + per-rank compute ms)"
  for mr in 100000 400000; do
  G4C_BX6I_MIN_ROWS=$mr G4C_BENCH_SAME_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); pc=d['partition_check']
print('min_rows', $mr, d['value'], pc['ok'], pc['max_abs_diff_vs_single_rank'], [ (r['owned_nodes'], r['compute_ms']) for r in pc['per_rank']])"
  done
done
