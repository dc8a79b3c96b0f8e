mkdir -p gpurun_out/f16
( for n in 4 8; do
    echo "--- headline, $n ranks on one GPU"
    G4C_BENCH_SAME_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 10 --warmup 2 2>gpurun_out/f16/rank_err_$n.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); pc=d['partition_check']
print(d['value'], d['n_gpus'], pc['ok'], pc['max_abs_diff_vs_single_rank'], pc.get('capture'), pc.get('capture_note'), [ (r['owned_nodes'], r['compute_ms'], r['in_exchanges_ms']) for r in pc['per_rank']])"
    tail -3 gpurun_out/f16/rank_err_$n.log
  done
) > gpurun_out/f16/run11.log 2>&1
cat gpurun_out/f16/run11.log
