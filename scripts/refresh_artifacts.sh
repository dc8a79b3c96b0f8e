#!/bin/bash
# Round artifacts, produced on the GPU box into gpurun_out/artifacts_<tag>/ (copy them into profiles/ afterwards):
#   pytest -m gpu tail, bench lines of every named workload, rocprofv3 kernel stats of the headline and REMuS benches, PMC HBM
#   traffic of both, MFMA ceiling, training benches.
# Usage: gpurun --timeout 3000 -- 'bash scripts/refresh_artifacts.sh r06'
TAG=${1:-r06}
cd "$GRAFT_REPO_ROOT"
A=gpurun_out/artifacts_$TAG; rm -rf $A; mkdir -p $A
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $A/${TAG}_pytest_gpu.log
# PMC first: bench.py's roofline block quotes the newest profiles/*_pmc_traffic*.json
timeout 1200 bash scripts/collect_pmc_traffic.sh $TAG > $A/pmc.log 2>&1; cp profiles/${TAG}_pmc_traffic.json $A/ 2>/dev/null
timeout 1200 bash scripts/collect_pmc_traffic.sh $TAG c3 > $A/pmc_c3.log 2>&1; cp profiles/${TAG}_pmc_traffic_c3.json $A/ 2>/dev/null
# pipe-utilisation counters of the shipped kernels in the DEFAULT arithmetic, BEFORE the bench (bench.py imports the newest
# profiles/*_pmc_mlp_ws.txt / *_pmc_mlp_bx6_node.txt and says whether they were collected from the kernel sources it runs): level-1 message
# launch (mlp_ws_kernel) and node launch (mlp_bx6_kernel)
bash scripts/pmc_ws.sh ws ${TAG}_ws util sq3 lds sq2 tcc mix mix2 coexec > profiles/${TAG}_pmc_mlp_ws.txt 2>&1; cp profiles/${TAG}_pmc_mlp_ws.txt $A/
PMC_EXTRA_ARGS=--node bash scripts/pmc_ws.sh tile ${TAG}_node util sq3 lds sq2 tcc mix mix2 coexec > profiles/${TAG}_pmc_mlp_bx6_node.txt 2>&1; cp profiles/${TAG}_pmc_mlp_bx6_node.txt $A/
timeout 900 python bench.py > $A/bench_stdout.log 2> $A/bench_stderr.log; tail -1 $A/bench_stdout.log > $A/${TAG}_bench_n1.json
for wl in c2 c3 c5-1gpu; do
  timeout 900 python bench.py --workload $wl 2> $A/bench_${wl}_stderr.log | tail -1 > $A/${TAG}_bench_${wl}.json
done
for wl in headline c3; do
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $A/prof_$wl -o p -- \
      python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-strict-range --no-side-configs > $A/prof_${wl}_stdout.log 2> $A/prof_${wl}_stderr.log )
  tail -1 $A/prof_${wl}_stdout.log > $A/${TAG}_bench_${wl}_under_rocprofv3.json
  cp $(find $A/prof_$wl -name '*kernel_stats.csv' | head -1) $A/${TAG}_rocprofv3_kernel_stats_${wl}.csv
  rm -rf $A/prof_$wl
done
hipcc -w --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/micro/mfma_peak.hip && /tmp/mfma_peak 2000 > $A/${TAG}_mfma_peak.log 2>&1
{
  echo "# side configurations (bench.py --no-cpu-baseline --no-roofline), one JSON line each"
  timeout 300 python bench.py --nodes 12500 --steps 50 --no-cpu-baseline --no-roofline | tail -1
  timeout 300 python bench.py --model NsFourScaleGNN --nodes 100000 --steps 20 --no-cpu-baseline --no-roofline | tail -1
  timeout 300 python bench.py --model NsRotEquiTreeScaleGNN --nodes 100000 --steps 20 --no-cpu-baseline --no-roofline | tail -1
  timeout 300 python scripts/bench_mugs.py 2>&1 | tail -2
  timeout 300 python bench.py --precision fp32 --steps 20 --no-cpu-baseline | tail -1
  timeout 300 python bench.py --precision bf16x6 --steps 100 --no-cpu-baseline | tail -1
  timeout 300 python bench.py --precision bf16 --steps 50 --no-cpu-baseline --no-roofline | tail -1
} > $A/${TAG}_side_configs.log 2>&1
# functional check of the partitioned path on the one GPU of the box (both ranks on cuda:0, gloo transport): partition_check in the line
for wl in headline c3; do
  G4C_BENCH_SAME_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --workload $wl --steps 20 --warmup 3 2> $A/bench_${wl}_2rank_stderr.log | tail -1 > $A/${TAG}_bench_${wl}_2rank_samegpu.json
done
# the driver's command shape: bench.py --gpus N as a PLAIN command starts its own ranks
for n in 2 4; do
  G4C_BENCH_SAME_GPU=1 timeout 900 python bench.py --gpus $n --steps 5 --warmup 2 2> $A/bench_${n}rank_plain_stderr.log | tail -1 > $A/${TAG}_bench_headline_${n}rank_plain_cmd.json
done
bash scripts/size_sweep.sh > $A/${TAG}_size_sweep.log 2>&1
timeout 300 python scripts/step_breakdown.py > $A/${TAG}_step_breakdown.log 2>&1
timeout 300 python scripts/step_breakdown.py --workload c3 > $A/${TAG}_step_breakdown_c3.log 2>&1
timeout 300 python scripts/bench_segment_reduce.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_segment_reduce_rotating.log
timeout 300 python scripts/bench_pool_edge.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_pool_edge_reductions.log
timeout 300 python scripts/step_breakdown.py --nodes 12500 2>&1 | grep -v amdgpu.ids > $A/${TAG}_step_breakdown_12k5.log
timeout 300 python scripts/step_breakdown.py --workload c2 2>&1 | grep -v amdgpu.ids > $A/${TAG}_step_breakdown_c2.log
timeout 300 python scripts/mlp_accuracy.py > $A/${TAG}_mlp_accuracy.log 2>&1
# the weight-stationary kernel (f16x3 stream and rounded-bf16 mode, 2 / 3 layers): checks against the tile kernel + same-process A/B; the dual-tile kernel (bf16x6 stream)
timeout 600 python scripts/ws_check.py --time 2>&1 | grep -v "^ok\|amdgpu.ids" > $A/${TAG}_ws_check_and_ab.log
G4C_MLP_PRECISION=bf16x6 timeout 300 python scripts/bx6i_check.py --time 2>&1 | tail -3 > $A/${TAG}_bx6i_check_and_ab.log
# cycle stamps of the weight-stationary kernel's pair loop (DESIGN.md 4.1 / 9 quote them)
bash scripts/build_ws_timing.sh > /dev/null 2>&1 && timeout 120 python scripts/ws_stamps.py graphs4cfd_amd/lib/libg4c_ws_timing.so 2>&1 | grep -v amdgpu.ids > $A/${TAG}_ws_stamps.log
timeout 120 python scripts/ws_stamps.py graphs4cfd_amd/lib/libg4c_ws_timing.so bf16 2>&1 | grep -v amdgpu.ids > $A/${TAG}_ws_stamps_bf16_mode.log
# round 5: the fused MP layer and the node update's own kernel against the separate launches / the tile kernel
timeout 600 python scripts/mp_layer_check.py --time 2>&1 | grep -v "^ok\|amdgpu.ids" > $A/${TAG}_mp_layer_check_and_ab.log
{ for f in 0 1; do for wl in "--workload c2" "--nodes 12500 --steps 100 --no-side-configs"; do
    echo "G4C_FUSE_LAYER=$f bench.py $wl: $(G4C_FUSE_LAYER=$f timeout 300 python bench.py $wl --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['value'],1))") steps/s"
  done; done; } > $A/${TAG}_ab_fused_layer.log 2>&1
timeout 300 python scripts/remus_bf16_err.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_remus_bf16_products_err.log
# round 6: the row-split kernel of the rounded-bf16 mode against mlp_ws_kernel<SP = 1> (parity + time, every row format), sender locality
timeout 300 python scripts/rs1_check.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_rs1_check.log
timeout 200 python scripts/rs1_gather_locality.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_rs1_gather_locality.log
timeout 300 python scripts/rs2_check.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_rs2_check.log
# co-issue microbenchmarks (what hides behind one MFMA, by shape, waves per SIMD and instruction kind)
hipcc -w --offload-arch=gfx950 -O3 -o /tmp/mfma_fillers scripts/micro/mfma_fillers.hip 2>/dev/null && /tmp/mfma_fillers > $A/${TAG}_mfma_fillers.log 2>&1
for m in mfma_gap_patterns mfma_chain_probe mfma_lds_probe; do
  hipcc -w --offload-arch=gfx950 -O3 -o /tmp/$m scripts/micro/$m.hip 2>/dev/null && /tmp/$m > $A/${TAG}_$m.log 2>&1
done
# what a hipGraph-capture failure on the first real RCCL run would cost: captured against eager partitioned step, world size 1, collectives entered
{ for n in 12500 100000; do timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$((n % 7)) scripts/dist_check.py \
    --backend nccl --capture 1 --force-exchange --nodes $n --time 50 2>&1 | grep "world=\|eager fallback"; done; } > $A/${TAG}_dist_single.log
timeout 300 python scripts/bench_mugs.py 2>&1 | tail -3 > $A/${TAG}_bench_mugs.log
# training path: step time + per-phase HIP-event times + the CPU leg
timeout -k 10 900 python scripts/bench_train.py --steps 10 --cpu-steps 1 --phases 2> $A/train_stderr.log | tail -1 > $A/${TAG}_train_bench_100k.json
# small launches: the tile kernel's deep-ring limit over four meshes (same process, bit-identity checked), the message launch over the
# row count on the three kernels, per-phase stamps of a one-tile launch, idle time between the kernels of a replayed step
timeout 600 python scripts/ab_small_launch.py --limits 0,512 2>&1 | grep -v amdgpu.ids > $A/${TAG}_ab_small_launch.log
timeout 300 python scripts/sweep_message_launch.py 2>&1 | grep -v amdgpu.ids > $A/${TAG}_sweep_message_launch.log
bash scripts/build_variant.sh "$GRAFT_REPO_ROOT/graphs4cfd_amd/lib/libg4c_timing.so" -DG4C_TIMING > /dev/null 2>&1 && \
  G4C_LIB_PATH=graphs4cfd_amd/lib/libg4c_timing.so timeout 300 python scripts/small_launch_stamps.py 400 12500 100000 2>&1 | grep -v amdgpu.ids > $A/${TAG}_small_launch_stamps.log
rm -f graphs4cfd_amd/lib/libg4c_timing.so
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $A/trace_small -o t -- \
    python bench.py --nodes 12500 --steps 60 --warmup 5 --no-side-configs --no-cpu-baseline --no-strict-range --no-roofline --no-partition-check > /dev/null 2>&1 )
python scripts/trace_gaps.py $A/trace_small > $A/${TAG}_trace_gaps_12k5.log 2>&1
python scripts/trace_step_positions.py $A/trace_small > $A/${TAG}_step_positions_12k5.log 2>&1; rm -rf $A/trace_small
# where a replayed step goes, launch by launch (headline and config 2)
for wl in headline c2; do
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $A/trace_$wl -o t -- \
      python bench.py --workload $wl --steps 20 --warmup 3 --no-side-configs --no-cpu-baseline --no-strict-range --no-roofline > /dev/null 2>&1 )
  sfx=""; [ "$wl" != headline ] && sfx="_$wl"
  python scripts/trace_step_positions.py $A/trace_$wl > $A/${TAG}_step_positions$sfx.log 2>&1; rm -rf $A/trace_$wl
done
ls -la $A
