#!/bin/bash
# Runs on the GPU box (through gpurun): the round's closing measurements, everything into gpurun_out/final_RR/ (copy to profiles/ afterwards:
# tools/final_collect.sh RR).   usage: tools/final_measure.sh r03 [part ...]     parts: tests prof profc4 sq sqc3 tcc ipc bench configs gloo2 scaling hard rows
set -e
R=$1; shift
PARTS=${*:-"tests prof profc4 sq sqc3 tcc ipc bench configs gloo2 scaling hard rows"}
O=gpurun_out/final_$R
mkdir -p $O
for part in $PARTS; do
  echo "== $part $(date +%T)"
  case $part in
    tests)   python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt ;;
    prof)    bash tools/profile.sh final_$R/prof > $O/prof.txt 2>&1 ;;
    profc4)  bash tools/profile.sh final_$R/prof_c4 --config C4 > $O/prof_c4.txt 2>&1 ;;
    sq)      bash tools/pmc_sq.sh final_$R/sq --config C4 > $O/sq.txt 2>&1 ;;
    sqc3)    bash tools/pmc_sq.sh final_$R/sq_c3 > $O/sq_c3.txt 2>&1 ;;
    tcc)     bash tools/pmc_tcc.sh final_$R/tcc_c4 --config C4 > $O/tcc_c4.txt 2>&1 ;;
    ipc)     timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29701 tools/ipc_exchange_bench.py $O/ipc_exchange.json > $O/ipc.txt 2>&1; tail -c 300 $O/ipc.txt ;;
    gloo2)   timeout -k 10 600 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu > $O/bench_gloo2.json 2> $O/bench_gloo2.err; cut -c1-200 $O/bench_gloo2.json ;;
    bench)   python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json ;;
    configs) for c in C4 C5 C5chain; do python bench.py --config $c --no-cpu --no-side-leg > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-200 $O/bench_$c.json; done
             python bench.py --force-sharded --no-cpu --no-side-leg > $O/bench_force_sharded_1rank.json 2> $O/bench_fs.err ;;
    scaling) python tools/predict_scaling.py C3 C4 > $O/predicted_scaling.json 2> $O/predicted_scaling.err; tail -8 $O/predicted_scaling.err ;;
    hard)    python tools/hard_workloads.py > $O/hard_workloads.json 2> $O/hard.err ;;
    rows)    python tools/next_rows_bench.py > $O/next_rows.json 2> $O/rows.err ;;
  esac
done
echo "== done $(date +%T)"
