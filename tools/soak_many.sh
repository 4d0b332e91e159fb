#!/bin/bash
# Runs on the GPU box (through gpurun): N concurrent soak.py processes on the one card (they perturb each other's workgroup order),
# every step of every process compared with its first and with the recorded oracle result.  At most 5 (the box allows 6 GPU processes).
# usage: tools/soak_many.sh N CONFIG STEPS [opt=value ...]      (logs under gpurun_out/soak/)
N=$1; CFG=$2; STEPS=$3; shift 3
mkdir -p gpurun_out/soak
pids=()
for i in $(seq 1 "$N"); do
    python3 tools/soak.py "$CFG" "$STEPS" "$@" > "gpurun_out/soak/${CFG}_$i.log" 2>&1 &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
tail -n 2 gpurun_out/soak/${CFG}_*.log
exit $rc
