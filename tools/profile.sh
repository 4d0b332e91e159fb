#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace and the two HBM counter passes of the bench command.
# usage: tools/profile.sh OUTDIR [bench.py arguments]        (outputs under gpurun_out/OUTDIR)
set -e
P=gpurun_out/$1; shift
mkdir -p "$P"
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu --no-side-leg $*"   # (the C3 steps only: the side legs run other workloads through the same kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 bench.py $ARGS > "$P/bench_trace.json" 2> "$P/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$P/pmc_fetch" -- python3 bench.py $ARGS > "$P/bench_fetch.json" 2> "$P/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$P/pmc_write" -- python3 bench.py $ARGS > "$P/bench_write.json" 2> "$P/write.err"
find "$P" -name '*.csv' | head -20
