#!/bin/bash
# Runs on the GPU box (through gpurun): tools/next_rows_bench.py plain (wall times) and under rocprofv3 (kernel durations).
# usage: tools/profile_next_rows.sh OUTDIR        (outputs under gpurun_out/OUTDIR)
set -e
P=gpurun_out/$1
mkdir -p "$P"
export TMPDIR=/tmp
python3 tools/next_rows_bench.py > "$P/next_rows.jsonl" 2> "$P/next_rows.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -- python3 tools/next_rows_bench.py > "$P/next_rows_traced.jsonl" 2> "$P/trace.err"
find "$P" -name '*kernel_stats.csv' | head -3
