#!/bin/bash
# Runs on the GPU box (through gpurun): the same bench command on the product library and on variants built by tools/build_variant.py, twice each,
# interleaved (a box's clocks drift).   usage: tools/ab_bench.sh "bench.py arguments" NAME [NAME ...]     (NAME = tscode_amd/ab_libs/NAME.so; "base" = the product)
ARGS=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset TSCODE_AMD_LIB; else export TSCODE_AMD_LIB=$PWD/tscode_amd/ab_libs/$v.so; fi
    python3 bench.py $ARGS --no-cpu --no-side-leg > gpurun_out/ab/$v.$rep.json 2> gpurun_out/ab/$v.$rep.err || { echo "$v failed"; tail -3 gpurun_out/ab/$v.$rep.err; exit 1; }
    python3 - gpurun_out/ab/$v.$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(f'{sys.argv[2]:12s} ms_per_step {d["ms_per_step"]:.4f}  events_off {d["events_off"]["ms_per_step"]:.4f}  parity {d["config"].get("parity_vs_recorded_oracle")}  '
      f'pair kernel us {d["roofline"]["avg_launch_us"]:.1f}')
PY
  done
done
