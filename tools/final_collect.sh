#!/bin/bash
# In the build container, after tools/final_measure.sh RR ran on the GPU box: gpurun_out/final_RR/* -> profiles/RR_*, then the doc tables.
# usage: tools/final_collect.sh r03
set -e
R=$1
O=gpurun_out/final_$R
python tools/collect_profiles.py $O/prof $R > /dev/null
python tools/collect_profiles.py $O/prof_c4 $R _C4 > /dev/null
cp $O/sq/sq_counters.json profiles/${R}_sq_counters_c4.json
[ -f $O/sq_c3/sq_counters.json ] && cp $O/sq_c3/sq_counters.json profiles/${R}_sq_counters_c3.json
[ -f $O/tcc_c4/tcc_counters.json ] && cp $O/tcc_c4/tcc_counters.json profiles/${R}_tcc_counters_C4.json
[ -f $O/ipc_exchange.json ] && cp $O/ipc_exchange.json profiles/${R}_ipc_exchange.json
[ -f $O/bench_gloo2.json ] && cp $O/bench_gloo2.json profiles/${R}_bench_gloo2_rehearsal.json
cp $O/bench.json profiles/${R}_final_bench.json
for c in C4 C5 C5chain; do cp $O/bench_$c.json profiles/${R}_final_bench_$c.json; done
cp $O/bench_force_sharded_1rank.json profiles/${R}_final_bench_force_sharded_1rank.json
cp $O/predicted_scaling.json profiles/${R}_predicted_scaling.json
cp $O/hard_workloads.json profiles/${R}_hard_workloads.json
cp $O/next_rows.json profiles/${R}_next_rows.json
python tools/make_tables.py
