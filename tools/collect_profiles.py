"""Summarise a tools/profile.sh output directory into profiles/ (kernel stats csv + per-kernel HBM counters).

usage: python tools/collect_profiles.py gpurun_out/DIR PREFIX [SUFFIX]   -> profiles/PREFIX_kernel_stats[SUFFIX].csv, ..._pmc_hbm_counters[SUFFIX].json
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB (MI355X_MICROARCH.md, HBM traffic section).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
suffix = sys.argv[3] if len(sys.argv) > 3 else ""          # e.g. "_C4": profiles/PREFIX_kernel_stats_C4.csv ...
# (gpurun merges into an existing gpurun_out/: take the newest run of each kind)
newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)
stats = newest(f"{src}/trace/**/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{prefix}_kernel_stats{suffix}.csv")
shutil.copy(f"{src}/bench_trace.json", f"profiles/{prefix}_bench_under_rocprof{suffix}.json")
# digest of what the library is built from -- sources, headers, flags: tscode_amd/build.py -- (bench.py prints `traffic` only while it still matches)
sys.path.insert(0, os.getcwd())
from tscode_amd.build import csrc_digest
_now = csrc_digest()
# ... and the digest the BINARY of the profiled run carried (bench.py prints it): the summary is labelled with that one, and refused when
# the sources have moved on since (a profile collected after an edit would otherwise pass for a profile of the edited kernels)
_line = [l for l in open(f"{src}/bench_trace.json").read().strip().splitlines() if l.startswith("{")][-1]
_ran = json.loads(_line)["roofline"].get("binary_csrc_sha256_16")
if _ran != _now:
    sys.exit(f"the profiled run was built from csrc {_ran}, the sources now are {_now}: run tools/profile.sh again")
out = {"csrc_sha256_16": _ran}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = newest(f"{src}/{name}/**/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    out[ctr] = {k: {"calls": c, "total_KB": v, "per_call_KB": v / c} for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    # steps the profiled command ran (k_init_run runs once per prune run = once per step): totals / steps = per step
    out["steps_profiled"] = next((v["calls"] for k, v in out[ctr].items() if "k_init_run" in k), None)
json.dump(out, open(f"profiles/{prefix}_pmc_hbm_counters{suffix}.json", "w"), indent=1)
for k, v in list(out["FETCH_SIZE"].items())[:8]:
    print(k, v)
