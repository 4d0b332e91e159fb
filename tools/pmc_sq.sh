#!/bin/bash
# Runs on the GPU box (through gpurun): SQ counters of the pair kernel, three --pmc passes of a short bench run.
# usage: tools/pmc_sq.sh OUTDIR [bench.py arguments]
set -e
P=gpurun_out/$1; shift
mkdir -p "$P"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu --no-side-leg $*"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$P/p1" -- python3 bench.py $ARGS > "$P/b1.json" 2> "$P/p1.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d "$P/p2" -- python3 bench.py $ARGS > "$P/b2.json" 2> "$P/p2.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d "$P/p3" -- python3 bench.py $ARGS > "$P/b3.json" 2> "$P/p3.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d "$P/p4" -- python3 bench.py $ARGS > "$P/b4.json" 2> "$P/p4.err"
# (the matrix-core screen of the large runs, csrc/mm.hpp: how busy the MFMA pipe is; a pass of its own that may fail where a counter is not offered)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d "$P/p5" -- python3 bench.py $ARGS > "$P/b5.json" 2> "$P/p5.err" || echo "pass p5 (MFMA counters) failed: see $P/p5.err"
python3 - "$P" <<'PY'
import csv, glob, sys, collections, json
P = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in glob.glob(f"{P}/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
out = {k: dict(v) for k, v in agg.items() if "sieve" in k or "open_rows" in k or "apply" in k or "pass_chunks" in k}
# launches of every kernel in one pass (the passes run the same command) and the digest of the binary that ran: what issue_frac is computed from
launches = collections.defaultdict(set)
for f in glob.glob(f"{P}/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        launches[r["Kernel_Name"].split("(")[0].split("<")[0]].add(r["Dispatch_Id"])
for k in out:
    out[k]["launches"] = len(launches.get(k, ()))
try:
    line = [l for l in open(f"{P}/b1.json").read().strip().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    out["_run"] = {"csrc_sha256_16": b["roofline"].get("binary_csrc_sha256_16"), "workload": b["config"]["workload"],
                   "what": "SQ counters summed over every launch of the kernel in `bench.py --steps 2 --warmup 1` (8 steps with the events-off and detail "
                           "steps); VALU issue: SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES (a CU issues one wave64 VALU instruction per cycle: four SIMDs, "
                           "four cycles each)"}
except Exception as exc:
    out["_run"] = {"error": str(exc)}
json.dump(out, open(f"{P}/sq_counters.json", "w"), indent=1)
out.pop("_run", None)
for k, v in out.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   ", c, f"{x:.4g}")
PY
