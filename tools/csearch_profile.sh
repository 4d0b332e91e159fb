#!/bin/bash
# Runs on the GPU box (through gpurun): tools/csearch_profile.py + the SQ counters of k_csearch_rotate (separate --pmc passes).
# usage: tools/csearch_profile.sh OUTDIR        -> gpurun_out/OUTDIR/csearch_profile.json
set -e
P=gpurun_out/$1
mkdir -p "$P"
export TMPDIR=/tmp
python3 tools/csearch_profile.py > "$P/plain.json" 2> "$P/plain.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$P/p1" -- python3 tools/csearch_profile.py --quick > "$P/b1.json" 2> "$P/p1.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d "$P/p2" -- python3 tools/csearch_profile.py --quick > "$P/b2.json" 2> "$P/p2.err"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d "$P/p3" -- python3 tools/csearch_profile.py --quick > "$P/b3.json" 2> "$P/p3.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d "$P/p4" -- python3 tools/csearch_profile.py --quick > "$P/b4.json" 2> "$P/p4.err"
python3 - "$P" <<'PY'
import csv, glob, sys, collections, json
P = sys.argv[1]
agg = collections.defaultdict(float)
disp = collections.defaultdict(set)
for f in glob.glob(f"{P}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_csearch_rotate" not in r["Kernel_Name"]:
            continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
out = json.load(open(f"{P}/plain.json"))
# the --quick runs launch the kernel several times (same size every time): per-launch means
c = {k: v / max(1, len(disp[k])) for k, v in agg.items()}
out["sq_counters_per_launch (20000 candidates, --quick)"] = c
g = lambda k: c.get(k, 0.0)
if g("SQ_WAVE_CYCLES"):
    out["derived"] = {"valu_issue_share_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES") * 4,
                      "lds_issue_share_of_wave_cycles": g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES") * 4,
                      "wait_inst_any_share": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), "wait_inst_lds_share": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
                      "wait_any_share": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                      "lds_bank_conflict_cycles_per_lds_active_cycle": g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")),
                      "insts_per_wave": {k[9:]: g(k) / max(1.0, g("SQ_WAVES")) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH")},
                      "note": "SQ_ACTIVE_INST_* count quad-cycles on this part (x4 = cycles), as in profiles/r03_sq_counters_c4.json"}
json.dump(out, open(f"{P}/csearch_profile.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
