"""Join tools/next_rows_bench.py's lines with the rocprofv3 kernel stats of the same script: profiles/PREFIX_next_rows.{json,md}.
usage: python tools/join_next_rows.py gpurun_out/OUTDIR PREFIX"""
import csv, glob, json, sys
src, prefix = sys.argv[1], sys.argv[2]
rows = [json.loads(l) for l in open(f"{src}/next_rows.jsonl") if l.startswith("{")]
stats = {}
for f in glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].split("::")[-1].split("<")[0].strip()
        stats[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
HBM = 8000.0  # GB/s, MI355X_MICROARCH.md
md = ["| row | entry point | size | GPU call, host arrays in/out (ms) | rate | oracle on the host cores (OpenMP) | kernel (largest launch, us) | algorithmic MB | GB/s in the kernel | of 8 TB/s | equal on the oracle's sample |",
      "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    ks = [stats[k] for k in r["kernels"] if k in stats]
    r["kernel_stats"] = {k: stats[k] for k in r["kernels"] if k in stats}
    kus = sum(k["max_us"] for k in ks) if ks else None          # the full-size call is the longest launch of each kernel
    single = len(r["kernels"]) == 1 and kus
    r["kernel_GBs"] = r["algorithmic_bytes"] / kus / 1e3 if single else None
    r["hbm_frac"] = r["kernel_GBs"] / HBM if single else None
    md.append(f"| {r['row']} | `{r['entry']}` | {r['size']} | {r['gpu_wall_ms_pcie_inclusive']:.2f} | {r['gpu_units_per_s']:.3g} {r['unit'].split(' ')[0]}/s | "
              f"{r.get('oracle_units_per_s', r.get('oracle_units_per_s_1core')):.3g}/s | {kus and round(kus, 1)} | {r['algorithmic_bytes'] / 1e6:.1f} | "
              f"{r['kernel_GBs'] and round(r['kernel_GBs'])} | {r['hbm_frac'] and round(r['hbm_frac'], 3)} | {r['equal_on_sample']} |")
json.dump(rows, open(f"profiles/{prefix}_next_rows.json", "w"), indent=1)
open(f"profiles/{prefix}_next_rows.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
