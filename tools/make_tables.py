#!/usr/bin/env python
"""Writes the measurement tables of README.md, DESIGN.md and MEASURED.md FROM the files under profiles/ (hand-kept tables
drift).  Everything between `<!-- tables:NAME:begin -->` and `<!-- tables:NAME:end -->` in the two documents is replaced; the
numbers' sources are named in each table's caption.

usage: python tools/make_tables.py [--check]      (--check: exit 1 if a document would change)"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
R = "r05"


def kernel_name(raw):
    """rocprofv3's kernel name without arguments; names it could not demangle (a _Float16 parameter: DF16_) as namespace tsc's plain name."""
    m = re.match(r"_ZN3tsc(\d+)", raw)
    if m:
        n = int(m.group(1))
        return raw[m.end():m.end() + n] + ("<...>" if raw[m.end() + n] == "I" else "")
    return re.sub(r"\(.*", "", raw).replace("void ", "").replace("tsc::", "")


def last_json_line(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def load(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else None


def f(x, nd=3):
    return "—" if x is None else (f"{x:.{nd}f}" if isinstance(x, float) else str(x))


def headline():
    b = last_json_line(f"{R}_final_bench.json")
    if not b:
        return f"(profiles/{R}_final_bench.json missing)"
    rows = ["| | ms / step | conformers/s |", "|---|---|---|"]
    rows.append(f"| C3 (100 000 x 50), timed region as the bench contract asks (HIP events on the pair-kernel dispatches of every 4th step): `value` | {b['ms_per_step']:.3f} | **{b['value'] / 1e6:.1f} M** |")
    if b.get("events_off"):
        rows.append(f"| the same K steps with the library's events off (`events_off`) | {b['events_off']['ms_per_step']:.3f} | {b['events_off']['value'] / 1e6:.1f} M |")
    if b.get("steps_in_flight") and "ms_per_step" in b["steps_in_flight"]:
        s = b["steps_in_flight"]
        rows.append(f"| {s['in_flight']} independent steps in flight (a side figure, never `value`) | {s['ms_per_step']:.3f} | {s['value'] / 1e6:.1f} M |")
    cb = b.get("cpu_baseline")
    if cb:
        rows.append(f"| CPU beside it: the oracle (\"{cb['kind']}\") on the WHOLE C3 workload, {cb['cores']} threads of {cb['cpu_model']} | {cb['seconds'] * 1e3:.0f} | {cb['value']:.0f} "
                    f"(GPU / CPU {cb.get('gpu_over_cpu', 0):.0f} x, survivors identical: {cb.get('survivors_identical_to_gpu')}) |")
    for name, label in ((f"{R}_final_bench_C4.json", "C4 (1M x 50) on one GPU"), (f"{R}_final_bench_C5.json", "C5 (500k x 200 atoms, 3 fragments)"),
                        (f"{R}_final_bench_C5chain.json", "C5 as a chain (20 000 csearch candidates -> conformers -> 500k poses -> clash -> prune)"),
                        (f"{R}_final_bench_force_sharded_1rank.json", "C3 through the multi-rank protocol with ONE rank (`--force-sharded`, RCCL collectives of one rank)")):
        d = last_json_line(name)
        if d:
            eo = d.get("events_off") or {}
            rows.append(f"| {label} | {d['ms_per_step']:.3f} ({f(eo.get('ms_per_step'))} events off) | {d['value'] / 1e6:.1f} M (parity with the recorded oracle run: {d['config'].get('parity_vs_recorded_oracle')}) |")
    return "\n".join(rows) + f"\n\n(from `profiles/{R}_final_bench*.json`; parity = packed survivor mask, counts and conformer count equal to `tests/golden/expected_full.json`)"


def roofline():
    b = last_json_line(f"{R}_final_bench.json")
    if not b or not b.get("roofline"):
        return f"(no roofline in profiles/{R}_final_bench.json)"
    r = b["roofline"]
    ex = r.get("executed", {})
    kname = r.get("kernel", "k_rmsd_sieve").split(" ")[0].split("<")[0]
    rows = [f"| `roofline` of the dominant kernel (`{kname}`, per launch) | |", "|---|---|",
            f"| algorithmic bytes per launch (SURVEY 8d: `A_p h 24 + 2 N`, mean over the {r['launches_per_step']} launches of a step) | {r['algorithmic_bytes_per_launch'] / 1e6:.2f} MB |",
            f"| average launch (HIP events on the dispatches of the timed region) | {r['avg_launch_us']:.2f} us |",
            f"| achieved / peak | {r['achieved']:.0f} / {r['peak']:.0f} GB/s = **{r['frac']:.4f}** |",
            f"| HBM traffic per launch (rocprofv3 --pmc, 2 x FETCH_SIZE + WRITE_SIZE) | {f(r['traffic'] / 1e6 if r.get('traffic') else None, 1)} MB"
            + (f" = {r['traffic'] / r['algorithmic_bytes_per_launch']:.2f} x the algorithmic bytes" if r.get("traffic") else f" ({r.get('traffic_source')})") + " |",
            f"| what binds the kernel (`roofline.bound`) and how close it comes (`roofline.issue_frac`: wave64 VALU instructions per CU-cycle of its launches, ceiling 1; SQ counters) | {r.get('bound')} -- VALU issue at "
            + (f"{r['issue_frac']:.3f} ({r['issue']['issue_frac_while_busy']:.3f} of the cycles a CU is busy; {r['issue']['fma_f32_share_of_valu']:.2f} of the instructions are fp32 FMAs)" if r.get('issue_frac') else f"— ({(r.get('issue') or {}).get('source', 'no SQ profile of these kernels')})") + " |",
            ("| what the instructions execute | " + (f"{ex['f16_mfma_TFLOPs']:.1f} TFLOP/s float16 MFMA (the screen: 64 flop per pair) = {ex['f16_mfma_frac']:.3f} of {ex['f16_mfma_peak_TFLOPs']:.0f}"
                                                   if ex.get("f16_mfma_TFLOPs") else f"{ex.get('fp32_TFLOPs') or 0:.1f} TFLOP/s packed fp32 = {ex.get('fp32_frac') or 0:.3f} of {ex.get('fp32_peak_TFLOPs')}")
             + f" + {ex.get('fp64_TFLOPs') or 0:.2f} TFLOP/s fp64 |"),
            f"| pairs screened / H formed per step | {ex.get('pairs_screened_per_step', 0):.3g} / {ex.get('pairs_with_H_formed_per_step', 0):.3g} |"]
    fr = b.get("roofline_front")
    if fr:
        rows.append(f"| front half (`roofline_front`: fused embed + clash, compaction, the passing poses embedded + described): B_K12 = {fr['algorithmic_bytes'] / 1e6:.1f} MB "
                    f"over {fr['ms'] * 1e3:.0f} us of stage events | {fr['achieved']:.0f} / {fr['peak']:.0f} GB/s = {fr['frac']:.3f} |")
    c4 = b.get("c4") or {}
    if c4.get("traffic_over_algorithmic"):
        rows.append(f"| C4 (1M x 50), pair kernels (`k_rmsd_sieve_mm` + `k_rmsd_sieve_sorted_mm`): HBM traffic / algorithmic bytes per step | "
                    f"{c4['pair_kernels_traffic_bytes_per_step'] / 1e9:.2f} GB / {c4['pair_kernels_algorithmic_bytes_per_step'] / 1e9:.2f} GB = {c4['traffic_over_algorithmic']:.2f} x |")
    ph = b.get("pipeline_hbm")
    if ph:
        rows.append(f"| whole step: algorithmic bytes over the step | {ph['algorithmic_bytes'] / 1e6:.0f} MB -> {ph['achieved_GBs']:.0f} GB/s = {ph['frac']:.3f} of peak |")
    return "\n".join(rows) + f"\n\n(from `profiles/{R}_final_bench.json`, `profiles/{R}_pmc_hbm_counters.json`)"


def kernels():
    path = os.path.join(P, f"{R}_kernel_stats.csv")
    if not os.path.exists(path):
        return f"(profiles/{R}_kernel_stats.csv missing)"
    rows_in = list(csv.DictReader(open(path)))
    steps = max([int(r["Calls"]) for r in rows_in if "k_init_run" in r["Name"]] or [1])
    rows = [f"| kernel (C3, {steps} steps under `rocprofv3 --kernel-trace --stats`) | launches / step | average us | us / step |", "|---|---|---|---|"]
    tot = 0.0
    for r in sorted(rows_in, key=lambda r: -int(r["TotalDurationNs"])):
        per = int(r["TotalDurationNs"]) / steps / 1e3
        tot += per
        if per >= 3.0:
            name = kernel_name(r["Name"])
            rows.append(f"| `{name}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.2f} | {per:.1f} |")
    rows.append(f"| sum over all kernels | | | {tot:.0f} |")
    return "\n".join(rows) + f"\n\n(from `profiles/{R}_kernel_stats.csv`)"


def kernels_c4():
    path = os.path.join(P, f"{R}_kernel_stats_C4.csv")
    if not os.path.exists(path):
        return f"(profiles/{R}_kernel_stats_C4.csv missing)"
    rows_in = list(csv.DictReader(open(path)))
    steps = max([int(r["Calls"]) for r in rows_in if "k_init_run" in r["Name"]] or [1])
    rows = [f"| kernel (C4, {steps} steps under `rocprofv3 --kernel-trace --stats`) | launches / step | average us | us / step |", "|---|---|---|---|"]
    tot = 0.0
    for r in sorted(rows_in, key=lambda r: -int(r["TotalDurationNs"])):
        per = int(r["TotalDurationNs"]) / steps / 1e3
        tot += per
        if per >= 40.0:
            name = kernel_name(r["Name"])
            rows.append(f"| `{name}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.2f} | {per:.1f} |")
    rows.append(f"| sum over all kernels | | | {tot:.0f} |")
    return "\n".join(rows) + f"\n\n(from `profiles/{R}_kernel_stats_C4.csv`)"


def hard():
    d = load(f"{R}_hard_workloads.json")
    if not d:
        return f"(profiles/{R}_hard_workloads.json missing)"
    rows = ["| workload (100 000 structures) | automatic choice: ms (kernels) | sieve | all-pairs kernel | conformers/s (automatic) | screen lets through | H formed | CPU port on a sample |",
            "|---|---|---|---|---|---|---|---|"]
    for name, l in d["legs"].items():
        g = l["gpu"]
        kern = {1: "tile", 2: "sieve", 3: "chunk-local"}
        cpu = l.get("cpu") or {}
        rows.append(f"| `{name}`: {l['workload']} | {g['auto']['ms_per_step']:.2f} ({', '.join(kern.get(k, str(k)) for k in g['auto']['kernels'])}) | {g['sieve']['ms_per_step']:.2f} | "
                    f"{g['tile']['ms_per_step']:.2f} | {d['n'] / g['auto']['ms_per_step'] * 1e3 / 1e6:.1f} M | {f(g['sieve'].get('screen_pass_rate'), 4)} | {g['sieve']['H_formed']:.3g} | "
                    f"{cpu.get('structures_per_s', 0):.0f}/s on {cpu.get('structures', 0)} ({cpu.get('threads')} threads) |")
    return "\n".join(rows) + f"\n\n(from `profiles/{R}_hard_workloads.json`, `tools/hard_workloads.py`)"


def scaling():
    d = load(f"{R}_predicted_scaling.json")
    if not d:
        return f"(profiles/{R}_predicted_scaling.json missing)"
    rows = ["| config | ranks | front: shard / replicate / hybrid (compute + modelled exchange, ring) | prune: setup (a run that builds its descriptors; `hybrid` leaves them beside the heavy atoms) + partitioned (local / close) + row tiles (local / close) + small | per-pass collectives ring - all links | step (best front) ring - all links | speed-up vs one rank | the per-pass messages through the library's own exchange (`tsc_xchg_*`) | the ring figure at 50 / 100 us fixed per collective (`collectives_per_step` of them) |",
            "|---|---|---|---|---|---|---|---|---|"]
    for cfg, rws in d["configs"].items():
        for r in rws:
            fr = r["fronts"]
            fs = " / ".join(f"{fr[k]['front_compute_ms'] + fr[k]['front_comm_ms_ring']:.2f}" if k in fr else "—" for k in ("shard", "replicate", "hybrid"))
            bk = r["by_kind"]
            g = lambda k, w: bk.get(k, {}).get(w, 0.0)
            best = min(fr.values(), key=lambda v: v["predicted_ms_per_step"])
            best_al = min(fr.values(), key=lambda v: v["predicted_ms_per_step_all_links"])
            sb = r.get("setup_borrowed_ms")
            setup = f"{r['setup_ms']:.2f}" + (f" (hybrid: {sb:.2f})" if (sb is not None and r["n_ranks"] > 1) else "")
            names = {id(v): k for k, v in fr.items()}
            rows.append(f"| {cfg} | {r['n_ranks']} | {fs} | {setup} + ({g('partitioned', 'local_ms'):.2f} / {g('partitioned', 'close_ms'):.2f}) + "
                        f"({g('row_tiles', 'local_ms'):.2f} / {g('row_tiles', 'close_ms'):.2f}) + {g('replicated', 'local_ms'):.2f} | {r['pass_comm_ms_ring']:.2f} - {r['pass_comm_ms_all_links']:.2f} | "
                        f"**{best['predicted_ms_per_step']:.2f}** ({names[id(best)]}) - {best_al['predicted_ms_per_step_all_links']:.2f} ({names[id(best_al)]}) | "
                        f"{best['speedup_vs_1_rank_protocol']:.2f} - {best_al['speedup_vs_1_rank_protocol_all_links']:.2f} x | "
                        + (lambda bi: f"{bi['predicted_ms_per_step_ipc']:.2f} ({names[id(bi)]}; {bi.get('speedup_vs_1_rank_protocol_ipc', 0):.2f} x)")(min(fr.values(), key=lambda v: v.get("predicted_ms_per_step_ipc", 1e9)))
                        + " | " + (f"{best['by_collective_fixed_us']['50']['ring']:.2f} / {best['by_collective_fixed_us']['100']['ring']:.2f} ({best['collectives_per_step']})"
                                   if "by_collective_fixed_us" in best else "—") + " |")
    return "\n".join(rows) + (f"\n\n(ms per step; from `profiles/{R}_predicted_scaling.json`, `tools/predict_scaling.py`: every rank's library calls timed on ONE GPU, collectives modelled: "
                              f"{d['model']['xgmi_link_GBs']} GB/s per link x {d['model']['link_efficiency']}, {d['model']['collective_fixed_us']} us fixed per collective)")


def culling():
    d = load("r03_culling_study.json")       # (offline CPU study of round 3: not re-run)
    if not d:
        return f"(profiles/{R}_culling_study.json missing)"
    rows = ["| config | pass k | active structures per chunk | tile pairs (16 x 128) of a chunk | visited by the ordered walk (if no cache stop) | within the limit, each unordered pair once | ratio | pairs the screen lets through |",
            "|---|---|---|---|---|---|---|---|"]
    for cfg, rws in d["configs"].items():
        for r in rws:
            rows.append(f"| {cfg} | {r['k']} | {r['structures']} | {r['tile_pairs_all']:.0f} | {r['tile_pairs_present_walk']:.0f} | {r['tile_pairs_within_limit']:.0f} | {r['culled_over_walk']:.2f} | {r['screen_pass_rate_of_pairs']:.5f} |")
    return "\n".join(rows) + f"\n\n(from `profiles/r03_culling_study.json`, `tools/culling_study.py`, offline on the CPU)"


TABLES = {"headline": headline, "roofline": roofline, "kernels": kernels, "kernels_c4": kernels_c4, "hard": hard, "scaling": scaling, "culling": culling}


def main():
    check = "--check" in sys.argv
    changed = False
    for doc in ("README.md", "DESIGN.md", "MEASURED.md"):
        path = os.path.join(ROOT, doc)
        text = open(path).read()
        new = text
        for name, fn in TABLES.items():
            pat = re.compile(rf"(<!-- tables:{name}:begin -->\n).*?(<!-- tables:{name}:end -->)", re.S)
            if pat.search(new):
                body = fn()
                new = pat.sub(lambda m, body=body: m.group(1) + body + "\n" + m.group(2), new)
        if new != text:
            changed = True
            if not check:
                open(path, "w").write(new)
                print("updated", doc)
    if check and changed:
        print("tables are stale: run python tools/make_tables.py")
        sys.exit(1)


if __name__ == "__main__":
    main()
