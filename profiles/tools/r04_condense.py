#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of r04_profile_all.sh into the small text / json files that are committed under profiles/.
usage: r04_condense.py <raw dir> <out dir>"""
import collections
import csv
import glob
import json
import os
import sys

raw, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def find(d, pat):
    f = glob.glob(os.path.join(raw, d, "**", pat), recursive=True)
    return f[0] if f else None


def stats_table(d, top=12):
    f = find(d, "*kernel_stats.csv")
    if not f:
        return "(no kernel_stats.csv)\n", {}
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"{'kernel':96s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}   min / max us"]
    avg = {}
    for r in rows[:top]:
        n = r["Name"].replace("void ", "")
        avg[n] = float(r["AverageNs"]) / 1e3
        lines.append(f"{n[:96]:96s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['AverageNs']) / 1e3:9.1f} "
                     f"{float(r['Percentage']):6.2f}   {float(r['MinNs']) / 1e3:7.1f} / {float(r['MaxNs']) / 1e3:7.1f}")
    lines.append(f"(all kernels: {tot / 1e6:.1f} ms of kernel time)")
    return "\n".join(lines) + "\n", avg


def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(raw, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


for case in ("dw", "ffn1", "attn"):
    txt, avg = stats_table(case + "_stats")
    plain = open(os.path.join(raw, case + "_plain.log")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(raw, case + "_plain.log")) else ""
    counters = {}
    for grp in ("sq", "fetch", "write", "tcc"):
        for k, cs in pmc(f"{case}_{grp}").items():
            counters.setdefault(k, {}).update(cs)
    with open(os.path.join(out, f"r04_{case}_kernel_stats.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 profiles/tools/dominant.py {case} 20   (MI355X, round 4)\n")
        f.write(f"# un-profiled HIP-event timing of the same command: {plain}\n")
        f.write(txt)
        f.write("\n# rocprofv3 --pmc, separate passes (SQ group | FETCH_SIZE + GRBM | WRITE_SIZE | TCC + instruction mix), average per launch:\n")
        for k, cs in counters.items():
            if any(s in k for s in ("gemm_", "attn_", "splitk")):
                f.write(k + "\n")
                for c, v in sorted(cs.items()):
                    f.write(f"    {c:32s} {v:18.1f}\n")
    if case == "attn":
        att = {}
        for k, c in counters.items():
            if "attn_" not in k:
                continue
            e = {"avg_us_rocprof": next((v for kk, v in avg.items() if k.replace("void ", "")[:22] in kk), None)}
            for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                         "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
                if name in c:
                    e[name] = c[name]
            if "SQ_INSTS_VALU" in e and e.get("SQ_INSTS_MFMA"):
                e["valu_per_mfma_instruction"] = e["SQ_INSTS_VALU"] / e["SQ_INSTS_MFMA"]
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["hbm_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024
            att[k] = e
        tot = sum(v.get("hbm_bytes_per_launch", 0.0) for k, v in att.items() if "dropmask" not in k)
        json.dump({"kernels": att, "hbm_bytes_per_launch": tot,
                   "note": "rocprofv3 --pmc in separate passes over profiles/tools/dominant.py attn (one layer: keep-bit generator, forward, dQ, dK/dV; "
                           "B=64, A=12, L=512, 24,643 packed rows, dropout 0.1); hbm_bytes_per_launch = forward + dQ + dK/dV, FETCH_SIZE x2 + WRITE_SIZE "
                           "(gfx950 tallies wide streaming reads at half: MI355X_MICROARCH.md, HBM)"},
                  open(os.path.join(out, "r04_attn_pmc.json"), "w"), indent=1)
    # HBM bytes per launch of the case's main kernel: FETCH_SIZE (KB) x2 (gfx950 tallies wide streaming reads at half) + WRITE_SIZE (KB)
    main = [k for k in counters if ("gemm_pring" in k or "gemm_ring" in k)]
    if main:
        c = counters[main[0]]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            json.dump({"kernel": main[0], "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
                       "hbm_bytes_per_launch": c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024,
                       "avg_us_rocprof": next((v for k, v in avg.items() if main[0].replace("void ", "")[:20] in k), None),
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over profiles/tools/dominant.py " + case +
                               "; FETCH_SIZE x2: gfx950 tallies wide streaming reads at half (MI355X_MICROARCH.md, HBM)"},
                      open(os.path.join(out, f"r04_{case}_pmc.json"), "w"), indent=1)
txt, _ = stats_table("bench_stats", top=30)
with open(os.path.join(out, "r04_bench_kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-table   (MI355X, round 4; 7 steps)\n")
    f.write("# side-stream kernels (weight gradients, reductions, column sums, keep-bit generator, AdamW) overlap the main chain: durations are\n# inflated by sharing the chip and their sum exceeds the wall time\n")
    f.write(txt)
print("wrote", sorted(os.listdir(out)))
