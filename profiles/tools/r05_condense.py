#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of r05_profile_all.sh into the files committed under profiles/:
  r05_kernels.json               per kernel family: the HIP-event timing and the rocprofv3 timing OF THE SAME LEASE, HBM bytes per launch
                                 from the FETCH_SIZE / WRITE_SIZE passes -- bench.py reads it (roofline.kernels[*].profile)
  r05_<case>_kernel_stats.txt    the rocprofv3 --stats table of each case + its counters
usage: r05_condense.py <raw dir> <out dir>"""
import collections
import csv
import glob
import json
import os
import re
import statistics
import sys

raw, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
MAIN = re.compile(r"gemm_|attn_fwd|attn_bwd|attn_dropmask|splitk_reduce")     # kernels that belong to a case's call (torch fills do not)
WARM = 50


def find(d, pat):
    f = glob.glob(os.path.join(raw, d, "**", pat), recursive=True)
    return f[0] if f else None


def stats_table(d, top=10):
    f = find(d, "*kernel_stats.csv")
    if not f:
        return "(no kernel_stats.csv)\n"
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"{'kernel':96s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}   min / max us"]
    for r in rows[:top]:
        n = r["Name"].replace("void ", "")
        lines.append(f"{n[:96]:96s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:9.2f} {float(r['AverageNs']) / 1e3:9.1f} "
                     f"{float(r['Percentage']):6.2f}   {float(r['MinNs']) / 1e3:7.1f} / {float(r['MaxNs']) / 1e3:7.1f}")
    lines.append(f"(all kernels: {tot / 1e6:.1f} ms of kernel time)")
    return "\n".join(lines) + "\n"


def trace(d):
    """per kernel name: list of durations (us) in dispatch order"""
    f = find(d, "*kernel_trace.csv")
    per = collections.OrderedDict()
    if not f:
        return per
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        n = r["Kernel_Name"].replace("void ", "")
        if MAIN.search(n):
            per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def pmc(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(raw, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].replace("void ", "").split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


cases = sorted({os.path.basename(p)[:-len("_plain.log")] for p in glob.glob(os.path.join(raw, "*_plain.log"))})
summary = {}
for case in cases:
    plain = open(os.path.join(raw, case + "_plain.log")).read().strip().splitlines()
    plain = plain[-1] if plain else ""
    m = re.search(r"median ([\d.]+) us per call \(min ([\d.]+), max ([\d.]+), mean ([\d.]+)", plain)
    e = {"hip_event_line": plain}
    if m:
        e.update(hip_event_median_us=float(m.group(1)), hip_event_min_us=float(m.group(2)), hip_event_max_us=float(m.group(3)), hip_event_mean_us=float(m.group(4)))
    per = trace(case + "_stats")
    kern = {}
    call_all = call_warm = call_med = 0.0
    ncalls = None
    for n, d in per.items():
        calls_per = max(1, round(len(d) / 250.0))                       # launches of this kernel per call of the case (250 calls: 50 + 200)
        ncalls = len(d) // calls_per
        w = d[WARM * calls_per:] or d
        kern[n[:110]] = dict(launches=len(d), avg_us_all=sum(d) / len(d), avg_us_after_warmup=sum(w) / len(w), median_us_after_warmup=statistics.median(w),
                             min_us=min(d), max_us=max(d), launches_per_call=calls_per)
        call_all += calls_per * sum(d) / len(d)
        call_warm += calls_per * sum(w) / len(w)
        call_med += calls_per * statistics.median(w)
    e["rocprof_kernels"] = kern
    e["rocprof_call_us_avg_all_launches"] = call_all            # what `rocprofv3 --stats` prints as the average, summed over the call's kernels
    e["rocprof_call_us_avg_after_warmup"] = call_warm
    e["rocprof_call_us_median_after_warmup"] = call_med
    counters = {}
    for grp in ("sq", "fetch", "write", "tcc"):
        for k, cs in pmc(f"{case}_{grp}").items():
            counters.setdefault(k, {}).update(cs)
    hbm, perk = 0.0, {}
    for k, c in counters.items():
        if not MAIN.search(k):
            continue
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024          # FETCH_SIZE x2: gfx950 tallies wide streaming reads at half
            perk[k] = dict(fetch_size_kb=c["FETCH_SIZE"], write_size_kb=c["WRITE_SIZE"], hbm_bytes_per_launch=b)
            calls_per = next((v["launches_per_call"] for n, v in kern.items() if n.startswith(k[:40])), 1)
            hbm += b * calls_per
        for name in ("TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
                     "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"):
            if name in c:
                perk.setdefault(k, {})[name] = c[name]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            perk[k]["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            perk[k]["wait_any_frac_of_wave_cycles"] = c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0)
    e["pmc"] = perk
    e["hbm_bytes_per_call"] = hbm if hbm > 0 else None
    e["note"] = ("one lease, one box: HIP events (profiles/tools/dominant.py <case> 200 50, un-profiled) and rocprofv3 --kernel-trace --stats of the same command; "
                 "--pmc FETCH_SIZE / WRITE_SIZE in separate passes (dominant.py <case> 10 5), hbm bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 per launch, "
                 "summed over the kernels of one call")
    summary[case] = e
    with open(os.path.join(out, f"r05_{case}_kernel_stats.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 profiles/tools/dominant.py {case} 200 50   (MI355X, round 5: 50 warm-up + 200 timed calls)\n")
        f.write(f"# un-profiled HIP-event timing of the same command, same lease: {plain}\n")
        f.write(f"# per call of the case, from the kernel trace: average over all launches {call_all:.1f} us | after the 50 warm-up calls: average {call_warm:.1f} us, median {call_med:.1f} us\n")
        f.write(stats_table(case + "_stats"))
        if perk:
            f.write("\n# rocprofv3 --pmc, separate passes, average per launch:\n")
            for k, cs in perk.items():
                f.write(k + "\n")
                for c_, v in sorted(cs.items()):
                    f.write(f"    {c_:34s} {v:18.4f}\n" if isinstance(v, float) and v < 10 else f"    {c_:34s} {v:18.1f}\n")
json.dump(summary, open(os.path.join(out, "r05_kernels.json"), "w"), indent=1)
if find("bench_stats", "*kernel_stats.csv"):
    with open(os.path.join(out, "r05_bench_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-table   (MI355X, round 5; 7 steps)\n")
        f.write("# side-stream kernels (weight gradients, reductions, column sums, keep-bit generator, AdamW) overlap the main chain: durations are\n"
                "# inflated by sharing the chip and their sum exceeds the wall time\n")
        f.write(stats_table("bench_stats", top=32))
print("wrote", sorted(os.listdir(out)))
