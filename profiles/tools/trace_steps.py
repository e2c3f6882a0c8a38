#!/usr/bin/env python3
"""Per-step, per-queue summary of a rocprofv3 --kernel-trace CSV of bench.py: finds the embedding-forward launches (one per step), cuts
the trace into steps and prints, for the chosen step, wall time, per-queue busy time, time with >= 2 kernels in flight, and the
per-kernel totals split by queue.   usage: trace_steps.py <kernel_trace.csv> [step_index]"""
import csv
import collections
import sys


def short(n):
    n = n.replace("void ", "")
    for k in ("gemm_pring_kernel", "gemm_ring_kernel", "gemm_mfma_kernel", "attn_bwd_dkv", "attn_bwd_dq", "attn_fwd", "ln_bwd_kernel",
              "ln_fwd_kernel", "splitk_reduce", "colsum_kernel", "adamw_kernel", "embed_bwd", "embed_fwd", "ce_vec", "gather_rows",
              "scatter_rows", "cast_kernel", "dact", "mask_build", "mask_tileinfo", "pack_plan", "FillFunctor", "attn_bwd_fused", "gemm_p8"):
        if k in n:
            if k.startswith("gemm_"):
                return n[n.index(k):].split("(")[0][:60]
            return k
    return n[:50]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"],
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    # one embedding-forward launch per step (the optimizer may run as one kernel or range by range on the side stream)
    opt = [i - 1 for i, r in enumerate(rows) if "embed_fwd_kernel" in r[3] and i > 0]
    k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(opt) // 2 - 1
    lo, hi = (opt[k - 1] + 1 if k > 0 else 0), opt[k] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][0], step[-1][1]
    print(f"steps found: {len(opt)}; step {k}: {len(step)} launches, wall {(t1 - t0) / 1e6:.2f} ms")
    ev = []
    for s, e, q, n, g in step:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, busy, multi = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    print(f"any kernel running {busy / 1e6:.2f} ms, >= 2 running {multi / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
    perq = collections.defaultdict(float)
    perk = collections.defaultdict(lambda: [0, 0.0])
    for s, e, q, n, g in step:
        perq[q] += (e - s) / 1e6
        key = (q, short(n))
        perk[key][0] += 1
        perk[key][1] += (e - s) / 1e3
    print("busy ms per queue:", {q: round(v, 2) for q, v in perq.items()})
    if "--gaps" in sys.argv:
        print("largest intervals with NO kernel running (us), with the kernel that ended before and the one that started after:")
        iv = sorted(step, key=lambda r: r[0])
        gaps, end, last = [], iv[0][1], iv[0]
        for r in iv[1:]:
            if r[0] > end:
                gaps.append((r[0] - end, short(last[3]), short(r[3]), (end - t0) / 1e6))
            if r[1] > end:
                end, last = r[1], r
        for g_, a_, b_, at in sorted(gaps, reverse=True)[:14]:
            print(f"   {g_ / 1e3:7.1f} us at {at:6.2f} ms  after {a_[:44]:44s} before {b_[:44]}")
        print(f"   ({len(gaps)} gaps, {sum(g[0] for g in gaps) / 1e6:.2f} ms in total)")
    if "--timeline" in sys.argv:
        # every launch of the step in start order: start / end (ms from the step's first launch), queue, kernel -- where the two queues wait for each other
        print("timeline (start ms, end ms, queue, workgroups, kernel):")
        for s_, e_, q, n, g_ in sorted(step, key=lambda r: r[0]):
            print(f"   {(s_ - t0) / 1e6:8.3f} {(e_ - t0) / 1e6:8.3f}  q{q:<3} {g_:6d}  {short(n)}")
    if "--small" in sys.argv:
        print("launches with fewer than 512 workgroups that run longer than 25 us (under-occupied kernels):")
        for s_, e_, q, n, g_ in sorted(step, key=lambda r: r[0] - r[1]):
            if g_ < 512 and (e_ - s_) > 25000:
                print(f"   queue {q} {short(n):60s} workgroups {g_:5d}  {(e_ - s_) / 1e3:8.1f} us")
    print(f"{'queue':>5} {'kernel':60s} {'calls':>5} {'total_us':>10} {'avg_us':>8}")
    for (q, n), (c, t) in sorted(perk.items(), key=lambda x: -x[1][1]):
        print(f"{q:>5} {n:60s} {c:5d} {t:10.1f} {t / c:8.1f}")


if __name__ == "__main__":
    main()
