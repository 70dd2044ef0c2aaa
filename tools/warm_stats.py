#!/usr/bin/env python3
"""Per-kernel statistics of the WARM steps of a profiled bench.py run.

    python tools/warm_stats.py <..._kernel_trace.csv> [steps to keep, default 2] > stats.csv

`rocprofv3 --kernel-trace --stats` sums over the whole process: the first step of a bench.py run materialises ~100 packed weight forms
and their device tables (one small launch and one copy each), which inflates the per-step launch count of a 4-step profile by ~130
copyBuffer and ~25 pack_run launches that a warm step does not have.  This tool cuts the kernel trace into steps at
`data_bn_stats_kernel` (the first kernel of every forward) and aggregates the last N steps only; the output has the columns of
rocprofv3's *_kernel_stats.csv plus `CallsPerStep` / `NsPerStep`, and a final `TOTAL` row.
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(csv.DictReader(open(path)))
    name_k = next(k for k in rows[0] if k.lower() in ("kernel_name", "name"))
    start_k = next(k for k in rows[0] if k.lower().startswith("start"))
    end_k = next(k for k in rows[0] if k.lower().startswith("end"))
    rows.sort(key=lambda r: int(r[start_k]))
    marks = [int(r[start_k]) for r in rows if "data_bn_stats_kernel" in r[name_k]]
    if len(marks) < keep + 1:
        raise SystemExit(f"only {len(marks)} steps in the trace; need {keep + 1} (the last step has no closing mark and is dropped)")
    lo, hi = marks[-keep - 1], marks[-1]
    agg = defaultdict(list)
    for r in rows:
        t = int(r[start_k])
        if lo <= t < hi:
            agg[r[name_k]].append(int(r[end_k]) - t)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "CallsPerStep", "NsPerStep", "MinNs", "MaxNs"])
    tot_calls = tot_ns = 0
    for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(d), sum(d), round(sum(d) / len(d), 1), round(len(d) / keep, 2), round(sum(d) / keep, 1), min(d), max(d)])
        tot_calls += len(d)
        tot_ns += sum(d)
    w.writerow([f"TOTAL ({keep} warm steps, cut at data_bn_stats_kernel)", tot_calls, tot_ns, "", round(tot_calls / keep, 2), round(tot_ns / keep, 1), "", ""])


if __name__ == "__main__":
    main()
