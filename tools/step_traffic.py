#!/usr/bin/env python3
"""HBM-side traffic of every kernel of the step from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
WRITE_SIZE cannot share a pass; FETCH_SIZE tallies 16-byte-per-lane reads at half their bytes on gfx950 -> doubled; counter unit KiB).

    tools/step_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps in the run> [title]

Prints one row per kernel name (calls, GB read / written per call, ms per call from the dispatch timestamps, TB/s) and the per-step
totals; kernels whose reads are dword loads are over-stated by the doubling (noted in the header)."""
import collections
import csv
import sys


def load(path, counter):
    rows = collections.defaultdict(lambda: [0, 0.0, 0.0])       # name -> [calls, counter sum, ns]
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            e = rows[r["Kernel_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
            e[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows


def main():
    fetch, write, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), float(sys.argv[3])
    title = sys.argv[4] if len(sys.argv) > 4 else ""
    print(title)
    print("read = 2 x FETCH_SIZE (gfx950 tallies 16-byte-per-lane reads at half their bytes; dword readers are over-stated), write = WRITE_SIZE;")
    print("KiB -> bytes; Infinity-Cache hits are counted (MI355X_MICROARCH.md); ms per call from the dispatch timestamps of the FETCH pass.")
    print(f"{'kernel':72s} {'calls/step':>10s} {'GB rd/call':>10s} {'GB wr/call':>10s} {'ms/call':>8s} {'TB/s':>6s} {'GB/step':>8s}")
    tot_r = tot_w = tot_ms = 0.0
    table = []
    for name, (calls, fsum, ns) in fetch.items():
        wsum = write.get(name, [0, 0.0, 0.0])[1]
        rd, wr = 2.0 * fsum * 1024 / 1e9, wsum * 1024 / 1e9
        table.append((rd + wr, name, calls, rd, wr, ns))
        tot_r, tot_w, tot_ms = tot_r + rd, tot_w + wr, tot_ms + ns / 1e6
    for total, name, calls, rd, wr, ns in sorted(table, reverse=True)[:48]:
        ms = ns / 1e6 / calls
        print(f"{name[:72]:72s} {calls / steps:10.1f} {rd / calls:10.3f} {wr / calls:10.3f} {ms:8.3f} {(rd + wr) / calls / ms:6.2f} {total / steps:8.2f}")
    print(f"per step: {tot_r / steps:.1f} GB read + {tot_w / steps:.1f} GB written = {(tot_r + tot_w) / steps:.1f} GB in {tot_ms / steps:.1f} ms of kernel time "
          f"(profiled, serial)")


if __name__ == "__main__":
    main()
