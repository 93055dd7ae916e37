"""Where a training step's wall time goes that is NOT inside a kernel: reads a rocprofv3 `--kernel-trace` CSV
(`*_kernel_trace.csv`) of `bench.py` and, for the last full training steps, reports

  * the step's wall time (first kernel start -> last kernel end, steps cut at the `match_pass1` launch that opens each one),
  * per hardware queue: busy time, number of launches, and the idle time BETWEEN consecutive kernels of that queue,
  * the union busy time over all queues (any kernel running) and its complement = time the whole chip sat idle,
  * the longest gaps with the kernels on either side.

Usage: python tools/trace_gaps.py <kernel_trace.csv> [--steps 3]"""
import argparse
import csv
import json
import sys
from collections import defaultdict


def load(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                             r.get("Stream_Id", "")))
            except (KeyError, ValueError):
                continue
    rows.sort()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--marker", default="match_pass1", help="kernel that opens a training step")
    a = ap.parse_args()
    rows = load(a.csv)
    starts = [i for i, r in enumerate(rows) if r[2].startswith(a.marker)]
    if len(starts) < a.steps + 1:
        print(json.dumps({"error": f"only {len(starts)} '{a.marker}' launches in the trace"}))
        return 1
    out = []
    for k in range(len(starts) - a.steps - 1, len(starts) - 1):
        seg = rows[starts[k]:starts[k + 1]]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        by_q = defaultdict(list)
        for r in seg:
            by_q[r[3]].append(r)
        queues = {}
        for q, rs in by_q.items():
            busy = sum(r[1] - r[0] for r in rs)
            gaps = [(rs[i + 1][0] - rs[i][1], rs[i][2][:60], rs[i + 1][2][:60]) for i in range(len(rs) - 1)]
            pos = [g for g in gaps if g[0] > 0]
            queues[q] = {"launches": len(rs), "busy_ms": round(busy / 1e6, 3),
                         "idle_between_ms": round(sum(g[0] for g in pos) / 1e6, 3),
                         "median_gap_us": round(sorted(g[0] for g in pos)[len(pos) // 2] / 1e3, 2) if pos else 0.0,
                         "gaps_over_10us": sum(1 for g in pos if g[0] > 10000),
                         "longest": [(round(g[0] / 1e3, 1), g[1], g[2]) for g in sorted(pos, reverse=True)[:5]]}
        # union of busy intervals over all queues
        ev = sorted((r[0], r[1]) for r in seg)
        union, cs, ce = 0, ev[0][0], ev[0][1]
        for s, e in ev[1:]:
            if s > ce:
                union += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        union += ce - cs
        out.append({"step_wall_ms": round((t1 - t0) / 1e6, 3), "to_next_step_ms": round((rows[starts[k + 1]][0] - t0) / 1e6, 3),
                    "kernel_sum_ms": round(sum(r[1] - r[0] for r in seg) / 1e6, 3), "any_kernel_running_ms": round(union / 1e6, 3),
                    "chip_idle_ms": round((t1 - t0 - union) / 1e6, 3), "launches": len(seg), "queues": queues})
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
