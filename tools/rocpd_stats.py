#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (SQLite) result: per-kernel calls / total / average / min / max (ns)
plus register and LDS use -- the '--stats' view as a small text table for profiles/.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [skip_first_n_per_kernel | -last_n] > profiles/rNN_x.txt

A negative argument keeps only the LAST n launches of each kernel (bench.py's timed steps, leaving out its
clock spin-up and warm-up launches).
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = db.execute("select name, duration, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, start "
                      "from kernels order by start").fetchall()
    per = {}
    for name, dur, vg, sg, lds, gx, wx, _ in rows:
        per.setdefault(name, []).append((dur, vg, sg, lds, gx, wx))
    print("%-48s %6s %14s %12s %12s %12s %5s %5s %7s %9s %5s" %
          ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "vgpr", "sgpr", "lds", "grid", "wg"))
    for name, v in sorted(per.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
        w = v[skip:] if (skip < 0 or len(v) > skip) else v
        d = [x[0] for x in w]
        print("%-48s %6d %14d %12.0f %12d %12d %5d %5d %7d %9d %5d" %
              (name[:48], len(d), sum(d), sum(d) / len(d), min(d), max(d), w[0][1], w[0][2], w[0][3], w[-1][4], w[-1][5]))
    if skip > 0:
        print("(first %d launches of each kernel skipped as warm-up)" % skip)
    if skip < 0:
        print("(last %d launches of each kernel: the timed steps; spin-up and warm-up launches left out)" % -skip)


if __name__ == "__main__":
    main()
