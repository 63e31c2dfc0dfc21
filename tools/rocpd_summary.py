#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel table (calls, total, avg, min, max, %).

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.kernel_stats.md
"""
import re
import sqlite3
import sys


def short_name(name):
    """Demangled kernel name without its ARGUMENT list: the last balanced (...) group is dropped, template arguments stay (the modes
    of one kernel template are different kernels), and "(anonymous namespace)::" is removed first - round 3 cut at the FIRST "(",
    which collapsed every `gnnome::(anonymous namespace)::k<...>` into one row."""
    name = name.replace("(anonymous namespace)::", "").strip()
    if name.endswith(")"):
        depth = 0
        for i in range(len(name) - 1, -1, -1):
            depth += name[i] == ")"
            depth -= name[i] == "("
            if depth == 0:
                name = name[:i]
                break
    name = re.sub(r"^void ", "", name).strip()
    return re.sub(r"\s*\[clone .*$", "", name)


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels").fetchall() if _has(c, "kernels") else []
    if not rows:
        rows = c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                         "on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, start, end in rows:
        name = short_name(name or "?")
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = end - start
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# kernel-trace summary of {path.split('/')[-1]} (durations in microseconds)\n")
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{name[:140]}` | {a[0]} | {a[1] / 1e3:.1f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / total:.1f} |")


def _has(c, view):
    return bool(c.execute("select 1 from sqlite_master where name = ?", (view,)).fetchone())


if __name__ == "__main__":
    main(sys.argv[1])
