#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db or *_kernel_stats.csv) into a short text
table: calls, total ms, average us, share.  Usage: tools/rocprof_summary.py <results.db|kernel_stats.csv> [out.txt]"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*?>)?)\(", name)
    if name.startswith("at::native") or "at::native" in name[:40]:
        k = re.search(r"at::native::(?:\(anonymous namespace\)::)?([a-zA-Z_0-9]+)", name)
        f = re.search(r"(sqrt|FillFunctor|MulFunctor|DivFunctor|CUDAFunctor_add|CUDAFunctorOnSelf_add|direct_copy|normal_kernel|CatArray|and_kernel|CompareEq)", name)
        return "torch:" + (k.group(1) if k else "?") + ("/" + f.group(1) if f else "")
    return (m.group(1) if m else name)[:110]


def rows_from_db(path):
    db = sqlite3.connect(path)
    # the top_kernels view reports microseconds
    return [(r[0], int(r[1]), float(r[2]) / 1e3, float(r[3]), float(r[4]))
            for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    src = sys.argv[1]
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    lines = [f"{'kernel':<112} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'share%':>7}"]
    for n, c, t, a, p in sorted(rows, key=lambda r: -r[2]):
        lines.append(f"{short(n):<112} {c:>6} {t:>10.3f} {a:>10.2f} {p:>7.2f}")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
