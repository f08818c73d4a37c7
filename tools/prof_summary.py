#!/usr/bin/env python
"""rocprofv3 results (.db rocpd or *_kernel_stats.csv) -> a compact per-kernel table for profiles/."""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)E+v", name)
    if m:
        tag = m.group(2)
        tag = tag.replace("DF16b", "bf16,").replace("DF16_", "f16,").replace("Lb0E", "0,").replace("Lb1E", "1,")
        tag = re.sub(r"Li(\d+)E", r"\1,", tag)
        return "%s<%s>" % (m.group(1), tag.strip(",")[:64])
    return name[:100]


def from_db(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [(short(n), calls, tot * 1e3, avg * 1e3, pct) for n, calls, tot, avg, pct in rows]   # rocpd views are in us


def from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    return out


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}")
    for n, calls, tot, avg, pct in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
        print(f"{n:70s} {calls:7d} {tot / 1e6:10.2f} {avg / 1e3:10.2f} {pct:7.2f}")


if __name__ == "__main__":
    main()
