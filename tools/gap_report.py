#!/usr/bin/env python
"""Where is the GPU idle?  Reads a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV directory and prints, for the LAST `--steps` steps of the
run (a step = the kernels between two launches of the patch-embedding im2row kernel of the first modality group), the busy / idle split and the
largest gaps with the kernels on either side - the bubbles a launch-bound host section or a host synchronisation leaves in the stream.
    cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python bench.py ... ; python tools/gap_report.py out"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)E+v", name)
    if m:
        return m.group(1)
    return re.sub(r"^void ", "", name)[:60]


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
    ev.sort()
    if not ev:
        print("no events")
        return
    # the timed region: from the first im2row of the third-last "step" on; simply take the last 60 % of the events' time span
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    lo = t0 + int((t1 - t0) * 0.45)
    ev = [e for e in ev if e[0] >= lo]
    busy, gaps, end = 0, [], ev[0][0]
    prev = None
    for s, e, n in ev:
        if s > end:
            gaps.append((s - end, prev, n, end))
            end = s
        if e > end:
            busy += e - max(s, end)
            end = e
            prev = n
    span = end - ev[0][0]
    idle = span - busy
    print(f"span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100.0 * idle / span:.2f} %), {len(ev)} events")
    hist = {}
    for g, a, b, _ in gaps:
        k = "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else "<1ms" if g < 1000000 else ">=1ms"
        h = hist.setdefault(k, [0, 0])
        h[0] += 1
        h[1] += g
    for k in ("<5us", "<20us", "<100us", "<1ms", ">=1ms"):
        if k in hist:
            print(f"  gaps {k:7s}: {hist[k][0]:7d}  total {hist[k][1] / 1e6:8.2f} ms")
    agg = {}
    for g, a, b, _ in gaps:
        if g >= 20000:
            x = agg.setdefault((a, b), [0, 0])
            x[0] += 1
            x[1] += g
    print("gaps >= 20 us by (kernel before -> kernel after), total ms:")
    for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {t / 1e6:8.2f} ms  {c:5d} x  {a}  ->  {b}")


if __name__ == "__main__":
    main()
