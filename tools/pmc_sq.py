"""tools/pmc_sq.py <rocprofv3 --pmc output dir> <out.txt>: per-kernel SQ counter summary of one bench step (tools/profile_round.sh).

One rocprofv3 pass with the 8 SQ slots + GRBM_GUI_ACTIVE (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); no trace shares the run.  Per kernel
(launch-count weighted sums over the step):
  mfma%   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)   (the gfx94x MfmaUtil formula - no gfx950 section ships -
            with GRBM_GUI_ACTIVE taken per XCD: rocprofv3 reports the sum over the 8 XCDs, 8 x clock x wall time)
  wait%   = SQ_WAIT_ANY / SQ_WAVE_CYCLES            waves parked on s_waitcnt / s_barrier
  stall%  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES       issue stalls (MFMA dependency, busy pipes)
  issue%  = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  ldsconf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  GHz     = GRBM_GUI_ACTIVE / 8 / kernel wall time  effective shader clock while the kernel ran (launches of a few 10 us: GUI_ACTIVE also
            counts the dispatch lead-in, read those rows' mfma% / GHz with care)
"""
import csv
import glob
import os
import re
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def short(name):
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    m = re.match(r"\(anonymous namespace\)::(\w+)", name) or re.match(r"(\w+?_kernel)", name)
    base = m.group(1) if m else name[:40]
    if base in ("gemm_kernel", "gemm_mid_kernel", "gemm_pc_kernel", "gemm_p8_kernel", "gemm_p8p_kernel"):
        tb = ["T" if b == "1" else "N" for b in re.findall(r"Lb([01])E", name)]
        tile = re.search(r"TileCfgILi(\d+)ELi(\d+)", name)
        last = re.search(r"ELi(\d+)EEEvNS", name)
        orient = "".join(tb) if len(tb) == 2 else "B=" + tb[0]          # the mid kernel only has the B-operand switch
        base += "<" + orient + (f",{tile.group(1)}x{tile.group(2)}" if tile else "")
        base += (f",act{last.group(1)}" if last and base != "gemm_pc_kernel" else "") + ">"
    return base


def main():
    d, out = sys.argv[1:3]
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, {"_disp": set(), "_ns": 0.0})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r["Dispatch_Id"] not in a["_disp"]:
                a["_disp"].add(r["Dispatch_Id"])
                a["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = []
    for k, a in acc.items():
        g = lambda n: a.get(n, 0.0)
        wc, gui = g("SQ_WAVE_CYCLES"), g("GRBM_GUI_ACTIVE")
        if not wc or not gui:
            continue
        rows.append((a["_ns"], k, len(a["_disp"]), 100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / XCDS * CUS * SIMDS), 100.0 * g("SQ_WAIT_ANY") / wc,
                     100.0 * g("SQ_WAIT_INST_ANY") / wc, 100.0 * g("SQ_ACTIVE_INST_ANY") / wc,
                     g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), gui / XCDS / max(a["_ns"], 1.0)))
    rows.sort(reverse=True)
    with open(out, "w") as fh:
        fh.write(__doc__.split("\n\n", 1)[1])
        fh.write(f"\n{'kernel':58s} {'n':>5s} {'ms':>8s} {'mfma%':>6s} {'wait%':>6s} {'stall%':>6s} {'issue%':>6s} {'ldsconf':>7s} {'GHz':>5s}\n")
        for ns, k, n, mf, wa, st, iss, lc, ghz in rows[:40]:
            fh.write(f"{k[:58]:58s} {n:5d} {ns / 1e6:8.2f} {mf:6.1f} {wa:6.1f} {st:6.1f} {iss:6.1f} {lc:7.3f} {ghz:5.2f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
