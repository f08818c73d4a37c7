"""Per-kernel register / spill table from a hipcc -Rpass-analysis=kernel-resource-usage report (stderr of the compile)."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in txt.split("Function Name: ")[1:]:
    name = b.split()[0]
    if pat and pat not in name: continue
    f = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(f"{name[:110]:110s} sgpr {f('TotalSGPRs'):>3s} vgpr {f('VGPRs'):>3s} agpr {f('AGPRs'):>3s} scratch {f('ScratchSize .bytes/lane.'):>4s} sspill {f('SGPRs Spill'):>3s} vspill {f('VGPRs Spill'):>3s}")
