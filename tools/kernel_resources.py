"""Condense hipcc's `-Rpass-analysis=kernel-resource-usage` remarks (stdin) into one line per kernel:
`make -C periodicity_amd/csrc resources` -> profiles/rNN_kernel_resources.txt.  Kernels that spill come first."""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                  r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    key, val = m.groups()
    if key == "Function Name":
        cur = {"file": line.split(":", 1)[0], "name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key.split(" [")[0]] = val
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
except OSError:
    dem = names
for r, d in zip(rows, dem):
    d = d.replace("(anonymous namespace)::", "")
    r["kernel"] = d[:d.find("(")] if "(" in d else d
rows.sort(key=lambda r: (-int(r.get("VGPRs Spill", 0)), -int(r.get("ScratchSize", 0)), r["file"], r["kernel"]))
print("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, every kernel of libperiodicity_hip.so")
print(f"# {'kernel':86s} {'file':18s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch B':>9s} {'occ':>4s} {'LDS B':>7s}")
for r in rows:
    print(f"{r['kernel'][:88]:88s} {r['file']:18s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} {r.get('ScratchSize', '?'):>9s} {r.get('Occupancy', '?'):>4s} {r.get('LDS Size', '?'):>7s}")
print(f"# {len(rows)} kernels, {sum(1 for r in rows if int(r.get('VGPRs Spill', 0)) > 0)} with spilled VGPRs")
