#!/bin/bash
# Run on the GPU box (through gpurun): per-kernel totals of ONE command under rocprofv3 --kernel-trace --stats.
#   tools/prof_kernels.sh gpurun_out/ss_prof python3 tools/ss_timing.py
# (the program itself follows: no env / bash -c hop between rocprofv3 and it)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- "$@" > "$out/run.log" 2>&1
grep -v "^[WEI]2026" "$out/run.log" | tail -8
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))
if not f:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(f[0])))
print(f"{'kernel':72s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
for r in rows[:int(__import__('os').environ.get('TOP', '16'))]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):6.2f}")
PY
