#!/bin/bash
# Run on the GPU box (through gpurun): one rocprofv3 --pmc pass per counter set over ONE command, summed per kernel.
#   tools/pmc_kernels.sh gpurun_out/ss_pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" -- python3 tools/ss_timing.py
# (counter sets in separate passes: per-block slot limits, MI355X_MICROARCH.md; no env / bash -c hop before the program)
out=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "${sets[@]}"; do
    rocprofv3 --pmc $set --kernel-trace -d "$out/p$i" -o run --output-format csv -- "$@" > "$out/p$i.log" 2>&1
    i=$((i + 1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            if f.find("/p0/") >= 0:
                calls[k] += 1
names = sorted({c for k in tot for c in tot[k]})
print("kernel".ljust(60), "calls", *[n[:18].rjust(18) for n in names])
for k in sorted(tot, key=lambda k: -sum(tot[k].values()))[:14]:
    print(k.ljust(60), str(calls[k]).rjust(5), *[f"{tot[k].get(n, 0):18.4g}" for n in names])
PY
