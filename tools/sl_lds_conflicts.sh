#!/bin/bash
# LDS bank-conflict cycles of the StringLength kernel at C5, phase by phase (on the GPU box, through gpurun):
# the product library and two experiment builds that stop every period after P1 / after P2
#   tools/ab_build.sh slstop1 "-DPDC_SL_STOP=1" stringlength.hip ; tools/ab_build.sh slstop2 "-DPDC_SL_STOP=2" stringlength.hip
#   gpurun -- bash tools/sl_lds_conflicts.sh gpurun_out/sl_lds
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
export SHAPES=50000x100000
for v in slstop1 slstop2 full; do
    if [ $v = full ]; then unset PDC_LIBRARY; else export PDC_LIBRARY=periodicity_amd/libpdc_ab_$v.so; fi
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
        --kernel-trace -d "$out/$v" -o run --output-format csv -- python3 tools/sl_shapes.py > "$out/$v.log" 2>&1
    grep "N= 50000" "$out/$v.log"
done
python3 - "$out" <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
res = {}
for v in ("slstop1", "slstop2", "full"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    with open(f"{out}/{v}/run_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if "sl_fast_kernel" not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    res[v] = {k: sorted(d.values())[len(d) // 2] for k, d in acc.items()}   # median dispatch
    r = res[v]
    print(v, {k: f"{x:.4g}" for k, x in r.items()},
          "conflict share of LDS cycles %.3f" % (r["SQ_LDS_BANK_CONFLICT"] / r["SQ_LDS_IDX_ACTIVE"]))
json.dump(res, open(f"{out}/sl_lds_conflicts.json", "w"), indent=1)
PY
