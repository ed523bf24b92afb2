#!/bin/bash
# Two experiment builds of the C5 StringLength kernel against the product (on the GPU box, through gpurun; results of
# the experiment builds are garbage, only time and counters count):
#   tools/ab_build.sh dblg "-DPDC_SL_EXP_DOUBLE_GATHER=1" stringlength.hip - every record of a range gathered twice
#   tools/ab_build.sh swz  "-DPDC_SL_EXP_SWIZZLE=1"    stringlength.hip   - fine-bucket counter words XOR-swizzled
#   gpurun -- bash tools/sl_experiments.sh gpurun_out/sl_exp
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
export SHAPES=50000x100000
for v in product dblg swz; do
    if [ $v = product ]; then unset PDC_LIBRARY; else export PDC_LIBRARY=periodicity_amd/libpdc_ab_$v.so; fi
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum \
        --kernel-trace -d "$out/$v" -o run --output-format csv -- python3 tools/sl_shapes.py > "$out/$v.log" 2>&1
    echo "$v: $(grep 'N= 50000' "$out/$v.log")"
done
python3 - "$out" <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
res = {}
for v in ("product", "dblg", "swz"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(f"{out}/{v}/run_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if "sl_fast_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    res[v] = {k: sorted(d.values())[len(d) // 2] for k, d in acc.items()}   # median dispatch
    r = res[v]
    print(v, f"conflict share {r['SQ_LDS_BANK_CONFLICT'] / r['SQ_LDS_IDX_ACTIVE']:.3f}", f"L2 read requests {r['TCP_TCC_READ_REQ_sum']:.3e}",
          f"GUI_ACTIVE/8 {r['GRBM_GUI_ACTIVE'] / 8:.3e}")
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
