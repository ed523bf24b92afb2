#!/bin/bash
# Run on the GPU box (through gpurun): one rocprofv3 pass per counter set over the bench command,
# plus a plain --kernel-trace --stats pass; then tools/pmc_summary.py merges them.
#   tools/collect_pmc.sh gpurun_out/r02_pmc [bench args...]
# Counter sets are kept apart because of the per-block slot limits (SQ 8, TCC 4: FETCH_SIZE takes 3,
# WRITE_SIZE 2) - see /opt/skills/guides/MI355X_MICROARCH.md.
set -u
out=$1; shift
args=${*:---steps 3 --warmup 1 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
pass() {
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace -d "$out/$name" -o run --output-format csv -- python3 bench.py $args > "$out/$name.log" 2>&1
    tail -1 "$out/$name.log" | cut -c1-200
}
pass valu  SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
# the VALU mix by instruction type (round 4): fp64 arithmetic issues a wave64 in 4 cycles on the SIMD-32, 32-bit
# VALU in 2 - bench.py prices the executed-issue fraction with these instead of 4 cycles for everything
pass mix64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS
pass mix32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass lds   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2    TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- python3 bench.py $args > "$out/stats.log" 2>&1
# the headline workload alone, 20 timed steps + 3 warm-up, so that the statistics of gls_scan_kernel are
# the C2 launch's; the bench line of THIS run (same lease, same process) is kept beside the trace
rocprofv3 --kernel-trace --stats -d "$out/stats_c2" -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$out/stats_c2.log" 2>&1
grep '^{"metric"' "$out/stats_c2.log" | tail -1 > "$out/bench_under_kernel_trace.json"
python3 tools/kernel_median.py "$out/stats_c2/run_kernel_trace.csv" "$out/bench_c2_kernel_median.json"
python3 tools/pmc_summary.py "$out" --out "$out/pmc_summary.json" --command "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py $args"
