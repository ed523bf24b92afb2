#!/bin/bash
# Run on the GPU box (through gpurun): everything profiles/ needs from ONE lease - the PMC passes + kernel statistics
# (tools/collect_pmc.sh), the default bench line, the loopback-8 line (strong-scaling entries), the StringLength
# shapes, the PDM shapes, the Supersmoother timings and the bootstrap end-to-end comparison.   tools/collect_round.sh gpurun_out/r06_final
out=${1:-gpurun_out/round}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
bash tools/collect_pmc.sh "$out/pmc" > "$out/pmc.log" 2>&1
cp "$out/pmc/pmc_summary.json" profiles/r06_pmc_summary.json     # bench.py reads it from there (this lease only)
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
python3 bench.py --loopback 8 > "$out/bench_loopback8.json" 2> "$out/bench_loopback8.err"
(LARGE=1 python3 tools/sl_shapes.py; SHAPES=29000x300,50000x1000,74326x1000,60000x20000,131000x8192,250000x4096,300000x4096,400000x2048,1000000x8192,2000000x512,3000000x256 python3 tools/sl_shapes.py; echo "== lists mode only (PDC_SL_SLICES=0: what samples in any order got up to round 4)"; PDC_SL_SLICES=0 SHAPES=300000x4096,400000x2048,1000000x2048,2000000x512 python3 tools/sl_shapes.py; echo "== samples in a random order (ordered by time on the device first: timesort.inc)"; SHUFFLE=1 SHAPES=300000x4096,400000x2048,1000000x2048,2000000x512 python3 tools/sl_shapes.py; echo "== no streamed kernels, no bit planes (round-3 dispatch)"; PDC_SL_STREAM=0 PDC_SL_P17=0 LARGE=1 python3 tools/sl_shapes.py) > "$out/sl_shapes.txt" 2>&1
python3 tools/bootstrap_e2e.py 20000 1000 > "$out/bootstrap_e2e.txt" 2>&1
python3 tools/pdm_shapes.py > "$out/pdm_shapes.txt" 2>&1
SHAPES=2000x10000,4096x20000,10000x8192,50000x4096,74326x2048,200000x512,1000000x96 python3 tools/ss_timing.py > "$out/ss_timing.txt" 2>&1
python3 tools/peaks_timing.py > "$out/peaks_timing.txt" 2>&1
# where a row of peaks_topk_kernel spends its time: stamps of all rows from a -DPDC_PK_DBG build (built here, removed again)
(bash tools/ab_build.sh pkdbg "-DPDC_PK_DBG=1" peaks.hip && for a in "4 1" "4 0" "1 0"; do PDC_LIBRARY=periodicity_amd/libpdc_ab_pkdbg.so python3 tools/peaks_stamps.py $a; done; rm -f periodicity_amd/libpdc_ab_pkdbg.so) > "$out/peaks_stamps.txt" 2>&1
# bench.py launched the way the driver launches --gpus N (one rank per GPU under torch.distributed.run), at one rank: over RCCL,
# and with the injected communicator failure through the gloo fallback
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --force-dist --force-sharded-extras --steps 5 --warmup 2 > "$out/bench_dist1_rccl.json" 2> "$out/bench_dist1_rccl.err"
PDC_FORCE_RCCL_FAIL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --force-dist --force-sharded-extras --steps 5 --warmup 2 > "$out/bench_dist1_gloo_fallback.json" 2> "$out/bench_dist1_gloo_fallback.err"
# a longer randomised parity run at these sources (bounded: ~12 minutes)
(timeout 900 python3 tools/fuzz_gpu.py --seeds ${FUZZ_SEEDS:-24} --start 60000 2>&1 | tail -4; timeout 300 python3 tools/fuzz_peaks.py --cases 400 --seed 6 2>&1 | tail -2; timeout 300 python3 tools/fuzz_sl.py --cases 100 --seed 6 2>&1 | tail -2) > "$out/fuzz.txt" 2>&1
cut -c1-160 "$out/bench_n1.json"
