mkdir -p gpurun_out/r03q
bash tools/collect_pmc.sh gpurun_out/r03q/pmc > gpurun_out/r03q/collect.log 2>&1
cp gpurun_out/r03q/pmc/pmc_summary.json profiles/r03_pmc_summary.json
python bench.py > gpurun_out/r03q/bench_n1.json 2> gpurun_out/r03q/bench_n1.err
python bench.py --loopback 3 --steps 5 --warmup 2 > gpurun_out/r03q/bench_loop3.json 2> gpurun_out/r03q/bench_loop3.err
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r03q/tests.log
cat gpurun_out/r03q/tests.log; tail -2 gpurun_out/r03q/collect.log; cut -c1-200 gpurun_out/r03q/bench_n1.json
