mkdir -p gpurun_out/r03p
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03p/smoke.log 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r03p/tests.log
bash tools/collect_pmc.sh gpurun_out/r03p/pmc > gpurun_out/r03p/collect.log 2>&1
cp gpurun_out/r03p/pmc/pmc_summary.json profiles/r03_pmc_summary.json
python bench.py > gpurun_out/r03p/bench_n1.json 2> gpurun_out/r03p/bench_n1.err
python bench.py --loopback 3 --steps 5 --warmup 2 > gpurun_out/r03p/bench_loop3.json 2> gpurun_out/r03p/bench_loop3.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 5 --warmup 2 > gpurun_out/r03p/bench_dist1.json 2> gpurun_out/r03p/bench_dist1.err
cp profiles/r03_pmc_summary.json gpurun_out/r03p/
tail -4 gpurun_out/r03p/smoke.log; tail -3 gpurun_out/r03p/tests.log; tail -2 gpurun_out/r03p/collect.log; cut -c1-250 gpurun_out/r03p/bench_n1.json; tail -2 gpurun_out/r03p/bench_n1.err
