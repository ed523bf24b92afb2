mkdir -p gpurun_out/r03b
python -m pytest tests/test_multi_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r03b/multi.log
python bench.py > gpurun_out/r03b/bench_n1.json 2> gpurun_out/r03b/bench_n1.err
python bench.py --loopback 3 --steps 5 --warmup 2 > gpurun_out/r03b/bench_loop3.json 2> gpurun_out/r03b/bench_loop3.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --steps 5 --warmup 2 > gpurun_out/r03b/bench_dist1.json 2> gpurun_out/r03b/bench_dist1.err
tools/ubench/gather_rate > gpurun_out/r03b/gather_rate.txt 2>&1
bash tools/collect_pmc.sh gpurun_out/r03b/pmc > gpurun_out/r03b/collect.log 2>&1
tail -3 gpurun_out/r03b/multi.log; cut -c1-300 gpurun_out/r03b/bench_n1.json; tail -3 gpurun_out/r03b/bench_n1.err; cut -c1-600 gpurun_out/r03b/bench_loop3.json; tail -3 gpurun_out/r03b/bench_loop3.err; cut -c1-300 gpurun_out/r03b/bench_dist1.json; tail -5 gpurun_out/r03b/bench_dist1.err; tail -5 gpurun_out/r03b/collect.log
