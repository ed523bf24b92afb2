mkdir -p gpurun_out/r03r
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --force-sharded-extras --steps 5 --warmup 2 > gpurun_out/r03r/bench_dist1.json 2> gpurun_out/r03r/bench_dist1.err
python bench.py --gpus 1 --force-plan --force-sharded-extras --steps 5 --warmup 2 > gpurun_out/r03r/bench_plan1.json 2> gpurun_out/r03r/bench_plan1.err
cut -c1-200 gpurun_out/r03r/bench_dist1.json; tail -3 gpurun_out/r03r/bench_dist1.err; cut -c1-200 gpurun_out/r03r/bench_plan1.json; tail -3 gpurun_out/r03r/bench_plan1.err
