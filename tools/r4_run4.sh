python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest2.log 2>&1; tail -4 gpurun_out/r4_gputest2.log
python bench.py > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err; echo "bench rc=$?"
IRIS_BENCH_FORCE_PG=1 python bench.py --steps 4 --warmup 1 --no-roofline --no-extras --cpu-seconds 0 --gather gather > gpurun_out/r4_bench_pg_gather.json 2> gpurun_out/r4_bench_pg_gather.err; echo "pg gather rc=$?"
IRIS_BENCH_FORCE_PG=1 python bench.py --steps 4 --warmup 1 --no-roofline --no-extras --cpu-seconds 0 --gather all_gather > gpurun_out/r4_bench_pg_allgather.json 2> gpurun_out/r4_bench_pg_allgather.err; echo "pg all_gather rc=$?"
IRIS_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --no-roofline --no-extras --cpu-seconds 0 --parity-pixels 0 > gpurun_out/r4_bench_gloo2.json 2> gpurun_out/r4_bench_gloo2.err; echo "gloo2 rc=$?"
