timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_tests3.log 2>&1; grep -E "passed|failed|error|exact_shade=" gpurun_out/r03_tests3.log | tail -12
python bench.py --cpu-frames 0 > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err || tail -20 gpurun_out/r03_bench_b.err
