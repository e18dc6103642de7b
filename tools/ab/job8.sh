timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/j8_gpu_suite.txt
cat gpurun_out/j8_gpu_suite.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/j8_bench.json 2> gpurun_out/j8_bench.err
tail -c 2500 gpurun_out/j8_bench.json; tail -5 gpurun_out/j8_bench.err
