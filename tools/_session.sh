mkdir -p gpurun_out/s15
L=sbwt_amd/lib
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py tests/test_gpu_config3.py -x -q 2>&1 | tail -5) > gpurun_out/s15/tests.log 2>&1
(timeout 400 python tools/fuzz_gpu.py 150 2>&1 | tail -5) > gpurun_out/s15/fuzz.log 2>&1
run3() { SBWTGPU_LIB=$L/libsbwtgpu$1.so SBWTGPU_SEARCH_VARIANT=$2 python bench.py --config 3 --steps 3 --warmup 1 > gpurun_out/s15/c3$1_v$2.json 2> gpurun_out/s15/c3$1_v$2.err; }
run3 _base 2
run3 "" 2
run3 "" 4
run3 _A 2
run3 _A 4
cat gpurun_out/s15/tests.log gpurun_out/s15/fuzz.log
