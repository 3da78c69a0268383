mkdir -p gpurun_out/s2
python -m pytest "tests/test_engine_gpu.py" -k "device_side_case_generator" -x -q -s > gpurun_out/s2/gen.log 2>&1; echo "rc=$?" >> gpurun_out/s2/gen.log
AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=1 python -m pytest "tests/test_engine_gpu.py" -k "device_side_case_generator and 0.0085" -x -q -s > gpurun_out/s2/gen_ser.log 2>&1; echo "rc=$?" >> gpurun_out/s2/gen_ser.log
python tools/bench_libs.py 3 pipe0 pipe1 > gpurun_out/s2/pipe.txt 2>&1
python tools/bench_libs.py 2 pipe0 pipe1 -- --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s2/pipe_w20.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --deselect "tests/test_engine_gpu.py::test_device_side_case_generator" > gpurun_out/s2/gputest_rest.log 2>&1; echo "rc=$?" >> gpurun_out/s2/gputest_rest.log
tail -3 gpurun_out/s2/gen.log; cat gpurun_out/s2/pipe.txt; tail -3 gpurun_out/s2/gputest_rest.log
