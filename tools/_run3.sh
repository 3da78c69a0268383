mkdir -p gpurun_out/s3
python tools/bench_libs.py 3 pipe1 pf1 pf2 pf3 pf3c1 pf3c2 > gpurun_out/s3/pf.txt 2>&1
cat gpurun_out/s3/pf.txt
AMD_LOG_LEVEL=1 timeout 1500 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV nostop noprint pass" -ex run -ex bt -ex "info threads" --args python -m pytest tests -m gpu -q -x -p no:faulthandler > gpurun_out/s3/gdb.log 2>&1; echo "rc=$?" >> gpurun_out/s3/gdb.log
grep -v "^\[New Thread\|^\[Thread\|^Extension" gpurun_out/s3/gdb.log | tail -60
