# Round 6, final tree, one gpurun call: the driver's bench command and the default one (full lines with the secondary objects), fresh fuzz and
# call-sequence generations (default, forced waves per tile, mailbox exchange, the RCCL double), the sizes table and the example step times.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/r06/bench_final_driver_window.json; cut -c1-260 gpurun_out/r06/bench_final_driver_window.json
python bench.py 2>/dev/null | grep '^{' > gpurun_out/r06/bench_final.json; cut -c1-260 gpurun_out/r06/bench_final.json
F=gpurun_out/r06/fuzz_generations_final_tree.txt; : > $F
for s in 261000 262000 263000; do echo "=== seed0 $s default" >> $F; SPHMI_FUZZ_SEED0=$s timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> $F; done
for w in 1 4 8; do echo "=== seed0 26${w}500 $w waves per tile forced" >> $F; SPHMI_WPT=$w SPHMI_FUZZ_SEED0=26${w}500 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> $F; done
echo "=== seed0 268000 mailbox exchange" >> $F; SPHMI_EXCHANGE=mailbox SPHMI_FUZZ_SEED0=268000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> $F
for s in 1000 2000 3000 4000; do echo "=== call sequences, seed0 $s" >> $F; SPHMI_SEQ_SEED0=$s timeout 600 python -m pytest tests/test_api_sequence_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> $F; done
cat $F
python tools/bench_examples.py 3000 2>&1 | grep -v "^\[" > gpurun_out/r06/examples_final.txt; cut -c1-100 gpurun_out/r06/examples_final.txt
python tools/time_sizes.py 0.02 0.0145 0.0115 0.0085 0.0065 0.0057 0.005 0.003 0.002125 > gpurun_out/r06/sizes_final.txt 2>&1; cat gpurun_out/r06/sizes_final.txt
