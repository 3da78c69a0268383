# Round 5, final tree, one gpurun call: the driver's bench command and the default one (full lines with the secondary objects), more fuzz generations
# (default, slabs, forced waves per tile), the sizes table and the example step times.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/r05/bench_final_driver_window.json; cut -c1-260 gpurun_out/r05/bench_final_driver_window.json
python bench.py 2>/dev/null | grep '^{' > gpurun_out/r05/bench_final.json; cut -c1-260 gpurun_out/r05/bench_final.json
for s in 161000 162000 163000 164000; do echo "=== seed0 $s default"; SPHMI_FUZZ_SEED0=$s timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed"; done > gpurun_out/r05/fuzz_generations_final_tree.txt 2>&1
echo "=== seed0 165000 four waves per tile forced" >> gpurun_out/r05/fuzz_generations_final_tree.txt; SPHMI_WPT=4 SPHMI_FUZZ_SEED0=165000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> gpurun_out/r05/fuzz_generations_final_tree.txt
echo "=== seed0 166000 eight waves per tile forced" >> gpurun_out/r05/fuzz_generations_final_tree.txt; SPHMI_WPT=8 SPHMI_FUZZ_SEED0=166000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> gpurun_out/r05/fuzz_generations_final_tree.txt
echo "=== seed0 167000 one wave per tile forced" >> gpurun_out/r05/fuzz_generations_final_tree.txt; SPHMI_WPT=1 SPHMI_FUZZ_SEED0=167000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> gpurun_out/r05/fuzz_generations_final_tree.txt
echo "=== seed0 168000 mailbox exchange" >> gpurun_out/r05/fuzz_generations_final_tree.txt; SPHMI_EXCHANGE=mailbox SPHMI_FUZZ_SEED0=168000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" >> gpurun_out/r05/fuzz_generations_final_tree.txt
cat gpurun_out/r05/fuzz_generations_final_tree.txt
python tools/bench_examples.py 3000 2>&1 | grep -v "^\[" > gpurun_out/r05/examples_final.txt; cut -c1-100 gpurun_out/r05/examples_final.txt
python tools/time_sizes.py 0.02 0.0145 0.0115 0.0085 0.0065 0.0057 0.005 0.003 0.002125 > gpurun_out/r05/sizes_final.txt 2>&1; cat gpurun_out/r05/sizes_final.txt
