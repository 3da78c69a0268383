# Round 5, one gpurun call: fresh fuzz generations on the final kernels (default and with the waves per tile forced) and every kernel
# instantiation next to round 4's tree (build/r4tree = `git archive 4dccbc0`, library built in place).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for s in 101000 102000 103000; do echo "=== seed0 $s default"; SPHMI_FUZZ_SEED0=$s timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3; done > gpurun_out/r05/fuzz_generations.txt 2>&1
echo "=== seed0 104000 four waves per tile forced" >> gpurun_out/r05/fuzz_generations.txt; SPHMI_WPT=4 SPHMI_FUZZ_SEED0=104000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r05/fuzz_generations.txt
echo "=== seed0 105000 eight waves per tile forced" >> gpurun_out/r05/fuzz_generations.txt; SPHMI_WPT=8 SPHMI_FUZZ_SEED0=105000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r05/fuzz_generations.txt
echo "=== seed0 106000 two waves per tile forced" >> gpurun_out/r05/fuzz_generations.txt; SPHMI_WPT=2 SPHMI_FUZZ_SEED0=106000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r05/fuzz_generations.txt
grep -E "===|passed|failed" gpurun_out/r05/fuzz_generations.txt
bash tools/variants_vs_previous.sh build/r4tree gpurun_out/r05/variants_vs_round4.md | tail -3
