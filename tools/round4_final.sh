# Round 4, final tree, one gpurun call: GPU suite, smoke, kernel stats, counters, fresh fuzz generations, the instantiation matrix against round 3's tree, examples and sizes.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
(time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r4h_suite.log 2>&1; head -3 gpurun_out/r4h_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile_round.sh r04h > gpurun_out/r4h_profile_round.txt 2>&1; tail -1 gpurun_out/r4h_profile_round.txt | cut -c1-200
bash tools/profile_round.sh r04h_driver --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r4h_profile_driver.txt 2>&1
bash $R/tools/pmc_passes.sh r4h_pmc_final > $R/gpurun_out/r4h_pmc_passes_final.txt 2>&1; grep -c "^###" $R/gpurun_out/r4h_pmc_passes_final.txt
cd $R
for s in 91000 92000; do echo "=== seed0 $s default"; SPHMI_FUZZ_SEED0=$s timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3; done > gpurun_out/r4h_fuzz_gen.txt 2>&1
echo "=== seed0 93000 four waves per tile forced" >> gpurun_out/r4h_fuzz_gen.txt; SPHMI_WPT=4 SPHMI_FUZZ_SEED0=93000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r4h_fuzz_gen.txt
echo "=== seed0 94000 two waves per tile forced" >> gpurun_out/r4h_fuzz_gen.txt; SPHMI_WPT=2 SPHMI_FUZZ_SEED0=94000 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r4h_fuzz_gen.txt
grep -E "===|passed|failed" gpurun_out/r4h_fuzz_gen.txt
bash tools/variants_vs_previous.sh build/r3tree gpurun_out/r4h_variants_vs_round3.md | tail -3
python tools/bench_examples.py 2000 2>&1 | grep -v "^\[" > gpurun_out/r4h_examples.txt; cat gpurun_out/r4h_examples.txt
python tools/time_sizes.py 0.02 0.0175 0.0145 0.0125 0.0115 0.0105 0.0085 0.0075 0.0065 0.0057 0.005 0.003 0.002125 > gpurun_out/r4h_sizes.txt 2>&1; cat gpurun_out/r4h_sizes.txt
