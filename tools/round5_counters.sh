# Round 5, one gpurun call: kernel stats of the bench command (100 steps and the driver's 20-step window) and the counter passes of the
# same kernels → gpurun_out/prof_r05*, gpurun_out/r05_pmc/ ; tools/pmc_derive.py turns the passes into profiles/r05_counters.json.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.txt 2>&1; tail -1 gpurun_out/r05_profile_round.txt | cut -c1-200
bash tools/profile_round.sh r05_driver --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r05_profile_driver.txt 2>&1; tail -1 gpurun_out/r05_profile_driver.txt | cut -c1-200
bash $R/tools/pmc_passes.sh r05_pmc > $R/gpurun_out/r05_pmc_passes.txt 2>&1; grep -c "^###" $R/gpurun_out/r05_pmc_passes.txt
