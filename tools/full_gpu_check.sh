# The driver's round-end sequence on one box: GPU suite, smoke, then an optional A/B of environment switches on the examples.
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
(time SPHMI_PERF_GUARDS=1 timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8) > gpurun_out/suite.log 2>&1
tail -12 gpurun_out/suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
if [ -n "$1" ]; then bash tools/ab_env.sh "$1" > gpurun_out/ab_env_stdout.txt 2>&1; grep -E "==|fp" gpurun_out/ab_env.txt | head -60; fi
