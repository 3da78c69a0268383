cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8) > gpurun_out/suite.log 2>&1
tail -6 gpurun_out/suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 100 --warmup 10 > gpurun_out/r4b_bench_100.json 2> gpurun_out/r4b_bench_100.err; cut -c1-400 gpurun_out/r4b_bench_100.json
python bench.py > gpurun_out/r4b_bench_default.json 2> gpurun_out/r4b_bench_default.err; cut -c1-400 gpurun_out/r4b_bench_default.json
