# Round 5: fp64 lane-pair fetch (-DSPHMI_PAIR_FETCH64=1) against the shipped fp64 kernels, prebuilt pf0 / pf1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for dp in 0.00425 0.0085 0.0115; do for model in default; do for v in pf0 pf1; do
  SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $dp $model 8 120 2>/dev/null | tail -1
done; done; done; done > gpurun_out/r05/pair_fetch64_ab.txt 2>&1
cat gpurun_out/r05/pair_fetch64_ab.txt
