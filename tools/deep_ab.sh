# Round 5: one more pair in flight (kDeep) — fp64 step times at four sizes / two models, and the fp32 bench, A/B builds prebuilt as deep0 / deep1 / deep3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for dp in 0.00425 0.0085 0.0145; do for model in default laminar; do for v in deep0 deep1; do
  SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $dp $model 8 120 2>/dev/null | tail -1
done; done; done; done > gpurun_out/r05/deep_ab_fp64.txt 2>&1
cat gpurun_out/r05/deep_ab_fp64.txt
python tools/bench_libs.py 2 deep0 deep3 > gpurun_out/r05/deep_ab_fp32.txt 2>&1; cat gpurun_out/r05/deep_ab_fp32.txt
