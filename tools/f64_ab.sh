# Round 6: the fp64 lane-pair fetch handed over packet by packet (GROUPS), for half tiles of two waves per half (WPT4) and for the run-time-model kernels (GENERIC):
# step times of fp64 handles, A/B builds prebuilt as f64base / f64g / f64w4g / f64all (tools/prebuild_variants.py), two interleaved repetitions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in 1 2; do
  for spec in "0.00425 default" "0.0085 default" "0.0115 default" "0.0145 default" "0.02 default" "0.0085 k1.5" "0.0085 laminar" "0.0115 sps+complex" "0.0085 shifting" "0.00425 laminar" "0.0145 laminar"; do
    for v in f64base f64g f64w4g f64all; do
      SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $spec 8 120 2>/dev/null | tail -1
    done
  done
done > gpurun_out/r06/f64_ab.txt 2>&1
cat gpurun_out/r06/f64_ab.txt
