cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in 1 2 3; do
  for spec in "0.0085 default 4" "0.0115 default 4" "0.0145 default 4" "0.02 default 4" "0.0085 laminar 4" "0.0115 sps+complex 4" "0.0145 default 8" "0.02 default 8" "0.0085 k1.5 4" "0.0115 shifting 4"; do
    for v in pfr0 pfr1; do
      SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $spec 300 2>/dev/null | tail -1
    done
  done
done > gpurun_out/r06/prefetch_ranges_sizes_ab.txt 2>&1
