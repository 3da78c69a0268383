# Round 5: which resource the fp64 kernels run out of — the shipped build next to the gather-only (-DSPHMI_DIAG=1) and arithmetic-only (=2) builds,
# fp64 handles at C3 and at the reference's published size, compiled-in and run-time models.  (prebuilt: tools/prebuild_variants.py "base:" "d1:-DSPHMI_DIAG=1" "d2:-DSPHMI_DIAG=2")
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for dp in 0.00425 0.0085; do for model in default laminar; do for v in base d1 d2; do
  SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $dp $model 8 120 2>/dev/null | tail -1
done; done; done; done > gpurun_out/r05/fp64_bounds.txt 2>&1
cat gpurun_out/r05/fp64_bounds.txt
