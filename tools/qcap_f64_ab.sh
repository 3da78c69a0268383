# Round 5: queue depth of the fp64 kernels (16 shipped) with the lane-pair fetch, prebuilt f16q / f10q / f8q / f6q
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for dp in 0.00425 0.0085 0.0115 0.02; do for v in f16q f10q f8q f6q; do
  SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $dp default 8 120 2>/dev/null | tail -1
done; done; done > gpurun_out/r05/qcap_f64_ab.txt 2>&1
sort gpurun_out/r05/qcap_f64_ab.txt
