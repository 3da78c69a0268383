# Round 5: queue depth of the fp32 compiled-in one-wave-per-half kernels (6 against 10) over the sizes that use them, prebuilt q6 / q10
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2; do for dp in 0.0075 0.0057 0.00425 0.003 0.002125; do for v in q10 q6; do
  SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $dp default 4 150 2>/dev/null | tail -1
done; done; done > gpurun_out/r05/qcap_sizes_ab.txt 2>&1
cat gpurun_out/r05/qcap_sizes_ab.txt
