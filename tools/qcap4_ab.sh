# Round 5: queue depth of the fp32 compiled-in kernels of two waves per half (launches of 330 … 3 000 tiles), prebuilt h12 / h8 / h6
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for rep in 1 2 3; do for dp in 0.0145 0.0115 0.0100 0.0085; do for v in h12 h8 h6; do
  SPHMI_LIB=$PWD/build/variants/libsphmi_$v.so python tools/variant_probe.py $dp default 4 400 2>/dev/null | tail -1
done; done; done > gpurun_out/r05/qcap4_ab.txt 2>&1
sort gpurun_out/r05/qcap4_ab.txt
