# Round 6: the XCD-share feedback ($SPHMI_XCD_FEEDBACK=0/1) over every instantiation of tools/bench_variants.py, the sizes table and the developed-flow window
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python tools/variant_probe.py 0.0085 default 4 3000 > /dev/null 2>&1
for fb in 1 0 1 0; do
  SPHMI_XCD_FEEDBACK=$fb python tools/bench_variants.py 200 2>/dev/null | sed "s/^/fb$fb /" 
done > gpurun_out/r06/xcd_feedback_variants.txt
for fb in 1 0 1 0; do SPHMI_XCD_FEEDBACK=$fb python tools/time_sizes.py 0.0065 0.0057 0.005 0.00425 0.003 0.002125 2>&1 | sed "s/^/fb$fb /"; done > gpurun_out/r06/xcd_feedback_sizes.txt
for fb in 1 0 1 0; do SPHMI_XCD_FEEDBACK=$fb python tools/bench_developed.py 2>/dev/null | tail -2 | sed "s/^/fb$fb /"; done > gpurun_out/r06/xcd_feedback_developed.txt
cat gpurun_out/r06/xcd_feedback_sizes.txt gpurun_out/r06/xcd_feedback_developed.txt
