# usage: bash tools/probe_run.sh "variant ..." "case:fb ..."      (prebuilt -DSPHMI_STATS / -DSPHMI_TRACE variants)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in ${2:-dam_break_3d_shipped:4 dam_break_2d:4}; do for v in ${1:-tr_nosplit tr_split}; do python tools/small_case_probe.py build/variants/libsphmi_$v.so ${c%%:*} ${c##*:}; done; done 2>&1 | tee gpurun_out/probe_trace.txt
