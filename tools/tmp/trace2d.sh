cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in dam_break_2d dam_break_3d_shipped; do
rm -rf /tmp/tr_$c; timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$c -o t -- python $R/tools/steps_example.py $c 4 300 > /dev/null 2>&1
db=$(find /tmp/tr_$c -name '*.db' | head -1)
echo "##### $c"; python $R/tools/trace_window.py $db k_neighbor_force 0 40 400 | head -80
done
