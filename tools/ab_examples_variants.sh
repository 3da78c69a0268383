for rep in 1 2; do
for v in notrims default; do
  if [ $v = default ]; then unset SPHMI_LIB; else export SPHMI_LIB=$GRAFT_REPO_ROOT/build/variants/libsphmi_$v.so; fi
  echo "== $v (rep $rep)"; python tools/bench_examples.py 3000 2>&1 | grep -v "^\[" | cut -c1-110
done; done
