cd $GRAFT_REPO_ROOT
for s in 21000 22000 23000 24000; do
  echo "=== seed0 $s default"; SPHMI_FUZZ_SEED0=$s timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -4
  echo "=== seed0 $s group kernel forced, host rebuild"; SPHMI_MDBC_GROUP=1 SPHMI_DEVICE_REBUILD=0 SPHMI_FUZZ_SEED0=$s timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -4
done
