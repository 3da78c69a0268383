#!/bin/bash
# Build libsphmi variants and run tools/bench_examples.py with each (GPU box).  usage: tools/sweep_examples.sh "name:-DFLAG ..." ...
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  python -c "from sphexample_amd import build; build.build(force=True, extra_flags='$flags'.split(), out='/tmp/libsphmi_$name.so')" > /dev/null 2>&1 || { echo "$name: build failed"; continue; }
  echo "== $name [$flags]"
  SPHMI_LIB=/tmp/libsphmi_$name.so python tools/bench_examples.py 500 2>&1 | grep -E "fp32" | cut -c1-140
done
