#!/bin/bash
# Same-box A/B of kernel builds (boxes of the pool differ by +-3 %): tools/ab.sh "<bench_igemm args>" libA.so libB.so ...
# (paths relative to the repo; "" = the in-tree libideas_hip.so).  Each library is measured twice, interleaved.
ARGS=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    if [ -n "$L" ] && [ "$L" != "-" ]; then export IDEAS_HIP_LIB=$PWD/$L; else unset IDEAS_HIP_LIB; fi
    echo "== ${L:-in-tree} (run $rep)"
    python tools/bench_igemm.py $ARGS 2>&1 | grep "TF/s"
  done
done
