#!/bin/bash
# Upper bound for the tap-fused weight gradient with NO operand split (probe library: the three planes get raw bits, results are
# garbage): how much of the kernel's time is the vector work of the split?   bash tools/probes/w3_nosplit.sh build   (here, hipcc)
#                                                                              bash tools/probes/w3_nosplit.sh run     (GPU box)
set -e
cd "$(dirname "$0")/../.."
B=tools/probes/_build
if [ "$1" = build ]; then
  sed 's/const Split4 s = split4(v);/Split4 s; s.p[0] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); s.p[1] = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w)); s.p[2] = s.p[0];/' \
      ideas_amd/csrc/conv_b3_wgrad3.hip > ideas_amd/csrc/_w3_nosplit.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c ideas_amd/csrc/_w3_nosplit.hip -o $B/w3_nosplit.o
  rm ideas_amd/csrc/_w3_nosplit.hip
  OBJS=$(ls ideas_amd/csrc/*.o | grep -v -e conv_b3_wgrad3.o -e dppb)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $B/w3_nosplit.o -o $B/libideas_hip_nosplit.so
  ls -la $B
else
  for lib in "" $B/libideas_hip_nosplit.so; do
    echo "== ${lib:-product library}"
    if [ -n "$lib" ]; then IDEAS_HIP_LIB=$PWD/$lib python tools/ab_wgrad3_s2.py; else python tools/ab_wgrad3_s2.py; fi
  done
fi
