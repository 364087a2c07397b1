#!/bin/bash
# tools/isa.sh <file.hip> <out.s> : device ISA of one kernel source (schedule / register inspection)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -S --cuda-device-only "$1" -o "$2" 2>&1 | grep -v "argument unused"
python3 - "$2" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", t, re.S):
    print("%-110s lds %6s vgpr %4s spill %s" % (m.group(2)[:110], m.group(1), m.group(3), m.group(4)))
PY
