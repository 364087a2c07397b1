#!/bin/bash
# Same-box A/B of the whole training step under environment switches: tools/ab_step.sh "<bench args>" "VAR=a" "VAR=b" ...
# (each setting measured twice, interleaved; prints images/s and ms per step)
ARGS=$1; shift
for rep in 1 2; do
  for S in "$@"; do
    echo -n "$S (run $rep): "
    env $S python bench.py $ARGS --cpu-baseline skip --roofline off --also-bf16 off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
