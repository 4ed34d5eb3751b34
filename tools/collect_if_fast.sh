#!/bin/bash
# The pool's boxes differ by several per cent for the same binary: probe the box with a short bench run and collect the evidence set
# only on a box at or above the given rate (tools/collect_if_fast.sh [threshold steps/s] [collect script]); the probe line is kept
# either way.
THR=${1:-600}
SCRIPT=${2:-tools/collect_r05.sh}
TAG=$(basename $SCRIPT .sh | sed 's/collect_//')
V=$(python bench.py --no-cpu-baseline --no-traffic --steps 100 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
echo "probe: $V steps/s (threshold $THR)" | tee gpurun_out/${TAG}_probe.txt
if python -c "import sys; sys.exit(0 if float('$V') >= float('$THR') else 1)"; then
  bash $SCRIPT > gpurun_out/${TAG}_collect.log 2>&1
  echo collected
else
  echo "box below the threshold: not collecting"
fi
