#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args]   -- runs bench.py once per value of an env knob
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 180 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$VAR=$v', d['ms_per_step'], d['stage_ms'])"
done
