#!/bin/bash
# Interleaved A/B of the whole step: usage  tools/ab_bench.sh ROUNDS STEPS "ENV_A" "ENV_B" ...   (each ENV a quoted list of VAR=value, may be empty)
rounds=$1; steps=$2; shift 2
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python bench.py --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-40s %8.2f img/s %7.3f ms  igemm %6.1f TF %6.3f ms' % ('$e' or '(default)', d['value'], d['ms_per_step'], r['achieved'], r.get('kernel_ms_per_step', 0)))"
  done
done
