#!/bin/bash
# same-box A/B of two trees: tools/ab_trees.sh ROUNDS "DIR_A" "DIR_B"  (default flags minus the CPU leg)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for d in "$@"; do
    (cd $d && python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; g = d['config']['step_graphs']
print('%-12s %8.2f img/s %7.3f ms  phase A %6.3f B %6.3f  igemm %6.1f TF %6.3f ms  wgrad %6.1f TF' % ('$d', d['value'], d['ms_per_step'], g.get('gpu_ms_phase_a', 0), g.get('gpu_ms_phase_b', 0), r['achieved'], r.get('kernel_ms_per_step', 0), r['wgrad_kernel']['achieved']))")
  done
done
