#!/bin/bash
# like ab_bench.sh, printing the fused step's device-side phase times:  tools/ab_phases.sh ROUNDS STEPS "ENV_A" "ENV_B" ...
rounds=$1; steps=$2; shift 2
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python bench.py --steps $steps --warmup 6 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['config']['step_graphs']
print('%-44s %8.2f img/s %7.3f ms  phase A %6.3f  gap %5.3f  phase B %6.3f  issue a/b %5.0f/%5.0f us  prefix ahead/inline %s/%s' % ('$e' or '(default)', d['value'], d['ms_per_step'], g.get('gpu_ms_phase_a', 0), g.get('gpu_ms_host_gap', 0), g.get('gpu_ms_phase_b', 0), g.get('host_us_issue_a', 0), g.get('host_us_issue_b', 0), g.get('prefix_ahead'), g.get('prefix_inline')))"
  done
done
