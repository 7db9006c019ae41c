#!/bin/bash
# A/B of an environment knob on ONE box: tools/ab_env.sh NAME v1 v2 ...   (one bench run per value, in the order given; STEPS=<n> timed steps)
name=$1; shift
for v in "$@"; do
  out=$(env $name=$v python bench.py --no-cpu-baseline --no-profile --steps ${STEPS:-20} 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['config']['step_graphs']; print('$name=$v', d['value'], d['ms_per_step'], s.get('gpu_ms_phase_a'), s.get('gpu_ms_phase_b'), 'err', d['config']['error_flag'])"
done
