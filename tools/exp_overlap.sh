run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['config']['step_graphs']; print(d['value'], d['ms_per_step'], 'A', g.get('gpu_ms_phase_a'), 'gap', g.get('gpu_ms_host_gap'), 'B', g.get('gpu_ms_phase_b'), d['final_losses']['loss_cls_source_strong'])"; }
run A=1
run ALDI_MASK_BITS=0
run A=1
run ALDI_MASK_BITS=0
