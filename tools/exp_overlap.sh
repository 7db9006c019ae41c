run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['config']['step_graphs']; print(d['value'], d['ms_per_step'], 'A', g.get('gpu_ms_phase_a'), 'gap', g.get('gpu_ms_host_gap'), 'B', g.get('gpu_ms_phase_b'))"; }
run A=1
run ALDI_WGRAD_BIG_GROUP=0
run ALDI_WGRAD_BIG_GROUP=0 ALDI_WGRAD_LDS_PAD_KB=24
run ALDI_WGRAD_BIG_GROUP=0 ALDI_WGRAD_LDS_PAD_KB=64
run ALDI_WGRAD_BIG_GROUP=0 ALDI_WGRAD_LDS_PAD_KB=24 ALDI_MAIN_PRIO=-1
run ALDI_WGRAD_BIG_GROUP=0 ALDI_WGRAD_LDS_PAD_KB=64 ALDI_MAIN_PRIO=-1
run ALDI_WGRAD_BIG_GROUP=0 ALDI_WGRAD_LDS_PAD_KB=64 ALDI_WGRAD_STREAM=0
