run() { for wl in vitdet_b convnext_l; do echo -n "$* $wl: "; env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --steps 12 --warmup 4 2>&1 | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; }
run X=1
run ALDI_WGRAD_BIG_MIN=8
run ALDI_WGRAD_BIG_MIN=16
run ALDI_WGRAD_BIG_MIN=60
run ALDI_WGRAD_SLOTS=256
run ALDI_WGRAD_SLOTS=512
run ALDI_WGRAD_SLOTS=768
run ALDI_IGEMM_LINTILE_MIN=512
run ALDI_IGEMM_BIGTILE_K=384
