for kc in 4 8 0; do for x in 0 1; do
echo "KC=$kc XCD=$x"; ALDI_IGEMM_KC=$kc ALDI_IGEMM_XCD=$x python bench.py --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'igemm', r['achieved'], r['kernel_ms_per_step'], 'wgrad', r['wgrad_kernel'])"
done; done
