for args in "--temper" "--map mesh-tin --temper" ""; do
for b in 16,32,8 32,32,4 16,16,16 8,32,16 32,16,8 16,64,4; do
  export MCL_VISIT_BINS=$b
  python bench.py --only-main --steps 100 --warmup 40 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$args] bins $b', d['ms_per_step'], 'main_us', d['roofline']['launch_us'])"
done; done
