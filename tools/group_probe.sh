for ns in 8 16 24 32 64 512; do
  for sp in 0 1; do
    if [ $sp = 1 ]; then export P3D_NO_SPARSE=1; else unset P3D_NO_SPARSE; fi
    python bench.py --nslices $ns --steps 100 --warmup 5 --only-main --no-cpu-baseline --no-dense --no-profile --repeats 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=$ns
print('nslices',ns,'nosparse',$sp,'ms_per_step',d['ms_per_step'],'us per slice-iter',d['ms_per_step']*1000/ns, 'equiv it/s of 512', 1000/(d['ms_per_step']*512/ns))"
  done
done
