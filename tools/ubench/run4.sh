timeout 600 python bench.py --no-knn --no-cpu-baseline --steps 5 --warmup 2 --default-prec-leg 0 2>&1 | tail -3 | cut -c1-3000 | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(json.dumps(d['train'],indent=1)[:2500])
    else: print(ln[:300])
"
