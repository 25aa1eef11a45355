for fl in 2 3 2 3; do echo -n "inflight $fl: "; timeout 300 python bench.py --no-knn --no-cpu-baseline --train-steps 0 --default-prec-leg 0 --steps 40 --warmup 10 --inflight $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['steps_in_flight'])"; done
