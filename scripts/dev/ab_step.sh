for rep in 1 2 3; do for v in head ""; do
  echo "== ${v:-new}"; PSCV_LIB=${v:+$PWD/scripts/dev/libpscv_$v.so} python scripts/dev/ab_bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,1), {k:v for k,v in d['kernels_us'].items()})"
done; done
