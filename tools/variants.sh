#!/bin/bash
# usage: tools/variants.sh <families> v1 v2 ...   (benchmarks fgumi_amd/variant_<v>.so builds)
FAM=$1; shift
for v in "$@"; do
  FGX_LIB=$PWD/fgumi_amd/variant_$v.so timeout 300 python bench.py --families $FAM --steps 5 --warmup 1 --no-cpu-baseline > /tmp/v_$v.log 2>&1
  python - "$v" /tmp/v_$v.log <<'PY'
import json,sys
v,f=sys.argv[1],sys.argv[2]
line=[l for l in open(f) if l.startswith('{')]
if not line: print(v,'FAILED'); print(open(f).read()[-600:])
else:
    d=json.loads(line[-1]); print(v, 'k_family_ms=%.2f k_emit_ms=%.2f reads/s=%.3g'%(d['roofline']['kernel_ms'], d['roofline']['k_emit_ms'], d['value']))
PY
done
