#!/bin/bash
# Same-box A/B of two builds of the library on the headline step: tools/ab_step.sh <libA.so> <libB.so> [rounds]
# (interleaved A B A B ..., `bench.py --steps 8 --warmup 2` each; prints ms per step and the top kernel rows of the last run of each)
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in $A $B; do
    CTK_LIB_PATH=$PWD/$L python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-lines 2>/dev/null | tail -1 > /tmp/ab.json
    python - $L <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
k = {r['name']: r for r in d.get('kernels', [])}
small = sum(r['total_ms'] for n, r in k.items() if n.startswith('gemm_sh_64'))
print(f"{sys.argv[1]:40s} {d['ms_per_step']:9.2f} ms/step  {d['value']:10.1f} pf/s  bit-identical steps: {d['steps_bit_identical']['identical']}  gemm_sh_64* {small:6.1f} ms  "
      f"corr {k.get('corr_volume_sh', {}).get('total_ms')}  ln {k.get('layernorm', {}).get('total_ms')}  parity px {d['parity']['timed_step']['coords_px']:.2e}")
PY
  done
done
