#!/bin/bash
# usage: tools/variants.sh <tag> "<ENV=.. ENV=..|flags>" ...   one short bench.py run per variant, value + per-kernel ms into gpurun_out/variants_<tag>.txt
TAG=$1; shift
OUT=gpurun_out/variants_$TAG.txt; mkdir -p gpurun_out; : > $OUT
for V in "$@"; do
  ENVS=""; FLAGS=""
  for W in $V; do case $W in --*) FLAGS="$FLAGS $W";; [0-9]*) FLAGS="$FLAGS $W";; *) ENVS="$ENVS $W";; esac; done
  echo "== $V" >> $OUT
  env $ENVS timeout 300 python bench.py --steps 3 --warmup 1 --frames 2 --no-cpu-baseline $FLAGS 2>> $OUT.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f M pts/s  ms_per_step %.3f' % (d['value'], d['ms_per_step']))
print({k.split('<')[0]: round(v['total_ms'],2) for k,v in d['kernels'].items() if k.startswith('k_')})
r=d['roofline']; print('dominant', r['kernel'], 'frac %.4f' % r['frac'], 'moved', r['moved_points'])
" >> $OUT 2>&1
done
cat $OUT
