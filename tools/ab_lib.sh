#!/bin/bash
# same-box A/B of two builds of the library: emo-disentanger_amd/_ab_old.so vs _ab_new.so (both built here, git-ignored), product bench loop
F="--steps 10 --warmup 3 --no-cpu-baseline --no-gen --no-stage1 --no-gpt2 --no-step0-check --no-b4 --no-fp32"
P=emo-disentanger_amd
for v in old new old new; do
  cp $P/_ab_$v.so $P/libemo_hip.so
  python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['median_ms_per_step'], [(round(r['achieved'],1), r['total_ms_per_step']) for r in [d['roofline']]+d['roofline']['roofline_others'][:3]])"
done
