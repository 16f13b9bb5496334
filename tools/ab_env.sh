#!/bin/bash
# same-box A/B of one environment setting inside the product training step: tools/ab_env.sh NAME VALUE_A VALUE_B  (VALUE "-" = unset)
F="--steps 10 --warmup 3 --no-cpu-baseline --no-gen --no-stage1 --no-gpt2 --no-step0-check --no-b4 --no-fp32"
N=$1; A=$2; B=$3
for v in "$A" "$B" "$A" "$B"; do
  if [ "$v" = "-" ]; then unset $N; else export $N="$v"; fi
  python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v', d['ms_per_step'], d['median_ms_per_step'], {k: (round(v['achieved_tflops']), v['total_ms_per_step']) for k, v in d['roofline']['gemm_families'].items()})"
done
