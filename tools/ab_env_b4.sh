#!/bin/bash
# same-box A/B of one environment setting on the reference-batch (B = 4) leg and the B = 64 leg: tools/ab_env_b4.sh NAME VALUE_A VALUE_B  ("-" = unset)
F="--steps 6 --warmup 2 --no-cpu-baseline --no-gen --no-stage1 --no-gpt2 --no-step0-check"
N=$1; A=$2; B=$3
for v in "$A" "$B" "$A" "$B"; do
  if [ "$v" = "-" ]; then unset $N; else export $N="$v"; fi
  python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N=$v', 'b4 ms/step', d['b4']['ms_per_step'], 'loss', d['b4']['mean_loss'], '| B=64', d['ms_per_step'])"
done
