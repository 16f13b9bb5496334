#!/bin/bash
# same-box A/B of the reference-batch (B = 4) training leg: A-stationary K = 512 kernel from 4096 tokens (column split) vs from 32768 only
F="--steps 6 --warmup 2 --no-cpu-baseline --no-gen --no-stage1 --no-gpt2 --no-step0-check"
for v in 32768 4096 32768 4096; do
  EMO_ASTAT_MIN_ROWS=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_rows $v', 'b4 ms/step', d['b4']['ms_per_step'], 'loss', d['b4']['mean_loss'], '| B=64', d['ms_per_step'])"
done
