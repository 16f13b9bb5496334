#!/bin/bash
# same-box per-kernel comparison of two library builds inside the training step (rocprofv3 kernel trace of the bench loop, old then new)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=emo-disentanger_amd
F="--steps 10 --warmup 3 --no-cpu-baseline --no-gen --no-stage1 --no-gpt2 --no-step0-check --no-b4 --no-fp32 --no-roofline"
for v in old new; do
  cp $P/_ab_$v.so $P/libemo_hip.so
  rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$v -o x -- python bench.py $F > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.log
  python tools/rocprof_summary.py gpurun_out/ab_$v/x_results.db gpurun_out/ab_${v}_stats.txt 13 > /dev/null
  echo "== $v"; grep -E "astat|w128|favor_fs|total kernel" gpurun_out/ab_${v}_stats.txt | cut -c1-60,96-140
  rm -rf gpurun_out/ab_$v
done
