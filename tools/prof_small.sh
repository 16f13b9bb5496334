#!/bin/bash
# rocprofv3 kernel-trace summaries of the small-grid training steps: stage-2 Performer at the reference YAML's batch size 4, and stage 1
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for job in "b4_step tools/pmc_step.py" "stage1 tools/bench_stage1.py"; do
  set -- $job
  BS=4 STEPS=9 rocprofv3 --kernel-trace --stats -d gpurun_out/ps_$1 -o x -- python $2 > gpurun_out/${R}_$1.log 2>&1
  python tools/rocprof_summary.py gpurun_out/ps_$1/x_results.db gpurun_out/${R}_$1_rocprof_stats.txt 1 > /dev/null
  rm -rf gpurun_out/ps_$1
  echo "== $1"; head -22 gpurun_out/${R}_$1_rocprof_stats.txt | cut -c1-75,93-140
done
