#!/bin/bash
# rocprofv3 kernel-trace summaries of the secondary paths (decode step, stage-1 step, GPT-2 step) -> gpurun_out/<round>_*_rocprof_stats.txt
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for job in "decode_step tools/gen_prof.py" "stage1 tools/bench_stage1.py" "gpt2_step tools/bench_gpt2.py"; do
  set -- $job
  rocprofv3 --kernel-trace --stats -d gpurun_out/sec_$1 -o x -- python $2 > gpurun_out/${R}_$1.log 2>&1
  python tools/rocprof_summary.py gpurun_out/sec_$1/x_results.db gpurun_out/${R}_$1_rocprof_stats.txt 1 > /dev/null
  rm -rf gpurun_out/sec_$1
  echo "== $1"; head -9 gpurun_out/${R}_$1_rocprof_stats.txt | cut -c1-140; tail -2 gpurun_out/${R}_$1.log | cut -c1-160
done
