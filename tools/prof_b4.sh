#!/bin/bash
# rocprofv3 kernel-trace summary of the reference-batch (B = 4) training step (tools/b4_cpu_probe.py's loop) -> gpurun_out/<round>_b4_step_rocprof_stats.txt
set -u
R=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/sec_b4 -o x -- python tools/b4_cpu_probe.py > gpurun_out/${R}_b4_step.log 2>&1
python tools/rocprof_summary.py gpurun_out/sec_b4/x_results.db gpurun_out/${R}_b4_step_rocprof_stats.txt 65 > /dev/null
rm -rf gpurun_out/sec_b4
head -45 gpurun_out/${R}_b4_step_rocprof_stats.txt | cut -c1-150
