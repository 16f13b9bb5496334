#!/bin/bash
# Round evidence, run on the GPU box from the repo root (gpurun): rocprofv3 kernel-trace summary of the timed bench command + the PMC
# passes over the training step (one counter group per pass; --pmc is never combined with other trace domains).  Outputs -> gpurun_out/,
# copy what is to be judged into profiles/.
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_stats -o x -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gen --no-stage1 --no-gpt2 --no-step0-check --no-b4 --no-fp32 > gpurun_out/${R}_stats_bench.json 2> gpurun_out/${R}_stats.log
python tools/rocprof_summary.py gpurun_out/${R}_stats/x_results.db gpurun_out/${R}_bench_train_rocprof_stats.txt 13 > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/${R}_pmc$i -o p -- python tools/pmc_step.py > gpurun_out/${R}_pmc$i.log 2>&1
done
python tools/pmc_classes.py gpurun_out/${R}_pmc_step.json gpurun_out/${R}_pmc1/p_results.db gpurun_out/${R}_pmc2/p_results.db gpurun_out/${R}_pmc3/p_results.db gpurun_out/${R}_pmc4/p_results.db > gpurun_out/${R}_pmc_classes.log 2>&1
rm -rf gpurun_out/${R}_pmc[1-4] gpurun_out/${R}_stats
head -30 gpurun_out/${R}_bench_train_rocprof_stats.txt
