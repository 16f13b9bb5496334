#!/bin/bash
# PMC passes over the FAVOR+ micro-benchmark (tools/bench_favor.py): per-kernel LDS activity / bank conflicts, VALU, MFMA-busy, waits
# -> gpurun_out/<round>_favor_pmc.txt.  One counter group per pass, --pmc never combined with other trace domains.
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/fp_t -o x -- env BS=64 python tools/bench_favor.py > gpurun_out/${R}_favor.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/fp_p$i -o p -- env BS=64 python tools/bench_favor.py > gpurun_out/${R}_favor_pmc$i.log 2>&1
done
python tools/pmc_kernels.py gpurun_out/${R}_favor_pmc.txt gpurun_out/fp_t/x_results.db gpurun_out/fp_p1/p_results.db gpurun_out/fp_p2/p_results.db gpurun_out/fp_p3/p_results.db --match favor --all > /dev/null 2> gpurun_out/${R}_favor_pmc.err
rm -rf gpurun_out/fp_t gpurun_out/fp_p1 gpurun_out/fp_p2 gpurun_out/fp_p3
grep -E "^[_a-z]|LDS|GRBM" gpurun_out/${R}_favor_pmc.txt | cut -c1-150
tail -3 gpurun_out/${R}_favor.log
