#!/bin/bash
# PMC passes over the GPT-2 attention micro-benchmark (tools/bench_sattn.py): per-kernel MFMA-busy, VALU / LDS activity, bank conflicts, waits
# -> gpurun_out/<round>_sattn_pmc.txt.  One counter group per pass, --pmc never combined with other trace domains.
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/sp_t -o x -- python tools/bench_sattn.py > gpurun_out/${R}_sattn.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/sp_p$i -o p -- python tools/bench_sattn.py > gpurun_out/${R}_sattn_pmc$i.log 2>&1
done
python tools/pmc_kernels.py gpurun_out/${R}_sattn_pmc.txt gpurun_out/sp_t/x_results.db gpurun_out/sp_p1/p_results.db gpurun_out/sp_p2/p_results.db gpurun_out/sp_p3/p_results.db gpurun_out/sp_p4/p_results.db --match sattn --all 2> gpurun_out/${R}_sattn_pmc.err
rm -rf gpurun_out/sp_t gpurun_out/sp_p1 gpurun_out/sp_p2 gpurun_out/sp_p3 gpurun_out/sp_p4
cat gpurun_out/${R}_sattn_pmc.txt
