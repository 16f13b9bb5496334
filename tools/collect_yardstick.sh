#!/bin/bash
# Vendor GEMM (hipBLASLt through torch.matmul: yardstick only) and emo_gemm on the same shapes, same counters: kernel trace + three PMC passes
# (never combined with other trace domains).  -> gpurun_out/<round>_gemm_yardstick.txt (copy to profiles/)
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/ys_t -o x -- python tools/gemm_yardstick.py > gpurun_out/${R}_yardstick.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  ITER=4 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/ys_p$i -o p -- python tools/gemm_yardstick.py > gpurun_out/${R}_yardstick_pmc$i.log 2>&1
done
python tools/pmc_kernels.py gpurun_out/${R}_gemm_yardstick.txt gpurun_out/ys_t/x_results.db gpurun_out/ys_p1/p_results.db gpurun_out/ys_p2/p_results.db gpurun_out/ys_p3/p_results.db --match Cijk,gemm_,astat,w128 2> gpurun_out/${R}_yardstick_summary.err
grep "max |" gpurun_out/${R}_yardstick.log
rm -rf gpurun_out/ys_t gpurun_out/ys_p1 gpurun_out/ys_p2 gpurun_out/ys_p3
