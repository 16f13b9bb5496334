#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA-busy of the persistent 256 x 256 kernel next to the kernels it replaces, on the
# workload of tools/bench_p256.py (ONLY=<substring> selects shapes).  -> gpurun_out/<tag>_p256_pmc.txt
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export ITER=3
rocprofv3 --kernel-trace --stats -d gpurun_out/pp_t -o x -- python tools/bench_p256.py > gpurun_out/${TAG}_p256_pmc.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/pp_p$i -o p -- python tools/bench_p256.py > /dev/null 2>&1
done
python tools/pmc_kernels.py gpurun_out/${TAG}_p256_pmc.txt gpurun_out/pp_t/x_results.db gpurun_out/pp_p1/p_results.db gpurun_out/pp_p2/p_results.db gpurun_out/pp_p3/p_results.db gpurun_out/pp_p4/p_results.db --all --match p256,w128,astat 2> gpurun_out/${TAG}_p256_pmc.err
rm -rf gpurun_out/pp_t gpurun_out/pp_p1 gpurun_out/pp_p2 gpurun_out/pp_p3 gpurun_out/pp_p4
cat gpurun_out/${TAG}_p256_pmc.txt
