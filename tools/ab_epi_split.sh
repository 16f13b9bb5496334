cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 2; do
  for job in "stage1 tools/bench_stage1.py"; do
    set -- $job
    EMO_GEMM_EPI_SPLIT=$v BS=4 STEPS=9 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$1_$v -o x -- python $2 > gpurun_out/ab_$1_$v.log 2>&1
    python tools/rocprof_summary.py gpurun_out/ab_$1_$v/x_results.db gpurun_out/r04_ab_epi_split${v}_$1_stats.txt 1 > /dev/null
    rm -rf gpurun_out/ab_$1_$v
    echo "== $1 EPI_SPLIT=$v"; head -12 gpurun_out/r04_ab_epi_split${v}_$1_stats.txt | cut -c1-70,93-140
  done
done
