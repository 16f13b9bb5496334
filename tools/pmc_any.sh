#!/bin/bash
# Per-kernel occupancy / pipe activity of any workload: tools/pmc_any.sh <tag> <command...>  -> gpurun_out/<tag>_counters.txt
# (kernel trace + two PMC passes; one counter group per pass, --pmc never combined with other trace domains)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/pa_t -o x -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pa_p1 -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d gpurun_out/pa_p2 -o p -- "$@" > /dev/null 2>&1
python tools/pmc_kernels.py gpurun_out/${TAG}_counters_raw.txt gpurun_out/pa_t/x_results.db gpurun_out/pa_p1/p_results.db gpurun_out/pa_p2/p_results.db --all > /dev/null 2>&1
rm -rf gpurun_out/pa_t gpurun_out/pa_p1 gpurun_out/pa_p2
python - "$TAG" <<'PY'
import sys
tag=sys.argv[1]
txt=open('gpurun_out/%s_counters_raw.txt'%tag).read().split('\n')
cur=None; rows={}
for l in txt:
    if l.startswith('    '):
        p=l.split(); rows[cur][p[0]]=float(p[1])
    elif l.strip() and not l.startswith('kernel'):
        cur=l[:88].strip(); f=l[88:].split(); rows[cur]={'calls':int(f[0]),'us':float(f[1])}
out=['%-64s %6s %9s %8s  %s' % ('kernel','calls','avg_us','total_ms','resident waves/SIMD | VALU busy | MFMA busy | waves waiting | LDS busy (per CU) | conflict share')]
for k,v in sorted(rows.items(), key=lambda kv:-kv[1]['calls']*kv[1]['us']):
    g=v.get('GRBM_GUI_ACTIVE')
    if not g: continue
    sc=g/8*1024
    out.append('%-64s %6d %9.2f %8.2f  %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f' % (k[:64], v['calls'], v['us'], v['calls']*v['us']/1e3, v.get('SQ_WAVE_CYCLES',0)*4/sc, v.get('SQ_ACTIVE_INST_VALU',0)*4/sc,
               v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/sc, v.get('SQ_WAIT_ANY',0)*4/sc, v.get('SQ_LDS_IDX_ACTIVE',0)/sc*4, v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1)))
open('gpurun_out/%s_counters.txt'%tag,'w').write('\n'.join(out)+'\n')
print('\n'.join(out[:28]))
PY
