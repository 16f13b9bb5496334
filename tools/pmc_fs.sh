cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS SQ_LDS_DATA_FIFO_FULL"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmcfs_$n -o x -- python $R/tools/pmc_favor.py > /dev/null 2>&1
  db=$(ls $R/gpurun_out/pmcfs_$n/*.db $R/gpurun_out/pmcfs_$n/*/*.db 2>/dev/null | head -1)
  python - "$db" <<'PY'
import sqlite3, sys, re, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in cur.execute('select * from counters_collection'):
    d = dict(zip(cols, row))
    k = d.get('kernel_name', d.get('name', '?'))
    if 'favor' not in k: continue
    m = re.search(r'favor_\w+', k); k = m.group(0)
    acc[k][d.get('counter_name')].append(d.get('value', d.get('counter_value')))
for k, cs in acc.items():
    print(k, ' '.join('%s=%.4g' % (c, sum(v)/len(v)) for c, v in sorted(cs.items())))
PY
done
