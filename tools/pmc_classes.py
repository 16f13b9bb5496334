#!/usr/bin/env python3
"""rocpd databases of the rocprofv3 --pmc passes over tools/pmc_step.py -> per kernel class: launches, HBM bytes per launch
(FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, + WRITE_SIZE; both are reported in KB), MFMA pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)) and the wave-state split (SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES), and the EFFECTIVE SHADER CLOCK of the class = GRBM_GUI_ACTIVE / dispatch wall time (MI355X guide, "DVFS
give-back"; the counter is summed over the 8 XCDs).  The result is stamped with a hash of the kernel sources (csrc_hash: bench.py prints the
counters only while the stamp matches the sources it runs).  usage: pmc_classes.py out.json db1 db2 ..."""
import collections, glob, hashlib, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    """sha256 over the kernel sources (names + contents) of emo-disentanger_amd/csrc and the C-ABI header: what 'git rev-parse HEAD:<dir>' would
    pin, computable on the GPU box (the snapshot has no .git)."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'emo-disentanger_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'emo-disentanger_amd', 'csrc', '*.h'))
                   + [os.path.join(ROOT, 'include', 'emo_hip.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]

CLASSES = [('NT/K=512/relu+drop+mask', ('gemm_astat_kernelIDF16bLi19E',)), ('NT/K=512/bits', ('gemm_astat_kernelIDF16bLi8E',)), ('NT/K=512/plain', ('gemm_astat_kernelIDF16bLi0E',)),
           ('NT/K=512/drop+res', ('gemm_astat_kernelIDF16bLi6E',)), ('NT/K=512', ('gemm_astat_kernel',)),
           ('TN/rs', ('gemm_w128_tn_kernel<1>', 'gemm_w128_tn_kernelILi1E')), ('TN/plain', ('gemm_w128_tn_kernel<0>', 'gemm_w128_tn_kernelILi0E')),
           ('TN/128', ('gemm_bf16_kernelILb0ELb0', 'gemm_bf16_kernel<false, false')), ('TN-reduce', ('splitk_reduce_kernel',)),
           ('NN', ('gemm_bf16_glds_kernelILb1ELb0', 'gemm_bf16_glds_kernel<true, false')),
           ('NT/K>1024', ('gemm_w128_kernel', 'gemm_bf16_glds_kernelILb1ELb1EDF16bLi64ELi2', 'gemm_bf16_glds_kernel<true, true, __bf16, 64, 2')),
           ('NT', ('gemm_bf16_glds_kernelILb1ELb1',  'gemm_bf16_glds_kernel<true, true')),
           ('favor_fwd', ('favor_fs_fwd_kernel', 'favor_fwd_kernel')), ('favor_bwd_dq', ('favor_bwd_dq_kernel', 'favor_fs_dq_kernel')),
           ('favor_bwd_dkv', ('favor_bwd_dkv_kernel', 'favor_fs_dkv_kernel')), ('layernorm_fwd', ('layernorm_fwd',)), ('layernorm_bwd', ('layernorm_bwd',))]


def klass(name):
    for k, pats in CLASSES:
        if any(p in name for p in pats):
            return k
    return None


def main(out, *dbs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    clk = collections.defaultdict(list)              # class -> [GRBM_GUI_ACTIVE cycles / (8 XCDs * dispatch ns)] = GHz
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
        have_t = 'start' in cols and 'end' in cols
        q = 'select kernel_name, counter_name, value%s from counters_collection' % (', start, end' if have_t else '')
        for row in cur.execute(q):
            name, cn, val = row[:3]
            k = klass(name)
            if k:
                acc[k][cn].append(val)
                if have_t and cn == 'GRBM_GUI_ACTIVE' and row[4] > row[3]:
                    clk[k].append(val / 8.0 / (row[4] - row[3]))
        if not have_t:
            print('[pmc_classes] counters_collection has no start / end columns (%s): no effective clock from %s' % (cols, db), file=sys.stderr)
    res = {}
    for k, cs in acc.items():
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {'launches_sampled': max(len(v) for v in cs.values())}
        if 'FETCH_SIZE' in mean:
            e['fetch_bytes_per_launch'] = round(mean['FETCH_SIZE'] * 1024 * 2)
        if 'WRITE_SIZE' in mean:
            e['write_bytes_per_launch'] = round(mean['WRITE_SIZE'] * 1024)
        if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
            e['traffic_bytes_per_launch'] = e['fetch_bytes_per_launch'] + e['write_bytes_per_launch']
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in mean and mean.get('GRBM_GUI_ACTIVE'):
            e['mfma_busy'] = round(mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        if mean.get('SQ_WAVE_CYCLES'):
            for c, key in (('SQ_WAIT_ANY', 'waves_parked'), ('SQ_WAIT_INST_ANY', 'waves_issue_stalled'), ('SQ_ACTIVE_INST_ANY', 'waves_issuing')):
                if c in mean:
                    e[key] = round(mean[c] / mean['SQ_WAVE_CYCLES'], 4)
        if clk.get(k):
            e['effective_clock_ghz'] = round(sum(clk[k]) / len(clk[k]), 3)
        if mean.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_conflict_share'] = round(mean.get('SQ_LDS_BANK_CONFLICT', 0.0) / mean['SQ_LDS_IDX_ACTIVE'], 4)
        res[k] = e
    json.dump({'classes': res, 'csrc_hash': csrc_hash(), 'how': 'rocprofv3 --kernel-trace --pmc <one group per pass> -- python tools/pmc_step.py; groups: FETCH_SIZE | WRITE_SIZE | '
               'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY | SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT; '
               'means over the dispatches of 3 training steps at B=64 x T=2048; effective_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (tools/pmc_classes.py)'}, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:])
