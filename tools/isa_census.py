#!/usr/bin/env python3
"""Instruction census of the hot loops of a kernel, from the compiler's assembly (`hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S`):
per loop body (label .. backward branch) the counts by class and, for the VALU, by what the instruction is there for.
usage: isa_census.py file.s [kernel-name-substring] [min MFMAs per loop]"""
import collections
import re
import sys

CATS = [('mfma', r'v_mfma'), ('exp / rcp / rsq (quarter rate)', r'v_(exp|log|rcp|rsq|sqrt)_'), ('convert (cvt, pack to bf16)', r'v_cvt'),
        ('move / select (mov, cndmask, perm, permlane, dpp, readlane)', r'v_(mov|cndmask|perm|permlane|readlane|readfirstlane|writelane|accvgpr|swap|bfi|alignbit)'),
        ('64-bit address arithmetic', r'v_(lshl_add_u64|add_co|addc_co|lshlrev_b64|mad_u64|mad_i64)'),
        ('32-bit integer (add, shift, and / or / xor, mul)', r'v_(add_u32|sub_u32|subrev_u32|lshlrev_b32|lshrrev_b32|ashrrev|and_b32|or_b32|or3|xor|xad|lshl_or|lshl_add|and_or|add3|add_lshl|mul_lo|mul_hi|mul_u32|mad_u32|bfe|bitop3|not_b32|mul_i32)'),
        ('compare', r'v_cmp'), ('packed fp32 (v_pk_*_f32)', r'v_pk_(mul|add|fma)_f32'), ('packed 16-bit', r'v_pk_'),
        ('fp32 arithmetic (add, mul, fma, max)', r'v_(add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|mac_f32|max_f32|min_f32|mad_f32|fmaak|fmamk|ldexp|max_i32|min_i32|med3)')]


def census(lines):
    c = collections.Counter()
    other = collections.Counter()
    for l in lines:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if not m:
            continue
        op = m.group(1)
        if op.startswith('v_'):
            for name, pat in CATS:
                if re.match(pat, op):
                    c['mfma' if name == 'mfma' else 'VALU: ' + name] += 1
                    break
            else:
                c['VALU: other'] += 1
                other[op] += 1
        elif op.startswith('ds_'):
            c['LDS (ds_*)'] += 1
        elif re.match(r'(global|buffer|scratch|flat)_', op):
            c['VMEM'] += 1
        elif op == 's_waitcnt':
            c['s_waitcnt'] += 1
        elif op == 's_barrier':
            c['s_barrier'] += 1
        elif op == 's_nop':
            c['s_nop'] += 1
        elif op.startswith('s_'):
            c['SALU / branch'] += 1
    return c, other


def main():
    s = open(sys.argv[1]).read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    for m in re.finditer(r'^(_Z\w+):\s', s, re.M):
        name = m.group(1)
        if filt and filt not in name:
            continue
        lines = s[m.end():s.find('s_endpgm', m.end())].split('\n')
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r'(\.LBB\d+_\d+):', l)
            if mm:
                labels[mm.group(1)] = i
        loops = []
        for j, l in enumerate(lines):
            mm = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < j:
                i = labels[mm.group(1)]
                if sum('v_mfma' in x for x in lines[i:j]) >= min_mfma:
                    loops.append((i, j))
        if not loops:
            continue
        i, j = max(loops, key=lambda t: t[1] - t[0])               # the outermost loop that holds the MFMAs = the chunk / tile loop
        c, other = census(lines[i:j + 1])
        valu = sum(v for k, v in c.items() if k.startswith('VALU'))
        print('%s\n  loop body: %d lines, %d MFMA, %d other VALU (%.1f per MFMA), %d LDS, %d VMEM, %d SALU, %d s_waitcnt, %d s_barrier' %
              (name, j - i + 1, c['mfma'], valu, valu / max(c['mfma'], 1), c['LDS (ds_*)'], c['VMEM'], c['SALU / branch'], c['s_waitcnt'], c['s_barrier']))
        for k, v in sorted(c.items(), key=lambda t: -t[1]):
            if k.startswith('VALU'):
                print('    %4d  %5.1f %%  %s' % (v, 100.0 * v / max(valu, 1), k[6:]))
        if other:
            print('          other: ' + ', '.join('%s x%d' % kv for kv in other.most_common(8)))


if __name__ == '__main__':
    main()
