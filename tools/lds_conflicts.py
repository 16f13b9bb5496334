#!/usr/bin/env python3
"""LDS bank-conflict model of the MI355X guide (lane groups and bank modulus per instruction) applied to per-lane byte addresses:
cycles(group) = max over banks of the number of DISTINCT dwords on the bank.  Used to check the operand layouts of the FAVOR+ slice kernels
and the attention kernels before spending a PMC run on them (SQ_LDS_BANK_CONFLICT is the arbiter)."""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]
GROUPS = {
    'read_b32': ([list(range(0, 32)), list(range(32, 64))], 32, 1),
    'read_b64': ([list(range(0, 32)), list(range(32, 64))], 64, 2),
    'read_b128': (G128, 64, 4),
    'read_tr_b64': ([list(range(0, 32)), list(range(32, 64))], 64, 2),
    'write_b32': ([list(range(0, 32)), list(range(32, 64))], 32, 1),
    'write_b64': ([list(range(16 * i, 16 * i + 16)) for i in range(4)], 32, 2),
    'write_b128': ([list(range(8 * i, 8 * i + 8)) for i in range(8)], 32, 4),
}


def cycles(kind, addr):
    """addr: function lane -> byte address.  Returns (cycles, conflict-free cycles)."""
    groups, mod, nd = GROUPS[kind]
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr(l)
            for d in range(nd):
                dw = a // 4 + d
                banks.setdefault(dw % mod, set()).add(dw)
        tot += max(len(v) for v in banks.values())
    return tot, len(groups)


if __name__ == '__main__':
    import sys
    LDF = int(sys.argv[1]) if len(sys.argv) > 1 else 136
    ROWB = 128
    sw = (lambda r: r & 7) if len(sys.argv) > 2 and sys.argv[2] == 'r03' else (lambda r: ((r >> 2) & 1) | (r & 2) | ((((r >> 2) ^ (r >> 3)) & 1) << 2))   # fs_sw (emo_favor_fs.hip)
    rep = []
    # ring fragment (b128): row = c, piece (4 s + g) ^ (c & 7)
    for s in range(2):
        rep.append(('fs_ring_frag s=%d' % s, cycles('read_b128', lambda l: (l & 15) * ROWB + ((((4 * s + (l >> 4)) ^ sw(l & 15))) << 4))))
    # feature image stores (b64): row c, column 16 w + 4 g (+64)
    for w in range(4):
        rep.append(('feature st4 w=%d' % w, cycles('write_b64', lambda l: ((l & 15) * LDF + 16 * w + 4 * (l >> 4)) * 2)))
    # load_perm (2 x b64): row c, elements 4 g + 32 step (+16)
    for st in range(4):
        rep.append(('load_perm step=%d' % st, cycles('read_b64', lambda l: ((l & 15) * LDF + 4 * (l >> 4) + 32 * st) * 2)))
    # load_perm_tr on the feature image: row = h 16 + 4 kc + i / 4, col = col0 + 4 (i % 4)
    for col0 in (0, 16, 64):
        for h in range(2):
            rep.append(('load_perm_tr(img) col0=%d h=%d' % (col0, h),
                        cycles('read_tr_b64', lambda l: ((h * 16 + 4 * (l >> 4) + ((l & 15) >> 2)) * LDF + col0 + 4 * (l & 3)) * 2)))
    # fs_ring_perm_tr on a swizzled ring tile
    def ring_tr(col0, h):
        def f(l):
            i, kc = l & 15, l >> 4
            row, col = h * 16 + kc * 4 + (i >> 2), col0 + (i & 3) * 4
            return row * ROWB + (((col >> 3) ^ sw(row)) << 4) + (col & 7) * 2
        return f
    for col0 in (0, 16, 32, 48):
        rep.append(('fs_ring_perm_tr col0=%d' % col0, cycles('read_tr_b64', ring_tr(col0, 0))))
    # fs_ring_perm: 8-B halves of pieces ch and ch + 2 of row c
    for st in range(2):
        for add in (0, 2):
            rep.append(('fs_ring_perm s=%d +%d' % (st, add), cycles('read_b64', lambda l: (l & 15) * ROWB + ((l >> 4) & 1) * 8 + ((((4 * st + ((l >> 4) >> 1)) + add) ^ sw(l & 15)) << 4))))
    for name, (c, ideal) in rep:
        print('%-36s %2d cycles (conflict-free %d)' % (name, c, ideal))
