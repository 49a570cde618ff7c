"""Instruction-class counts per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only ...):
    python tools/dbg/isa_count.py /tmp/enc.s [name-filter]
Dropout-hash / address integer multiplies (v_mul_lo_u32 & co: quarter rate) are listed apart from the full-rate vector ops."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and '@' in l]
for i, name in starts:
    if flt not in name:
        continue
    j = i
    while j < len(lines) and 's_endpgm' not in lines[j]:
        j += 1
    c = collections.Counter()
    for l in lines[i:j]:
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'):
            continue
        op = l.split()[0]
        k = ('mfma' if op.startswith('v_mfma') else 'lds' if op.startswith('ds_') else
             'vmem' if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch') else
             'v_mul32' if op.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64')) else
             'v_trans' if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')) else
             'valu' if op.startswith('v_') else 'wait' if op.startswith('s_waitcnt') else
             'barrier' if op.startswith('s_barrier') else 'salu' if op.startswith('s_') else 'other')
        c[k] += 1
    print(name[:70], dict(sorted(c.items())))


def segments(name_filter):
    """the same counts per barrier-to-barrier segment of the first kernel whose name contains `name_filter`"""
    for i, name in starts:
        if name_filter not in name:
            continue
        j = i
        seg, c = 0, collections.Counter()
        while j < len(lines) and 's_endpgm' not in lines[j]:
            l = lines[j].strip()
            j += 1
            if not l or l[0] in ';.' or l.endswith(':'):
                continue
            op = l.split()[0]
            if op.startswith('s_barrier'):
                print('  seg %2d' % seg, dict(sorted(c.items())))
                seg += 1
                c = collections.Counter()
                continue
            k = ('mfma' if op.startswith('v_mfma') else 'lds' if op.startswith('ds_') else
                 'vmem' if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch') else
                 'v_mul32' if op.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64')) else
                 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'other')
            c[k] += 1
        print('  seg %2d' % seg, dict(sorted(c.items())))
        return


if len(sys.argv) > 3 and sys.argv[3] == 'seg':
    segments(flt)
