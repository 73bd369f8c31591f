"""Instruction mix of the hottest loop of every kernel in a hipcc -S listing (python tools/loop_mix.py file.s [name-substring]): the
loop with the most MFMAs, by opcode -- what the round-6 entry-side work counted its vector instructions with."""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    for k in re.split(r'\n\s*\.globl\s+', s)[1:]:
        name = k.split('\n')[0].split(';')[0].strip()
        if want not in name:
            continue
        lines = k.split('\n')
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        if not loops:
            continue

        def ops(a, b):
            c = collections.Counter()
            for l in lines[a:b + 1]:
                l = l.strip()
                if not l or l[0] in '.;/' or l.endswith(':'):
                    continue
                c[l.split()[0]] += 1
            return c
        # innermost loop with the most MFMAs: smallest span among those with the maximum MFMA count
        def nm(ab):
            return sum(v for o, v in ops(*ab).items() if o.startswith('v_mfma'))
        top = max(nm(ab) for ab in loops)
        best = min((ab for ab in loops if nm(ab) == top), key=lambda ab: ab[1] - ab[0])
        c = ops(*best)
        valu = sum(v for o, v in c.items() if o.startswith('v_') and not o.startswith('v_mfma'))
        print(f'{name}\n   loop of {best[1] - best[0]} lines: {top} MFMA, {valu} VALU, '
              f'{sum(v for o, v in c.items() if o.startswith("ds_"))} LDS, '
              f'{sum(v for o, v in c.items() if o.startswith(("global_", "buffer_")))} VMEM, '
              f'{sum(v for o, v in c.items() if o.startswith("s_"))} scalar')
        print('   ' + ', '.join(f'{o} {v}' for o, v in c.most_common() if o.startswith('v_') and not o.startswith('v_mfma')))


if __name__ == '__main__':
    main()
