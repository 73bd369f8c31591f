#!/usr/bin/env python3
"""Build-time guard for kernels_wide.hip (run by `make`; takes any number of assembly files).

The kernel issues its MFMAs as inline asm (accumulators tied in place in the accumulator file) and its LDS-DMA as
inline asm too, so hipcc neither pads their hazards nor counts their memory operations.  This script reads the
generated assembly and fails the build unless, in every sepconv_wide*_kernel instance:
  * nothing uses scratch between the first and the last MFMA, i.e. inside the K loop (a spill reload is a vector-memory
    operation the loop's counted `s_waitcnt vmcnt(N)` does not expect; it would also sit in the hot loop).  The
    persistent kernel's once-per-workgroup prologue and per-tile epilogue may park a few per-lane constants there (the
    compiler counts those itself); more than 8 spilled dwords fail the build all the same;
  * every v_mfma keeps its accumulator in place (dst == C); a tile kept in ordinary registers is touched by no other
    instruction between the first and the last MFMA, and nothing but MFMAs touches the accumulator file in between;
  * no VALU instruction writes a register of an MFMA's A/B operands within the two instructions in front of it
    (the 2 wait states hipcc would have inserted had it seen the MFMA);
  * the epilogue's first read of an accumulator is at least 18 wait states behind the last MFMA (the explicit s_nops).
"""
import re
import sys

warn_only = '--warn' in sys.argv          # experiments builds (in-kernel stamps cost registers): report, do not fail
kernels = []
for path in [a for a in sys.argv[1:] if a != '--warn']:
    found = re.findall(r'^(_ZN\S*sepconv_wide\d*_kernel\S*):[^\n]*\n(.*?)s_endpgm', open(path).read(), re.S | re.M)
    if not found:
        sys.exit('check_wide: no sepconv_wide*_kernel instance found in ' + path)
    kernels += found
bad = []
rng = re.compile(r'(?<![0-9A-Za-z_])([va])\[(\d+):(\d+)\]|(?<![0-9A-Za-z_])([va])(\d+)\b')


def regs(tok):
    out = set()
    for m in rng.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


VALU = re.compile(r'^v_(?!mfma|accvgpr)')
for name, body in kernels:
    ins = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((';', '.'))]
    n_mfma = 0
    first_mfma = last_mfma = None
    vacc = set()                        # accumulator tiles kept in ordinary registers
    for k, ln in enumerate(ins):
        if ln.startswith('v_mfma'):
            n_mfma += 1
            first_mfma = k if first_mfma is None else first_mfma
            last_mfma = k
            ops = [o.strip() for o in ln.split(None, 1)[1].split(',')]
            if (ops[0] != ops[3] and ops[3] != '0') or not ops[0].startswith(('a[', 'v[')):      # C = 0: a tile's first MFMA
                bad.append(f'{name}: MFMA accumulator not in place: {ln}')
            if ops[0].startswith('v['):
                vacc |= regs(ops[0])
            ab = regs(ops[1]) | regs(ops[2])
            for back in (1, 2):
                if k - back >= 0 and VALU.match(ins[k - back]):
                    dst = regs(ins[k - back].split(None, 1)[1].split(',')[0])
                    if dst & ab:
                        bad.append(f'{name}: VALU write {ins[k - back]!r} {back} instruction(s) before {ln!r}')
    if n_mfma == 0:
        bad.append(f'{name}: no MFMA found')
        continue
    in_loop = [ln for ln in ins[first_mfma:last_mfma + 1] if ln.startswith('scratch_')]
    if in_loop:
        bad.append(f'{name}: {len(in_loop)} scratch access(es) inside the K loop, e.g. {in_loop[0]!r}')
    spilled = sum({'dword': 1, 'dwordx2': 2, 'dwordx3': 3, 'dwordx4': 4}.get(ln.split()[0].rsplit('_', 1)[-1], 4)
                  for ln in ins if ln.startswith('scratch_store'))
    if spilled > 8:
        bad.append(f'{name}: {spilled} dwords spilled outside the K loop (limit 8)')
    for ln in ins[first_mfma:last_mfma]:
        if not ln.startswith('v_mfma') and ln.split(None, 1)[0][:2] in ('v_', 'ds', 'gl', 'bu', 'fl') and len(ln.split(None, 1)) > 1 \
                and regs(ln.split(None, 1)[1]) & vacc:
            bad.append(f'{name}: {ln!r} touches an accumulator tile kept in vector registers inside the loop')
    # hipcc does not know an asm MFMA's latency: between the first and the last MFMA nothing else may touch a register
    # of an accumulator tile (no copy of a tile out of the accumulator file, no use of a tile as spill space; accumulator
    # registers that hold no tile are the compiler's to use)
    acc_regs = set()
    for ln in ins:
        if ln.startswith('v_mfma'):
            acc_regs |= regs(ln.split(None, 1)[1].split(',')[0])
    for ln in ins[first_mfma:last_mfma]:
        if not ln.startswith('v_mfma') and len(ln.split(None, 1)) > 1 and regs(ln.split(None, 1)[1]) & acc_regs:
            bad.append(f'{name}: {ln!r} touches an accumulator tile between MFMAs')
    # weight fragments loaded by inline asm (kernels named *wide32*): hipcc does not know the load is asynchronous, so nothing
    # but an MFMA (behind the hand-written s_waitcnt) may read a register between such a load and its next overwrite
    pending = set()                     # (the loop only: the prologue also has loads hipcc issues and tracks itself)
    for ln in ins[first_mfma:last_mfma + 1]:
        parts = ln.split(None, 1)
        if len(parts) < 2:
            continue
        ops = [o.strip() for o in parts[1].split(',')]
        if parts[0] == 'global_load_dwordx4' and 'offset' in ln and name.find('wide32') >= 0:
            pending |= regs(ops[0])
            continue
        if parts[0].startswith('v_mfma'):       # behind its hand-written wait: the data has arrived
            pending -= regs(ops[1]) | regs(ops[2])
            continue
        used = regs(parts[1])
        if parts[0].startswith(('v_', 'ds_read', 'global_load', 'buffer_load')):      # first operand is a destination
            src = regs(','.join(ops[1:]))
            if src & pending:
                bad.append(f'{name}: {ln!r} reads a register of an in-flight asm weight load')
            pending -= regs(ops[0])
        elif used & pending:
            bad.append(f'{name}: {ln!r} reads a register of an in-flight asm weight load')
    # the first accumulator read after the last MFMA must be behind >= 18 wait states of s_nop
    states = 0
    for ln in ins[last_mfma + 1:]:
        touched = len(ln.split(None, 1)) > 1 and bool(regs(ln.split(None, 1)[1]) & acc_regs)
        if touched:
            if states < 18:
                bad.append(f'{name}: accumulator read {states} wait states after the last MFMA')
            break
        m = re.match(r's_nop\s+(\d+)', ln)
        states += int(m.group(1)) + 1 if m else 1
if bad:
    print('check_wide: FAILED' + (' (warning only)' if warn_only else ''), file=sys.stderr)
    for b in bad[:20]:
        print('  ' + b, file=sys.stderr)
    sys.exit(0 if warn_only else 1)
print(f'check_wide: ok ({len(kernels)} instances: accumulators in place, no scratch in the K loop, no VALU->MFMA operand hazard)')
