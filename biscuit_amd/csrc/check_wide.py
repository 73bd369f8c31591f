#!/usr/bin/env python3
"""Build-time guard for kernels_wide.hip (run by `make`).

The kernel issues its MFMAs as inline asm (accumulators tied in place in the accumulator file) and its LDS-DMA as
inline asm too, so hipcc neither pads their hazards nor counts their memory operations.  This script reads the
generated assembly and fails the build unless, in every sepconv_wide_kernel instance:
  * nothing uses scratch (a spill reload is a vector-memory operation the loop's counted `s_waitcnt vmcnt(N)` does not
    expect; it would also sit in the hot loop);
  * every v_mfma keeps its accumulator in place (dst == C) in the accumulator file;
  * no VALU instruction writes a register of an MFMA's A/B operands within the two instructions in front of it
    (the 2 wait states hipcc would have inserted had it seen the MFMA);
  * every v_accvgpr_read of the epilogue is at least 16 wait states behind the last MFMA (the explicit s_nops).
"""
import re
import sys

path = sys.argv[1]
text = open(path).read()
kernels = re.findall(r'^(_ZN\S*sepconv_wide_kernel\S*):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M)
if not kernels:
    sys.exit('check_wide: no sepconv_wide_kernel instance found in ' + path)
meta = dict(re.findall(r'\.name:\s+(\S*sepconv_wide_kernel\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)', text))
bad = []
rng = re.compile(r'([va])\[(\d+):(\d+)\]|([va])(\d+)\b')


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
    if 'scratch_' in body:
        bad.append(f'{name}: scratch access ({body.count("scratch_")} instructions)')
    ins = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((';', '.'))]
    n_mfma = 0
    last_mfma = None
    for k, ln in enumerate(ins):
        if ln.startswith('v_mfma'):
            n_mfma += 1
            last_mfma = k
            ops = [o.strip() for o in ln.split(None, 1)[1].split(',')]
            if ops[0] != ops[3] or not ops[0].startswith('a['):
                bad.append(f'{name}: MFMA accumulator not in place: {ln}')
            ab = regs(ops[1]) | regs(ops[2])
            for back in (1, 2):
                if k - back >= 0 and VALU.match(ins[k - back]):
                    dst = regs(ins[k - back].split(None, 1)[1].split(',')[0])
                    if dst & ab:
                        bad.append(f'{name}: VALU write {ins[k - back]!r} {back} instruction(s) before {ln!r}')
    if n_mfma == 0:
        bad.append(f'{name}: no MFMA found')
    # the first accumulator read after the last MFMA must be behind >= 16 wait states of s_nop
    states = 0
    for ln in ins[last_mfma + 1:]:
        if ln.startswith('v_accvgpr_read'):
            if states < 16:
                bad.append(f'{name}: accumulator read {states} wait states after the last MFMA')
            break
        m = re.match(r's_nop\s+(\d+)', ln)
        states += int(m.group(1)) + 1 if m else 1
if bad:
    print('check_wide: FAILED', file=sys.stderr)
    for b in bad[:20]:
        print('  ' + b, file=sys.stderr)
    sys.exit(1)
print(f'check_wide: ok ({len(kernels)} instances: accumulators in place, no scratch, no VALU->MFMA operand hazard)')
