"""Static check of the persistent GEMM's hand-counted global loads (developer tool).

For every inline-asm `global_load_dwordx4 / dwordx2 v[a:b], ...` in the dense_bf16_w4 kernels of an
assembly listing, the first later instruction that names one of v[a..b] must come AFTER an
`s_waitcnt vmcnt(...)`: the compiler believes the asm output is valid at once, so a copy or a
use placed before the wait would read the register before the load has landed.
Usage: hipcc -S --cuda-device-only ... -o w4.s gemm_bf16_w4.hip; python check_async_regs.py w4.s
"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def main(path):
    lines = open(path).read().split('\n')
    bad = 0
    checked = 0
    kernel = None
    for i, ln in enumerate(lines):
        m = re.match(r'^(_ZN4gnlp20dense_bf16_w4_kernel\w+):', ln)
        if m:
            kernel = m.group(1)
        if ln.startswith('.Lfunc_end'):
            kernel = None
        if kernel is None:
            continue
        t = ln.strip()
        if not (t.startswith('global_load_dwordx4') or t.startswith('global_load_dwordx2')):
            continue
        dst = regs_of(t.split()[1].rstrip(','))
        checked += 1
        waited = False
        for j in range(i + 1, min(i + 4000, len(lines))):
            u = lines[j].strip()
            if u.startswith('.Lfunc_end'):
                break
            if u.startswith('s_waitcnt') and 'vmcnt' in u:
                waited = True
                break
            if u.startswith(';') or u.startswith('.') or not u:
                continue
            if u.startswith('s_cbranch') or u.startswith('s_branch'):
                # straight-line check only; a branch before the wait is reported
                print('%s:%d branch before wait after load at line %d' % (kernel, j + 1, i + 1))
                bad += 1
                break
            toks = re.findall(r'v\[\d+:\d+\]|v\d+', u)
            # the load's own address operand may alias later instructions' operands; any mention
            # of a destination register before the wait is an error
            if any(regs_of(tk) & dst for tk in toks):
                print('%s: line %d uses %s before a vmcnt wait (load at line %d)' % (
                    kernel, j + 1, u, i + 1))
                bad += 1
                break
        if not waited and not bad:
            pass
    print('checked %d asm loads, %d problems' % (checked, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
