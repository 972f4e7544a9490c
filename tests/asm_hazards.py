"""Test helper: scan a -S assembly of csrc/gemm6.hip for instructions that touch the destination registers of a vector-memory
load that is still in flight by the hand-counted vmcnt discipline (inline-asm loads the compiler believes complete).
Linear scan per kernel; loop bodies (label .. backward branch to it) are scanned a second time with the state at the
branch.  usage: asm_hazards.py file.s [kernel-name-substring]"""
import re, sys

def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
def all_regs(line):
    out = set()
    for tok in re.findall(r'v\[\d+:\d+\]|v\d+', line):
        out |= regs_of(tok)
    return out
def scan(lines, fifo, name, report):
    for ln, line in lines:
        s = line.strip()
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
        op = s.split()[0]
        m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', s)
        if op == 's_waitcnt' and 'vmcnt' in s:
            n = int(re.search(r'vmcnt\((\d+)\)', s).group(1))
            while len(fifo) > n: fifo.pop(0)
            continue
        touched = all_regs(s)
        inflight = set().union(*[d for d, _ in fifo]) if fifo else set()
        if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
            toks = s.split(None, 1)[1].split(',')
            if 'lds' in op:
                dest, addr = set(), all_regs(s)
            else:
                dest, addr = regs_of(toks[0].strip()), all_regs(','.join(toks[1:]))
            bad = (addr | dest) & inflight
            if bad: report.append((name, ln, s, sorted(bad)))
            fifo.append((dest, ln))
            continue
        if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic')):
            bad = touched & inflight
            if bad: report.append((name, ln, s, sorted(bad)))
            fifo.append((set(), ln))
            continue
        bad = touched & inflight
        if bad: report.append((name, ln, s, sorted(bad)))
    return fifo
def scan_file(path, pat='k_gemm6'):
    """-> {kernel symbol: [(where, line number, instruction, registers in flight)]}"""
    text = open(path).read().split('\n')
    out = {}
    starts = [i for i, l in enumerate(text) if re.match(r'^_Z\w+:', l) and pat in l]
    for st in starts:
        name = text[st].split(':')[0]
        end = next(i for i in range(st, len(text)) if 's_endpgm' in text[i])
        body = [(i + 1, text[i]) for i in range(st, end)]
        report = []
        scan(body, [], name, report)
        # loops: label L at index a, a later branch to L at index b > a -> rescan [a, b] with the state at b
        labels = {l.strip()[:-1]: k for k, (_, l) in enumerate(body) if re.match(r'^\.LBB\w+:', l.strip())}
        for k, (_, l) in enumerate(body):
            m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\w+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                a = labels[m.group(1)]
                f = scan(body[:k], [], name, [])
                scan(body[a:k], f, name + ' (loop again)', report)
        out[name] = report
    return out


if __name__ == '__main__':
    for name, report in scan_file(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'k_gemm6').items():
        print(f'{name[:90]}: {len(report)} suspicious instruction(s)')
        for r in report[:12]:
            print('   line', r[1], r[2][:90], 'in flight:', r[3][:8])
