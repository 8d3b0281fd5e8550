"""Per basic block of one kernel's ISA: how many vector / scalar / wait / branch / LDS / memory instructions it holds, with LLVM's loop annotations.
usage: python scripts/isa_blocks.py <file.s> <substring of the kernel's mangled name> [min instructions per block]
(the .s comes from `hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only`; used to see where a score loop's scalar instructions sit)"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 6
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l.split(':')[0]][0]
end = [i for i, l in enumerate(lines) if i > start and '.end_amdhsa_kernel' in l][0]
blocks = []
cur = {'name': 'entry', 'hdr': '', 'ops': {}, 'line': 0}
for n, l in enumerate(lines[start:end]):
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?', l)
    if m:
        blocks.append(cur)
        cur = {'name': m.group(1), 'hdr': (m.group(2) or ''), 'ops': {}, 'line': n}
        continue
    if l.strip().startswith(';') and 'Loop' in l:
        cur['hdr'] += ' ' + l.strip()
    t = l.strip().split()
    if not t:
        continue
    op = t[0]
    if op.startswith('v_'):
        k = 'valu'
    elif op.startswith('s_'):
        k = 'wait' if op in ('s_waitcnt', 's_nop') else 'br' if op.startswith(('s_cbranch', 's_branch')) else 'salu'
    elif op.startswith('ds_'):
        k = 'lds'
    elif op.startswith(('global_', 'buffer_', 'flat_')):
        k = 'vmem'
    else:
        continue
    cur['ops'][k] = cur['ops'].get(k, 0) + 1
blocks.append(cur)
tot = {}
for b in blocks:
    for k, v in b['ops'].items():
        tot[k] = tot.get(k, 0) + v
    if sum(b['ops'].values()) >= minn:
        print(b['line'], b['name'], b['ops'], re.sub(r'\s+', ' ', b['hdr'])[:110])
print('total', tot)
