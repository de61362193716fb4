#!/usr/bin/env python
"""Where a kernel's scratch accesses are: counts scratch loads / stores of the named functions in a device assembly file by the LLVM
loop depth of the enclosing basic block.  usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPCP_TU=0 --cuda-device-only -S pcp_kernels.hip -o k0.s;
       python tools/scratch_depth.py k0.s <mangled kernel name> ..."""
import re, sys, collections
t = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    i = t.find('\n' + name + ':')
    if i < 0:
        print(name, 'not found'); continue
    j = t.index('.Lfunc_end', i)
    depth = 0; hdr = ''
    per = collections.Counter(); ex = {}
    n_instr = 0
    for line in t[i:j].split('\n'):
        s_ = line.strip()
        m = re.match(r'^\.LBB\S+:\s*;\s*(.*)$', s_)
        if s_.startswith('.LBB') or s_.startswith('; %bb'):
            mm = re.search(r'Depth=(\d+)', s_)
            depth = int(mm.group(1)) if mm else 0
            continue
        if not s_ or s_.startswith((';', '.')): continue
        n_instr += 1
        op = s_.split()[0]
        if op.startswith('scratch_') or (op.startswith('buffer_') and 'offen' in s_ and 's[0:3]' in s_):
            per[(op.split('_')[1], depth)] += 1
    print(name[:60], 'instructions', n_instr, 'scratch accesses by (kind, loop depth):', dict(sorted(per.items())))
