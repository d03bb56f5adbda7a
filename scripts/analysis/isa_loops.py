"""Loops of one kernel in a device assembly listing (hipcc --cuda-device-only -S): registers, and per backward branch the number of
instructions, VALU, fp64 and LDS instructions between the label and the branch.
python scripts/analysis/isa_loops.py listing.s <mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
a = next(i for i, l in enumerate(s) if l.startswith(name + ':'))
b = next(i for i in range(a, len(s)) if '.amdhsa_kernel ' + name in s[i])
body = s[a:b]
meta = '\n'.join(s[b:b + 120])
for key in ('next_free_vgpr', 'next_free_sgpr', 'accum_offset', 'private_segment_fixed_size', 'group_segment_fixed_size'):
    mm = re.search(r'\.amdhsa_' + key + r'\s+(\S+)', meta)
    print(key, mm.group(1) if mm else None)
print('instructions', sum(1 for l in body if re.match(r'\s+[sv]_|\s+ds_|\s+global_|\s+buffer_|\s+scratch_|\s+flat_', l)),
      'scratch', sum(1 for l in body if re.match(r'\s+scratch_', l)))
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < i:
            seg = body[labels[t]:i + 1]
            ins = [x for x in seg if re.match(r'\s+[sv]_|\s+ds_|\s+global_|\s+buffer_|\s+scratch_|\s+flat_', x)]
            print(t, 'lines', labels[t], i, 'instr', len(ins), 'valu', sum(1 for x in ins if re.match(r'\s+v_', x)),
                  'f64', sum(1 for x in ins if '_f64' in x), 'lds', sum(1 for x in ins if re.match(r'\s+ds_', x)),
                  'rcp/sqrt/div', sum(1 for x in ins if re.search(r'v_rcp|v_rsq|v_sqrt|v_div', x)))
