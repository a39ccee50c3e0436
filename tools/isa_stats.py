#!/usr/bin/env python3
"""Per-kernel instruction counts and resource use from `hipcc -S --cuda-device-only` output.
usage: isa_stats.py k.s [substring-of-kernel-name ...]"""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2:]
pat = re.compile(r'^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel', re.S | re.M)
ops = ['global_load_dwordx4', 'global_load_dwordx3', 'global_load_dwordx2', 'global_load_dword ', 'buffer_load',
       'ds_read_b128', 'ds_read_b64', 'ds_read_b32', 'ds_read2', 'ds_write_b128', 'ds_write_b32', 's_barrier',
       'scratch_', 's_waitcnt vmcnt(0)', 's_load_dword', 'v_cndmask', 's_cbranch']
for m in pat.finditer(text):
    name, body = m.group(1), m.group(2)
    if want and not all(w in name for w in want):
        continue
    print(name)
    print('   ' + '  '.join(f'{k.strip()}={body.count(k)}' for k in ops if body.count(k)))
    res = []
    for k in ['.amdhsa_next_free_vgpr', '.amdhsa_next_free_sgpr', '.amdhsa_group_segment_fixed_size',
              '.amdhsa_private_segment_fixed_size']:
        mm = re.search(re.escape(k) + r'\s+(\S+)', body)
        res.append(f'{k[8:]}={mm.group(1) if mm else None}')
    mm = re.search(re.escape(name) + r'\.num_vgpr, (\d+)', text)
    res.append(f'num_vgpr={mm.group(1) if mm else None}')
    print('   ' + '  '.join(res))
