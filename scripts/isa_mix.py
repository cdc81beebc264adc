"""Instruction mix of a kernel from hipcc's saved assembly.
usage: isa_mix.py <file.s> <substring of the mangled kernel name> [...]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
for pat in sys.argv[2:]:
    i = 0
    while i < len(lines):
        l = lines[i]
        if re.match(r'^_Z\w+:', l) and pat in l.split(':')[0]:
            name = l.split(':')[0]
            ops = collections.Counter()
            j = i + 1
            while not lines[j].startswith('.Lfunc_end'):
                t = lines[j].strip()
                if t and not t.startswith(('.', ';', '//')) and not t.endswith(':'):
                    ops[t.split()[0]] += 1
                j += 1
            cat = collections.Counter()
            for o, c in ops.items():
                k = ('valu' if o.startswith('v_') else 'salu' if o.startswith('s_') else
                     'gmem' if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'lds' if o.startswith('ds_') else 'other')
                cat[k] += c
            print(name[:110], sum(ops.values()), dict(cat))
            print('   ', ops.most_common(14))
            i = j
        i += 1
    for m in re.finditer(r'\.amdhsa_kernel (\S*' + re.escape(pat) + r'\S*)\n(.*?)\.end_amdhsa_kernel', '\n'.join(lines), re.S):
        vals = {k: v for k, v in re.findall(r'\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|group_segment_fixed_size) (\d+)', m.group(2))}
        print('   ', m.group(1)[:60], vals)
