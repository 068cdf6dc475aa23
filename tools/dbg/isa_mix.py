"""instruction mix of the kernels in a hipcc -S listing: python3 tools/dbg/isa_mix.py file.s [filter]"""
import re, collections, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else 'ILi5ELi7ELi5ELi8'
for m in re.finditer(r'^(_ZN2z3[^\n:]*):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ops = collections.Counter(l.split()[0] for l in body.split('\n') if re.match(r'\t(v_|ds_|global_|s_|buffer_|scratch_)', l))
    v = sum(c for o, c in ops.items() if o.startswith('v_'))
    pk = sum(c for o, c in ops.items() if o.startswith('v_pk_'))
    print(re.sub(r'INS_4Plan.*', '', name)[7:], 'total', sum(ops.values()), 'valu', v, 'pk', pk, 'ds', sum(c for o, c in ops.items() if o.startswith('ds_')),
          'scratch', sum(c for o, c in ops.items() if o.startswith('scratch_')))
    print('   ', ', '.join('%s %d' % (o, c) for o, c in ops.most_common(24)))
