"""Instruction mix per kernel of an assembly dump (hipcc -S --cuda-device-only): tools/asm_mix.py file.s [name-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('s_endpgm')[0]
    if pat not in name: continue
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    g = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))
    print(name[:60], 'total', len(ins), 'mfma', g('v_mfma'), 'valu', g('v_') - g('v_mfma'), 'salu', g('s_') - c['s_waitcnt'] - c['s_nop'], 'ds', g('ds_'),
          'gload', g('global_load'), 'gstore', g('global_store'), 'scratch', g('scratch'), 'waitcnt', c['s_waitcnt'], 'nop', c['s_nop'], 'mov', c['v_mov_b32'] + g('v_accvgpr'))
