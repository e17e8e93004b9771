import re
from collections import Counter
s=open('/tmp/capi.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S*ps_ransac_score_(?:mfma)\S*)(.*?)\.end_amdhsa_kernel', s, re.S):
    body=m.group(2)
    g=lambda k: re.search(k+r'\s+(\S+)', body).group(1)
    print(m.group(1)[:60], 'vgpr', g(r'\.amdhsa_next_free_vgpr'), 'sgpr', g(r'\.amdhsa_next_free_sgpr'), 'scratch', g(r'\.amdhsa_private_segment_fixed_size'), 'lds', g(r'\.amdhsa_group_segment_fixed_size'))
i=s.index('\n_ZN5psdev20ps_ransac_score_mfma'); j=s.index('.end_amdhsa_kernel',i)
body=s[i:j]
blocks=re.split(r"\n(\.LBB\d+_\d+):", body)
for k in range(1,len(blocks),2):
    b=blocks[k+1]
    n=b.count('v_mfma'); sc=len(re.findall(r'scratch_(?:load|store)', b))
    ins=[l.split()[0] for l in b.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.',';'))]
    if n or sc:
        c=Counter(ins)
        print(blocks[k], 'mfma',n,'instrs',len(ins),'valu',sum(v for q,v in c.items() if q.startswith('v_') and 'mfma' not in q),'scratch',sc, 'v_mov', c.get('v_mov_b32_e32',0))
