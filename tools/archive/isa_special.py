import re,sys
kernel=sys.argv[1] if len(sys.argv)>1 else 'decode_single_kernel'
text=open('/tmp/isa/dint_hip-hip-amdgcn-amd-amdhsa-gfx950.s').read()
body=text[text.index(kernel+'ENS_11decode_argsE:'):]
body=body[:body.index('s_endpgm')]
cur='prologue'; cnt={}; order=[]
for line in body.split('\n'):
    m=re.search(r'; MARK (\w+)',line)
    if m: cur=m.group(1); continue
    t=line.split()
    if not t: continue
    key=None
    if t[0] in('v_writelane_b32','v_readlane_b32','v_readfirstlane_b32','ds_bpermute_b32','v_mul_lo_u32'): key=t[0]
    if t[0].startswith('scratch_'): key=t[0]
    if key:
        if cur not in order: order.append(cur)
        cnt.setdefault(cur,{}).setdefault(key,0); cnt[cur][key]+=1
for k in order: print(k,cnt[k])
m=re.search(r'\.amdhsa_kernel \S*'+kernel+r'.*?\.end_amdhsa_kernel',text,re.S)
for key in ('next_free_vgpr','next_free_sgpr','private_segment_fixed_size'):
    mm=re.search(r'\.amdhsa_'+key+r'\s+(\S+)',m.group(0)); print(key,mm.group(1) if mm else None)
