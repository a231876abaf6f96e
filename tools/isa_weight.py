"""tools/isa_weight.py FILE.s — executed instructions per site of a generated lane kernel, from its listing: every
backward branch closes a loop; loops nested inside the chunk loop run 3 times per level (the looped members' digits)."""
import re,sys,collections
f=sys.argv[1]
lines=[l for l in open(f).read().split('\n')]
addr={}
for i,l in enumerate(lines):
    m=re.search(r'// ([0-9A-F]{12}):',l)
    if m: addr[int(m.group(1),16)]=i
base=min(addr)
loops=[]
for i,l in enumerate(lines):
    m=re.search(r's_cbranch_\w+ (\d+)\s+// ([0-9A-F]{12}):.*<\w+\+0x([0-9a-f]+)>',l)
    if m:
        tgt=base+int(m.group(3),16); here=int(m.group(2),16)
        if tgt<here and tgt in addr: loops.append((addr[tgt],i))
loops.sort(key=lambda ab:(ab[0],-ab[1]))
# the chunk loop = the largest; digit loops = loops inside it with > 300 instructions
big=max(loops,key=lambda ab:ab[1]-ab[0])
digit=[ab for ab in loops if ab!=big and ab[0]>=big[0] and ab[1]<=big[1] and ab[1]-ab[0]>300]
print("chunk loop",big,"digit loops",digit)
w=collections.Counter(); tot=0
kinds=collections.defaultdict(collections.Counter)
for i in range(big[0],big[1]+1):
    l=lines[i].strip()
    if not l or l.startswith('//') or ':' in l.split()[0]: continue
    d=sum(1 for a,b in digit if a<=i<=b)
    w[d]+=1
    op=l.split()[0]
    k='fp64' if op.startswith(('v_fma','v_mul_f64','v_add_f64','v_div','v_rcp_f64')) else ('acc' if 'accvgpr' in op else ('lds' if op.startswith('ds_') else ('smem' if op.startswith('s_load') else ('wait' if op=='s_waitcnt' else 'other'))))
    kinds[d][k]+=1
N=len(digit)
total=sum(c*3**d for d,c in w.items())
for d in sorted(w):
    print("depth",d,"static",w[d],"x",3**d,"=",w[d]*3**d,dict(kinds[d]))
print("executed per site ~",total,"; per configuration at 3^%d x block"%N)
