"""tools/loopstat.py FILE.s [K] — instruction mix of the K smallest loops of more than 400 instructions in an
llvm-objdump -d listing of a generated kernel (the unrolled block's loop is the hot one)."""
import re,sys,collections
f=sys.argv[1]
lines=open(f).read().split('\n')
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
big=sorted((b-a,a,b) for a,b in loops if b-a>400)
for size,a,b in big[:int(sys.argv[2]) if len(sys.argv)>2 else 1]:
    c=collections.Counter(l.split()[0] for l in lines[a:b+1] if l.strip())
    fp=sum(v for k,v in c.items() if k.startswith(('v_fma','v_mul_f64','v_add_f64')))
    print(f, "lines",a,b,"instr",size+1,"fp64",fp, dict(c.most_common(16)))
