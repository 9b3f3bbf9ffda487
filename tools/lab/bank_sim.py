import itertools
def f_old(r): return (r ^ (r>>3)) & 7
def f_new(r): return (((r>>1)&3) ^ ((r>>4)&3)) | (((r ^ (r>>3)) & 1) << 2)
def off(f,row,slot,extra_bytes=0): return row*128 + ((slot ^ f(row))&7)*16 + extra_bytes
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 += [[l+32 for l in g] for g in G128]
def cycles(addrs, width, nbanks):
    # addrs: byte addresses for the lanes in one group; each covers `width` bytes; returns max distinct addresses per bank
    bank = {}
    for a in addrs:
        for d in range(width//4):
            b = ((a//4)+d) % nbanks
            bank.setdefault(b,set()).add((a//4)+d)
    return max(len(v) for v in bank.values())
for name,f in (("old",f_old),("new",f_new)):
    worst=0; tot=0
    for base in range(0,256,16):
        for ks in (0,1):
            for grp in G128:
                addrs=[off(f, base+(l&15), ks*4+(l>>4)) for l in grp]
                c=cycles(addrs,16,64); worst=max(worst,c); tot+=c
    print(name,"frag read b128 worst",worst,"avg",tot/(16*2*4))
    # GPTQ B write: thread tid -> col tid%128, slot = brow + r, brow = 2*(tid//128)?? WPT=2: brow = WPT*(tid/BN) ; 8 contiguous lanes per group
    worst=0
    for tid0 in range(0,512,8):
        for r in range(2):
            addrs=[off(f,(t%128), 2*(t//128)+r) for t in range(tid0,tid0+8)]
            worst=max(worst,cycles(addrs,16,32))
    print(name,"gptq write b128 worst",worst)
    worst=0
    for q in range(4):
        for tid0 in range(0,512,8):
            addrs=[off(f,(t+512*q)>>3,(t+512*q)&7) for t in range(tid0,tid0+8)]
            worst=max(worst,cycles(addrs,16,32))
    print(name,"A write b128 worst",worst)
    worst=0
    for c in range(8):
        for h in range(1):
            for tid0 in range(0,512,32):
                addrs=[]
                for t in range(tid0,tid0+32):
                    bcol=8*(t%16); brow=2*(t//16)
                    addrs.append(off(f,bcol+c,brow>>3,((brow&7)+2*h)*2))
                worst=max(worst,cycles(addrs,4,32))
    print(name,"awq write b32 worst",worst)
