# LDS bank-conflict model of the bilinear tap reads of a ring job (ds_read2_b32: two 32-lane groups per dword, bank =
# dword address mod 32; MI355X_MICROARCH.md, LDS): average cycles per 32-lane access for candidate row strides of the padded
# image and for alternative lane orders.  python scripts/bank_sim.py
import numpy as np
rng=np.random.default_rng(0)
def conflicts(pst, r0, LR=8, R1=16, n=256, trials=40, lanemap=None):
    # code-6 job: 8 instances (radii r0..r0+7), lanes = sub*8+t
    tot=0; base=0
    for tr in range(trials):
        cx=46+rng.integers(-3,4)+rng.random()*0; cy=46+rng.integers(-3,4)
        cx+=rng.random()*2-1; cy+=rng.random()*2-1   # accumulated fractional centre
        for a in range(R1):
            for u in range(2):
                lanes=np.arange(64); sub=lanes//LR; t=lanes%LR
                j=2*(LR*a+t)+u
                phi=2*np.pi*j/n
                r=r0+sub
                x=cx+r*np.sin(phi); y=cy+r*np.cos(phi)
                ix=np.floor(x).astype(int); iy=np.floor(y).astype(int)
                for dy in (0,1):
                    for dx in (0,1):
                        addr=(iy+dy)*pst+ix+dx
                        for grp in (slice(0,32),slice(32,64)):
                            ad=np.unique(addr[grp]); b=ad%32
                            c=np.bincount(b,minlength=32).max()
                            tot+=c; base+=1
    return tot/base
for pst in range(100,118):
    print(pst, round(np.mean([conflicts(pst,r0) for r0 in (21,29)]),3))
print("--- slot-major within ring: half-wave = 1 ring x 4 x-adjacent slots x 8 lanes")
def conflicts2(pst, r0, LR=8, R1=16, n=256, trials=40, slots=((0,0),(1,0),(2,0),(3,0))):
    tot=0; base=0
    for tr in range(trials):
        cx=46+rng.integers(-3,1)+rng.random()*2-1; cy=46+rng.integers(-3,4)+rng.random()*2-1
        for a in range(R1):
            for u in range(2):
                lanes=np.arange(64); sub=lanes//LR; t=lanes%LR
                ringl=sub//4; s=sub%4
                sx=np.array([slots[k][0] for k in s]); sy=np.array([slots[k][1] for k in s])
                j=2*(LR*a+t)+u
                phi=2*np.pi*j/n
                r=r0+ringl
                x=cx+sx+r*np.sin(phi); y=cy+sy+r*np.cos(phi)
                ix=np.floor(x).astype(int); iy=np.floor(y).astype(int)
                for dy in (0,1):
                    for dx in (0,1):
                        addr=(iy+dy)*pst+ix+dx
                        for grp in (slice(0,32),slice(32,64)):
                            ad=np.unique(addr[grp]); b=ad%32
                            tot+=np.bincount(b,minlength=32).max(); base+=1
    return tot/base
for pst in (100,101,103,105,109):
    print(pst, 'x-adjacent', round(np.mean([conflicts2(pst,r0) for r0 in (21,25,29,33)]),3),
          'straddle(3+1)', round(np.mean([conflicts2(pst,r0,slots=((1,0),(2,0),(3,0),(-3,1))) for r0 in (21,29)]),3),
          'straddle(2+2)', round(np.mean([conflicts2(pst,r0,slots=((2,0),(3,0),(-3,1),(-2,1))) for r0 in (21,29)]),3))
print("--- hypothetical: half-wave = 32 consecutive samples (stride 1 or 2) of one ring")
def conflicts3(pst, r, stride, n=256, trials=200):
    tot=0; base=0
    for tr in range(trials):
        cx=46+rng.integers(-3,4)+rng.random()*2-1; cy=46+rng.integers(-3,4)+rng.random()*2-1
        j0=rng.integers(0,n)
        j=j0+stride*np.arange(32)
        phi=2*np.pi*j/n
        x=cx+r*np.sin(phi); y=cy+r*np.cos(phi)
        ix=np.floor(x).astype(int); iy=np.floor(y).astype(int)
        for dy in (0,1):
            for dx in (0,1):
                addr=(iy+dy)*pst+ix+dx
                ad=np.unique(addr); b=ad%32
                tot+=np.bincount(b,minlength=32).max(); base+=1
    return tot/base
for pst in (100,101,103,105,109,113):
    print(pst, [round(conflicts3(pst,r,s),2) for r in (22,30,36) for s in (1,2)])
