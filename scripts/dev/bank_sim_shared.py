import numpy as np
rng=np.random.default_rng(1)
def run(pst, layout, r0s=(21,25,29,33), trials=30, n=256):
    tot=0; base=0
    for r0 in r0s:
      for tr in range(trials):
        cx=46+rng.integers(-3,4)+rng.random()*2-1; cy=46+rng.integers(-3,4)+rng.random()*2-1
        lanes=np.arange(64)
        for step in range(16 if layout=='old' else 4):
          for u in range(2):
            if layout=='old':
                sub=lanes//8; t=lanes%8; a=step
                j=2*(8*a+t)+u; r=r0+sub
            elif layout=='sh':
                g=lanes>>4; rr=(lanes>>3)&1; t=lanes&7; i=step
                j=64*i+2*(8*g+t)+u; r=r0+rr
            elif layout=='sh_rot':   # group g processes quadrant (i+g)%4
                g=lanes>>4; rr=(lanes>>3)&1; t=lanes&7; i=(step+g)%4
                j=64*i+2*(8*g+t)+u; r=r0+rr
            phi=2*np.pi*j/n
            x=cx+r*np.sin(phi); y=cy+r*np.cos(phi)
            ix=np.floor(x).astype(int); iy=np.floor(y).astype(int)
            for dy in (0,1):
                for dx in (0,1):
                    addr=(iy+dy)*pst+ix+dx
                    for grp in (slice(0,32),slice(32,64)):
                        ad=np.unique(addr[grp]); b=ad%32
                        tot+=np.bincount(b,minlength=32).max(); base+=1
    return tot/base
for pst in (101,103,105,107,109,111,113,115,117):
    print(pst, 'old %.3f  sh %.3f  sh_rot %.3f' % (run(pst,'old'), run(pst,'sh'), run(pst,'sh_rot')))
