"""Which FMA contractions does ATen's CPU grid_sample (4-D, vectorised kernel) have?  Emulates the candidates with exact
rational arithmetic and counts bit mismatches against torch.nn.functional.grid_sample (0 = that variant is what ATen does).
Result on torch 2.10: align_corners=False unnormalise is an fma, the four-corner blend is nw*w -> fma -> fma -> fma."""
import numpy as np, torch, torch.nn.functional as F
from fractions import Fraction
f32=np.float32
def rnd32(fr):
    # correctly rounded Fraction -> float32
    if fr == 0: return f32(0)
    d = float(fr)  # rounds to f64 (RNE); double rounding possible -> fix by checking neighbours
    x = f32(d)
    # ensure nearest: compare neighbours exactly
    cands=[x, np.nextafter(x, f32(np.inf)), np.nextafter(x, f32(-np.inf))]
    best=min(cands, key=lambda c:(abs(Fraction(float(c))-fr), int(c.view(np.uint32))&1))
    return best
def fma32(a,b,c): return rnd32(Fraction(float(a))*Fraction(float(b))+Fraction(float(c)))

torch.manual_seed(0)
H,W=7,9
inp=torch.randn(1,1,H,W)
g=torch.Generator().manual_seed(1)
P=400
coords=torch.rand(1,1,P,2,generator=g)*torch.tensor([W+2.0,H+2.0])-1.0   # pixel-ish coords incl. out of range
for align in (True,False):
  for pad in ("zeros","border"):
    sc=torch.tensor([2/max(W-1,1),2/max(H-1,1)]) if align else torch.tensor([2/W,2/H])
    gr=coords*sc; gr=gr-1
    ref=F.grid_sample(inp,gr,align_corners=align,padding_mode=pad)[0,0,0].numpy()
    inn=inp[0,0].numpy(); G=gr[0,0].numpy()
    res={}
    for unn in ("plain","fms"):
      for blend in ("fma_chain","plain","fma_rev"):
        out=np.zeros(P,f32)
        for p in range(P):
          ax=[]
          for a,size in ((0,W),(1,H)):
            gg=f32(G[p,a])
            if align:
              u=f32(f32(gg+f32(1))*f32(f32(size-1)/f32(2)))
            else:
              sf=f32(f32(size)/f32(2))
              if unn=="plain": u=f32(f32(f32(gg+f32(1))*sf)-f32(0.5))
              else: u=fma32(f32(gg+f32(1)),sf,f32(-0.5))
            if pad=="border": u=min(f32(size-1),max(u,f32(0)))
            fl=np.floor(u); w1=f32(u-fl); w0=f32(f32(1)-w1)
            ax.append((int(fl),w0,w1))
          (x0,e,w),(y0,s,n)=ax
          nw=f32(s*e); ne=f32(s*w); sw=f32(n*e); se=f32(n*w)
          def val(y,x): return inn[y,x] if (0<=x<W and 0<=y<H) else f32(0)
          v=[val(y0,x0),val(y0,x0+1),val(y0+1,x0),val(y0+1,x0+1)]; ws=[nw,ne,sw,se]
          if blend=="plain":
            o=f32(f32(f32(v[0]*ws[0])+f32(v[1]*ws[1]))+f32(v[2]*ws[2])); o=f32(o+f32(v[3]*ws[3]))
          elif blend=="fma_chain":
            o=f32(v[0]*ws[0]); o=fma32(v[1],ws[1],o); o=fma32(v[2],ws[2],o); o=fma32(v[3],ws[3],o)
          else:
            o=f32(v[1]*ws[1]); o=fma32(v[0],ws[0],o); o=fma32(v[2],ws[2],o); o=fma32(v[3],ws[3],o)
          out[p]=o
        res[(unn,blend)]=int((out.view(np.uint32)!=ref.view(np.uint32)).sum())
    print(align,pad,res)
