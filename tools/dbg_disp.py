import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_amd
from oracle import oracle as o
ctx = stereo_amd.Context(0)
def pair(seed,h,w,zero=0.05):
    rng=np.random.RandomState(seed); a=rng.randint(0,256,(h,w)).astype(np.uint8); b=rng.randint(0,256,(h,w)).astype(np.uint8)
    a[rng.rand(h,w)<zero]=0; b[rng.rand(h,w)<zero]=0; return a,b
for win,D,h,w in [(3,24,37,150),(5,24,37,150),(5,16,37,150),(5,8,20,40),(9,16,37,150),(9,32,37,150),(7,16,20,100),(1,24,10,100),(1,40,10,300)]:
    a,b=pair(1,h,w)
    want=o.disparity_ext(a,b,win,D,0); got=ctx.disparity(a,b,win=win,max_disp=D)
    bad=np.argwhere(got!=want)
    print("win",win,"D",D,"bad",len(bad),"of",h*w)
    if len(bad):
        print("  col%4 hist",np.bincount(bad[:,1]%4,minlength=4),"rows",np.unique(bad[:,0])[:12], "cols", np.unique(bad[:,1])[:20])
        for y,x in bad[:6]: print("   ",y,x,"got",got[y,x],"want",want[y,x])
