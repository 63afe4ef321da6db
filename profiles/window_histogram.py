"""Distribution of search-window shapes on the benchmark scene (CPU, numpy): which lane layout / segment class of k_invert
the pixels fall into.  Uses the executable specification of the window logic (tests/prune_model.py).

    python profiles/window_histogram.py
"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, bench
from prune_model import search_window
lut, co = bench.build_product_lut()
db = np.asarray(co["db"]); incax=np.asarray(co["inc"]); w=np.asarray(co["wspd"]); phi=np.asarray(co["phi"])
n_w,n_phi=len(w),len(phi)
cp,sp=np.cos(np.radians(phi)),np.sin(np.radians(phi))
w0, inv_wstep = w[0], (n_w - 1) / (w[-1] - w[0]); phi0, inv_dphi = phi[0], (n_phi - 1) / (phi[-1] - phi[0])
rng=np.random.default_rng(0)
cols=[];rows=[]
for l0 in (1000, 6000, 11000, 16000):
    inc,s,anc = bench.make_scene(8, 20000, 20000, l0, 20260322, torch.device("cpu"))
    inc,s,anc=[t.numpy() for t in (inc,s,anc)]
    for k in rng.integers(0, inc.size, 700):
        i,ss,a_=inc.flat[k],s.flat[k],anc.flat[k]
        if not (np.isfinite(i) and np.isfinite(ss) and np.isfinite(a_.real)): continue
        sdb=float(10*np.log10(np.float32(ss)+np.float32(1e-15)))
        ii=int(np.argmin(np.abs(incax-i))); a,b=float(a_.real),abs(float(a_.imag))
        mag=np.hypot(a,b); th=np.degrees(np.arctan2(b,a)); ipr=int(np.clip(np.rint((th-phi0)*inv_dphi),0,n_phi-1))
        Jc=((w*cp[ipr]-a)/2)**2+((w*sp[ipr]-b)/2)**2+((db[ii][:,ipr]-sdb)/0.1)**2
        jub=Jc.min()
        wl,wh,pl,ph=search_window(mag,th,jub,w0,inv_wstep,n_w,phi0,phi[-1],inv_dphi,n_phi)
        cols.append(ph-pl+1); rows.append(wh-wl+1)
cols=np.array(cols); rows=np.array(rows)
print("n",len(cols),"mean cols %.1f rows %.1f area %.0f"%(cols.mean(),rows.mean(),(cols*rows).mean()))
for lo,hi in ((1,4),(5,8),(9,12),(13,16),(17,21),(22,32),(33,64),(65,999)):
    m=(cols>=lo)&(cols<=hi); print(f"cols {lo}-{hi}: {m.mean()*100:.1f}% of pixels, mean rows {rows[m].mean() if m.any() else 0:.1f}, share of candidates {(cols*rows)[m].sum()/(cols*rows).sum()*100:.1f}%")
