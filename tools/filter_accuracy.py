import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from mosaicmagnifique_amd.engine import Engine
from mosaicmagnifique_amd import synth
e=Engine(0)
rng=np.random.default_rng(7)
def report(name,a,b):
    a=np.ascontiguousarray(a,dtype=np.float32).reshape(-1,3); b=np.ascontiguousarray(b,dtype=np.float32).reshape(-1,3)
    ex=e.colour_diff_batch(2,a,b); fa=e.filter_diff_batch(2,a,b).astype(np.float64)
    edge=fa<0
    err=np.abs(fa-ex)[~edge]
    rel=err/np.maximum(ex[~edge],1e-9)
    i=np.argmax(err)
    print('%-28s n=%9d edge=%8d (%.2e) max_abs=%.3e mean_abs=%.3e max_rel(ex>1)=%.3e  worst: a=%s b=%s ex=%.6f fa=%.6f'%(name,len(a),edge.sum(),edge.mean(),err.max(),err.mean(),rel[ex[~edge]>1].max() if (ex[~edge]>1).any() else 0,a[~edge][i],b[~edge][i],ex[~edge][i],fa[~edge][i]))
    return err
lo,hi=np.array([0,-128,-128]),np.array([100,127,127])
N=1<<24
report('uniform Lab', rng.uniform(lo,hi,(N,3)), rng.uniform(lo,hi,(N,3)))
a=rng.uniform(lo,hi,(N,3)); report('near-identical', a, a+rng.normal(0,0.5,(N,3)))
a=rng.uniform(lo,hi,(N,3)); report('near-identical tiny', a, a+rng.normal(0,0.01,(N,3)))
a=rng.uniform(lo,hi,(N,3)); a[:,1:]*=1e-3; report('near-grey first', a, rng.uniform(lo,hi,(N,3)))
a=rng.uniform(lo,hi,(N,3)); a[:,1:]*=1e-3; b=rng.uniform(lo,hi,(N,3)); b[:,1:]*=1e-3; report('both near-grey', a, b)
a=rng.uniform(lo,hi,(N,3)); a[:,1:]=0; report('exact grey first', a, rng.uniform(lo,hi,(N,3)))
a=rng.uniform(lo,hi,(N,3)); b=rng.uniform(lo,hi,(N,3)); b[:,1:]=0; report('exact grey second', a, b)
# near-opposite hues
th=rng.uniform(0,2*np.pi,N); r1=rng.uniform(1,120,N); r2=rng.uniform(1,120,N); d=rng.normal(0,1e-3,N)
a=np.stack([rng.uniform(0,100,N), r1*np.cos(th), r1*np.sin(th)],1); b=np.stack([rng.uniform(0,100,N), r2*np.cos(th+np.pi+d), r2*np.sin(th+np.pi+d)],1)
report('near-opposite hue 1e-3', a, b)
d=rng.normal(0,1e-4,N); b=np.stack([rng.uniform(0,100,N), r2*np.cos(th+np.pi+d), r2*np.sin(th+np.pi+d)],1)
report('near-opposite hue 1e-4', a, b)
# blue region high chroma (R_T active)
th=np.deg2rad(rng.uniform(230,320,N)); th2=th+rng.normal(0,0.2,N)
a=np.stack([rng.uniform(0,100,N), r1*np.cos(th), r1*np.sin(th)],1); b=np.stack([rng.uniform(0,100,N), r2*np.cos(th2), r2*np.sin(th2)],1)
report('blue region', a, b)
# same hue, chroma differs (dH ~ 0)
b=np.stack([rng.uniform(0,100,N), r2*np.cos(th), r2*np.sin(th)],1); report('same hue', a, b)
# image-like: synthetic library vs target pixels
lib=synth.preprocess(synth.library_u8(2000,64),2).reshape(-1,3); img=synth.preprocess(synth.target_u8(2048,2048),2).reshape(-1,3)
n=min(len(lib),len(img)); idx=rng.permutation(n)[:N]
err=report('synthetic images', img[idx], lib[rng.permutation(n)[:N]])
print('sum-level: mean signed err', )
