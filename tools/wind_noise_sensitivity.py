"""How much one AcousticDynamics call of the REFERENCE ALGORITHM (oracle/dyn_core.py, bit-identical to the reference run on
these inputs) moves when 1e-13 m/s of noise is added to the initial winds of the baroclinic C12 case: the bound an
end-to-end comparison from independently generated initial winds can have (tests/helpers.GENERATED_TOL).  Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
from helpers import DSW_CFG, acoustic_errors, acoustic_fixture, golden, oracle_grid
from oracle import dyn_core
n,nz=12,79
fixes=[acoustic_fixture(t) for t in range(6)]
grids=[oracle_grid({k[5:]:v for k,v in fx.items() if k.startswith("grid_")},n,nz) for fx in fixes]
col={k:v for k,v in golden("column_namelist_c12.npz").items()}
cfg=dict(DSW_CFG,p_fac=0.05,rf_cutoff=3000.0,tau=10.0,delt_max=0.002,hord_tm=6)
rng=np.random.default_rng(0)
states=[{k[3:]:v.copy() for k,v in fx.items() if k.startswith("in_") and k!="in_cappa"} for fx in fixes]
for st in states:
    for k in ("u","v"): st[k]=st[k]+1e-13*rng.standard_normal(st[k].shape)
cappas=[fx["in_cappa"].copy() for fx in fixes]
tmp=dyn_core.acoustic_dynamics(grids,col,cfg,states,cappas,float(fixes[0]["timestep"]),int(fixes[0]["n_split"]),n,nz)
worst={}
for t in range(6):
    out=dict(states[t]); out["heat_source"]=tmp[t].heat_source
    for k,e in acoustic_errors(fixes[t],out).items(): worst[k]=max(worst.get(k,0),e)
print({k:f"{v:.1e}" for k,v in worst.items()})
w2={}
for t in range(6):
    out=dict(states[t])
    ks=fixes[t]["k_sel"]
    for k in ("u","v","delp","pt","w","ua","va"):
        di = 1 if k in ("v",) else 0
        dj = 1 if k in ("u",) else 0
        kk=[x for x in ks if x<79]; idx=[list(ks).index(x) for x in kk]
        got=out[k][3:15+di,3:15+dj][:,:,kk]; ref=fixes[t]["out_"+k][:12+di,:12+dj][:,:,idx]
        w2[k]=max(w2.get(k,0), float(np.abs(ref-got).max()/np.abs(ref).max()))
print("scaled", {k:f"{v:.1e}" for k,v in w2.items()})
