import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
import oracle_lib as ol
from openekfmonoslam_amd import engine
from openekfmonoslam_amd.synth import SyntheticSequence
N=int(sys.argv[1]); F=int(sys.argv[2]); PREC=int(sys.argv[3]) if len(sys.argv)>3 else 1  # 1 fast fp32, 2 exact
seq=SyntheticSequence(N,F)
e=engine.EkfEngine(seq.cam,seq.par,N,max_keypoints=2*N+64,precision=PREC)
o=ol.Oracle(seq.cam,seq.par,N+8)
e.set_state(seq.x13,seq.feature_pos,seq.feature_type,seq.feature_desc,seq.P0)
o.set_state(seq.x13,seq.feature_pos,seq.feature_type,seq.feature_desc,seq.P0)
for t in range(F):
    e.step(*seq.frames[t]); o.step(*seq.frames[t], ol.ALGORITHMIC)
    if t % 25 == 24 or t==F-1:
        x,fp,P=e.get_state(); xo,fpo,Po=o.x13(),o.feature_pos(),o.P()
        blk={k: float(np.abs(x[s]-xo[s]).max()/max(np.abs(xo[s]).max(),1e-9)) for k,s in (("r",slice(0,3)),("v",slice(7,10)),("w",slice(10,13)))}
        print(t, "Pmax %.2e fro %.2e"%(np.abs(P-Po).max()/np.abs(Po).max(), np.linalg.norm(P-Po)/np.linalg.norm(Po)), {k:"%.1e"%v for k,v in blk.items()}, "feat %.2e"%(np.abs(fp-fpo)/np.maximum(np.abs(fpo),1e-4)).max())
