import numpy as np, torch, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden_cfg, load_golden
from gpu_util import rel_l2
import plbert_amd
from plbert_amd.engine import HipEngine
for name in ("small_h128","small_h128_dualloss"):
    g=load_golden(name); ocfg,pcfg,sd=golden_cfg(g)
    B,S=g["labels"].shape
    eng=HipEngine(pcfg,188,int(g["num_tokens"]),max_batch=B,max_seq=S); eng.load_state_dict(sd)
    idx=[list(map(int,x)) for x in g["index"]]; off,flat=plbert_amd.masked_indices_to_csr(idx)
    tok=g["token_ids"] if "token_ids" in g.files and int(g["num_tokens"]) and "dual" in name else None
    for step in range(1,len(g["losses"])+1):
        eng.loss_fwd_bwd(g["masked"],g["labels"],g["lengths"].astype(np.int32),off,flat,int(off[-1]),token_ids=tok); eng.adamw_step(step,lr=1e-3)
    torch.cuda.synchronize()
    for k in g.files:
        if k.startswith("final/"):
            kk=k[6:]
            d_got=eng.view(kk).cpu()-torch.from_numpy(sd[kk]); d_ref=torch.from_numpy(g[k]-sd[kk])
            if float(d_ref.norm())>0: print(name,kk[-40:],round(rel_l2(d_got,d_ref),4))
