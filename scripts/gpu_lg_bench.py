import importlib, sys, json, torch
sys.path.insert(0,'.')
lg=importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights=importlib.import_module('deep-image-matching_amd.weights')
conf={"depth_confidence":-1,"width_confidence":-1,"filter_threshold":0.1}
res={}
sd=weights.synthetic_lightglue_state_dict(0,256)
for B in (1,4,8):
    net=lg.LightGlueHIP(sd,conf,max_pairs=B,max_kpts=2048)
    kt=torch.rand(2*B,2048,2,device='cuda')*1024; dt=torch.nn.functional.normalize(torch.randn(2*B,2048,256,device='cuda'),dim=-1)
    nt=torch.full((2*B,),2048,dtype=torch.int32,device='cuda'); st=torch.full((2*B,2),1024.0,device='cuda')
    out=None
    for _ in range(2): out=net.match_batch(kt,dt,nt,st,out=out)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out=net.match_batch(kt,dt,nt,st,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    res[f'B{B}_ms_per_pair']=ms/B; res[f'B{B}_TF']=229.8*B/ms
    del net
print(json.dumps(res,indent=1))
