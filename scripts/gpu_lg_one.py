import importlib, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lg=importlib.import_module('deep-image-matching_amd.lightglue_hip'); weights=importlib.import_module('deep-image-matching_amd.weights')
conf={"depth_confidence":-1,"width_confidence":-1,"filter_threshold":0.1}
B=8
net=lg.LightGlueHIP(weights.synthetic_lightglue_state_dict(0,256),conf,max_pairs=B,max_kpts=2048)
kt=torch.rand(2*B,2048,2,device='cuda')*1024; dt=torch.nn.functional.normalize(torch.randn(2*B,2048,256,device='cuda'),dim=-1)
nt=torch.full((2*B,),2048,dtype=torch.int32,device='cuda'); st=torch.full((2*B,2),1024.0,device='cuda')
out=None
for _ in range(2): out=net.match_batch(kt,dt,nt,st,out=out)
torch.cuda.synchronize()
