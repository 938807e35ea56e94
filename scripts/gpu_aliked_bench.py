import importlib, sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
al=importlib.import_module('deep-image-matching_amd.aliked_hip'); weights=importlib.import_module('deep-image-matching_amd.weights'); capi=importlib.import_module('deep-image-matching_amd.capi')
for a in sys.argv[1:]:   # KEY=VALUE -> dim_tune_set (e.g. 9=0: no BatchNorm folding; 1=0: the fp32 paths)
    if '=' in a and a.split('=')[0].isdigit(): capi.load().dim_tune_set(int(a.split('=')[0]), int(a.split('=')[1]))
cfg={"model_name":"aliked-n16rot","max_num_keypoints":4000,"detection_threshold":0.2,"nms_radius":2}
res={}
for (B,H,W) in ((1,1024,1024),(8,1024,1024),(4,1000,1500)):
    net=al.AlikedHIP(weights.synthetic_aliked_state_dict(7),cfg,max_batch=B,max_hw=(H,W),capacity=4000)
    imgs=torch.rand(B,H,W,3,device='cuda')
    for _ in range(2): out=net.extract_batch(imgs)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out=net.extract_batch(imgs)
    e1.record(); torch.cuda.synchronize()
    res[f'B{B}_{H}x{W}_ms_per_image']=e0.elapsed_time(e1)/5/B; res[f'B{B}_{H}x{W}_n']=out[3].tolist()
    del net
print(json.dumps(res))
