import importlib, sys, json, torch
sys.path.insert(0,'.')
sp=importlib.import_module('deep-image-matching_amd.superpoint_hip'); weights=importlib.import_module('deep-image-matching_amd.weights')
cfg={"nms_radius":3,"keypoint_threshold":0.0005,"max_keypoints":2048,"remove_borders":4}
res={}
for B in (1,4,8):
    net=sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234),cfg,max_batch=B,max_hw=(1024,1024),capacity=2048)
    imgs=torch.rand(B,1024,1024,device='cuda')
    for _ in range(2): net.extract_batch(imgs)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out=net.extract_batch(imgs)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    res[f'B{B}_ms_per_image']=ms/B; res[f'B{B}_TF']=177.85*B/ms
    del net
print(json.dumps(res,indent=1))
