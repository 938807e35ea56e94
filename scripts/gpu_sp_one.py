import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sp=importlib.import_module('deep-image-matching_amd.superpoint_hip'); weights=importlib.import_module('deep-image-matching_amd.weights')
cfg={"nms_radius":3,"keypoint_threshold":0.0005,"max_keypoints":2048,"remove_borders":4}
B=16
net=sp.SuperPointHIP(weights.synthetic_superpoint_state_dict(1234),cfg,max_batch=B,max_hw=(1024,1024),capacity=2048)
imgs=torch.rand(B,1024,1024,device='cuda')
for _ in range(2): net.extract_batch(imgs)
torch.cuda.synchronize()
