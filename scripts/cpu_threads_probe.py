import importlib, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import superpoint_ref, lightglue_ref
weights = importlib.import_module('deep-image-matching_amd.weights')
sp_sd = weights.synthetic_superpoint_state_dict(1234); lg_sd = weights.synthetic_lightglue_state_dict(0,256)
cfg = {"nms_radius": 3, "keypoint_threshold": 0.0005, "max_keypoints": 2048, "remove_borders": 4}
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}
img = torch.rand(1,1,1024,1024); size=torch.tensor([1024.0,1024.0])
print('os.cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for th in (8,16,32,64,128):
    torch.set_num_threads(th)
    f = superpoint_ref.superpoint_forward(img, sp_sd, cfg)
    t0=time.perf_counter(); f = superpoint_ref.superpoint_forward(img, sp_sd, cfg); t1=time.perf_counter()
    lightglue_ref.lightglue_forward(f["keypoints"], f["descriptors"].t().contiguous(), size, f["keypoints"], f["descriptors"].t().contiguous(), size, lg_sd, conf)
    t2=time.perf_counter()
    print(th, 'SP %.2fs LG %.2fs -> %.3f pairs/s'%(t1-t0, t2-t1, 1/(2*(t1-t0)+(t2-t1))))
