"""CPU study (no GPU): is Winograd F(2x2, 3x3) numerically admissible for SuperPoint's 3x3 convolutions at fp32 product precision?
Runs the encoder + detector head on a 256^2 image with direct fp32 convolutions and with fp32 Winograd convolutions (transforms and
products in fp32; the fp16x3 products of the HIP path are fp32-class) against an fp64 evaluation, and prints the range growth of the
transformed activations (the fp16x3 range guard limit shrinks by that factor).  Result recorded in DESIGN.md section 9."""
import sys, importlib, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
weights = importlib.import_module("deep-image-matching_amd.weights")
torch.manual_seed(0); torch.set_num_threads(8)
sd = weights.synthetic_superpoint_state_dict(1234)
G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]])
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1.]])
At = torch.tensor([[1,1,1,0],[0,1,-1,-1.]])
def wino_conv(x, w, b, dtype):
    # x [1,C,H,W] (H,W even), w [O,C,3,3]; F(2x2,3x3); everything in `dtype`
    x = x.to(dtype); w = w.to(dtype); Gd, Btd, Atd = G.to(dtype), Bt.to(dtype), At.to(dtype)
    U = torch.einsum('ij,ocjk,lk->ocil', Gd, w, Gd)            # [O,C,4,4]
    xp = F.pad(x, (1,1,1,1))
    H, W = x.shape[2], x.shape[3]
    tiles = xp.unfold(2,4,2).unfold(3,4,2)                     # [1,C,H/2,W/2,4,4]
    V = torch.einsum('ij,nchwjk,lk->nchwil', Btd, tiles, Btd)   # [1,C,th,tw,4,4]
    M = torch.einsum('ocil,nchwil->nohwil', U, V)
    Y = torch.einsum('ij,nohwjk,lk->nohwil', Atd, M, Atd)       # [1,O,th,tw,2,2]
    y = Y.permute(0,1,2,4,3,5).reshape(1, w.shape[0], H, W)
    return y + b.to(dtype).view(1,-1,1,1)
def encoder(img, conv):
    x = img
    names = ['conv1a','conv1b','conv2a','conv2b','conv3a','conv3b','conv4a','conv4b']
    for i, n in enumerate(names):
        x = torch.relu(conv(x, sd[n+'.weight'], sd[n+'.bias']))
        if n in ('conv1b','conv2b','conv3b'): x = F.max_pool2d(x, 2, 2)
    cPa = torch.relu(conv(x, sd['convPa.weight'], sd['convPa.bias']))
    logits = F.conv2d(cPa.to(torch.float64), sd['convPb.weight'].double(), sd['convPb.bias'].double())
    return logits
def wino1d_conv(x, w, b, dtype):
    # F(2, 3) along x only: 4 transform positions x 3 kernel rows (12 instead of 18 products per 2 outputs)
    x = x.to(dtype); w = w.to(dtype); Gd, Btd, Atd = G.to(dtype), Bt.to(dtype), At.to(dtype)
    U = torch.einsum('ij,ocrj->ocri', Gd, w)                    # [O,C,3 rows,4]
    xp = F.pad(x, (1,1,1,1))
    H, W = x.shape[2], x.shape[3]
    tiles = xp.unfold(3,4,2)                                    # [1,C,H+2,W/2,4]
    V = torch.einsum('ij,nchwj->nchwi', Btd, tiles)             # [1,C,H+2,tw,4]
    M = sum(torch.einsum('oci,nchwi->nohwi', U[:, :, r], V[:, :, r:r + H]) for r in range(3))   # [1,O,H,tw,4]
    Y = torch.einsum('ij,nohwj->nohwi', Atd, M)                 # [1,O,H,tw,2]
    return Y.reshape(1, w.shape[0], H, W) + b.to(dtype).view(1,-1,1,1)
direct = lambda dt: (lambda x,w,b: F.conv2d(x.to(dt), w.to(dt), b.to(dt), padding=1))
img = torch.rand(1,1,256,256)
ref64 = encoder(img, direct(torch.float64))
d32 = encoder(img, direct(torch.float32))
w32 = encoder(img, lambda x,w,b: wino_conv(x,w,b,torch.float32) if w.shape[1] > 1 else F.conv2d(x, w, b, padding=1))
w64 = encoder(img, lambda x,w,b: wino_conv(x,w,b,torch.float64) if w.shape[1] > 1 else F.conv2d(x.double(), w.double(), b.double(), padding=1))
sc = lambda l: torch.softmax(l, 1)[:, :-1]
print("logit scale", ref64.abs().max().item())
print("winograd fp64 vs direct fp64 (algebra check):", (w64 - ref64).abs().max().item())
print("direct fp32  vs fp64: logits", (d32 - ref64).abs().max().item(), " scores", (sc(d32) - sc(ref64)).abs().max().item())
print("winograd fp32 vs fp64: logits", (w32 - ref64).abs().max().item(), " scores", (sc(w32) - sc(ref64)).abs().max().item())
w1 = encoder(img, lambda x,w,b: wino1d_conv(x,w,b,torch.float32) if w.shape[1] > 1 else F.conv2d(x, w, b, padding=1))
w1d = encoder(img, lambda x,w,b: wino1d_conv(x,w,b,torch.float64) if w.shape[1] > 1 else F.conv2d(x.double(), w.double(), b.double(), padding=1))
print("1-D winograd fp64 vs direct fp64 (algebra check):", (w1d - ref64).abs().max().item())
print("1-D winograd fp32 vs fp64: logits", (w1 - ref64).abs().max().item(), " scores", (sc(w1) - sc(ref64)).abs().max().item())
# range growth of the transformed activations
x = torch.relu(F.conv2d(img, sd['conv1a.weight'], sd['conv1a.bias'], padding=1))
tiles = F.pad(x,(1,1,1,1)).unfold(2,4,2).unfold(3,4,2)
V = torch.einsum('ij,nchwjk,lk->nchwil', Bt, tiles, Bt)
print("max |activation|", x.abs().max().item(), " max |B^T d B|", V.abs().max().item())
