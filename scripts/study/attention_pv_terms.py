"""CPU study (no GPU): what does the P_lo * V_hi cross term of the attention's P V product buy?  attn_x6_kernel issues three fp16 MFMAs
per P V step (lo*hi, hi*lo, hi*hi); the kernel is MFMA-bound, so dropping one would remove 1/6 of its MFMAs.  This
script runs the oracle LightGlue (512 x 512 keypoints, 9 layers, synthetic weights) with the P V product emulated as
(a) fp32, (b) the three-term fp16 split, (c) the split WITHOUT P_lo * V_hi, and reports the max |delta log-assignment| against an
fp64 evaluation.  Budget: 1e-3, of which the fp32 path itself uses ~1-3e-4."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lightglue_ref as L
from tests import golden_cases as gc
weights = importlib.import_module("deep-image-matching_amd.weights")
torch.set_num_threads(8)

def split16(x, scale):
    xs = (x * scale).clamp(-65504, 65504)
    hi = xs.half().float()
    lo = (xs - hi).half().float()
    return hi, lo

MODE = {"v": "fp32"}
def pv(attn, v):
    if MODE["v"] == "fp32" or attn.dtype == torch.float64:
        return attn @ v
    if MODE["v"] == "two_consistent":
        # round 5: the probabilities ROUNDED to fp16 and used as they are in the numerator AND in the normaliser (the kernel's l_run would sum the
        # rounded values): P' V / sum P' — two MFMA terms (P' V_hi, P' V_lo), no P_lo plane, no residual split
        pr = attn / attn.amax(dim=-1, keepdim=True) * 16.0          # the kernel's p = 2^(s - m_run + 4): <= 16 right after a rescale
        ph = pr.half().float()
        vh, vl = split16(v, 16.0)
        return (ph @ vh + ph @ vl) / ph.sum(dim=-1, keepdim=True) / 16.0
    ph, pl = split16(attn, 4096.0)        # probabilities are produced scaled by 2^12 (<= 4096) in the kernel's log2-domain softmax
    vh, vl = split16(v, 16.0)
    acc = ph @ vh + ph @ vl
    if MODE["v"] == "three":
        acc = acc + pl @ vh
    return acc / (4096.0 * 16.0)

def self_block(x, enc, sd, i, heads=4):
    p = f"transformers.{i}.self_attn"; n, d = x.shape
    qkv = L._lin(x, sd, p + ".Wqkv").reshape(n, heads, d // heads, 3).permute(1, 0, 2, 3)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q, k = L._rotary(enc[:, None], q), L._rotary(enc[:, None], k)
    attn = torch.softmax((q @ k.transpose(-1, -2)) * (d // heads) ** -0.5, dim=-1)
    msg = L._lin(pv(attn, v).permute(1, 0, 2).reshape(n, d), sd, p + ".out_proj")
    return x + L._ffn(x, msg, sd, p)

def cross_block(x0, x1, sd, i, heads=4):
    p = f"transformers.{i}.cross_attn"; d = x0.shape[-1]; dh = d // heads
    split = lambda t: t.reshape(t.shape[0], heads, dh).permute(1, 0, 2)
    qk0, qk1 = split(L._lin(x0, sd, p + ".to_qk")), split(L._lin(x1, sd, p + ".to_qk"))
    v0, v1 = split(L._lin(x0, sd, p + ".to_v")), split(L._lin(x1, sd, p + ".to_v"))
    s = (dh ** -0.5) ** 0.5
    sim = (qk0 * s) @ (qk1 * s).transpose(-1, -2)
    m0 = pv(torch.softmax(sim, dim=-1), v1)
    m1 = pv(torch.softmax(sim.transpose(-1, -2).contiguous(), dim=-1), v0)
    m0 = L._lin(m0.permute(1, 0, 2).reshape(-1, d), sd, p + ".to_out")
    m1 = L._lin(m1.permute(1, 0, 2).reshape(-1, d), sd, p + ".to_out")
    return x0 + L._ffn(x0, m0, sd, p), x1 + L._ffn(x1, m1, sd, p)

L.self_block, L.cross_block = self_block, cross_block
conf = {"depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.0, "pruning_min_kpts": -1}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for gain in (2.0,):           # the gain of the parity tests' synthetic weights
    sd = weights.synthetic_lightglue_state_dict(0, 256, gain=gain)
    f0, f1 = gc.lg_inputs(dict(gc.LG_CASES["fixed"], m=N, n=N, seed=21))
    sd64 = {k: v.double() for k, v in sd.items()}
    la64 = L.lightglue_forward(f0["kpts"].double(), f0["desc"].double(), f0["size"].double(), f1["kpts"].double(), f1["desc"].double(),
                               f1["size"].double(), sd64, {**conf, "dtype": torch.float64}, taps=True)["log_assignment"][:N, :N]
    out = {}
    for mode in ("fp32", "three", "two", "two_consistent"):
        MODE["v"] = mode
        la = L.lightglue_forward(f0["kpts"], f0["desc"], f0["size"], f1["kpts"], f1["desc"], f1["size"], sd, conf, taps=True)["log_assignment"][:N, :N]
        out[mode] = float((la.double() - la64).abs().max())
    print(f"gain {gain}: max |delta log-assignment| vs fp64 — fp32 P V {out['fp32']:.2e}, three-term split {out['three']:.2e}, without P_lo*V_hi {out['two']:.2e}, P rounded to fp16 in numerator AND normaliser {out['two_consistent']:.2e}")
