"""GPU check of the fused ffn.0 + LayerNorm + GELU kernel against fp64 at production sizes, twice (determinism)."""
import importlib, os, sys, ctypes, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_ops_emu import _ffn_ln_gelu_case
capi = importlib.import_module('deep-image-matching_amd.capi'); lib = capi.load()
res = {}
for variant in (1,):
  lib.dim_tune_set(11, variant)
  for M in (2048, 32768, 65536, 204800):
      C1, ref = _ffn_ln_gelu_case(lib, M, 512, seed=M, device="cuda")
      C2, _ = _ffn_ln_gelu_case(lib, M, 512, seed=M, device="cuda")
      err = (C1.double() - ref).abs()
      bad = (err.max(1).values > 1e-5).nonzero().reshape(-1)
      res[f'{variant}_{M}'] = {"max_err": float(err.max()), "bad_rows": int(bad.numel()), "bit_equal_rerun": bool(torch.equal(C1, C2)),
              "first_bad_rows": bad[:8].tolist()}
print(json.dumps(res))
