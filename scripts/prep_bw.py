"""K1/K2 (k_prep_rows) against the HBM roofline: rows already on the device appended to an index, HIP events around the
launches (hx_profile slot 4).  python scripts/prep_bw.py [rows=1048576]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_application_amd import engine as eng
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
X = torch.rand((n, 768), device="cuda") * 2 - 1
ix = eng.HxIndex(768, (64, 128, 256))
ix.reserve(3 * n)
ix.add_device(X)
ix.profile(True); ix.profile_read()
ix.add_device(X); ix.add_device(X)
p = ix.profile_read()["prep_rows"]
print(json.dumps(dict(lib=os.environ.get("HX_LIB_PATH"), launches=p["launches"], ms=p["ms"], gb=p["bytes"] / 1e9, gbs=p["bytes"] / p["ms"] / 1e6,
                      frac=p["bytes"] / p["ms"] / 1e6 / 8000)))
