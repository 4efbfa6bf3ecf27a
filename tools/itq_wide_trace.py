#!/usr/bin/env python3
"""Measurement helper: a few sq_itq_hash calls at 2 M x 512 -> 256 bits (float32 rows unless DT=f64; NORM=2 for
normalize=2) for rocprofv3 --kernel-trace / tools/prof_pmc.sh runs of itq_wide_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smqtk_indexing_amd import _lib
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
n, d, bits = int(os.environ.get("N", 2_000_000)), int(os.environ.get("D", 512)), int(os.environ.get("BITS", 256))
dt = torch.float64 if os.environ.get("DT") == "f64" else torch.float32
norm = _lib.SQ_NORM_L2 if os.environ.get("NORM") == "2" else _lib.SQ_NORM_NONE
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.empty((n, d), dtype=dt, device=dev)
for s in range(0, n, 1 << 19): x[s:s + (1 << 19)].normal_(generator=g)
rot_np, _ = np.linalg.qr(np.random.default_rng(5).standard_normal((d, d)))
rot = torch.from_numpy(np.ascontiguousarray(rot_np[:, :bits])).to(dev)
mean = x[:100_000].double().mean(dim=0).contiguous()
out = torch.empty((n, (bits + 63) // 64), dtype=torch.int64, device=dev)
code = _lib.SQ_DTYPE_F32 if dt == torch.float32 else _lib.SQ_DTYPE_F64
import time
if os.environ.get("DEBUG"):
    _lib.set_option("dense_debug", int(os.environ["DEBUG"]))
ts = []
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.itq_hash_device(x.data_ptr(), code, n, d, mean.data_ptr(), rot.data_ptr(), bits, norm, out.data_ptr(), st)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"DEBUG={os.environ.get('DEBUG', 0)} DT={os.environ.get('DT', 'f32')} NORM={os.environ.get('NORM', 'none')}: call {min(ts[1:]) * 1e3:.3f} ms")
