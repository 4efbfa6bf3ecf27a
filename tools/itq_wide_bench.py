#!/usr/bin/env python3
"""Measurement helper: sq_itq_hash on device-resident rows at the shapes the wide filter covers (2 M x 512 -> 256 b,
float32 and float64 rows; 10 M x 128 float64 -> 64 b) against the float64 MFMA kernel (option itq_exact), ms per call
and the fraction of the HBM peak the algorithmic bytes (N d sizeof + N bits / 8) reach."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def run(n, d, bits, dt, norm):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.empty((n, d), dtype=dt, device=dev)
    for s in range(0, n, 1 << 19):
        x[s:s + (1 << 19)].normal_(generator=g)
    rot_np, _ = np.linalg.qr(np.random.default_rng(5).standard_normal((d, d)))
    rot = torch.from_numpy(np.ascontiguousarray(rot_np[:, :bits])).to(dev)
    mean = x[:100_000].double().mean(dim=0).contiguous()
    words = (bits + 63) // 64
    out = torch.empty((n, words), dtype=torch.int64, device=dev)
    code = _lib.SQ_DTYPE_F32 if dt == torch.float32 else _lib.SQ_DTYPE_F64
    res = {}
    for tag, exact in (("filter", 0), ("float64", 1)):
        _lib.set_option("itq_exact", exact)
        ts = []
        for i in range(4 if exact else 8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.itq_hash_device(x.data_ptr(), code, n, d, mean.data_ptr(), rot.data_ptr(), bits, norm, out.data_ptr(), st)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[tag] = float(np.median(ts[1:]))
        if not exact:
            keep = out.clone()
    _lib.set_option("itq_exact", 0)
    same = bool(torch.equal(keep, out))
    bytes_ = n * d * x.element_size() + n * words * 8
    print(f"n={n} d={d} bits={bits} {str(dt).split('.')[-1]} norm={norm}: filter {res['filter']*1e3:.3f} ms = {bytes_/res['filter']/1e12:.2f} TB/s = "
          f"{bytes_/res['filter']/8e12:.3f} of peak; float64 kernel {res['float64']*1e3:.3f} ms; identical codes: {same}", flush=True)


for (n, d, bits, dt) in ((2_000_000, 512, 256, torch.float32), (2_000_000, 512, 256, torch.float64), (2_000_000, 512, 64, torch.float32),
                         (10_000_000, 128, 64, torch.float64), (4_000_000, 256, 256, torch.float32), (10_000_000, 128, 64, torch.float32)):
    for norm in (_lib.SQ_NORM_NONE, _lib.SQ_NORM_L2):
        run(n, d, bits, dt, norm)
