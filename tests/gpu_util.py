"""Helpers shared by the GPU parity tests: every search goes through the C ABI
(art_amd.engine -> ctypes -> libartmips.so); the oracle is only the checker."""
import numpy as np
import torch

from art_amd.engine import MipsEngine


def build_engine(rows, ids, storage=torch.float16, devices=(0,)):
    """rows: numpy float16 [N, D] (or float32 for bf16 storage); ids int32."""
    eng = MipsEngine(rows.shape[1], storage, list(devices))
    if storage == torch.float16:
        eng.build(np.ascontiguousarray(rows, dtype=np.float16), ids)
    elif storage == torch.float32:
        eng.build(np.ascontiguousarray(rows, dtype=np.float32), ids)
    else:
        eng.build(torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).to(storage), ids)
    return eng


def search_np(eng, queries, k, q_dtype=torch.float16):
    q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)).to(q_dtype).cuda(eng.devices[0])
    s, i = eng.search(q, k, score_dtype=torch.float32)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy()
