"""INTEGRATION.md §2 shows the ctypes stub a maintainer of the reference would add to bind the C ABI
directly.  This test runs that code block VERBATIM (extracted from the markdown) and checks its answer,
so that the document cannot drift from the library."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _section2_code():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1, "INTEGRATION.md §2 is expected to hold exactly one python block"
    return blocks[0]


def test_documented_ctypes_stub_runs_and_is_exact(monkeypatch):
    monkeypatch.chdir(ROOT)  # the snippet loads art_amd/lib/libartmips.so by a relative path
    rng = np.random.Generator(np.random.PCG64(2))
    n_rows = 30000
    rows_f16 = (rng.standard_normal((n_rows, 768)) * 0.05).astype(np.float16)
    ids_i32 = (rng.permutation(n_rows) + 1).astype(np.int32)
    queries = rng.standard_normal((64, 768)).astype(np.float16)
    ns = {"n_rows": n_rows, "rows_f16": rows_f16, "ids_i32": ids_i32}
    exec(compile(_section2_code(), "INTEGRATION.md#2", "exec"), ns)
    dist_, idx = ns["search_mips_index"](torch.from_numpy(queries).cuda(), 33, reconstruct=False)
    torch.cuda.synchronize()
    assert dist_.dtype == torch.float16 and idx.dtype == torch.int32 and tuple(idx.shape) == (64, 33)
    es, ei, _ = orc.exact_topk(queries.astype(np.float64), rows_f16.astype(np.float64), ids_i32, 33)
    got = idx.cpu().numpy()
    # fp32 accumulation order may swap rows whose scores differ below fp32 resolution; nothing else
    assert (got != ei).sum() <= 2
    np.testing.assert_allclose(dist_.float().cpu().numpy(), es, rtol=2e-3)
