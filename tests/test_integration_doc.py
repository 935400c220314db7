"""INTEGRATION.md section 2 holds the ctypes stubs a maintainer of the reference would paste into its modules.  They are executed
here AS WRITTEN (the fenced python blocks are extracted from the document): the struct mirrors against the library's ud_struct_size
on CPU, the two bindings against independent statements of the op on the GPU -- the document cannot drift from include/unidepth_hip.h."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec2 = text.split("## 2.")[1].split("## 3.")[0]
    return re.findall(r"```python\n(.*?)```", sec2, flags=re.S)


def _namespace():
    """Execute the document's blocks with the in-tree library path substituted for the bare soname."""
    from unidepth_amd import _lib
    ns = {}
    for code in _blocks():
        code = code.replace('ctypes.CDLL("libunidepth_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
        if "_lib = " not in code:
            code = "import ctypes, torch\n" + code
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    return ns


def test_integration_doc_struct_mirrors_match_the_header():
    ns = _namespace()                                   # the blocks assert ud_struct_size == ctypes.sizeof themselves
    from unidepth_amd import _lib
    assert ctypes.sizeof(ns["UdAttention"]) == ctypes.sizeof(_lib.UdAttention) == _lib.lib.ud_struct_size(2)
    assert [f[0] for f in ns["UdAttention"]._fields_] == [f[0] for f in _lib.UdAttention._fields_]
    assert ctypes.sizeof(ns["UdKnn"]) == ctypes.sizeof(_lib.UdKnn) == _lib.lib.ud_struct_size(11)
    assert [f[0] for f in ns["UdKnn"]._fields_] == [f[0] for f in _lib.UdKnn._fields_]


@pytest.mark.gpu
def test_integration_doc_hip_sdpa_snippet_runs():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ns = _namespace()
    g = torch.Generator().manual_seed(0)
    B, N, H = 2, 1376, 16
    q, k, v = (torch.randn(B, N, H * 64, generator=g).half().cuda() for _ in range(3))
    o = ns["hip_sdpa"](q, k, v)
    torch.cuda.synchronize()
    sp = lambda t: t.float().view(B, N, H, 64).permute(0, 2, 1, 3)
    ref = torch.nn.functional.scaled_dot_product_attention(sp(q), sp(k), sp(v)).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    err = ((o.float() - ref).norm() / ref.norm()).item()
    assert err < 2e-3, err


@pytest.mark.gpu
def test_integration_doc_knn_snippet_runs():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import numpy as np
    from oracle import restate_eval
    ns = _namespace()
    g = torch.Generator().manual_seed(1)
    p1, p2 = torch.randn(2, 700, 3, generator=g), torch.randn(2, 900, 3, generator=g)
    l1, l2 = torch.tensor([700, 650]), torch.tensor([900, 512])
    idx, dists = ns["knn_points_idx"](p1.cuda(), p2.cuda(), l1.cuda(), l2.cuda(), 2, 4)
    torch.cuda.synchronize()
    d_ref, i_ref = restate_eval.knn_points(p1.numpy(), p2.numpy(), l1.numpy(), l2.numpy(), K=4)
    assert np.array_equal(idx.cpu().numpy(), i_ref) and np.array_equal(dists.cpu().numpy(), d_ref)
