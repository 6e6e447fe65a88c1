"""The plain-C FSQ restatement (oracle/fsq_oracle.c) against the reference's known-answer vectors and the torch oracle."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import l3ac_oracle as O
from tests.helpers import GOLDEN, index_mismatch_report

ORACLE_DIR = Path(__file__).resolve().parent.parent / "oracle"


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)
    return C.CDLL(str(ORACLE_DIR / "libfsq_oracle.so"))


def _quantize(lib, z, levels):
    z = np.ascontiguousarray(z, dtype=np.float32)
    n, d = z.shape
    lv = (C.c_int32 * d)(*levels)
    q = np.empty_like(z)
    idx = np.empty(n, dtype=np.int32)
    li = np.empty_like(z)
    lib.fsq_oracle_quantize(z.ctypes.data_as(C.c_void_p), C.c_int64(n), d, lv, q.ctypes.data_as(C.c_void_p),
                            idx.ctypes.data_as(C.c_void_p), li.ctypes.data_as(C.c_void_p))
    return q, idx, li


def test_c_fsq_matches_reference_vectors(lib):
    kat = np.load(GOLDEN / "fsq_kat.npz")
    for tag in ("l7", "l9977", "even", "tiny"):
        levels = kat[f"{tag}_levels"].tolist()
        z = kat[f"{tag}_z"]
        q, idx, li = _quantize(lib, z, levels)
        n_bad, ok = index_mismatch_report(idx, kat[f"{tag}_indices"], z, levels, tau=1e-6)  # libm vs Sleef tanh: <= 1 ulp
        assert ok and n_bad <= 1
        if n_bad == 0:
            np.testing.assert_array_equal(li, kat[f"{tag}_level_indices"])
            np.testing.assert_array_equal(q, kat[f"{tag}_q"])
        sel = np.ascontiguousarray(kat[f"{tag}_dec_idx"], dtype=np.int32)
        codes = np.empty((sel.size, len(levels)), dtype=np.float32)
        lib.fsq_oracle_indices_to_codes(sel.ctypes.data_as(C.c_void_p), C.c_int64(sel.size), len(levels),
                                        (C.c_int32 * len(levels))(*levels), codes.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(codes, kat[f"{tag}_dec_codes"])
    # half-to-even at exact ties
    _, idx, li = _quantize(lib, np.zeros((1, 4), np.float32), [2, 4, 6, 8])
    assert li.tolist() == [[0.0, 2.0, 2.0, 4.0]] and idx.tolist() == [212]


def test_c_fsq_rounding_boundaries_bit_exact(lib):
    """tests/golden/fsq_boundary_kat.npz: exact k + 0.5 products and their +-1 / +-2 ulp neighbours, answers from the
    reference's SuperFSQ.quantize_act_value.  No transcendental is involved, so the C restatement must agree bit for bit."""
    kat = np.load(GOLDEN / "fsq_boundary_kat.npz")
    for tag in ("l7", "l9977", "even", "tiny"):
        levels = kat[f"{tag}_levels"].tolist()
        act = np.ascontiguousarray(kat[f"{tag}_act"], dtype=np.float32)
        n, d = act.shape
        q, li = np.empty_like(act), np.empty_like(act)
        idx = np.empty(n, dtype=np.int32)
        lib.fsq_oracle_quantize_act(act.ctypes.data_as(C.c_void_p), C.c_int64(n), d, (C.c_int32 * d)(*levels),
                                    q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), li.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(li, kat[f"{tag}_level_indices"])
        np.testing.assert_array_equal(idx, kat[f"{tag}_indices"])
        np.testing.assert_array_equal(q, kat[f"{tag}_q"])


def test_c_argmin_agrees_with_closed_form(lib):
    levels = [5, 3, 4]
    g = torch.Generator().manual_seed(3)
    z = torch.randn(500, 3, generator=g)
    _, idx_ref, _ = O.fsq_quantize(z, levels)
    cb = np.ascontiguousarray(O.codebook(levels).numpy())
    qv = np.ascontiguousarray(torch.tanh(z).numpy())
    out = np.empty(500, dtype=np.int32)
    lib.fsq_oracle_argmin(qv.ctypes.data_as(C.c_void_p), C.c_int64(500), cb.ctypes.data_as(C.c_void_p), cb.shape[0], 3,
                          out.ctypes.data_as(C.c_void_p))
    lv = torch.tensor(levels, dtype=torch.float32)
    scaled = (torch.tanh(z) + 1) / 2 * (lv - 1)
    clear = (((scaled - scaled.floor()) - 0.5).abs().min(dim=1).values > 1e-4).numpy()
    np.testing.assert_array_equal(out[clear], idx_ref.numpy()[clear])
