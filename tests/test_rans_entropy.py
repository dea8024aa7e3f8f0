"""CPU tests: the native range-ANS coder (libevc_rans.so, through its C ABI) against the pure-Python oracle
restatement -- byte-identical streams, cross decoding, bypass (out-of-table) symbols, corrupt input -- and
the entropy-table builders.  Integer work: everything must be exact."""
import numpy as np
import pytest

import evc_amd  # noqa: F401
from evc_amd import entropy, lib
from oracle import entropy as OE
from oracle import rans as OR


def toy_tables(rng, n_cdfs=5, max_len=12):
    cdfs = np.zeros((n_cdfs, max_len + 2), dtype=np.int32)
    sizes = np.zeros(n_cdfs, dtype=np.int32)
    offsets = np.zeros(n_cdfs, dtype=np.int32)
    for i in range(n_cdfs):
        L = int(rng.integers(3, max_len + 1))
        pmf = rng.random(L).astype(np.float32) + 0.01
        pmf /= pmf.sum()
        c = lib.pmf_to_quantized_cdf(pmf)
        cdfs[i, :len(c)] = c
        sizes[i] = len(c)
        offsets[i] = -int(rng.integers(0, L))
    return cdfs, sizes, offsets


@pytest.mark.parametrize("n", [0, 1, 7, 1000])
def test_native_coder_matches_oracle_bytes_and_roundtrips(n):
    rng = np.random.default_rng(n)
    cdfs, sizes, offsets = toy_tables(rng)
    idx = rng.integers(0, len(sizes), n).astype(np.int32)
    sym = np.array([rng.integers(offsets[i] - 3, offsets[i] + sizes[i] + 2) for i in idx], dtype=np.int32)
    s_native = lib.rans_encode(sym, idx, cdfs, sizes, offsets)
    s_oracle = OR.encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, sizes, offsets)
    assert s_native == s_oracle
    assert len(s_native) >= 8 and len(s_native) % 4 == 0
    np.testing.assert_array_equal(lib.rans_decode(s_native, idx, cdfs, sizes, offsets), sym)
    assert OR.decode_with_indexes(s_native, idx.tolist(), cdfs, sizes, offsets) == sym.tolist()


def test_bypass_extremes_and_large_values():
    rng = np.random.default_rng(3)
    cdfs, sizes, offsets = toy_tables(rng, n_cdfs=2)
    idx = np.zeros(8, dtype=np.int32)
    sym = np.array([-100000, 100000, -1 + offsets[0], offsets[0] + sizes[0] - 2, 2 ** 20, -2 ** 20, 0, 5], dtype=np.int32)
    s = lib.rans_encode(sym, idx, cdfs, sizes, offsets)
    assert s == OR.encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, sizes, offsets)
    np.testing.assert_array_equal(lib.rans_decode(s, idx, cdfs, sizes, offsets), sym)


def test_decoder_rejects_truncated_and_bad_arguments():
    rng = np.random.default_rng(4)
    cdfs, sizes, offsets = toy_tables(rng)
    idx = rng.integers(0, len(sizes), 500).astype(np.int32)
    sym = np.array([offsets[i] + 1 for i in idx], dtype=np.int32)
    s = lib.rans_encode(sym, idx, cdfs, sizes, offsets)
    with pytest.raises(lib.EvcKernelError):
        lib.rans_decode(s[:12], idx, cdfs, sizes, offsets)          # stream ends early
    with pytest.raises(lib.EvcKernelError):
        lib.rans_decode(s, idx + 100, cdfs, sizes, offsets)         # index outside the table set
    with pytest.raises(lib.EvcKernelError):
        lib.rans_decode(s[:4], idx, cdfs, sizes, offsets)           # shorter than the 8-byte state


def test_pmf_to_quantized_cdf_matches_oracle_and_is_monotone():
    rng = np.random.default_rng(5)
    for n in (2, 5, 33, 200):
        pmf = rng.random(n).astype(np.float32) ** 4 + 1e-7
        pmf /= pmf.sum()
        c = lib.pmf_to_quantized_cdf(pmf)
        np.testing.assert_array_equal(c, OE.pmf_to_quantized_cdf(pmf))
        assert c[0] == 0 and c[-1] == 1 << 16 and np.all(np.diff(c) > 0)


def test_gaussian_conditional_tables_shape_and_oracle_agreement():
    table = entropy.get_scale_table()
    np.testing.assert_allclose(table.numpy(), OE.get_scale_table(), rtol=1e-6)
    t = entropy.gaussian_conditional_tables(table)
    qcdf, length, offset = OE.gaussian_tables(table.numpy())
    assert t.cdf.shape == qcdf.shape and t.cdf.shape[0] == 64
    np.testing.assert_array_equal(t.length, length)
    np.testing.assert_array_equal(t.offset, offset)
    # float32 erfc (product) vs float64 erf (oracle): which near-zero tail bins get a stolen count differs,
    # and a steal shifts every boundary between the two bins, so compare as distributions (16-bit counts)
    assert np.abs(t.cdf.astype(np.int64) - qcdf.astype(np.int64)).max() <= 0.005 * (1 << 16)
    for i in range(64):
        row = t.cdf[i, :t.length[i]]
        assert row[0] == 0 and row[-1] == 1 << 16 and np.all(np.diff(row) > 0)


def test_entropy_bottleneck_codec_roundtrip():
    rng = np.random.default_rng(6)
    C = 12
    tables = entropy.logistic_bottleneck_tables(0.5 + 2 * rng.random(C))
    med = rng.standard_normal(C).astype(np.float32)
    codec = entropy.EntropyBottleneckCodec(tables, med)
    z = (4 * rng.standard_normal((3, C, 2, 3))).astype(np.float32)
    strings = codec.compress(z)
    z_hat = codec.decompress(strings, (2, 3))
    np.testing.assert_array_equal(z_hat, np.rint(z - med[None, :, None, None]) + med[None, :, None, None])


def test_library_exports_every_declared_symbol():
    """Every function declared in include/*.h resolves in the built libraries (no compute call)."""
    import os
    import re
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for header, table, loader in (("evc_hip.h", lib.HIP_SYMBOLS, lambda: lib.hip_lib(require_device=False)),
                                  ("evc_rans.h", lib.RANS_SYMBOLS, lib.rans_lib)):
        text = open(os.path.join(repo, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared = set(re.findall(r"\b(evc_[a-z0-9_]+)\s*\(", text))
        assert declared == set(table), (declared ^ set(table))
        so = loader()
        for name in declared:
            assert getattr(so, name) is not None
