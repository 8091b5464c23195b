"""Shared test helpers: column comparison and seeded synthetic batches."""
import numpy as np

from naive_query_engine_amd import Column, DType


class F:
    """stand-in for a schema field (only .name is used by ColumnExpr name resolution)"""

    def __init__(self, name):
        self.name = name


def fields(*names):
    return [F(n) for n in names]


def assert_column_equal(got: Column, exp: Column, rtol=None, what=""):
    assert got.dtype == exp.dtype, f"{what}: dtype {got.dtype} != {exp.dtype}"
    assert got.length == exp.length, f"{what}: length {got.length} != {exp.length}"
    gm, em = got.valid_mask(), exp.valid_mask()
    assert (gm == em).all(), f"{what}: validity differs at {np.nonzero(gm != em)[0][:8]}"
    if got.dtype == DType.UTF8:
        assert got.to_list() == exp.to_list(), what
        return
    g, e = got.to_numpy()[em], exp.to_numpy()[em]
    if got.dtype == DType.FLOAT64:
        if rtol is None:
            same = (g.view(np.uint64) == e.view(np.uint64)) | (np.isnan(g) & np.isnan(e)) | ((g == 0) & (e == 0))
            assert same.all(), f"{what}: float values differ at {np.nonzero(~same)[0][:8]}: {g[~same][:4]} vs {e[~same][:4]}"
        else:
            assert np.allclose(g, e, rtol=rtol, atol=0, equal_nan=True), f"{what}: float values differ beyond rtol={rtol}"
    else:
        assert (g == e).all(), f"{what}: values differ at {np.nonzero(g != e)[0][:8]}"


def assert_batches_equal(got_cols, exp_cols, rtol=None, what=""):
    assert len(got_cols) == len(exp_cols), f"{what}: {len(got_cols)} columns vs {len(exp_cols)}"
    for i, (g, e) in enumerate(zip(got_cols, exp_cols)):
        assert_column_equal(g, e, rtol=rtol, what=f"{what} col {i}")


def rows_sorted(cols):
    """rows of a batch as a sorted float matrix (multiset compare for aggregates)"""
    m = np.stack([c.to_numpy().astype(np.float64) for c in cols], axis=1) if cols else np.zeros((0, 0))
    if m.size == 0:
        return m
    # NaN sorts behind +inf (a tie between the two would leave such rows in input order: found by a group whose max is NaN next
    # to one whose max is +inf, everything else equal)
    nan = np.isnan(m)
    val = np.where(nan, np.inf, m)
    keys = []
    for c in range(m.shape[1]):
        keys += [val[:, c], nan[:, c].astype(np.float64)]
    return m[np.lexsort(keys[::-1])]


def assert_rows_multiset_equal(got_cols, exp_cols, rtol=1e-9, exact_cols=(), what=""):
    assert [c.dtype for c in got_cols] == [c.dtype for c in exp_cols], what
    g, e = rows_sorted(got_cols), rows_sorted(exp_cols)
    assert g.shape == e.shape, f"{what}: shape {g.shape} vs {e.shape}"
    for c in exact_cols:
        assert (g[:, c] == e[:, c]).all(), f"{what}: exact column {c} differs"
    assert np.allclose(g, e, rtol=rtol, atol=0, equal_nan=True), f"{what}: rows differ beyond rtol={rtol}"


def random_batch(rng, n, null_frac=0.0, key_mod=None, with_bool=False):
    """[id Int64 (0..n-1 shuffled or mod), k Int64 (small domain, negatives), v Float64, u UInt64, (b Boolean)]"""
    def mask():
        return None if null_frac == 0 else rng.random(n) >= null_frac

    ids = rng.permutation(n).astype(np.int64)
    k = rng.integers(-50, 50, n).astype(np.int64) if key_mod is None else (rng.integers(0, key_mod, n)).astype(np.int64)
    v = (rng.random(n) * 200.0 - 100.0).astype(np.float64)
    u = rng.integers(0, 1 << 40, n).astype(np.uint64)
    cols = [Column.from_numpy(ids, mask()), Column.from_numpy(k, mask()), Column.from_numpy(v, mask()), Column.from_numpy(u, mask())]
    if with_bool:
        cols.append(Column.from_numpy(rng.random(n) < 0.5, mask()))
    return cols


def random_utf8(rng, n, null_frac=0.0):
    """strings of mixed lengths (empty, ascii, multi-byte), optional nulls"""
    alphabet = ["", "a", "bob", "alice", "véé", "日本語", "x" * 70, "lynne", "grandmaster"]
    items = [alphabet[int(i)] + (str(int(k)) if k % 3 else "") for i, k in zip(rng.integers(0, len(alphabet), n), rng.integers(0, 1000, n))]
    if null_frac:
        items = [None if rng.random() < null_frac else s for s in items]
    return Column.from_list(items, DType.UTF8)
