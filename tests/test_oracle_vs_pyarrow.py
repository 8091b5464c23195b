"""Widens the oracle's pin (VERDICT round 1 "What's weak" #1): the reference's own tests assert one `+`, one `>`, three
aggregate groups and five join rows; everything else the oracle states about arrow-rs 13 came from its published semantics.
Here the oracle's arrow-level functions are cross-checked against `pyarrow.compute` — a DIFFERENT Arrow implementation, so a
sanity check, not an oracle — on seeded random and edge inputs, and every place where the two are EXPECTED to differ is listed
with the arrow-rs 13 behaviour the oracle follows:

  D1  integer and float `divide` by zero: arrow-rs 13 `math_checked_divide_op` raises DivideByZero for any VALID zero divisor,
      floats included (`right.is_zero()`, arrow-13 arithmetic.rs `divide`); pyarrow's unchecked float divide yields ±inf/NaN.
  D2  i64::MIN / -1 (and % -1): Rust's checked division panics ("attempt to divide with overflow"); pyarrow's unchecked divide
      returns 0.  The oracle reports NQE_ERR_ARROW.
  D3  `modulus`: pyarrow.compute has no remainder kernel; the oracle's truncated remainder (sign of the dividend: Rust `%` on
      integers, `fmod` on f64) is checked against numpy's `fmod`, which has the same C semantics.
  D4  a NULL divisor slot is never a divide-by-zero (the check looks at valid slots only): same in both.

Also hand-derived vectors for the reference's own aggregate code that no library can witness: `Avg` counts in a u32 and divides
`sum / cnt as f64` (avg.rs:27-29, 123-125), `Max`/`Min` run on `OrderedFloat<f64>` started at f64::MIN / f64::MAX (max.rs:38-50):
NaN is greater than every number, so one NaN makes `max` NaN and never changes `min`; ±inf never beats the f64::MIN/MAX start
in the direction of the start value."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
pc = pytest.importorskip("pyarrow.compute")

from naive_query_engine_amd import AggregateFunc, Column, DType, ErrorCode, Operator, Status  # noqa: E402
from naive_query_engine_amd.expression import binop, col  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import fields  # noqa: E402

F2 = fields("a", "b")
I64_EDGE = [0, 1, -1, 2, -2, 7, -7, 2**31, -(2**31), 2**62, -(2**62), 2**63 - 1, -(2**63), 1000003, -999983]
U64_EDGE = [0, 1, 2, 7, 2**32, 2**63, 2**64 - 1, 2**64 - 2, 1000003]
F64_EDGE = [0.0, -0.0, 1.0, -1.0, 0.5, -2.5, 1e308, -1e308, 5e-324, float("inf"), float("-inf"), float("nan"), 3.0000000000000004]


def pairs(vals, rng, dtype, extra=200):
    """every edge value against every edge value, plus seeded random rows"""
    a = [x for x in vals for _ in vals]
    b = [y for _ in vals for y in vals]
    if dtype == np.int64:
        a += [int(v) for v in rng.integers(-2**62, 2**62, extra)]
        b += [int(v) for v in rng.integers(-1000, 1000, extra)]
    elif dtype == np.uint64:
        a += [int(v) for v in rng.integers(0, 2**63, extra)]
        b += [int(v) for v in rng.integers(0, 1000, extra)]
    else:
        a += [float(v) for v in rng.normal(0, 1e3, extra)]
        b += [float(v) for v in rng.normal(0, 10, extra)]
    return np.array(a, dtype=dtype), np.array(b, dtype=dtype)


def with_nulls(rng, n, frac=0.2):
    return rng.random(n) >= frac


def to_pa(c: Column):
    m = c.valid_mask()
    return pa.array(c.to_numpy(), mask=~m) if not m.all() else pa.array(c.to_numpy())


def orc_eval(op, ca: Column, cb: Column) -> Column:
    return orc.expr_evaluate([[ca, cb]], binop(col(0), op, col(1)).flatten(F2))


def assert_same(got: Column, exp, what):
    """oracle Column vs a pyarrow array: validity identical; valid slots bit-identical (NaN == NaN, -0.0 == 0.0 allowed for sums of zeros only where noted)"""
    exp = exp.combine_chunks() if isinstance(exp, pa.ChunkedArray) else exp
    gm = got.valid_mask()
    em = ~np.asarray(exp.is_null())
    assert (gm == em).all(), f"{what}: validity differs at {np.nonzero(gm != em)[0][:5]}"
    g = got.to_numpy()[gm]
    e = np.asarray(exp.filter(pa.array(em)).to_numpy(zero_copy_only=False))
    if g.dtype == np.float64:
        same = (g.view(np.uint64) == e.astype(np.float64).view(np.uint64)) | (np.isnan(g) & np.isnan(e))
    else:
        same = g == e.astype(g.dtype)
    assert same.all(), f"{what}: values differ at {np.nonzero(~same)[0][:5]}: {g[~same][:3]} vs {e[~same][:3]}"


DTYPES = [(np.int64, I64_EDGE), (np.uint64, U64_EDGE), (np.float64, F64_EDGE)]


@pytest.mark.parametrize("npdt,edge", DTYPES)
@pytest.mark.parametrize("op,fn", [(Operator.Plus, "add"), (Operator.Minus, "subtract"), (Operator.Multiply, "multiply")])
def test_wrapping_arithmetic_and_null_propagation(npdt, edge, op, fn):
    """add / subtract / multiply: wrapping on integers (arrow-rs 13 math_op = plain `a op b`), IEEE on f64, validity = AND"""
    rng = np.random.default_rng(1)
    a, b = pairs(edge, rng, npdt)
    ca, cb = Column.from_numpy(a, with_nulls(rng, len(a))), Column.from_numpy(b, with_nulls(rng, len(b)))
    assert_same(orc_eval(op, ca, cb), getattr(pc, fn)(to_pa(ca), to_pa(cb)), f"{fn} {npdt.__name__}")


@pytest.mark.parametrize("npdt,edge", DTYPES)
def test_divide_truncates_toward_zero(npdt, edge):
    rng = np.random.default_rng(2)
    a, b = pairs(edge, rng, npdt)
    keep = b != 0  # D1: zero divisors are an error in arrow-rs 13 (checked below)
    if npdt == np.int64:
        keep &= ~((a == -(2**63)) & (b == -1))  # D2
    a, b = a[keep], b[keep]
    ca, cb = Column.from_numpy(a, with_nulls(rng, len(a))), Column.from_numpy(b, with_nulls(rng, len(b)))
    assert_same(orc_eval(Operator.Divide, ca, cb), pc.divide(to_pa(ca), to_pa(cb)), f"divide {npdt.__name__}")


@pytest.mark.parametrize("npdt,edge", DTYPES)
def test_modulus_is_the_truncated_remainder(npdt, edge):
    """D3: no pyarrow kernel; numpy.fmod has the C semantics Rust's `%` has on integers and f64 (sign of the dividend)"""
    rng = np.random.default_rng(3)
    a, b = pairs(edge, rng, npdt)
    keep = b != 0
    if npdt == np.int64:
        keep &= ~((a == -(2**63)) & (b == -1))
    a, b = a[keep], b[keep]
    got = orc_eval(Operator.Modulos, Column.from_numpy(a), Column.from_numpy(b)).to_numpy()
    with np.errstate(invalid="ignore"):
        exp = np.fmod(a, b)
    if npdt == np.float64:
        assert ((got.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(got) & np.isnan(exp))).all()
    else:
        assert (got == exp).all()
        # and the defining identity a == (a / b) * b + a % b with the truncated quotient
        q = orc_eval(Operator.Divide, Column.from_numpy(a), Column.from_numpy(b)).to_numpy()
        assert ((q * b + got) == a).all()


@pytest.mark.parametrize("npdt", [np.int64, np.uint64, np.float64])
@pytest.mark.parametrize("op", [Operator.Divide, Operator.Modulos])
def test_divide_by_zero_is_an_error_on_valid_slots_only(npdt, op):
    """D1 + D4"""
    a = np.array([5, 6, 7], dtype=npdt)
    z = np.array([1, 0, 2], dtype=npdt)
    with pytest.raises(ErrorCode) as e:
        orc_eval(op, Column.from_numpy(a), Column.from_numpy(z))
    assert e.value.status == Status.ArrowError and "ivide by zero" in str(e.value)
    # the zero divisor in a NULL slot (either side null) is not looked at
    out = orc_eval(op, Column.from_numpy(a), Column.from_numpy(z, np.array([True, False, True])))
    assert out.valid_mask().tolist() == [True, False, True]
    out = orc_eval(op, Column.from_numpy(a, np.array([True, False, True])), Column.from_numpy(z))
    assert out.valid_mask().tolist() == [True, False, True]
    if npdt == np.float64:  # pyarrow: unchecked float division by zero is not an error (the documented divergence)
        assert np.isinf(pc.divide(pa.array([1.0]), pa.array([0.0])).to_numpy()[0])
        with pytest.raises(ErrorCode):  # -0.0 is zero too (`is_zero`)
            orc_eval(op, Column.from_numpy(np.array([1.0])), Column.from_numpy(np.array([-0.0])))


def test_int64_min_divided_by_minus_one_is_an_error():
    """D2"""
    a, b = np.array([-(2**63)], dtype=np.int64), np.array([-1], dtype=np.int64)
    for op in (Operator.Divide, Operator.Modulos):
        with pytest.raises(ErrorCode) as e:
            orc_eval(op, Column.from_numpy(a), Column.from_numpy(b))
        assert e.value.status == Status.ArrowError and "overflow" in str(e.value)
    assert pc.divide(pa.array(a), pa.array(b)).to_pylist() == [0]  # pyarrow's unchecked kernel: no error


CMP = [(Operator.Eq, "equal"), (Operator.NotEq, "not_equal"), (Operator.Lt, "less"), (Operator.LtEq, "less_equal"), (Operator.Gt, "greater"),
       (Operator.GtEq, "greater_equal")]


@pytest.mark.parametrize("npdt,edge", DTYPES)
@pytest.mark.parametrize("op,fn", CMP)
def test_six_comparisons(npdt, edge, op, fn):
    """eq_dyn … gt_eq_dyn: exact on integers (signed vs unsigned order!), IEEE on f64 — NaN compares false except `!=`, -0.0 == 0.0"""
    rng = np.random.default_rng(4)
    a, b = pairs(edge, rng, npdt, extra=100)
    ca, cb = Column.from_numpy(a, with_nulls(rng, len(a))), Column.from_numpy(b, with_nulls(rng, len(b)))
    got = orc_eval(op, ca, cb)
    assert got.dtype == DType.BOOLEAN
    assert_same(got, getattr(pc, fn)(to_pa(ca), to_pa(cb)), f"{fn} {npdt.__name__}")


@pytest.mark.parametrize("op,fn", [(Operator.And, "and_kleene"), (Operator.Or, "or_kleene")])
def test_kleene_logic(op, fn):
    vals = [True, False, None]
    a = [x for x in vals for _ in vals] * 3
    b = [y for _ in vals for y in vals] * 3
    ca, cb = Column.from_list(a, DType.BOOLEAN), Column.from_list(b, DType.BOOLEAN)
    got = orc_eval(op, ca, cb)
    exp = getattr(pc, fn)(pa.array(a, pa.bool_()), pa.array(b, pa.bool_()))
    assert got.to_list() == exp.to_pylist()


def test_take_with_nulls_matches_pyarrow():
    """compute::take as the hash join uses it (hash_join.rs:239,245): gather by Int64 indices, validity travels with the value"""
    rng = np.random.default_rng(5)
    n = 500
    left = [Column.from_numpy(np.arange(n, dtype=np.int64)), Column.from_numpy(rng.normal(size=n), with_nulls(rng, n)),
            Column.from_numpy(rng.random(n) < 0.5, with_nulls(rng, n))]
    keys = rng.integers(0, n, 2000).astype(np.int64)
    right = [Column.from_numpy(keys)]
    out = orc.hash_join([left], [right], 0, 0)[0]  # every probe row matches exactly one build row: out = take(left, keys) ++ right
    idx = pa.array(keys)
    for c, name in ((1, "float64"), (2, "boolean")):
        assert_same(out[c], pc.take(to_pa(left[c]), idx), f"take {name}")


def test_concat_with_nulls_matches_pyarrow():
    """concat_batches (hash_join.rs:258-273) as the aggregate applies it to its input: un-grouped count/sum over three batches"""
    rng = np.random.default_rng(6)
    parts = [Column.from_numpy(rng.normal(size=m), with_nulls(rng, m, 0.3)) for m in (5, 130, 64)]
    whole = pa.concat_arrays([to_pa(p) for p in parts])
    out = orc.aggregate([[p] for p in parts], [(AggregateFunc.Count, 0), (AggregateFunc.Sum, 0)])[0]
    assert out[0].to_list() == [len(whole) - whole.null_count]
    exp_sum = 0.0
    for v in whole.to_pylist():  # the reference adds row by row in batch order
        if v is not None:
            exp_sum += v
    assert out[1].to_list() == [exp_sum]


# ---- the reference's own aggregate arithmetic: hand-derived, no library involved
def agg1(values, func, dtype=DType.FLOAT64):
    return orc.aggregate([[Column.from_list(values, dtype)]], [(func, 0)])[0][0].to_list()[0]


def test_avg_is_sum_over_u32_count_as_f64():
    # avg.rs:123-125: sum / cnt as f64; nulls are skipped by both (`.flatten()` / `is_null`)
    assert agg1([1.0, 2.0, None, 4.0], AggregateFunc.Avg) == (1.0 + 2.0 + 4.0) / 3.0
    assert agg1([1, 2, 4], AggregateFunc.Avg, DType.INT64) == 7.0 / 3.0          # `val as f64` first (avg.rs:49)
    big = [2**63, 2**63]                                                           # UInt64 beyond i64: u64 as f64
    assert agg1(big, AggregateFunc.Avg, DType.UINT64) == (float(2**63) + float(2**63)) / 2.0
    assert np.isnan(agg1([None, None], AggregateFunc.Avg))                         # 0.0 / 0 as f64 = NaN


def test_ordered_float_rules_of_max_and_min():
    nan, inf = float("nan"), float("inf")
    # max.rs:30,48: starts at f64::MIN, `val > self.val` on OrderedFloat: NaN > everything, NaN == NaN
    assert np.isnan(agg1([1.0, nan, 3.0], AggregateFunc.Max))
    assert np.isnan(agg1([nan, 1.0], AggregateFunc.Max))
    assert agg1([1.0, nan, 3.0], AggregateFunc.Min) == 1.0                          # NaN is never less than anything
    assert agg1([nan], AggregateFunc.Min) == np.finfo(np.float64).max               # …so the f64::MAX start survives
    assert agg1([-inf], AggregateFunc.Max) == -np.finfo(np.float64).max             # -inf < f64::MIN: the start survives
    assert agg1([inf], AggregateFunc.Min) == np.finfo(np.float64).max
    assert agg1([inf, 1.0], AggregateFunc.Max) == inf
    assert agg1([None, None], AggregateFunc.Max) == -np.finfo(np.float64).max       # no valid row: the start value
    assert agg1([-5, 3, None], AggregateFunc.Max, DType.INT64) == 3.0               # integers compare as f64
    assert agg1([2**63 + 1024, 7], AggregateFunc.Max, DType.UINT64) == float(2**63 + 1024)
