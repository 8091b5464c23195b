"""Differential fuzzing of the hot path against the oracle: random table sizes around the 64-row / 4096-row tile edges,
null fractions, predicates (integer range, float range, Boolean column, expression trees), group keys (column, modulo by
literal, trees), value columns of every numeric type, joins with random key multiplicities.  Seeds are fixed: a failure
names its case."""
import numpy as np
import pytest

from naive_query_engine_amd import AggregateFunc, Column, DType, ErrorCode, Operator
from naive_query_engine_amd.expression import binop, col, lit_bool, lit_f64, lit_i64, lit_u64
from oracle import oracle as orc
from tests.helpers import assert_batches_equal, assert_rows_multiset_equal, fields, random_batch
from tests.test_gpu_parity import _random_tree

import os

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("NQE_FUZZ_EXTRA_SEEDS", "0"))  # a longer hunt: NQE_FUZZ_EXTRA_SEEDS=200 pytest tests/test_gpu_fuzz.py
BASE = int(os.environ.get("NQE_FUZZ_SEED_BASE", "0"))      # ... and NQE_FUZZ_SEED_BASE=100000 for fresh cases
FLD = fields("id", "k", "v", "u", "b")
SIZES = [0, 1, 2, 63, 64, 65, 127, 1000, 4095, 4096, 4097, 8193, 20000, 70001]
RTOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    from naive_query_engine_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def flat(e):
    return e.flatten(FLD) if e is not None else None


def random_chain_key(rng, c=0, unsigned=False):
    """one or two integer operations with literals over one column (the fast kernels' interpreted key variant when it cannot fault,
    the general kernel when it can: zero / -1 divisors, literal-on-the-left divisions)"""
    lit = lit_u64 if unsigned else lit_i64
    lo = 0 if unsigned else -9
    e = col(c)
    for _ in range(int(rng.integers(1, 3))):
        op = [Operator.Plus, Operator.Minus, Operator.Multiply, Operator.Divide, Operator.Modulos][int(rng.integers(0, 5))]
        v = int(rng.choice([lo, 0, 1, 2, 3, 7, 16, 1000, 4097])) if op in (Operator.Divide, Operator.Modulos) else int(rng.integers(lo, 50))
        if op in (Operator.Divide, Operator.Modulos) and v == 0 and rng.random() < 0.8:
            v = 5
        e = binop(lit(abs(v) if unsigned else v), op, e) if rng.random() < 0.15 else binop(e, op, lit(abs(v) if unsigned else v))
    return e


def _random_leaf(rng):
    """`col cmp lit` / `lit cmp col` over the batch's Int64 / Float64 / UInt64 columns (ids, k, v, u)"""
    c = int(rng.choice([0, 1, 2, 3]))
    cmp_op = [Operator.Lt, Operator.LtEq, Operator.Gt, Operator.GtEq, Operator.Eq, Operator.NotEq][int(rng.integers(0, 6))]
    if c == 2:
        lit = lit_f64(float(rng.choice([0.0, -0.0, 25.5, -60.0, 99.0, float("nan"), float("inf"), float("-inf")])))
    elif c == 3:
        lit = lit_u64(int(rng.choice([0, 1, 1 << 20, 1 << 39, (1 << 40) - 1])))
    else:
        lit = lit_i64(int(rng.choice([-50, -1, 0, 1, 7, 49, 500, 3000])))
    return binop(lit, cmp_op, col(c)) if rng.random() < 0.25 else binop(col(c), cmp_op, lit)


def random_pred(rng):
    kind = int(rng.integers(0, 12))
    if kind == 9 or kind == 10:  # two to five tests joined by and / or (any association): in-kernel when the columns allow
        op = Operator.And if kind == 9 else Operator.Or
        parts = [_random_leaf(rng) for _ in range(int(rng.integers(2, 6)))]
        while len(parts) > 1:
            i = int(rng.integers(0, len(parts) - 1))
            parts[i:i + 2] = [binop(parts[i], op, parts[i + 1])]
        return parts[0]
    if kind == 11:  # Float64 chain ending in a comparison
        e = col(2)
        for _ in range(int(rng.integers(1, 4))):
            op = [Operator.Plus, Operator.Minus, Operator.Multiply, Operator.Divide][int(rng.integers(0, 4))]
            v = float(rng.choice([0.5, -2.0, 3.0, 100.0, 1e-3])) if op == Operator.Divide else float(rng.choice([0.0, -0.0, 0.5, -2.0, 3.0, 100.0, float("inf")]))
            e = binop(lit_f64(v), op, e) if (rng.random() < 0.2 and op != Operator.Divide) else binop(e, op, lit_f64(v))
        cmp_op = [Operator.Lt, Operator.LtEq, Operator.Gt, Operator.GtEq, Operator.Eq, Operator.NotEq][int(rng.integers(0, 6))]
        lit = lit_f64(float(rng.choice([0.0, 10.0, -75.0, 250.0, float("nan")])))
        return binop(lit, cmp_op, e) if rng.random() < 0.2 else binop(e, cmp_op, lit)
    if kind >= 7:  # integer chain ending in a comparison (the fast kernels' interpreted predicate when it cannot fault)
        unsigned = kind == 8
        e = random_chain_key(rng, 3 if unsigned else 0, unsigned=unsigned)
        cmp_op = [Operator.Lt, Operator.LtEq, Operator.Gt, Operator.GtEq, Operator.Eq, Operator.NotEq][int(rng.integers(0, 6))]
        lit = (lit_u64(int(rng.integers(0, 2000))) if unsigned else lit_i64(int(rng.integers(-50, 2000))))
        return binop(lit, cmp_op, e) if rng.random() < 0.2 else binop(e, cmp_op, lit)
    if kind == 0:
        return None
    if kind == 1:
        return binop(col(0), [Operator.Lt, Operator.GtEq, Operator.NotEq, Operator.Eq][int(rng.integers(0, 4))], lit_i64(int(rng.integers(-5, 3000))))
    if kind == 2:
        return binop(col(2), [Operator.Gt, Operator.LtEq, Operator.NotEq][int(rng.integers(0, 3))], lit_f64(float(rng.choice([0.0, -0.0, 25.5, -60.0, float("nan"), float("inf")]))))
    if kind == 3:
        return col(4)
    if kind == 4:
        return binop(lit_u64(int(rng.integers(0, 1 << 39))), Operator.Lt, col(3))
    return _random_tree(rng, int(rng.integers(1, 4)), "b")


def random_key(rng):
    kind = int(rng.integers(0, 8))
    if kind == 6:
        return random_chain_key(rng, 0)
    if kind == 7:
        return random_chain_key(rng, 3, unsigned=True)
    if kind == 0:
        return col(1)
    if kind == 1:
        return binop(col(0), Operator.Modulos, lit_i64(int(rng.choice([2, 7, 64, 1000, 1024, -16, 5000]))))
    if kind == 2:
        return binop(col(3), Operator.Modulos, lit_u64(int(rng.choice([3, 256, 4097]))))
    if kind == 3:
        return col(3)
    if kind == 4:
        return _random_tree(rng, int(rng.integers(1, 3)), "i")
    return col(0)


def both(fn_gpu, fn_orc, what):
    """runs both sides; an oracle error must be the same error on the device"""
    try:
        exp = fn_orc()
    except ErrorCode as e:
        with pytest.raises(ErrorCode) as g:
            fn_gpu()
        assert g.value.status == e.status, what
        return None, None
    return fn_gpu(), exp


@pytest.mark.parametrize("seed", range(12 + EXTRA))
def test_fuzz_selection_projection_aggregate(ctx, seed):
    rng = np.random.default_rng(1000 + BASE + seed)
    for case in range(14):
        n = int(rng.choice(SIZES))
        null_frac = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
        key_mod = int(rng.choice([3, 50, 3000])) if rng.random() < 0.7 else None
        cols = random_batch(rng, n, null_frac, key_mod=key_mod, with_bool=True)
        t = ctx.table_from_host(cols)
        pred, key = random_pred(rng), (random_key(rng) if rng.random() < 0.8 else None)
        what = f"seed {seed} case {case}: n={n} nulls={null_frac} pred={pred!r} key={key!r}"
        # selection + projection
        if pred is not None:
            proj = [col(0), _random_tree(rng, int(rng.integers(1, 4)), str(rng.choice(["i", "f", "u", "b"]))), col(2)]
            got, exp = both(lambda: ctx.selection_projection(t, flat(pred), [flat(e) for e in proj]).to_host(),
                            lambda: orc.projection(orc.selection([cols], flat(pred)), [flat(e) for e in proj])[0], what)
            if exp is not None:
                assert_batches_equal(got, exp, what=what + " [selection+projection]")
        # aggregate (values of every numeric type; count over Boolean)
        vcols = [int(c) for c in rng.choice([0, 1, 2, 3], size=int(rng.integers(1, 4)), replace=False)]
        funcs = [AggregateFunc.Count, AggregateFunc.Sum, AggregateFunc.Avg, AggregateFunc.Min, AggregateFunc.Max]
        aggs = [(f, c) for c in vcols for f in rng.choice(funcs, size=int(rng.integers(1, 6)), replace=False)]
        if rng.random() < 0.3:
            aggs.append((AggregateFunc.Count, 4))
        if n == 0 and key is None:
            continue  # un-grouped aggregate over an empty batch list is a separate (tested) error path
        got, exp = both(lambda: ctx.aggregate(t, aggs, group_nodes=flat(key), pred_nodes=flat(pred)).to_host(),
                        lambda: orc.aggregate([cols], aggs, group_nodes=flat(key), pred_nodes=flat(pred))[0], what)
        if exp is not None:
            exact = [i for i, (f, _) in enumerate(aggs) if f == AggregateFunc.Count]
            assert_rows_multiset_equal(got, exp, RTOL, exact_cols=exact, what=what + f" [aggregate {aggs}]")


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_fuzz_hash_join(ctx, seed):
    rng = np.random.default_rng(2000 + BASE + seed)
    for case in range(10):
        nb, npr = int(rng.choice([0, 1, 5, 64, 1000, 4097, 30000])), int(rng.choice([0, 1, 63, 65, 4096, 50001]))
        domain = int(rng.choice([4, 100, 5000, 1 << 20, 1 << 40]))
        unique = rng.random() < 0.5
        if unique and nb:
            bk = np.unique(rng.integers(0, max(domain, 2 * nb), 2 * nb + 8))[:nb]
            rng.shuffle(bk)
            nb = bk.size
        else:
            bk = rng.integers(0, domain, nb)
        pk = rng.integers(0, max(domain, 1), npr) if rng.random() < 0.5 or nb == 0 else rng.choice(bk, npr) + rng.integers(0, 2, npr)
        dt = np.uint64 if rng.random() < 0.3 else np.int64
        shift = 0 if dt == np.uint64 else int(rng.choice([0, -1000]))
        left = [Column.from_numpy((bk + shift).astype(dt)), Column.from_numpy(rng.random(nb), None if rng.random() < 0.5 else rng.random(nb) > 0.2)]
        right = [Column.from_numpy(rng.integers(0, 9, npr).astype(np.int64)), Column.from_numpy((pk + shift).astype(dt))]
        what = f"seed {seed} case {case}: nb={nb} npr={npr} domain={domain} unique={unique} dtype={dt.__name__}"
        got, exp = both(lambda: ctx.hash_join(ctx.table_from_host(left), ctx.table_from_host(right), 0, 1).to_host(),
                        lambda: orc.hash_join([left], [right], 0, 1)[0], what)
        if exp is not None:
            assert_batches_equal(got, exp, what=what)


@pytest.mark.parametrize("seed", range(4 + EXTRA // 4))
def test_fuzz_many_groups_partitioned_paths(ctx, seed):
    """enough rows and distinct keys to leave the single LDS table: partitioned (dense and hashed) and two-level paths,
    one and several value-column passes, nullable values"""
    rng = np.random.default_rng(3000 + BASE + seed)
    n = int(rng.choice([300_000, 700_001, 1_500_000, 4_000_000]))
    # 1500 / 3000: a key RANGE that fits a workgroup table (two / one value column); 3300 .. 7000: between one and two tables' worth (hashed
    # tables hand over at three quarters of their slots; two key subsets over a direct-mapped table, the range tier, hashed subsets)
    groups = int(rng.choice([1500, 3000, 3300, 5000, 7000, 40_000, 900_000]))
    null_frac = float(rng.choice([0.0, 0.0, 0.1]))
    shape = rng.random()
    if shape < 0.35:
        ids = rng.integers(-groups // 2, groups // 2, n).astype(np.int64)
    elif shape < 0.6:  # the same spread over 2 .. 9 times the range
        ids = (rng.integers(-groups // 2, groups // 2, n) * int(rng.integers(2, 10)) + int(rng.integers(-50, 50))).astype(np.int64)
    else:
        ids = (rng.integers(0, 1 << 50, groups)[rng.integers(0, groups, n)]).astype(np.int64)
    mask = (lambda: None if null_frac == 0 else rng.random(n) >= null_frac)
    cols = [Column.from_numpy(ids), Column.from_numpy(rng.integers(-1000, 1000, n).astype(np.int64), mask()), Column.from_numpy(rng.random(n) * 10, mask()),
            Column.from_numpy(rng.integers(0, 1 << 30, n).astype(np.uint64)), Column.from_numpy(rng.random(n) < 0.5)]
    t = ctx.table_from_host(cols)
    for case in range(3):
        pred = random_pred(rng) if case else None
        vcols = [1, 2, 3][: int(rng.integers(1, 4))]
        funcs = (AggregateFunc.Count, AggregateFunc.Sum, AggregateFunc.Min, AggregateFunc.Max, AggregateFunc.Avg) if rng.random() < 0.5 else \
            (AggregateFunc.Count, AggregateFunc.Sum, AggregateFunc.Avg)  # without min / max: the instances that leave those arrays out, three columns in one pass
        aggs = [(f, c) for c in vcols for f in funcs]
        key = col(0) if case < 2 else [binop(binop(col(0), Operator.Multiply, lit_i64(3)), Operator.Plus, lit_i64(1)),
                                       binop(binop(col(0), Operator.Plus, lit_i64(7)), Operator.Modulos, lit_i64(max(groups // 2, 1) + 1)),
                                       binop(col(0), Operator.Divide, lit_i64(2))][int(rng.integers(0, 3))]
        what = f"seed {seed} case {case}: n={n} groups={groups} nulls={null_frac} pred={pred!r} vcols={vcols} key={key!r}"
        got, exp = both(lambda: ctx.aggregate(t, aggs, group_nodes=flat(key), pred_nodes=flat(pred)).to_host(),
                        lambda: orc.aggregate([cols], aggs, group_nodes=flat(key), pred_nodes=flat(pred))[0], what)
        if exp is not None:
            exact = [i for i, (f, _) in enumerate(aggs) if f == AggregateFunc.Count]
            assert_rows_multiset_equal(got, exp, RTOL, exact_cols=exact, what=what)
            # the second execution starts from what the first one remembered (plan hint, key range)
            again = ctx.aggregate(t, aggs, group_nodes=flat(key), pred_nodes=flat(pred)).to_host()
            assert_rows_multiset_equal(again, exp, RTOL, exact_cols=exact, what=what + " (second execution)")


@pytest.mark.parametrize("seed", range(4 + EXTRA // 4))
def test_fuzz_utf8_everywhere(ctx, seed):
    """Utf8 as payload through selection / join, as group key, as join key, in comparisons; partial + merge of aggregates"""
    from naive_query_engine_amd.expression import lit_utf8
    from tests.helpers import random_utf8

    rng = np.random.default_rng(4000 + BASE + seed)
    n = int(rng.choice([1, 64, 1000, 4097, 30000]))
    null_frac = float(rng.choice([0.0, 0.2]))
    s1, s2 = random_utf8(rng, n, null_frac), random_utf8(rng, n, 0.0)
    ids = Column.from_numpy(rng.integers(0, 60, n).astype(np.int64))
    v = Column.from_numpy(rng.random(n), None if null_frac == 0 else rng.random(n) > null_frac)
    cols = [s1, s2, ids, v]
    f = fields("s1", "s2", "id", "v")
    t = ctx.table_from_host(cols)
    op = [Operator.Eq, Operator.NotEq, Operator.Lt, Operator.GtEq][int(rng.integers(0, 4))]
    pred = binop(binop(col(0), op, lit_utf8(str(rng.choice(["bob", "", "lynne", "x"])))), Operator.Or, binop(col(2), Operator.Lt, lit_i64(int(rng.integers(0, 60)))))
    what = f"seed {seed}: n={n} nulls={null_frac} pred={pred!r}"
    assert_batches_equal(ctx.selection(t, pred.flatten(f)).to_host(), orc.selection([cols], pred.flatten(f))[0], what=what + " [selection]")
    aggs = [(AggregateFunc.Count, 3), (AggregateFunc.Sum, 3), (AggregateFunc.Max, 3), (AggregateFunc.Count, 0)]
    for key in (col(1), col(0)):
        exp = orc.aggregate([cols], aggs, group_nodes=key.flatten(f), pred_nodes=pred.flatten(f))[0]
        got = ctx.aggregate(t, aggs, group_nodes=key.flatten(f), pred_nodes=pred.flatten(f))
        assert_rows_multiset_equal(got.to_host(), exp, RTOL, exact_cols=[0, 3], what=what + f" [group by {key!r}]")
    # partial + merge over two halves == single pass
    if n >= 2:
        h = n // 2
        halves = [[Column.from_list(c.to_list()[a:b], c.dtype) if c.dtype == DType.UTF8 else Column.from_numpy(c.to_numpy()[a:b], c.valid_mask()[a:b]) for c in cols] for a, b in ((0, h), (h, n))]
        parts = [ctx.aggregate_partial(ctx.table_from_host(hc), aggs[:3], group_nodes=col(2).flatten(f), pred_nodes=None) for hc in halves]
        merged, _ = ctx.aggregate_merge([p[0] for p in parts], [p[1] for p in parts], aggs[:3])
        exp = orc.aggregate([cols], aggs[:3], group_nodes=col(2).flatten(f))[0]
        assert_rows_multiset_equal(merged.to_host(), exp, RTOL, exact_cols=[0], what=what + " [partial+merge]")
    # join on a Utf8 key with Utf8 payload on both sides
    nb = int(rng.choice([1, 50, 2000]))
    left = [random_utf8(rng, nb, 0.0), Column.from_numpy(rng.integers(0, 9, nb).astype(np.int64)), random_utf8(rng, nb, null_frac)]
    exp = orc.hash_join([left], [cols], 0, 1)[0]
    got = ctx.hash_join(ctx.table_from_host(left), t, 0, 1).to_host()
    assert_batches_equal(got, exp, what=what + f" [utf8 join nb={nb}]")


@pytest.mark.parametrize("seed", range(4 + EXTRA // 4))
def test_fuzz_csv(ctx, seed):
    """random CSV images (quotes, escapes, embedded terminators, empty fields and lines, CR/LF/CRLF, odd delimiters)"""
    rng = np.random.default_rng(5000 + BASE + seed)
    for case in range(12):
        nrows, ncols = int(rng.choice([0, 1, 3, 4, 50, 700])), int(rng.integers(1, 6))
        delim = str(rng.choice([",", ";", "|", "\t"]))
        kinds = [str(rng.choice(["int", "float", "str", "bool", "mixed"])) for _ in range(ncols)]
        def field(kind):
            r = rng.random()
            if r < 0.06:
                return ""
            if kind == "int":
                return str(int(rng.integers(-10**15, 10**15)))
            if kind == "float":
                return str(rng.choice([repr(float(rng.normal() * 1e3)), f"{rng.integers(-999, 999)}.{rng.integers(0, 99999)}", "0.0", "-12.5"]))
            if kind == "bool":
                return str(rng.choice(["true", "false", "TRUE", "False"]))
            if kind == "mixed":
                return str(rng.choice(["1", "2.5", "x", "true", "2020-01-01x"]))
            raw = str(rng.choice(["a", "bob", 'say "hi"', "x" + delim + "y", "line\nbreak", "cr\rhere", "日本", ' lead', 'q"mid', ""]))
            if any(ch in raw for ch in (delim, '"', "\n", "\r")) and not raw.startswith('q"'):
                return '"' + raw.replace('"', '""') + '"'
            return raw if raw != "" or rng.random() < 0.5 else '""'
        lines = [delim.join(f"c{j}" for j in range(ncols))] if rng.random() < 0.8 else []
        has_header = bool(lines)
        for _ in range(nrows):
            lines.append(delim.join(field(k) for k in kinds))
            if rng.random() < 0.05:
                lines.append("")
        term = str(rng.choice(["\n", "\r\n", "\r"]))
        data = term.join(lines).encode() + (term.encode() if rng.random() < 0.7 else b"")
        kw = dict(has_header=has_header, delimiter=delim, max_read_records=int(rng.choice([-1, 1, 3, 1000])), batch_size=int(rng.choice([1_000_000, 5, 0])))
        what = f"seed {seed} case {case}: rows={nrows} kinds={kinds} kw={kw} data={data[:120]!r}"
        try:
            names, nullable, exp = orc.csv_read(data, **kw)
        except ErrorCode as e:
            with pytest.raises(ErrorCode) as g:
                gn, gd, _ = ctx.csv_infer_schema(data, kw["has_header"], delim, kw["max_read_records"], kw["batch_size"])
                ctx.csv_read(data, gd, kw["has_header"], delim, kw["batch_size"])
            assert g.value.status == e.status, what
            continue
        gn, gd, gnull = ctx.csv_infer_schema(data, kw["has_header"], delim, kw["max_read_records"], kw["batch_size"])
        assert (gn, gnull, gd) == (names, nullable, [c.dtype for c in exp]), what
        assert_batches_equal(ctx.csv_read(data, gd, kw["has_header"], delim, kw["batch_size"]).to_host(), exp, what=what)
