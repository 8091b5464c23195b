"""Host mirror of the reference's physical operators (src/physical_plan/*.rs).

Same class names, constructor arguments, `schema()/execute()/children()` surface and error
behaviour as the reference, so plans are written as in its tests:

    scan = ScanPlan.create(MemTable.try_create(schema, [batch]), None)
    sel  = SelectionPlan.create(scan, PhysicalBinaryExpr.create(ColumnExpr.try_create("id", None),
                                                                Operator.Gt, PhysicalLiteralExpr.create(ScalarValue.Int64(1))))
    batches = sel.execute()          # Vec<RecordBatch> — here: device-resident batches

`execute()` returns `DeviceRecordBatch`es (columns stay in HBM between operators; `.to_host()`
downloads).  All computation goes through the C ABI (capi → libnqe_hip.so); there is no host
fallback.  Every operator here does exactly what its reference namesake does, one operator at a
time; the fused device entry points (Projection∘Selection, Aggregate∘Selection) are introduced
by a separate pass over the tree, `rewrite()` in rewrite.py — `rewrite(tree).execute()` equals
`tree.execute()`, which tests/test_gpu_plans.py checks.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

from . import capi
from .arrow_host import AggregateFunc, Column, DType, ErrorCode, Field, RecordBatch, Status
from .expression import ColumnExpr, PhysicalExpr

NaiveSchema = List[Field]  # src/logical_plan/schema.rs (qualifiers are ignored at physical planning, Q12)


def field_with_unqualified_name(schema: Sequence[Field], name: str) -> Field:
    """NaiveSchema::field_with_unqualified_name (schema.rs:116-125): first match."""
    for f in schema:
        if f.name == name:
            return f
    raise ErrorCode(Status.PlanError, f"No field named '{name}'")


class DeviceRecordBatch:
    """A RecordBatch whose columns live in HBM (nqe_table handle + field names)."""

    def __init__(self, fields: Sequence[Field], table: "capi.Table"):
        self.fields = list(fields)
        self.table = table

    @property
    def num_rows(self) -> int:
        return self.table.num_rows

    @property
    def num_columns(self) -> int:
        return self.table.num_columns

    def to_host(self) -> RecordBatch:
        cols = self.table.to_host()
        return RecordBatch([Field(f.name, c.dtype, c.validity is not None) for f, c in zip(self.fields, cols)], cols)

    def column(self, i: int) -> Column:
        return self.table.download_column(i)


# ----------------------------------------------------------------------------- data sources
class MemTable:
    """src/datasource/memory.rs:14-46 — batches are uploaded to HBM once, at creation."""

    def __init__(self, schema: NaiveSchema, batches: List[DeviceRecordBatch]):
        self._schema = list(schema)
        self.batches = batches

    @staticmethod
    def try_create(schema: NaiveSchema, batches: Sequence[RecordBatch], ctx: Optional["capi.Context"] = None) -> "MemTable":
        ctx = ctx or capi.default_context()
        dev = [DeviceRecordBatch(schema, ctx.table_from_host(b.columns)) for b in batches]
        return MemTable(schema, dev)

    @staticmethod
    def from_device(schema: NaiveSchema, tables: Sequence["capi.Table"]) -> "MemTable":
        return MemTable(schema, [DeviceRecordBatch(schema, t) for t in tables])

    def schema(self) -> NaiveSchema:
        return self._schema

    def scan(self, projection: Optional[Sequence[int]]) -> List[DeviceRecordBatch]:
        if projection is None:
            return list(self.batches)  # Arc clones (memory.rs:41)
        out = []
        for b in self.batches:  # RecordBatch::project (memory.rs:33-38)
            t = b.table.ctx.project(b.table, list(projection))
            out.append(DeviceRecordBatch([b.fields[i] for i in projection], t))
        return out

    def source_name(self) -> str:
        return "MemTable"


class CsvConfig:
    """src/datasource/csv.rs:23-43 (file_projection / datetime_format are not mirrored)"""

    def __init__(self, has_header: bool = True, delimiter: str = ",", max_read_records: Optional[int] = 3, batch_size: int = 1_000_000):
        self.has_header, self.delimiter, self.max_read_records, self.batch_size = has_header, delimiter, max_read_records, batch_size


class CsvTable(MemTable):
    """src/datasource/csv.rs:46-103 — schema inferred from the first max_read_records records, only the first batch is
    loaded (Q1); scan ignores projection (Q2).  The file image is parsed on the GPU (nqe_csv_read)."""

    @staticmethod
    def try_create(filename: str, csv_config: Optional[CsvConfig] = None, ctx: Optional["capi.Context"] = None) -> "CsvTable":
        cfg = csv_config or CsvConfig()
        ctx = ctx or capi.default_context()
        with open(filename, "rb") as f:
            data = f.read()
        mrr = -1 if cfg.max_read_records is None else cfg.max_read_records
        names, dtypes, nullable = ctx.csv_infer_schema(data, cfg.has_header, cfg.delimiter, mrr, cfg.batch_size)
        table = ctx.csv_read(data, dtypes, cfg.has_header, cfg.delimiter, cfg.batch_size)
        fields = [Field(n, d, nl) for n, d, nl in zip(names, dtypes, nullable)]
        return CsvTable(fields, [DeviceRecordBatch(fields, table)])

    def scan(self, projection):
        return list(self.batches)

    def source_name(self) -> str:
        return "CsvTable"


# ----------------------------------------------------------------------------- plans
class PhysicalPlan:
    """trait PhysicalPlan (src/physical_plan/plan.rs:14-21)."""

    def schema(self) -> NaiveSchema:
        raise NotImplementedError

    def execute(self) -> List[DeviceRecordBatch]:
        raise NotImplementedError

    def children(self) -> List["PhysicalPlan"]:
        raise NotImplementedError


class ScanPlan(PhysicalPlan):
    """src/physical_plan/scan.rs:18-41"""

    def __init__(self, source, projection):
        self.source, self.projection = source, projection

    @staticmethod
    def create(source, projection: Optional[Sequence[int]] = None) -> "ScanPlan":
        return ScanPlan(source, projection)

    def schema(self):
        return self.source.schema()

    def execute(self):
        return self.source.scan(self.projection)

    def children(self):
        return []


def _ctx_of(batches: Sequence[DeviceRecordBatch]) -> "capi.Context":
    return batches[0].table.ctx if batches else capi.default_context()


class SelectionPlan(PhysicalPlan):
    """src/physical_plan/selection.rs:23-112"""

    def __init__(self, input: PhysicalPlan, expr: PhysicalExpr):
        self.input, self.expr = input, expr

    @staticmethod
    def create(input: PhysicalPlan, expr: PhysicalExpr) -> "SelectionPlan":
        return SelectionPlan(input, expr)

    def schema(self):
        return self.input.schema()

    def execute(self):
        batches = self.input.execute()
        if not batches:  # input[0] panics (selection.rs:60)
            raise ErrorCode(Status.NotSupported, "index out of bounds: input[0] (selection.rs:60 panics)")
        ctx = _ctx_of(batches)
        pred = self.expr.flatten(batches[0].fields)
        if len(batches) == 1:
            return [DeviceRecordBatch(batches[0].fields, ctx.selection(batches[0].table, pred))]
        # quirk Q3: the predicate is evaluated on batch 0 only and zipped against every batch
        mask = ctx.expr_evaluate(batches[0].table, pred)
        return [DeviceRecordBatch(b.fields, ctx.filter(b.table, mask, 0)) for b in batches]

    def children(self):
        return [self.input]


class ProjectionPlan(PhysicalPlan):
    """src/physical_plan/projection.rs:18-75"""

    def __init__(self, input: PhysicalPlan, schema: NaiveSchema, expr: Sequence[PhysicalExpr]):
        self.input, self._schema, self.expr = input, list(schema), list(expr)

    @staticmethod
    def create(input: PhysicalPlan, schema: NaiveSchema, expr: Sequence[PhysicalExpr]) -> "ProjectionPlan":
        return ProjectionPlan(input, schema, expr)

    def schema(self):
        return self._schema

    def _out_fields(self, cols_dtypes):
        return [Field(f.name, dt, True) for f, dt in zip(self._schema, cols_dtypes)]

    def execute(self):
        if not self._schema:  # projection.rs:47-48: pass-through above an aggregate
            return self.input.execute()
        batches = self.input.execute()
        out = []
        for b in batches:
            ctx = b.table.ctx
            t = ctx.projection(b.table, [e.flatten(b.fields) for e in self.expr])
            out.append(DeviceRecordBatch(self._out_fields(t.dtypes()), t))
        return out

    def children(self):
        return [self.input]


class _Materialized(PhysicalPlan):
    """already-executed child (avoids executing a subtree twice when a fusion attempt falls through)"""

    def __init__(self, batches, schema):
        self._batches, self._schema = batches, schema

    def schema(self):
        return self._schema

    def execute(self):
        return self._batches

    def children(self):
        return []


class PhysicalLimitPlan(PhysicalPlan):
    """src/physical_plan/limit.rs:32-49"""

    def __init__(self, input, n):
        self.input, self.n = input, n

    @staticmethod
    def create(input, n: int):
        return PhysicalLimitPlan(input, n)

    def schema(self):
        return self.input.schema()

    def execute(self):
        n, ret = self.n, []
        for b in self.input.execute():
            if n == 0:
                break
            if b.num_rows <= n:
                ret.append(b)
                n -= b.num_rows
            else:
                ret.append(DeviceRecordBatch(b.fields, b.table.ctx.slice(b.table, 0, n)))
                n = 0
        return ret

    def children(self):
        return [self.input]


class PhysicalOffsetPlan(PhysicalPlan):
    """src/physical_plan/offset.rs:30-51"""

    def __init__(self, input, n):
        self.input, self.n = input, n

    @staticmethod
    def create(input, n: int):
        return PhysicalOffsetPlan(input, n)

    def schema(self):
        return self.input.schema()

    def execute(self):
        n, ret = self.n, []
        for b in self.input.execute():
            if n == 0:
                ret.append(b)
                continue
            if n >= b.num_rows:
                n -= b.num_rows
                continue
            ret.append(DeviceRecordBatch(b.fields, b.table.ctx.slice(b.table, n, b.num_rows - n)))
            n = 0
        return ret

    def children(self):
        return [self.input]


# ----------------------------------------------------------------------------- aggregates
class AggregateOperator:
    """trait AggregateOperator (src/physical_plan/aggregate/mod.rs:225-235).  The per-row
    update()/update_batch() loops run on the device; the host object only names the function,
    its column and the output field."""

    func: AggregateFunc = AggregateFunc.Count
    label = "count"
    out_dtype = DType.FLOAT64

    def __init__(self, col_expr: ColumnExpr):
        self.col_expr = col_expr

    @classmethod
    def create(cls, col_expr: ColumnExpr):
        return cls(col_expr)

    def data_field(self, schema: NaiveSchema) -> Field:
        if self.col_expr.name is not None:
            f = field_with_unqualified_name(schema, self.col_expr.name)
            return Field(f"{self.label}({f.name})", self.out_dtype, False)
        if self.col_expr.idx is not None:
            return Field(f"{self.label}({schema[self.col_expr.idx].name})", self.out_dtype, False)
        raise ErrorCode(Status.LogicalError, "ColumnExpr must has name or idx")


class Sum(AggregateOperator):
    """src/physical_plan/aggregate/sum.rs"""
    func, label = AggregateFunc.Sum, "sum"


class Avg(AggregateOperator):
    """src/physical_plan/aggregate/avg.rs"""
    func, label = AggregateFunc.Avg, "avg"


class Count(AggregateOperator):
    """src/physical_plan/aggregate/count.rs"""
    func, label, out_dtype = AggregateFunc.Count, "count", DType.UINT64


class Max(AggregateOperator):
    """src/physical_plan/aggregate/max.rs"""
    func, label = AggregateFunc.Max, "max"


class Min(AggregateOperator):
    """src/physical_plan/aggregate/min.rs"""
    func, label = AggregateFunc.Min, "min"


class PhysicalAggregatePlan(PhysicalPlan):
    """src/physical_plan/aggregate/mod.rs:31-223"""

    def __init__(self, group_expr, aggr_ops, input):
        self.group_expr, self.aggr_ops, self.input = list(group_expr), list(aggr_ops), input
        self._schema = list(input.schema())
        self._ungrouped_state = None  # quirk Q9: un-grouped state is never cleared between execute() calls

    @staticmethod
    def create(group_expr: Sequence[PhysicalExpr], aggr_ops: Sequence[AggregateOperator], input: PhysicalPlan):
        return PhysicalAggregatePlan(group_expr, aggr_ops, input)

    def schema(self):
        return self._schema  # the INPUT schema (quirk Q8/Q13)

    def children(self):
        return [self.input]

    def _input_batches(self):
        """(input batches, predicate expression the aggregation kernel applies itself): the plain operator has no predicate"""
        return self.input.execute(), None

    def execute(self):
        out_fields = [op.data_field(self._schema) for op in self.aggr_ops]
        batches, pred_expr = self._input_batches()
        ctx = _ctx_of(batches)
        if not batches:
            if self.group_expr:
                # concat_batches of an empty Vec → empty batch → zero groups
                raise ErrorCode(Status.NotSupported, "aggregate over an empty batch list is not supported on the device path")
            raise ErrorCode(Status.NotSupported, "aggregate over an empty batch list is not supported on the device path")
        fields = batches[0].fields
        aggs = [(op.func, op.col_expr.resolve(fields)) for op in self.aggr_ops]
        pred = pred_expr.flatten(fields) if pred_expr is not None else None
        if not self.group_expr:
            # update_batch per batch (:123-139): partial state per batch, merged with the running state
            parts = [ctx.aggregate_partial(b.table, aggs, pred_nodes=pred)[0] for b in batches]
            if self._ungrouped_state is not None:
                parts.insert(0, self._ungrouped_state)
            if len(parts) > 1 or self._ungrouped_state is not None:
                # keep the merged raw state for the next execute() (Q9)
                merged_state = _merge_raw_state(ctx, parts, aggs)
            else:
                merged_state = parts[0]
            self._ungrouped_state = merged_state
            out, _ = ctx.aggregate_merge([merged_state], None, aggs)
            return [DeviceRecordBatch(out_fields, out)]
        # grouped: concat_batches, then group by group_expr[0] only (:143-148)
        table = batches[0].table if len(batches) == 1 else ctx.concat([b.table for b in batches])
        key = self.group_expr[0].flatten(fields)
        out = ctx.aggregate(table, aggs, group_nodes=key, pred_nodes=pred)
        return [DeviceRecordBatch(out_fields, out)]


def _merge_raw_state(ctx, parts, aggs):
    """Un-grouped raw states are single rows of (count,sum,min,max) per aggregate and the merge is
    associative, so the running state is simply the concatenation of the partial rows; it is folded
    by nqe_aggregate_merge whenever a result is needed."""
    return ctx.concat(parts)


# ----------------------------------------------------------------------------- hash join
@dataclass(frozen=True)
class ColumnRef:
    """logical_plan::expression::Column (expression.rs:167-170)"""

    table: Optional[str]
    name: str


class JoinType:
    Inner, Left, Right, Cross = range(4)


class HashJoin(PhysicalPlan):
    """src/physical_plan/hash_join.rs:44-289 — LEFT child = build side, RIGHT = probe side (Q11)."""

    def __init__(self, left, right, on, join_type, schema):
        self.left, self.right, self.on, self.join_type, self._schema = left, right, list(on), join_type, list(schema)
        self._executions = 0  # the reference never clears its hash table between execute() calls (quirk Q11)

    @staticmethod
    def create(left: PhysicalPlan, right: PhysicalPlan, on: Sequence[Tuple[ColumnRef, ColumnRef]], join_type, schema: NaiveSchema):
        return HashJoin(left, right, on, join_type, schema)

    def schema(self):
        return self._schema

    def children(self):
        return [self.left, self.right]

    def execute(self):
        if not self.on:  # hash_join.rs:125-129
            raise ErrorCode(Status.PlanError, "Inner Join on Conditions can't not be empty")
        lb = self.left.execute()
        rb = self.right.execute()
        ctx = _ctx_of(lb or rb)
        if not lb:
            raise ErrorCode(Status.NotSupported, "join with an empty left batch list is not supported on the device path")
        ltab = lb[0].table if len(lb) == 1 else ctx.concat([b.table for b in lb])  # concat_batches (:132)
        lkey = ColumnExpr.try_create(self.on[0][0].name, None).resolve(lb[0].fields)  # by NAME, first match (:134-136)
        # Q11: build() pushes the row indices into the SAME map again on every execute(), so the k-th execute() emits each
        # match k times, in the order [matches of the first build..., of the second...].  A build side of k copies of the
        # left batch has exactly that match order (row i + c*n carries the payload of row i).
        self._executions += 1
        if self._executions > 1:
            ltab = ctx.concat([ltab] * self._executions)
        jt = ctx.hash_join_build(ltab, lkey)
        out = []
        for b in rb:  # one output batch per probe batch (:177-250)
            rkey = ColumnExpr.try_create(self.on[0][1].name, None).resolve(b.fields)
            t = ctx.hash_join_probe(jt, b.table, rkey)
            fields = self._schema if len(self._schema) == t.num_columns else list(lb[0].fields) + list(b.fields)
            out.append(DeviceRecordBatch(fields, t))
        return out
