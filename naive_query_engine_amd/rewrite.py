"""Plan-level rewrite pass: takes the UNFUSED operator tree the reference's planner builds
(`QueryPlanner::create_physical_plan`, src/planner/mod.rs:42-182: a `ProjectionPlan` over a `SelectionPlan` over a `ScanPlan`, an
aggregate over a selection, …) and substitutes the subtrees the device executes in one go:

    ProjectionPlan(SelectionPlan(x))          →  FusedSelectionProjectionPlan(x)      nqe_selection_projection_execute
    PhysicalAggregatePlan(SelectionPlan(x))   →  FusedSelectionAggregatePlan(x)       nqe_aggregate_execute with its predicate argument

Everything else keeps its operator, with rewritten children.  The pass walks the tree through `children()` exactly as the
reference's `_visit_physical_plan` does (src/physical_plan/visitor.rs:12-24: pre_visit, children in order, post_visit) —
`PhysicalPlanVisitor` / `visit_physical_plan` mirror that hook — and rebuilds bottom-up.  `rewrite(tree).execute()` returns what
`tree.execute()` returns (tests/test_gpu_plans.py); the fused operators fall back to the plain chain wherever the reference's
semantics depend on the tree being unfused (several input batches: quirk Q3's predicate-from-batch-0).

Also here: `Catalog` / `NaiveDB`, the table-registration surface of src/catalog.rs:28-62 and src/db.rs:19-47 (the SQL front end that
sits on top of them in the reference is out of scope; `NaiveDB.run_plan` takes the physical tree the planner would build).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

from .arrow_host import ErrorCode, Field, RecordBatch, Status
from .physical_plan import (CsvConfig, CsvTable, DeviceRecordBatch, HashJoin, MemTable, NaiveSchema, PhysicalAggregatePlan, PhysicalLimitPlan,
                            PhysicalOffsetPlan, PhysicalPlan, ProjectionPlan, ScanPlan, SelectionPlan, _ctx_of, _Materialized)


# ----------------------------------------------------------------------------- visitor.rs:4-24
class PhysicalPlanVisitor:
    """trait PhysicalPlanVistor (visitor.rs:4-10)"""

    def pre_visit(self, plan: PhysicalPlan) -> None:  # "Invoke before visit PhysicalPlan"
        pass

    def post_visit(self, plan: PhysicalPlan) -> None:  # "Invoke before after PhysicalPlan"
        pass


def visit_physical_plan(plan: PhysicalPlan, visitor: PhysicalPlanVisitor) -> None:
    """_visit_physical_plan (visitor.rs:12-24): children() first, then pre_visit, the children in order, post_visit"""
    children = plan.children()
    visitor.pre_visit(plan)
    for child in children:
        visit_physical_plan(child, visitor)
    visitor.post_visit(plan)


# ----------------------------------------------------------------------------- fused operators
class FusedSelectionProjectionPlan(PhysicalPlan):
    """ProjectionPlan(SelectionPlan(input)) as one device pass: only the columns the projection references are compacted and the
    expressions are evaluated in the compaction kernel (C2 of BASELINE.json)."""

    def __init__(self, input: PhysicalPlan, predicate, schema: NaiveSchema, expr):
        self.input, self.predicate, self._schema, self.expr = input, predicate, list(schema), list(expr)

    def schema(self):
        return self._schema

    def children(self):
        return [self.input]

    def unfused(self, child: PhysicalPlan) -> PhysicalPlan:
        return ProjectionPlan.create(SelectionPlan.create(child, self.predicate), self._schema, self.expr)

    def execute(self):
        below = self.input.execute()
        if len(below) != 1 or not self._schema:
            # several batches: the predicate comes from batch 0 (Q3) — the plain operators reproduce that; an empty projection
            # schema is the reference's pass-through (projection.rs:47-48)
            return self.unfused(_Materialized(below, self.input.schema())).execute()
        ctx = _ctx_of(below)
        fields = below[0].fields
        t = ctx.selection_projection(below[0].table, self.predicate.flatten(fields), [e.flatten(fields) for e in self.expr])
        return [DeviceRecordBatch([Field(f.name, dt, True) for f, dt in zip(self._schema, t.dtypes())], t)]


class FusedSelectionAggregatePlan(PhysicalAggregatePlan):
    """PhysicalAggregatePlan(SelectionPlan(input)): the filter is the aggregation kernel's predicate (the headline query).  The
    aggregate's own state and quirks (Q8, Q9, Q10) are inherited unchanged."""

    def __init__(self, group_expr, aggr_ops, predicate, input: PhysicalPlan):
        super().__init__(group_expr, aggr_ops, input)
        self.predicate = predicate

    def unfused(self, child: PhysicalPlan) -> PhysicalPlan:
        return PhysicalAggregatePlan.create(self.group_expr, self.aggr_ops, SelectionPlan.create(child, self.predicate))

    def _input_batches(self):
        below = self.input.execute()
        if len(below) == 1:
            return below, self.predicate
        # several batches (Q3), or none (the selection's own error): the plain selection over the batches already produced
        return SelectionPlan.create(_Materialized(below, self.input.schema()), self.predicate).execute(), None


def rewrite(plan: PhysicalPlan) -> PhysicalPlan:
    """the substitution pass (see the module docstring); returns a NEW tree, the input tree is left as it was"""
    if isinstance(plan, (FusedSelectionProjectionPlan, FusedSelectionAggregatePlan, ScanPlan, _Materialized)):
        return plan
    if isinstance(plan, ProjectionPlan):
        if isinstance(plan.input, SelectionPlan) and plan._schema:
            return FusedSelectionProjectionPlan(rewrite(plan.input.input), plan.input.expr, plan._schema, plan.expr)
        return ProjectionPlan.create(rewrite(plan.input), plan._schema, plan.expr)
    if isinstance(plan, PhysicalAggregatePlan):
        if isinstance(plan.input, SelectionPlan):
            return FusedSelectionAggregatePlan(plan.group_expr, plan.aggr_ops, plan.input.expr, rewrite(plan.input.input))
        return PhysicalAggregatePlan.create(plan.group_expr, plan.aggr_ops, rewrite(plan.input))
    if isinstance(plan, SelectionPlan):
        return SelectionPlan.create(rewrite(plan.input), plan.expr)
    if isinstance(plan, PhysicalLimitPlan):
        return PhysicalLimitPlan.create(rewrite(plan.input), plan.n)
    if isinstance(plan, PhysicalOffsetPlan):
        return PhysicalOffsetPlan.create(rewrite(plan.input), plan.n)
    if isinstance(plan, HashJoin):
        return HashJoin.create(rewrite(plan.left), rewrite(plan.right), plan.on, plan.join_type, plan._schema)
    return plan  # an operator this pass does not know: left alone, children included


class _Shape(PhysicalPlanVisitor):
    def __init__(self):
        self.names: List[str] = []

    def pre_visit(self, plan):
        self.names.append(type(plan).__name__)


def plan_shape(plan: PhysicalPlan) -> List[str]:
    """operator names in visit order (pre-order): what a plan looks like before / after `rewrite`"""
    v = _Shape()
    visit_physical_plan(plan, v)
    return v.names


# ----------------------------------------------------------------------------- catalog.rs / db.rs
class Catalog:
    """src/catalog.rs:21-62"""

    def __init__(self):
        self.tables: Dict[str, object] = {}

    def add_csv_table(self, table: str, csv_file: str, csv_conf: Optional[CsvConfig] = None) -> None:
        self.tables[table] = CsvTable.try_create(csv_file, csv_conf)

    def add_memory_table(self, table: str, schema: NaiveSchema, batches: Sequence[RecordBatch]) -> None:
        self.tables[table] = MemTable.try_create(schema, batches)

    def add_arrow_table(self, table: str, record_batches) -> None:
        """real Arrow data (pyarrow.RecordBatch objects) through the Arrow C Data Interface of the C ABI"""
        from . import capi

        ctx = capi.default_context()
        if not record_batches:
            raise ErrorCode(Status.Others, "add_arrow_table needs at least one RecordBatch (its schema is the table's)")
        tabs = [ctx.table_from_arrow(rb) for rb in record_batches]
        schema = [Field(n, dt, True) for n, dt in zip(record_batches[0].schema.names, tabs[0].dtypes())]
        self.tables[table] = MemTable.from_device(schema, tabs)

    def get_table(self, table: str):
        if table not in self.tables:
            raise ErrorCode(Status.NoSuchTable, f"Unable to get table named: {table}")
        return self.tables[table]


class NaiveDB:
    """src/db.rs:19-47 without its SQL front end: tables are registered as in the reference; `run_plan` executes the physical tree
    `QueryPlanner::create_physical_plan` would hand to `execute()` (db.rs:34-36), after the rewrite pass."""

    def __init__(self):
        self.catalog = Catalog()

    def create_csv_table(self, table: str, csv_file: str, csv_conf: Optional[CsvConfig] = None) -> None:
        self.catalog.add_csv_table(table, csv_file, csv_conf)

    def create_memory_table(self, table: str, schema: NaiveSchema, batches: Sequence[RecordBatch]) -> None:
        self.catalog.add_memory_table(table, schema, batches)

    def create_arrow_table(self, table: str, record_batches) -> None:
        self.catalog.add_arrow_table(table, record_batches)

    def scan(self, table: str, projection: Optional[Sequence[int]] = None) -> ScanPlan:
        return ScanPlan.create(self.catalog.get_table(table), projection)

    def run_plan(self, physical_plan: PhysicalPlan) -> List[DeviceRecordBatch]:
        return rewrite(physical_plan).execute()

    def run_sql(self, sql: str):
        raise ErrorCode(Status.NotSupported, "the SQL parser / logical planner of the reference are out of scope here (SURVEY §8): build the physical plan and call run_plan")
