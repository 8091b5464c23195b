"""Host mirror of the reference's physical expressions.

Same names and constructor arguments as src/physical_plan/expression/{column,literal,binary}.rs so
plans are written exactly as in the reference's tests (e.g. selection.rs:145-155):

    add_expr = PhysicalBinaryExpr.create(ColumnExpr.try_create("id", None), Operator.Plus,
                                         PhysicalLiteralExpr.create(ScalarValue.Int64(1)))

The host side only *describes* the tree; evaluation happens on the GPU through the C ABI
(nqe_expr_evaluate and the fused operator entry points), which take the flat post-order
encoding produced by `flatten()`.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from .arrow_host import (ErrorCode, Field, NqeExprNode, Operator, ScalarValue, Status, node_binary, node_column,
                         node_literal)


class PhysicalExpr:
    """trait PhysicalExpr (src/physical_plan/expression/mod.rs:25-29)."""

    def flatten(self, fields: Sequence[Field]) -> List[NqeExprNode]:
        raise NotImplementedError

    def referenced_columns(self, fields: Sequence[Field]) -> List[int]:
        return [n.column for n in self.flatten(fields) if n.kind == 0]


class ColumnExpr(PhysicalExpr):
    """src/physical_plan/expression/column.rs:18-58: prefers idx, then the FIRST field whose
    name matches (quirk Q12)."""

    def __init__(self, name: Optional[str], idx: Optional[int]):
        self.name = name
        self.idx = idx

    @staticmethod
    def try_create(name: Optional[str] = None, idx: Optional[int] = None) -> "ColumnExpr":
        if name is None and idx is None:
            raise ErrorCode(Status.LogicalError, "ColumnExpr must has name or idx")
        return ColumnExpr(name, idx)

    def resolve(self, fields: Sequence[Field]) -> int:
        if self.idx is not None:
            if not 0 <= self.idx < len(fields):
                # RecordBatch::column(idx) panics out of range
                raise ErrorCode(Status.NotSupported, f"column index {self.idx} out of range")
            return self.idx
        for i, f in enumerate(fields):
            if f.name == self.name:
                return i
        raise ErrorCode(Status.LogicalError, "ColumnExpr must has name or idx")

    def flatten(self, fields):
        return [node_column(self.resolve(fields))]

    def __repr__(self):
        return f"ColumnExpr(name={self.name!r}, idx={self.idx})"


class PhysicalLiteralExpr(PhysicalExpr):
    """src/physical_plan/expression/literal.rs:17-35."""

    def __init__(self, literal: ScalarValue):
        self.literal = literal

    @staticmethod
    def create(literal: ScalarValue) -> "PhysicalLiteralExpr":
        return PhysicalLiteralExpr(literal)

    def flatten(self, fields):
        return [node_literal(self.literal)]

    def __repr__(self):
        return f"Literal({self.literal})"


class PhysicalBinaryExpr(PhysicalExpr):
    """src/physical_plan/expression/binary.rs:91-156."""

    def __init__(self, left: PhysicalExpr, op: Operator, right: PhysicalExpr):
        self.left, self.op, self.right = left, Operator(op), right

    @staticmethod
    def create(left: PhysicalExpr, op: Operator, right: PhysicalExpr) -> "PhysicalBinaryExpr":
        return PhysicalBinaryExpr(left, op, right)

    def flatten(self, fields):
        return self.left.flatten(fields) + self.right.flatten(fields) + [node_binary(self.op)]

    def __repr__(self):
        return f"({self.left!r} {self.op.name} {self.right!r})"


# small conveniences for tests / bench (not part of the reference surface)
def col(i_or_name) -> ColumnExpr:
    return ColumnExpr.try_create(None, i_or_name) if isinstance(i_or_name, int) else ColumnExpr.try_create(i_or_name, None)


def lit_i64(v) -> PhysicalLiteralExpr:
    return PhysicalLiteralExpr.create(ScalarValue.Int64(v))


def lit_u64(v) -> PhysicalLiteralExpr:
    return PhysicalLiteralExpr.create(ScalarValue.UInt64(v))


def lit_f64(v) -> PhysicalLiteralExpr:
    return PhysicalLiteralExpr.create(ScalarValue.Float64(v))


def lit_bool(v) -> PhysicalLiteralExpr:
    return PhysicalLiteralExpr.create(ScalarValue.Boolean(v))


def lit_utf8(v) -> PhysicalLiteralExpr:
    return PhysicalLiteralExpr.create(ScalarValue.Utf8(v))


def binop(l, op, r) -> PhysicalBinaryExpr:
    return PhysicalBinaryExpr.create(l, op, r)
