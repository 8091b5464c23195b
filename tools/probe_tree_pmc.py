#!/usr/bin/env python
"""Small driver for PMC passes over the expression-tree kernel (run under rocprofv3 --pmc ... --kernel-trace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from naive_query_engine_amd import DType, Operator, capi
from naive_query_engine_amd.expression import binop, col, lit_i64


class F:
    def __init__(self, n):
        self.name = n


ctx = capi.Context(0)
dev = torch.device("cuda", 0)
n = 100_000_000
idt = torch.empty(n, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx.synth_fill(0, 0, 0, n, 1, 0, idt.data_ptr())
t = ctx.table_from_device([(DType.INT64, n, idt.data_ptr(), None)])
f = [F("id")]
cases = [binop(binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(5)),
         binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Plus, lit_i64(1)),
         __import__("functools").reduce(lambda a, _: binop(a, Operator.Plus, lit_i64(1)), range(8), col(0))]
for e in cases:
    for _ in range(2):
        r = ctx.expr_evaluate(t, e.flatten(f)); del r
ctx.synchronize()
