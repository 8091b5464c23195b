"""Host-side Arrow-layout containers and the ctypes mirror of include/nqe.h.

These are plumbing: numpy-backed stand-ins for the arrow `Array` / `RecordBatch` /
`ScalarValue` types that the reference's operators exchange
(src/physical_plan/plan.rs:18, src/logical_plan/expression.rs:174-187), laid out
exactly as the C ABI expects (64-bit little-endian values, LSB-first bitmaps).
No compute happens here.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np


class DType(enum.IntEnum):
    """nqe_dtype (include/nqe.h) — Arrow types on the hot path (selection.rs:69-99)."""

    NULL = 0
    BOOLEAN = 1
    INT64 = 2
    UINT64 = 3
    FLOAT64 = 4
    UTF8 = 5


class Operator(enum.IntEnum):
    """enum Operator (src/logical_plan/expression.rs:335-362), same order."""

    Eq = 0
    NotEq = 1
    Lt = 2
    LtEq = 3
    Gt = 4
    GtEq = 5
    Plus = 6
    Minus = 7
    Multiply = 8
    Divide = 9
    Modulos = 10
    And = 11
    Or = 12


class AggregateFunc(enum.IntEnum):
    """enum AggregateFunc (src/logical_plan/expression.rs:491-502), same order."""

    Count = 0
    Sum = 1
    Min = 2
    Max = 3
    Avg = 4


class Status(enum.IntEnum):
    """nqe_status; 1..13 mirror enum ErrorCode (src/error.rs:13-40)."""

    OK = 0
    ArrowError = 1
    IoError = 2
    NoSuchField = 3
    ColumnNotExists = 4
    LogicalError = 5
    NoSuchTable = 6
    ParserError = 7
    IntervalError = 8
    PlanError = 9
    NoMatchFunction = 10
    NotSupported = 11
    NotImplemented = 12
    Others = 13
    HipError = 100
    RcclError = 101
    InvalidArgument = 102
    OutOfMemory = 103


class ErrorCode(Exception):
    """The reference's `ErrorCode` (src/error.rs:13-40) as a Python exception."""

    def __init__(self, status: int, message: str = ""):
        try:
            self.status = Status(status)
        except ValueError:
            self.status = Status.Others
        self.message = message
        super().__init__(f"{self.status.name}: {message}")


# ----------------------------------------------------------------------------- ctypes structs
class NqeColumn(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("location", C.c_int32),
        ("length", C.c_int64),
        ("null_count", C.c_int64),
        ("values", C.c_void_p),
        ("validity", C.c_void_p),
        ("data", C.c_void_p),
        ("data_length", C.c_int64),
    ]


class _NodeValue(C.Union):
    _fields_ = [("i64", C.c_int64), ("u64", C.c_uint64), ("f64", C.c_double), ("boolean", C.c_int64), ("utf8", C.c_void_p)]


class NqeExprNode(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("op", C.c_int32),
        ("column", C.c_int32),
        ("dtype", C.c_int32),
        ("is_null", C.c_int32),
        ("utf8_length", C.c_int32),
        ("value", _NodeValue),
    ]


class NqeAggregate(C.Structure):
    _fields_ = [("func", C.c_int32), ("column", C.c_int32)]


class NqeCsvOptions(C.Structure):
    """nqe_csv_options (CsvConfig, src/datasource/csv.rs:23-43)"""

    _fields_ = [("has_header", C.c_int32), ("delimiter", C.c_int32), ("max_read_records", C.c_int64), ("batch_size", C.c_int64)]


EXPR_COLUMN, EXPR_LITERAL, EXPR_BINARY = 0, 1, 2
HOST, DEVICE = 0, 1

_WORD_NP = {DType.INT64: np.int64, DType.UINT64: np.uint64, DType.FLOAT64: np.float64}


def bitmap_bytes(n: int) -> int:
    return (n + 7) // 8


def pack_bits(mask: np.ndarray) -> np.ndarray:
    """bool[n] -> LSB-first packed uint8[ceil(n/8)]"""
    return np.packbits(np.asarray(mask, dtype=bool), bitorder="little")


def unpack_bits(buf: np.ndarray, n: int) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=bool)
    return np.unpackbits(np.asarray(buf, dtype=np.uint8), count=n, bitorder="little").astype(bool)


# ----------------------------------------------------------------------------- ScalarValue
@dataclass(frozen=True)
class ScalarValue:
    """enum ScalarValue (src/logical_plan/expression.rs:174-187)."""

    dtype: DType
    value: object = None  # None = X(None)

    @staticmethod
    def Null() -> "ScalarValue":
        return ScalarValue(DType.NULL, None)

    @staticmethod
    def Boolean(v: Optional[bool]) -> "ScalarValue":
        return ScalarValue(DType.BOOLEAN, v)

    @staticmethod
    def Float64(v: Optional[float]) -> "ScalarValue":
        return ScalarValue(DType.FLOAT64, v)

    @staticmethod
    def Int64(v: Optional[int]) -> "ScalarValue":
        return ScalarValue(DType.INT64, v)

    @staticmethod
    def UInt64(v: Optional[int]) -> "ScalarValue":
        return ScalarValue(DType.UINT64, v)

    @staticmethod
    def Utf8(v: Optional[str]) -> "ScalarValue":
        return ScalarValue(DType.UTF8, v)


# ----------------------------------------------------------------------------- Array
class Column:
    """One host Arrow array (a RecordBatch column).

    values: int64/uint64/float64[n] for the 8-byte types, packed uint8 for Boolean,
    int32 offsets[n+1] for Utf8; validity: packed uint8 or None (= no nulls).
    """

    __slots__ = ("dtype", "length", "values", "validity", "data")

    def __init__(self, dtype: DType, length: int, values: np.ndarray, validity: Optional[np.ndarray] = None,
                 data: Optional[np.ndarray] = None):
        self.dtype = DType(dtype)
        self.length = int(length)
        self.values = values
        self.validity = validity
        self.data = data

    # -- constructors
    @staticmethod
    def from_numpy(arr: np.ndarray, mask: Optional[np.ndarray] = None) -> "Column":
        """mask: bool[n], True = valid."""
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.bool_:
            col = Column(DType.BOOLEAN, arr.size, pack_bits(arr))
        elif arr.dtype == np.int64:
            col = Column(DType.INT64, arr.size, arr)
        elif arr.dtype == np.uint64:
            col = Column(DType.UINT64, arr.size, arr)
        elif arr.dtype == np.float64:
            col = Column(DType.FLOAT64, arr.size, arr)
        else:
            raise TypeError(f"unsupported numpy dtype {arr.dtype}")
        if mask is not None:
            col.validity = pack_bits(mask)
        return col

    @staticmethod
    def from_list(items: Sequence, dtype: DType) -> "Column":
        """Python list with None for nulls (like `Int64Array::from(vec![Some(1), None])`)."""
        dtype = DType(dtype)
        n = len(items)
        mask = np.array([x is not None for x in items], dtype=bool)
        has_null = not mask.all() if n else False
        if dtype == DType.UTF8:
            offs = np.zeros(n + 1, dtype=np.int32)
            chunks = []
            for i, s in enumerate(items):
                b = (s or "").encode("utf-8") if s is not None else b""
                chunks.append(b)
                offs[i + 1] = offs[i] + len(b)
            data = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()
            col = Column(dtype, n, offs, None, data)
        elif dtype == DType.BOOLEAN:
            col = Column(dtype, n, pack_bits(np.array([bool(x) if x is not None else False for x in items], dtype=bool)))
        else:
            npdt = _WORD_NP[dtype]
            col = Column(dtype, n, np.array([x if x is not None else 0 for x in items], dtype=npdt))
        if has_null:
            col.validity = pack_bits(mask)
        return col

    # -- accessors
    def valid_mask(self) -> np.ndarray:
        if self.validity is None:
            return np.ones(self.length, dtype=bool)
        return unpack_bits(self.validity, self.length)

    @property
    def null_count(self) -> int:
        return 0 if self.validity is None else int(self.length - self.valid_mask().sum())

    def to_numpy(self) -> np.ndarray:
        """Values as a numpy array (bool[n] for Boolean); null slots hold unspecified values."""
        if self.dtype == DType.BOOLEAN:
            return unpack_bits(self.values, self.length)
        if self.dtype == DType.UTF8:
            raise TypeError("Utf8 has no numpy value view; use to_list()")
        return self.values[: self.length]

    def to_list(self) -> list:
        m = self.valid_mask()
        if self.dtype == DType.UTF8:
            raw = bytes(self.data.tobytes()) if self.data is not None else b""
            out = []
            for i in range(self.length):
                out.append(raw[self.values[i]: self.values[i + 1]].decode("utf-8") if m[i] else None)
            return out
        vals = self.to_numpy().tolist()
        return [v if ok else None for v, ok in zip(vals, m.tolist())]

    def as_nqe(self, keepalive: list) -> NqeColumn:
        c = NqeColumn()
        c.dtype = int(self.dtype)
        c.location = HOST
        c.length = self.length
        c.null_count = -1 if self.validity is not None else 0
        vals = np.ascontiguousarray(self.values)
        keepalive.append(vals)
        c.values = vals.ctypes.data if vals.size else None
        if self.validity is not None:
            v = np.ascontiguousarray(self.validity, dtype=np.uint8)
            keepalive.append(v)
            c.validity = v.ctypes.data if v.size else None
        if self.dtype == DType.UTF8:
            d = np.ascontiguousarray(self.data if self.data is not None else np.zeros(0, np.uint8), dtype=np.uint8)
            keepalive.append(d)
            c.data = d.ctypes.data if d.size else None
            c.data_length = d.size
        return c

    @staticmethod
    def from_nqe_host(c: NqeColumn) -> "Column":
        """Copy out of an NQE_HOST column descriptor (oracle results)."""
        dt = DType(c.dtype)
        n = int(c.length)

        def grab(ptr, nbytes, npdt):
            if not ptr or nbytes == 0:
                return np.zeros(0, dtype=npdt)
            buf = (C.c_uint8 * nbytes).from_address(ptr)
            return np.frombuffer(buf, dtype=npdt).copy()

        validity = grab(c.validity, bitmap_bytes(n), np.uint8) if c.validity else None
        if dt == DType.BOOLEAN:
            return Column(dt, n, grab(c.values, bitmap_bytes(n), np.uint8), validity)
        if dt == DType.UTF8:
            return Column(dt, n, grab(c.values, (n + 1) * 4, np.int32), validity, grab(c.data, int(c.data_length), np.uint8))
        if dt == DType.NULL:
            return Column(dt, n, np.zeros(0, np.uint8), validity)
        return Column(dt, n, grab(c.values, n * 8, _WORD_NP[dt]), validity)

    def __repr__(self) -> str:
        return f"Column({self.dtype.name}, {self.to_list() if self.length <= 16 else f'len={self.length}'})"


@dataclass
class Field:
    """NaiveField (src/logical_plan/schema.rs): qualifier is ignored at physical planning (Q12)."""

    name: str
    dtype: DType
    nullable: bool = False


class RecordBatch:
    """Host RecordBatch: named columns of equal length."""

    def __init__(self, fields: Sequence[Field], columns: Sequence[Column]):
        if len(fields) != len(columns):
            raise ErrorCode(Status.ArrowError, "number of columns must match number of fields")
        n = columns[0].length if columns else 0
        for c in columns:
            if c.length != n:
                raise ErrorCode(Status.ArrowError, "all columns in a record batch must have the same length")
        self.fields: List[Field] = list(fields)
        self.columns: List[Column] = list(columns)

    @property
    def num_rows(self) -> int:
        return self.columns[0].length if self.columns else 0

    @property
    def num_columns(self) -> int:
        return len(self.columns)

    def column(self, i: int) -> Column:
        return self.columns[i]

    @staticmethod
    def from_pydict(d: dict) -> "RecordBatch":
        """{name: numpy array | (list, DType)}"""
        fields, cols = [], []
        for name, v in d.items():
            if isinstance(v, Column):
                col = v
            elif isinstance(v, tuple):
                col = Column.from_list(v[0], v[1])
            else:
                col = Column.from_numpy(np.asarray(v))
            fields.append(Field(name, col.dtype, col.validity is not None))
            cols.append(col)
        return RecordBatch(fields, cols)

    def to_pydict(self) -> dict:
        return {f.name: c.to_list() for f, c in zip(self.fields, self.columns)}

    # -- Arrow interchange (pyarrow is optional plumbing; the C ABI takes the same buffers directly)
    @staticmethod
    def from_arrow(batch) -> "RecordBatch":
        """pyarrow.RecordBatch / Table (single chunk) → host RecordBatch, zero interpretation: the Arrow buffers
        (validity bitmap, values / offsets+data) are exactly the layout of include/nqe.h."""
        import pyarrow as pa

        if isinstance(batch, pa.Table):
            batch = batch.combine_chunks().to_batches()[0] if batch.num_rows else pa.RecordBatch.from_pylist([], schema=batch.schema)
        amap = {pa.int64(): DType.INT64, pa.uint64(): DType.UINT64, pa.float64(): DType.FLOAT64, pa.bool_(): DType.BOOLEAN, pa.string(): DType.UTF8}
        fields, cols = [], []
        for name, arr in zip(batch.schema.names, batch.columns):
            if arr.type not in amap:
                raise ErrorCode(Status.NotSupported, f"Arrow type {arr.type} is not on the hot path")
            dt = amap[arr.type]
            n = len(arr)
            if arr.offset != 0:  # the ABI takes offset-free arrays
                arr = pa.concat_arrays([arr])
            bufs = arr.buffers()
            validity = None
            if arr.null_count and bufs[0] is not None:
                validity = np.frombuffer(bufs[0], dtype=np.uint8)[: bitmap_bytes(n)].copy()
            if dt == DType.BOOLEAN:
                col = Column(dt, n, np.frombuffer(bufs[1], dtype=np.uint8)[: bitmap_bytes(n)].copy() if n else np.zeros(0, np.uint8), validity)
            elif dt == DType.UTF8:
                offs = np.frombuffer(bufs[1], dtype=np.int32)[: n + 1].copy() if bufs[1] is not None else np.zeros(1, np.int32)
                data = np.frombuffer(bufs[2], dtype=np.uint8).copy() if bufs[2] is not None else np.zeros(0, np.uint8)
                col = Column(dt, n, offs, validity, data)
            else:
                col = Column(dt, n, np.frombuffer(bufs[1], dtype=_WORD_NP[dt])[:n].copy() if n else np.zeros(0, _WORD_NP[dt]), validity)
            fields.append(Field(name, dt, arr.null_count > 0))
            cols.append(col)
        return RecordBatch(fields, cols)

    def to_arrow(self):
        """host RecordBatch → pyarrow.RecordBatch built from the same buffers."""
        import pyarrow as pa

        tmap = {DType.INT64: pa.int64(), DType.UINT64: pa.uint64(), DType.FLOAT64: pa.float64(), DType.BOOLEAN: pa.bool_(), DType.UTF8: pa.string()}
        arrays = []
        for c in self.columns:
            vb = pa.py_buffer(np.ascontiguousarray(c.validity).tobytes()) if c.validity is not None else None
            if c.dtype == DType.UTF8:
                data = c.data if c.data is not None else np.zeros(0, np.uint8)
                bufs = [vb, pa.py_buffer(np.ascontiguousarray(c.values).tobytes()), pa.py_buffer(np.ascontiguousarray(data).tobytes())]
            else:
                bufs = [vb, pa.py_buffer(np.ascontiguousarray(c.values).tobytes())]
            arrays.append(pa.Array.from_buffers(tmap[c.dtype], c.length, bufs, null_count=c.null_count if c.validity is not None else 0))
        return pa.RecordBatch.from_arrays(arrays, names=[f.name for f in self.fields])


# ----------------------------------------------------------------------------- expression encoding
def node_column(idx: int) -> NqeExprNode:
    n = NqeExprNode()
    n.kind = EXPR_COLUMN
    n.column = idx
    return n


def node_literal(s: ScalarValue) -> NqeExprNode:
    n = NqeExprNode()
    n.kind = EXPR_LITERAL
    n.dtype = int(s.dtype)
    n.is_null = 1 if s.value is None else 0
    if s.value is not None:
        if s.dtype == DType.INT64:
            n.value.i64 = int(s.value)
        elif s.dtype == DType.UINT64:
            n.value.u64 = int(s.value)
        elif s.dtype == DType.FLOAT64:
            n.value.f64 = float(s.value)
        elif s.dtype == DType.BOOLEAN:
            n.value.boolean = 1 if s.value else 0
        elif s.dtype == DType.UTF8:
            raw = s.value.encode() if isinstance(s.value, str) else bytes(s.value)
            buf = C.create_string_buffer(raw, len(raw) + 1)
            n._keep = buf  # borrowed by the callee for the duration of the call
            n.value.utf8 = C.cast(buf, C.c_void_p).value
            n.utf8_length = len(raw)
        else:
            raise ErrorCode(Status.NotSupported, "literal type")
    return n


def node_binary(op: Operator) -> NqeExprNode:
    n = NqeExprNode()
    n.kind = EXPR_BINARY
    n.op = int(op)
    return n


def nodes_array(nodes: Sequence[NqeExprNode]):
    arr = (NqeExprNode * max(1, len(nodes)))()
    arr._keep = [getattr(nd, "_keep", None) for nd in nodes]  # Utf8 literal bytes referenced by the copied structs
    for i, nd in enumerate(nodes):
        arr[i] = nd
    return arr


def read_csv(path: str, has_header: bool = True, delimiter: str = ",", max_read_records: int = 3) -> RecordBatch:
    """Minimal stand-in for CsvTable::try_create (src/datasource/csv.rs:53-86): schema inferred
    from the first `max_read_records` rows (Int64, else Float64, else Utf8, CsvConfig default 3,
    csv.rs:33-43), non-nullable fields, one batch.  CSV parsing itself is out of scope for
    the GPU path; this only feeds the fixtures to the operators."""
    import csv

    with open(path, newline="") as f:
        rows = list(csv.reader(f, delimiter=delimiter))
    header = rows[0] if has_header else [f"column_{i + 1}" for i in range(len(rows[0]))]
    body = rows[1:] if has_header else rows
    fields, cols = [], []
    for j, name in enumerate(header):
        sample = [r[j] for r in body[:max_read_records]]

        def all_match(fn):
            try:
                for s in sample:
                    fn(s)
                return True
            except ValueError:
                return False

        if all_match(int):
            col = Column.from_numpy(np.array([int(r[j]) for r in body], dtype=np.int64))
        elif all_match(float):
            col = Column.from_numpy(np.array([float(r[j]) for r in body], dtype=np.float64))
        else:
            col = Column.from_list([r[j] for r in body], DType.UTF8)
        fields.append(Field(name, col.dtype, False))
        cols.append(col)
    return RecordBatch(fields, cols)
