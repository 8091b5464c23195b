//! UNCOMPILED SOURCE — the build image has no Rust toolchain.  `tools/check_rust_shim.py` (run by
//! tests/test_rust_shim.py) keeps it honest without one: every `extern "C"` declaration below is compared with
//! include/nqe.h (name, argument count, argument classes), every `Type::function(` / `.method(` this file calls on its own
//! types must be defined here, and brackets must balance.
//!
//! A module to drop into the reference crate as `src/physical_plan/gpu.rs`.  It implements the reference's own operator trait
//! (`PhysicalPlan`, `physical_plan/plan.rs:14-21`) and table trait (`TableSource`, `datasource/mod.rs:17-27`) on top of the
//! C ABI of include/nqe.h, written against arrow-rs 13 (`Cargo.lock`) and the crate's types as of the surveyed commit.
//!
//! Data stays in HBM between operators: a table registered through `GpuMemTable` is uploaded ONCE, at registration
//! (`MemTable::try_create`, `datasource/memory.rs:21-29`); every `Gpu*` operator hands device-resident batches (`GpuBatch`) to
//! its parent through `GpuExec::execute_device`, and only the root's `PhysicalPlan::execute()` — the call `NaiveDB::run_sql`
//! makes (`db.rs:36`) — downloads.  A `Gpu*` operator above a CPU operator uploads that child's `RecordBatch`es; a CPU
//! operator above a `Gpu*` operator calls its `execute()` and gets host batches: the two kinds mix freely.
//!
//! The edits to the reference crate this module needs (the whole patch; nothing else changes):
//!   1. `src/physical_plan/mod.rs`:      `mod gpu; pub use gpu::*;`
//!   2. `src/physical_plan/plan.rs:14`:  two more methods on `trait PhysicalPlan`:
//!          `fn as_any(&self) -> &dyn std::any::Any;`                         (each operator: `fn as_any(&self) -> &dyn Any { self }`)
//!          `fn as_gpu(&self) -> Option<&dyn crate::physical_plan::GpuExec> { None }`
//!   3. `src/datasource/mod.rs:17`:      one more method on `trait TableSource`:
//!          `fn scan_device(&self, _projection: Option<Vec<usize>>) -> Option<Result<Vec<crate::physical_plan::GpuBatch>>> { None }`
//!   4. `pub(crate)` on the fields of `ScanPlan` (scan.rs:19-22), `SelectionPlan` (selection.rs:23-27), `ProjectionPlan`
//!      (projection.rs:18-23), `HashJoin::{left,right,on,schema}` (hash_join.rs:44-56), `PhysicalLimitPlan` (limit.rs:15-19),
//!      `PhysicalOffsetPlan` (offset.rs:15-19) and `PhysicalBinaryExpr` (expression/binary.rs:91-96).
//!   5. `src/physical_plan/aggregate/mod.rs:225`: one more method on `trait AggregateOperator`, one line in each of
//!      sum.rs / avg.rs / count.rs / max.rs / min.rs:
//!          `fn describe(&self) -> (AggregateFunc, ColumnExpr);`             (e.g. `(AggregateFunc::Sum, self.col_expr.clone())`)
//!   6. `src/db.rs:34-36`:               `let physical_plan = match gpu::context() { Some(ctx) => gpu::rewrite(&ctx, physical_plan)?, None => physical_plan };`
//!      `NaiveDB::create_gpu_memory_table` forwarding to `gpu::add_gpu_memory_table` (below) next to `create_csv_table` (db.rs:39-46), and
//!      `Catalog::add_table_source(&mut self, table: &str, source: TableRef)` next to `add_memory_table` (catalog.rs:40-49).
//!   7. `build.rs`:                      `println!("cargo:rustc-link-lib=dylib=nqe_hip");` + the search path of libnqe_hip.so.
//!
//! The tested callers of the same ABI are `naive_query_engine_amd/capi.py` (ctypes) with `physical_plan.py` / `rewrite.py`, and
//! `naive_query_engine_amd/host/naive_db.hpp` (C++), which mirror these classes one to one.
#![allow(dead_code)]
use std::any::Any;
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};
use std::sync::{Arc, OnceLock};

use arrow::array::{make_array, Array, ArrayData, ArrayRef};
use arrow::buffer::Buffer;
use arrow::datatypes::{DataType, Field, Schema};
use arrow::record_batch::RecordBatch;

use crate::catalog::Catalog;
use crate::datasource::{TableRef, TableSource};
use crate::error::{ErrorCode, Result};
use crate::logical_plan::expression::{AggregateFunc, Column, ScalarValue};
use crate::logical_plan::schema::NaiveSchema;
use crate::physical_plan::{
    ColumnExpr, HashJoin, PhysicalAggregatePlan, PhysicalBinaryExpr, PhysicalExprRef, PhysicalLimitPlan, PhysicalLiteralExpr, PhysicalOffsetPlan,
    PhysicalPlan, PhysicalPlanRef, ProjectionPlan, ScanPlan, SelectionPlan,
};

// ------------------------------------------------------------------ FFI (1:1 with include/nqe.h)
#[repr(C)]
pub struct NqeColumn {
    pub dtype: i32, pub location: i32, pub length: i64, pub null_count: i64,
    pub values: *const c_void, pub validity: *const u8, pub data: *const c_void, pub data_length: i64,
}
#[repr(C)] #[derive(Clone, Copy)]
pub union NqeValue { pub i64_: i64, pub u64_: u64, pub f64_: f64, pub boolean: i64, pub utf8: *const c_char }
#[repr(C)] #[derive(Clone, Copy)]
pub struct NqeExprNode { pub kind: i32, pub op: i32, pub column: i32, pub dtype: i32, pub is_null: i32, pub utf8_length: i32, pub value: NqeValue }
#[repr(C)] #[derive(Clone, Copy)]
pub struct NqeAggregate { pub func: i32, pub column: i32 }
pub enum NqeCtx {}
pub enum NqeTable {}
pub enum NqeJoinTable {}
pub enum NqeComm {}

const NQE_BOOLEAN: i32 = 1; const NQE_INT64: i32 = 2; const NQE_UINT64: i32 = 3; const NQE_FLOAT64: i32 = 4; const NQE_UTF8: i32 = 5;

extern "C" {
    fn nqe_ctx_create(device: i32, stream: *mut c_void, out: *mut *mut NqeCtx) -> i32;
    fn nqe_ctx_destroy(ctx: *mut NqeCtx) -> i32;
    fn nqe_last_error(ctx: *const NqeCtx) -> *const c_char;
    fn nqe_table_create(ctx: *mut NqeCtx, columns: *const NqeColumn, num_columns: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_release(table: *mut NqeTable) -> i32;
    fn nqe_table_num_rows(table: *const NqeTable) -> i64;
    fn nqe_table_num_columns(table: *const NqeTable) -> i32;
    fn nqe_table_column(table: *const NqeTable, i: i32, out: *mut NqeColumn) -> i32;
    fn nqe_table_download_column(table: *const NqeTable, i: i32, values_out: *mut c_void, validity_out: *mut u8, data_out: *mut c_void) -> i32;
    fn nqe_table_project(ctx: *mut NqeCtx, input: *const NqeTable, indices: *const i32, n: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_slice(ctx: *mut NqeCtx, input: *const NqeTable, offset: i64, length: i64, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_concat(ctx: *mut NqeCtx, tables: *const *const NqeTable, n: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_expr_evaluate(ctx: *mut NqeCtx, input: *const NqeTable, nodes: *const NqeExprNode, num_nodes: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_filter(ctx: *mut NqeCtx, input: *const NqeTable, pred_table: *const NqeTable, pred_column: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_selection_execute(ctx: *mut NqeCtx, input: *const NqeTable, pred: *const NqeExprNode, pred_nodes: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_projection_execute(ctx: *mut NqeCtx, input: *const NqeTable, nodes: *const NqeExprNode, expr_offsets: *const i32, num_exprs: i32,
                              out: *mut *mut NqeTable) -> i32;
    fn nqe_selection_projection_execute(ctx: *mut NqeCtx, input: *const NqeTable, pred: *const NqeExprNode, pred_nodes: i32,
                                        nodes: *const NqeExprNode, expr_offsets: *const i32, num_exprs: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_aggregate_execute(ctx: *mut NqeCtx, input: *const NqeTable, pred: *const NqeExprNode, pred_nodes: i32, group: *const NqeExprNode,
                             group_nodes: i32, aggs: *const NqeAggregate, num_aggs: i32, out: *mut *mut NqeTable, keys_out: *mut *mut NqeTable) -> i32;
    fn nqe_hash_join_build(ctx: *mut NqeCtx, left: *const NqeTable, left_key: i32, out: *mut *mut NqeJoinTable) -> i32;
    fn nqe_hash_join_probe(ctx: *mut NqeCtx, build: *const NqeJoinTable, right: *const NqeTable, right_key: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_join_table_release(jt: *mut NqeJoinTable) -> i32;
}

// ------------------------------------------------------------------ context, device tables, upload, download
#[derive(Debug)]
pub struct GpuCtx(*mut NqeCtx);
// one host thread at a time (nqe.h conventions); the reference is single-threaded
unsafe impl Send for GpuCtx {}
unsafe impl Sync for GpuCtx {}
impl Drop for GpuCtx { fn drop(&mut self) { unsafe { nqe_ctx_destroy(self.0); } } }

/// owned device table handle (nqe_table)
#[derive(Debug)]
pub struct GpuTable(*mut NqeTable);
unsafe impl Send for GpuTable {}
unsafe impl Sync for GpuTable {}
impl Drop for GpuTable { fn drop(&mut self) { unsafe { nqe_table_release(self.0); } } }
impl GpuTable {
    pub fn num_rows(&self) -> usize { unsafe { nqe_table_num_rows(self.0) as usize } }
    pub fn num_columns(&self) -> usize { unsafe { nqe_table_num_columns(self.0) as usize } }
}

/// A RecordBatch whose columns live in HBM: what the `Gpu*` operators pass to each other (physical_plan.py: DeviceRecordBatch).
#[derive(Debug, Clone)]
pub struct GpuBatch { pub table: Arc<GpuTable> }
impl GpuBatch {
    fn wrap(raw: *mut NqeTable) -> Self { GpuBatch { table: Arc::new(GpuTable(raw)) } }
    pub fn num_rows(&self) -> usize { self.table.num_rows() }
}

/// The process-wide device context `NaiveDB::run_sql` consults (patch item 6): `Some` when NQE_DEVICE names a device and the
/// library initialises, `None` otherwise — the CPU operators then run as before.
pub fn context() -> Option<Arc<GpuCtx>> {
    static CTX: OnceLock<Option<Arc<GpuCtx>>> = OnceLock::new();
    CTX.get_or_init(|| {
        let device = std::env::var("NQE_DEVICE").ok()?.parse::<i32>().ok()?;
        GpuCtx::create(device).ok()
    }).clone()
}

impl GpuCtx {
    pub fn create(device: i32) -> Result<Arc<Self>> {
        let mut p = std::ptr::null_mut();
        if unsafe { nqe_ctx_create(device, std::ptr::null_mut(), &mut p) } != 0 { return Err(ErrorCode::Others); }
        Ok(Arc::new(GpuCtx(p)))
    }
    /// nqe_status → ErrorCode: codes 1..13 are the enum's variants in declaration order (error.rs:13-40)
    fn check(&self, st: i32) -> Result<()> {
        if st == 0 { return Ok(()); }
        let msg = unsafe { CStr::from_ptr(nqe_last_error(self.0)) }.to_string_lossy().into_owned();
        Err(match st {
            1 => ErrorCode::ArrowError(arrow::error::ArrowError::ComputeError(msg)),
            3 => ErrorCode::NoSuchField, 4 => ErrorCode::ColumnNotExists(msg), 5 => ErrorCode::LogicalError(msg), 6 => ErrorCode::NoSuchTable(msg),
            8 => ErrorCode::IntervalError(msg), 9 => ErrorCode::PlanError(msg), 10 => ErrorCode::NoMatchFunction(msg),
            11 => ErrorCode::NotSupported(msg), 12 => ErrorCode::NotImplemented, _ => ErrorCode::Others,
        })
    }
    /// RecordBatch → device table: the Arrow buffers are handed over as they are (nqe_table_create copies them to HBM, so the
    /// result is library-owned memory: operator outputs may share it).  Sliced arrays go through the C Data Interface, which
    /// honours offsets.
    pub fn upload(&self, batch: &RecordBatch) -> Result<GpuBatch> {
        if batch.columns().iter().any(|a| a.data().offset() != 0) { return self.upload_arrow(batch); }
        let cols: Vec<NqeColumn> = batch.columns().iter().map(|a| {
            let d = a.data();
            let dtype = match d.data_type() {
                DataType::Boolean => NQE_BOOLEAN, DataType::Int64 => NQE_INT64, DataType::UInt64 => NQE_UINT64,
                DataType::Float64 => NQE_FLOAT64, DataType::Utf8 => NQE_UTF8, _ => -1, // → NQE_ERR_NOT_SUPPORTED (selection.rs:98 panics)
            };
            let utf8 = dtype == NQE_UTF8;
            NqeColumn {
                dtype, location: 0, length: d.len() as i64, null_count: d.null_count() as i64,
                values: d.buffers()[0].as_ptr() as *const c_void,
                validity: d.null_buffer().map_or(std::ptr::null(), |b| b.as_ptr()),
                data: if utf8 { d.buffers()[1].as_ptr() as *const c_void } else { std::ptr::null() },
                data_length: if utf8 { d.buffers()[1].len() as i64 } else { 0 },
            }
        }).collect();
        let mut t = std::ptr::null_mut();
        self.check(unsafe { nqe_table_create(self.0, cols.as_ptr(), cols.len() as i32, &mut t) })?;
        Ok(GpuBatch::wrap(t))
    }
    /// device table → RecordBatch with the given schema (column types come from the table itself)
    pub fn download(&self, b: &GpuBatch, schema: &NaiveSchema) -> Result<RecordBatch> {
        let t = &b.table;
        let mut arrays: Vec<ArrayRef> = vec![];
        for i in 0..t.num_columns() as i32 {
            let mut info: NqeColumn = unsafe { std::mem::zeroed() };
            self.check(unsafe { nqe_table_column(t.0, i, &mut info) })?;
            let n = info.length as usize;
            let (dt, vbytes) = match info.dtype {
                NQE_BOOLEAN => (DataType::Boolean, (n + 7) / 8), NQE_INT64 => (DataType::Int64, n * 8), NQE_UINT64 => (DataType::UInt64, n * 8),
                NQE_FLOAT64 => (DataType::Float64, n * 8), _ => (DataType::Utf8, (n + 1) * 4),
            };
            let mut values = vec![0u8; vbytes];
            let mut validity = if info.validity.is_null() { vec![] } else { vec![0u8; (n + 7) / 8] };
            let mut data = vec![0u8; info.data_length as usize];
            self.check(unsafe { nqe_table_download_column(t.0, i, values.as_mut_ptr() as *mut c_void,
                if validity.is_empty() { std::ptr::null_mut() } else { validity.as_mut_ptr() },
                if data.is_empty() { std::ptr::null_mut() } else { data.as_mut_ptr() as *mut c_void }) })?;
            let mut builder = ArrayData::builder(dt.clone()).len(n).add_buffer(Buffer::from(values));
            if dt == DataType::Utf8 { builder = builder.add_buffer(Buffer::from(data)); }
            if !validity.is_empty() { builder = builder.null_bit_buffer(Some(Buffer::from(validity))); }
            arrays.push(make_array(builder.build()?));
        }
        // field names from the plan's schema, types and nullability from what the device produced
        let fields: Vec<Field> = arrays.iter().enumerate().map(|(i, a)| {
            let name = schema.fields().get(i).map_or(format!("c{}", i), |f| f.name().to_string());
            Field::new(name.as_str(), a.data_type().clone(), true)
        }).collect();
        Ok(RecordBatch::try_new(Arc::new(Schema::new(fields)), arrays)?)
    }
    fn download_all(&self, batches: &[GpuBatch], schema: &NaiveSchema) -> Result<Vec<RecordBatch>> {
        batches.iter().map(|b| self.download(b, schema)).collect()
    }
    /// concat_batches (hash_join.rs:258-273) on the device; one batch is returned as it is
    fn concat(&self, batches: &[GpuBatch]) -> Result<GpuBatch> {
        if batches.len() == 1 { return Ok(batches[0].clone()); }
        let raw: Vec<*const NqeTable> = batches.iter().map(|b| b.table.0 as *const NqeTable).collect();
        let mut out = std::ptr::null_mut();
        self.check(unsafe { nqe_table_concat(self.0, raw.as_ptr(), raw.len() as i32, &mut out) })?;
        Ok(GpuBatch::wrap(out))
    }
    fn slice(&self, b: &GpuBatch, offset: usize, length: usize) -> Result<GpuBatch> {
        let mut out = std::ptr::null_mut();
        self.check(unsafe { nqe_table_slice(self.0, b.table.0, offset as i64, length as i64, &mut out) })?;
        Ok(GpuBatch::wrap(out))
    }
}

// ------------------------------------------------------------------ the device side of the operator interface
/// What a `Gpu*` operator offers its parent besides `PhysicalPlan::execute()`: its result, still in HBM.
pub trait GpuExec {
    fn execute_device(&self) -> Result<Vec<GpuBatch>>;
}
/// A child's result as device batches: resident ones from a `Gpu*` child, uploaded ones from a CPU child.
fn child_device(ctx: &GpuCtx, child: &PhysicalPlanRef) -> Result<Vec<GpuBatch>> {
    match child.as_gpu() {
        Some(g) => g.execute_device(),
        None => child.execute()?.iter().map(|b| ctx.upload(b)).collect(),
    }
}

// ------------------------------------------------------------------ tables: uploaded once, at registration
/// MemTable (datasource/memory.rs:14-46) whose batches live in HBM.  `scan` — what a CPU operator sees — still returns the host
/// batches it was created from; `scan_device` (patch item 3) is what `GpuScanPlan` calls.
#[derive(Debug)]
pub struct GpuMemTable { schema: NaiveSchema, host: Vec<RecordBatch>, device: Vec<GpuBatch>, ctx: Arc<GpuCtx> }
impl GpuMemTable {
    pub fn try_create(ctx: Arc<GpuCtx>, schema: NaiveSchema, batches: Vec<RecordBatch>) -> Result<TableRef> {
        let device = batches.iter().map(|b| ctx.upload(b)).collect::<Result<Vec<_>>>()?; // the one upload
        Ok(Arc::new(Self { schema, host: batches, device, ctx }))
    }
    fn project_device(&self, b: &GpuBatch, projection: &[usize]) -> Result<GpuBatch> {
        let idx: Vec<i32> = projection.iter().map(|&i| i as i32).collect();
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { nqe_table_project(self.ctx.0, b.table.0, idx.as_ptr(), idx.len() as i32, &mut out) })?; // zero-copy (memory.rs:33-38)
        Ok(GpuBatch::wrap(out))
    }
}
impl TableSource for GpuMemTable {
    fn schema(&self) -> &NaiveSchema { &self.schema }
    fn scan(&self, projection: Option<Vec<usize>>) -> Result<Vec<RecordBatch>> {
        match projection {
            Some(p) => self.host.iter().map(|b| Ok(b.project(p.as_ref())?)).collect(),
            None => Ok(self.host.clone()),
        }
    }
    fn source_name(&self) -> String { "GpuMemTable".into() }
    fn scan_device(&self, projection: Option<Vec<usize>>) -> Option<Result<Vec<GpuBatch>>> {
        Some(match projection {
            Some(p) => self.device.iter().map(|b| self.project_device(b, &p)).collect(),
            None => Ok(self.device.clone()), // Arc clones (memory.rs:41)
        })
    }
}
/// Catalog::add_memory_table (catalog.rs:40-49) for a device-resident table; `NaiveDB::create_gpu_memory_table` forwards here.
pub fn add_gpu_memory_table(catalog: &mut Catalog, ctx: &Arc<GpuCtx>, table: &str, schema: NaiveSchema, batches: Vec<RecordBatch>) -> Result<()> {
    let source = GpuMemTable::try_create(ctx.clone(), schema, batches)?;
    catalog.add_table_source(table, source); // patch item 6: `self.tables.insert(table.to_string(), source)` (the map is private, catalog.rs:23)
    Ok(())
}

/// ScanPlan (scan.rs:18-41) over a source that holds its batches in HBM
#[derive(Debug)]
pub struct GpuScanPlan { source: TableRef, projection: Option<Vec<usize>>, ctx: Arc<GpuCtx> }
impl GpuScanPlan {
    pub fn create(ctx: Arc<GpuCtx>, source: TableRef, projection: Option<Vec<usize>>) -> PhysicalPlanRef { Arc::new(Self { source, projection, ctx }) }
}
impl GpuExec for GpuScanPlan {
    fn execute_device(&self) -> Result<Vec<GpuBatch>> {
        match self.source.scan_device(self.projection.clone()) {
            Some(r) => r,
            None => self.source.scan(self.projection.clone())?.iter().map(|b| self.ctx.upload(b)).collect(),
        }
    }
}
impl PhysicalPlan for GpuScanPlan {
    fn schema(&self) -> &NaiveSchema { self.source.schema() }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> { self.source.scan(self.projection.clone()) } // the host batches: no round trip
    fn as_any(&self) -> &dyn Any { self }
    fn as_gpu(&self) -> Option<&dyn GpuExec> { Some(self) }
}

// ------------------------------------------------------------------ expressions → flat post-order nodes
fn zero_node() -> NqeExprNode { NqeExprNode { kind: 0, op: 0, column: 0, dtype: 0, is_null: 0, utf8_length: 0, value: NqeValue { i64_: 0 } } }

/// ColumnExpr → column index: prefer idx, else the FIRST field with that name (column.rs:39-57, quirk Q12)
fn resolve(c: &ColumnExpr, schema: &NaiveSchema) -> Result<i32> {
    match (c.idx, &c.name) {
        (Some(i), _) => Ok(i as i32),
        (None, Some(n)) => Ok(schema.index_of(n)? as i32),
        _ => Err(ErrorCode::LogicalError("ColumnExpr must has name or idx".to_string())),
    }
}

/// `keep` holds the bytes of Utf8 literals for the duration of the call (the ABI borrows them)
fn flatten(e: &PhysicalExprRef, schema: &NaiveSchema, out: &mut Vec<NqeExprNode>, keep: &mut Vec<Vec<u8>>) -> Result<()> {
    if let Some(c) = e.as_any().downcast_ref::<ColumnExpr>() {
        out.push(NqeExprNode { kind: 0, column: resolve(c, schema)?, ..zero_node() });
    } else if let Some(l) = e.as_any().downcast_ref::<PhysicalLiteralExpr>() {
        let mut n = NqeExprNode { kind: 1, ..zero_node() };
        match &l.literal {
            ScalarValue::Null => { n.dtype = 0; n.is_null = 1; }
            ScalarValue::Boolean(v) => { n.dtype = NQE_BOOLEAN; n.is_null = v.is_none() as i32; n.value.boolean = v.unwrap_or(false) as i64; }
            ScalarValue::Int64(v) => { n.dtype = NQE_INT64; n.is_null = v.is_none() as i32; n.value.i64_ = v.unwrap_or(0); }
            ScalarValue::UInt64(v) => { n.dtype = NQE_UINT64; n.is_null = v.is_none() as i32; n.value.u64_ = v.unwrap_or(0); }
            ScalarValue::Float64(v) => { n.dtype = NQE_FLOAT64; n.is_null = v.is_none() as i32; n.value.f64_ = v.unwrap_or(0.0); }
            ScalarValue::Utf8(v) => {
                n.dtype = NQE_UTF8; n.is_null = v.is_none() as i32;
                if let Some(s) = v {
                    keep.push(s.as_bytes().to_vec());
                    let b = keep.last().unwrap(); // (the inner Vec's heap block does not move when `keep` grows)
                    n.value.utf8 = b.as_ptr() as *const c_char;
                    n.utf8_length = b.len() as i32;
                }
            }
        }
        out.push(n);
    } else if let Some(b) = e.as_any().downcast_ref::<PhysicalBinaryExpr>() {
        flatten(&b.left, schema, out, keep)?;
        flatten(&b.right, schema, out, keep)?;
        out.push(NqeExprNode { kind: 2, op: b.op.clone() as i32, ..zero_node() }); // Operator is declared in nqe_operator's order (expression.rs:335-362)
    } else {
        return Err(ErrorCode::NotSupported("expression kind has no device implementation (cast / unary)".to_string()));
    }
    Ok(())
}
/// a projection list as the ABI takes it: the expressions back to back + their offsets
fn flatten_list(exprs: &[PhysicalExprRef], schema: &NaiveSchema, keep: &mut Vec<Vec<u8>>) -> Result<(Vec<NqeExprNode>, Vec<i32>)> {
    let (mut nodes, mut offs) = (vec![], vec![0i32]);
    for e in exprs {
        flatten(e, schema, &mut nodes, keep)?;
        offs.push(nodes.len() as i32);
    }
    Ok((nodes, offs))
}

// ------------------------------------------------------------------ operators
/// SelectionPlan (selection.rs:24-107); with `project` set: Projection∘Selection fused into one device pass (C2)
#[derive(Debug)]
pub struct GpuSelectionPlan { input: PhysicalPlanRef, expr: PhysicalExprRef, project: Option<(NaiveSchema, Vec<PhysicalExprRef>)>, ctx: Arc<GpuCtx> }
impl GpuSelectionPlan {
    pub fn create(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, expr: PhysicalExprRef) -> PhysicalPlanRef { Arc::new(Self { input, expr, project: None, ctx }) }
    pub fn create_fused(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, expr: PhysicalExprRef, schema: NaiveSchema, exprs: Vec<PhysicalExprRef>) -> PhysicalPlanRef {
        Arc::new(Self { input, expr, project: Some((schema, exprs)), ctx })
    }
    /// the plain selection over already-executed batches, quirk Q3 included: the predicate is evaluated on batch 0 ONLY and
    /// zipped against every batch (selection.rs:60) — nqe_expr_evaluate once, nqe_filter per batch
    fn select(&self, input: &[GpuBatch]) -> Result<Vec<GpuBatch>> {
        if input.is_empty() { return Err(ErrorCode::NotSupported("SelectionPlan over no batches (selection.rs:60 indexes input[0])".to_string())); }
        let (mut pred, mut keep) = (vec![], vec![]);
        flatten(&self.expr, self.input.schema(), &mut pred, &mut keep)?;
        let mut out = std::ptr::null_mut();
        if input.len() == 1 {
            self.ctx.check(unsafe { nqe_selection_execute(self.ctx.0, input[0].table.0, pred.as_ptr(), pred.len() as i32, &mut out) })?;
            return Ok(vec![GpuBatch::wrap(out)]);
        }
        self.ctx.check(unsafe { nqe_expr_evaluate(self.ctx.0, input[0].table.0, pred.as_ptr(), pred.len() as i32, &mut out) })?;
        let mask = GpuBatch::wrap(out);
        input.iter().map(|b| {
            let mut o = std::ptr::null_mut();
            self.ctx.check(unsafe { nqe_filter(self.ctx.0, b.table.0, mask.table.0, 0, &mut o) })?;
            Ok(GpuBatch::wrap(o))
        }).collect()
    }
}
impl GpuExec for GpuSelectionPlan {
    fn execute_device(&self) -> Result<Vec<GpuBatch>> {
        let input = child_device(&self.ctx, &self.input)?;
        let in_schema = self.input.schema();
        match &self.project {
            // one batch: selection and projection in one pass over the referenced columns
            Some((schema, exprs)) if input.len() == 1 && !schema.fields().is_empty() => {
                let (mut pred, mut keep) = (vec![], vec![]);
                flatten(&self.expr, in_schema, &mut pred, &mut keep)?;
                let (nodes, offs) = flatten_list(exprs, in_schema, &mut keep)?;
                let mut out = std::ptr::null_mut();
                self.ctx.check(unsafe { nqe_selection_projection_execute(self.ctx.0, input[0].table.0, pred.as_ptr(), pred.len() as i32,
                                                                         nodes.as_ptr(), offs.as_ptr(), (offs.len() - 1) as i32, &mut out) })?;
                Ok(vec![GpuBatch::wrap(out)])
            }
            // several batches (Q3) or the pass-through projection (projection.rs:47-48): the plain operators, one after the other
            Some((schema, exprs)) => {
                let selected = self.select(&input)?;
                if schema.fields().is_empty() { return Ok(selected); }
                project_batches(&self.ctx, &selected, exprs, in_schema)
            }
            None => self.select(&input),
        }
    }
}
impl PhysicalPlan for GpuSelectionPlan {
    fn schema(&self) -> &NaiveSchema { self.project.as_ref().map_or(self.input.schema(), |p| &p.0) }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> { self.ctx.download_all(&self.execute_device()?, self.schema()) }
    fn as_any(&self) -> &dyn Any { self }
    fn as_gpu(&self) -> Option<&dyn GpuExec> { Some(self) }
}

/// ProjectionPlan::execute's loop (projection.rs:50-68): every expression over every batch
fn project_batches(ctx: &GpuCtx, input: &[GpuBatch], exprs: &[PhysicalExprRef], in_schema: &NaiveSchema) -> Result<Vec<GpuBatch>> {
    let mut keep = vec![];
    let (nodes, offs) = flatten_list(exprs, in_schema, &mut keep)?;
    input.iter().map(|b| {
        let mut out = std::ptr::null_mut();
        ctx.check(unsafe { nqe_projection_execute(ctx.0, b.table.0, nodes.as_ptr(), offs.as_ptr(), (offs.len() - 1) as i32, &mut out) })?;
        Ok(GpuBatch::wrap(out))
    }).collect()
}

/// ProjectionPlan (projection.rs:19-70)
#[derive(Debug)]
pub struct GpuProjectionPlan { input: PhysicalPlanRef, schema: NaiveSchema, exprs: Vec<PhysicalExprRef>, ctx: Arc<GpuCtx> }
impl GpuProjectionPlan {
    pub fn create(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, schema: NaiveSchema, exprs: Vec<PhysicalExprRef>) -> PhysicalPlanRef {
        Arc::new(Self { input, schema, exprs, ctx })
    }
}
impl GpuExec for GpuProjectionPlan {
    fn execute_device(&self) -> Result<Vec<GpuBatch>> {
        let input = child_device(&self.ctx, &self.input)?;
        if self.schema.fields().is_empty() { return Ok(input); } // projection.rs:47-48: pass-through above an aggregate
        project_batches(&self.ctx, &input, &self.exprs, self.input.schema())
    }
}
impl PhysicalPlan for GpuProjectionPlan {
    fn schema(&self) -> &NaiveSchema { &self.schema }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        // the pass-through case keeps the child's own schema (an aggregate's output fields)
        if self.schema.fields().is_empty() { return self.input.execute(); }
        self.ctx.download_all(&self.execute_device()?, &self.schema)
    }
    fn as_any(&self) -> &dyn Any { self }
    fn as_gpu(&self) -> Option<&dyn GpuExec> { Some(self) }
}

/// PhysicalAggregatePlan (aggregate/mod.rs:28-222); `filter` = a SelectionPlan predicate fused below it (the headline query)
#[derive(Debug)]
pub struct GpuAggregatePlan {
    input: PhysicalPlanRef, group_expr: Vec<PhysicalExprRef>, aggs: Vec<(AggregateFunc, ColumnExpr)>, filter: Option<PhysicalExprRef>,
    out_schema: NaiveSchema, comm: Option<Arc<GpuComm>>, ctx: Arc<GpuCtx>,
}
impl GpuAggregatePlan {
    /// from the operator the planner built (planner/mod.rs:95-170): its group expressions, the (function, column) pair of every
    /// aggregate (patch item 5: `AggregateOperator::describe`) and the output fields `data_field` names (`sum(x)`, …: sum.rs:58-84)
    /// `comm`: one process per GPU, each holding its row range of the input — the same query over the union of every rank's rows
    pub fn from_reference(ctx: Arc<GpuCtx>, a: &PhysicalAggregatePlan, input: PhysicalPlanRef, filter: Option<PhysicalExprRef>,
                          comm: Option<Arc<GpuComm>>) -> Result<PhysicalPlanRef> {
        let ops = a.aggr_ops.lock().unwrap();
        let aggs: Vec<(AggregateFunc, ColumnExpr)> = ops.iter().map(|op| op.describe()).collect();
        let fields = ops.iter().map(|op| op.data_field(&a.schema)).collect::<Result<Vec<_>>>()?;
        Ok(Arc::new(Self { input, group_expr: a.group_expr.clone(), aggs, filter, out_schema: NaiveSchema::new(fields), comm, ctx }))
    }
}
impl GpuExec for GpuAggregatePlan {
    fn execute_device(&self) -> Result<Vec<GpuBatch>> {
        let in_schema = self.input.schema();
        let mut batches = child_device(&self.ctx, &self.input)?;
        let mut filter = self.filter.clone();
        if batches.len() > 1 && filter.is_some() {
            // the fused predicate is per row; over several batches the reference's selection is not (Q3): run it unfused
            let sel = GpuSelectionPlan { input: self.input.clone(), expr: filter.take().unwrap(), project: None, ctx: self.ctx.clone() };
            batches = sel.select(&batches)?;
        }
        let single = self.ctx.concat(&batches)?; // concat_batches (aggregate/mod.rs:144); a single batch is used as it is
        let (mut pred, mut group, mut keep) = (vec![], vec![], vec![]);
        if let Some(f) = &filter { flatten(f, in_schema, &mut pred, &mut keep)?; }
        if let Some(g) = self.group_expr.first() { flatten(g, in_schema, &mut group, &mut keep)?; } // group_expr[0] only (:146)
        let aggs = self.aggs.iter().map(|(f, c)| Ok(NqeAggregate {
            func: f.clone() as i32, // AggregateFunc is declared in nqe_agg_func's order (expression.rs:491-502)
            column: resolve(c, in_schema)?,
        })).collect::<Result<Vec<_>>>()?;
        let mut out = std::ptr::null_mut();
        let st = unsafe {
            match &self.comm {
                Some(comm) => nqe_sharded_aggregate_execute(comm.raw, single.table.0, pred.as_ptr(), pred.len() as i32, group.as_ptr(), group.len() as i32,
                                                            aggs.as_ptr(), aggs.len() as i32, &mut out, std::ptr::null_mut()),
                None => nqe_aggregate_execute(self.ctx.0, single.table.0, pred.as_ptr(), pred.len() as i32, group.as_ptr(), group.len() as i32,
                                              aggs.as_ptr(), aggs.len() as i32, &mut out, std::ptr::null_mut()),
            }
        };
        self.ctx.check(st)?;
        Ok(vec![GpuBatch::wrap(out)]) // one row per group, sorted by key (the reference's order is the HashMap's)
    }
}
impl PhysicalPlan for GpuAggregatePlan {
    fn schema(&self) -> &NaiveSchema { self.input.schema() } // the INPUT schema, as the reference returns (aggregate/mod.rs:44, quirk Q8)
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> { self.ctx.download_all(&self.execute_device()?, &self.out_schema) }
    fn as_any(&self) -> &dyn Any { self }
    fn as_gpu(&self) -> Option<&dyn GpuExec> { Some(self) }
}

/// owned join table handle (nqe_join_table)
struct GpuJoinTable(*mut NqeJoinTable);
impl Drop for GpuJoinTable { fn drop(&mut self) { unsafe { nqe_join_table_release(self.0); } } }

/// HashJoin (hash_join.rs:44-289): LEFT = build side, RIGHT = probe side, on[0] only, inner only
#[derive(Debug)]
pub struct GpuHashJoin {
    left: PhysicalPlanRef, right: PhysicalPlanRef, on: Vec<(Column, Column)>, schema: NaiveSchema, comm: Option<(Arc<GpuComm>, bool)>, ctx: Arc<GpuCtx>,
}
impl GpuHashJoin {
    /// `comm` = (communicator, gather): build side replicated on every rank, `right` = this rank's row range; gather: the whole result
    /// on every rank (nqe_table_all_gather), else this rank's rows
    pub fn create(ctx: Arc<GpuCtx>, left: PhysicalPlanRef, right: PhysicalPlanRef, on: Vec<(Column, Column)>, schema: NaiveSchema,
                  comm: Option<(Arc<GpuComm>, bool)>) -> PhysicalPlanRef {
        Arc::new(Self { left, right, on, schema, comm, ctx })
    }
}
impl GpuExec for GpuHashJoin {
    fn execute_device(&self) -> Result<Vec<GpuBatch>> {
        let (lc, rc) = self.on.first().ok_or_else(|| ErrorCode::PlanError("Inner Join on Conditions can't not be empty".to_string()))?; // hash_join.rs:125-129
        let left = child_device(&self.ctx, &self.left)?;
        if left.is_empty() { return Err(ErrorCode::NotSupported("join with no left batches".to_string())); }
        let single = self.ctx.concat(&left)?; // concat_batches (:132)
        let lk = self.left.schema().index_of(&lc.name)? as i32; // by NAME, first match (:134-136)
        let rk = self.right.schema().index_of(&rc.name)? as i32;
        let mut jt = std::ptr::null_mut();
        self.ctx.check(unsafe { nqe_hash_join_build(self.ctx.0, single.table.0, lk, &mut jt) })?; // HashJoin::build (:124-166), once
        let jt = GpuJoinTable(jt);
        child_device(&self.ctx, &self.right)?.iter().map(|b| { // one output batch per probe batch (:177-250)
            let mut out = std::ptr::null_mut();
            let st = unsafe {
                match &self.comm {
                    Some((comm, gather)) => nqe_sharded_hash_join_probe(comm.raw, jt.0, b.table.0, rk, *gather as i32, &mut out),
                    None => nqe_hash_join_probe(self.ctx.0, jt.0, b.table.0, rk, &mut out),
                }
            };
            self.ctx.check(st)?;
            Ok(GpuBatch::wrap(out))
        }).collect()
    }
}
impl PhysicalPlan for GpuHashJoin {
    fn schema(&self) -> &NaiveSchema { &self.schema }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.left.clone(), self.right.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> { self.ctx.download_all(&self.execute_device()?, &self.schema) }
    fn as_any(&self) -> &dyn Any { self }
    fn as_gpu(&self) -> Option<&dyn GpuExec> { Some(self) }
}

/// PhysicalLimitPlan / PhysicalOffsetPlan (limit.rs:32-49, offset.rs:30-51) over device batches: whole batches are passed on,
/// a cut batch is nqe_table_slice — so a LIMIT above a device operator downloads `n` rows, not the operator's whole result
#[derive(Debug)]
pub struct GpuLimitPlan { input: PhysicalPlanRef, n: usize, offset: bool, ctx: Arc<GpuCtx> }
impl GpuLimitPlan {
    pub fn create_limit(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, n: usize) -> PhysicalPlanRef { Arc::new(Self { input, n, offset: false, ctx }) }
    pub fn create_offset(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, n: usize) -> PhysicalPlanRef { Arc::new(Self { input, n, offset: true, ctx }) }
}
impl GpuExec for GpuLimitPlan {
    fn execute_device(&self) -> Result<Vec<GpuBatch>> {
        let batches = child_device(&self.ctx, &self.input)?;
        let (mut n, mut ret) = (self.n, vec![]);
        for b in &batches {
            let rows = b.num_rows();
            if self.offset {
                if n == 0 { ret.push(b.clone()); continue; }
                if n >= rows { n -= rows; continue; }
                ret.push(self.ctx.slice(b, n, rows - n)?);
                n = 0;
            } else {
                if n == 0 { break; }
                if rows <= n { ret.push(b.clone()); n -= rows; } else { ret.push(self.ctx.slice(b, 0, n)?); n = 0; }
            }
        }
        Ok(ret)
    }
}
impl PhysicalPlan for GpuLimitPlan {
    fn schema(&self) -> &NaiveSchema { self.input.schema() }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> { self.ctx.download_all(&self.execute_device()?, self.schema()) }
    fn as_any(&self) -> &dyn Any { self }
    fn as_gpu(&self) -> Option<&dyn GpuExec> { Some(self) }
}

// ------------------------------------------------------------------ Arrow C Data Interface (nqe.h "Arrow C Data Interface")
// arrow-rs speaks the same interface (arrow::ffi): a RecordBatch goes over as ONE struct array, no per-buffer marshalling.
extern "C" {
    fn nqe_table_import_arrow(ctx: *mut NqeCtx, array: *mut arrow::ffi::FFI_ArrowArray, schema: *const arrow::ffi::FFI_ArrowSchema, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_export_arrow(table: *const NqeTable, names: *const *const c_char, out_array: *mut arrow::ffi::FFI_ArrowArray,
                              out_schema: *mut arrow::ffi::FFI_ArrowSchema) -> i32;
}
impl GpuCtx {
    /// RecordBatch → device table through the C Data Interface (the library copies the buffers to HBM and releases the array)
    pub fn upload_arrow(&self, batch: &RecordBatch) -> Result<GpuBatch> {
        let sa: arrow::array::StructArray = batch.clone().into();
        let (mut array, schema) = (arrow::ffi::FFI_ArrowArray::new(sa.data()), arrow::ffi::FFI_ArrowSchema::try_from(sa.data_type())?);
        let mut t = std::ptr::null_mut();
        self.check(unsafe { nqe_table_import_arrow(self.0, &mut array, &schema, &mut t) })?;
        Ok(GpuBatch::wrap(t))
    }
    /// device table → RecordBatch: the exported structs own host copies; arrow-rs calls their release callbacks on drop
    pub fn download_arrow(&self, b: &GpuBatch) -> Result<RecordBatch> {
        let (mut array, mut schema) = (arrow::ffi::FFI_ArrowArray::empty(), arrow::ffi::FFI_ArrowSchema::empty());
        self.check(unsafe { nqe_table_export_arrow(b.table.0, std::ptr::null(), &mut array, &mut schema) })?;
        let data = arrow::ffi::ArrowArray::new(array, schema).to_data()?;
        Ok(RecordBatch::from(&arrow::array::StructArray::from(data)))
    }
}

// ------------------------------------------------------------------ the rewrite pass (what rewrite.py / naive_db.hpp `rewrite` do)
// The planner keeps building the plain tree (planner/mod.rs:42-182); this pass (db.rs:34-36, between create_physical_plan and
// execute: patch item 6) substitutes the device operators bottom-up, fusing Projection∘Selection and Aggregate∘Selection.
// An operator it does not know (CrossJoin, NestedLoopJoin) is left as it is, children included: its `execute()` pulls host
// batches from whatever is below.
pub fn rewrite(ctx: &Arc<GpuCtx>, plan: PhysicalPlanRef) -> Result<PhysicalPlanRef> { rewrite_sharded(ctx, None, plan) }
/// the same pass for one rank of a multi-GPU deployment (`comm`: aggregates merge over the ranks, joins return this rank's rows)
pub fn rewrite_sharded(ctx: &Arc<GpuCtx>, comm: Option<&Arc<GpuComm>>, plan: PhysicalPlanRef) -> Result<PhysicalPlanRef> {
    let rewrite = |c: &Arc<GpuCtx>, p: PhysicalPlanRef| rewrite_sharded(c, comm, p);
    if plan.as_gpu().is_some() { return Ok(plan); }
    let any = plan.as_any();
    if let Some(s) = any.downcast_ref::<ScanPlan>() {
        return Ok(GpuScanPlan::create(ctx.clone(), s.source.clone(), s.projection.clone()));
    }
    if let Some(p) = any.downcast_ref::<ProjectionPlan>() {
        if let Some(sel) = p.input.as_any().downcast_ref::<SelectionPlan>() {
            if !p.schema.fields().is_empty() {
                return Ok(GpuSelectionPlan::create_fused(ctx.clone(), rewrite(ctx, sel.input.clone())?, sel.expr.clone(), p.schema.clone(), p.expr.clone()));
            }
        }
        return Ok(GpuProjectionPlan::create(ctx.clone(), rewrite(ctx, p.input.clone())?, p.schema.clone(), p.expr.clone()));
    }
    if let Some(a) = any.downcast_ref::<PhysicalAggregatePlan>() {
        let (input, filter) = match a.input.as_any().downcast_ref::<SelectionPlan>() {
            Some(sel) => (rewrite(ctx, sel.input.clone())?, Some(sel.expr.clone())),
            None => (rewrite(ctx, a.input.clone())?, None),
        };
        return GpuAggregatePlan::from_reference(ctx.clone(), a, input, filter, comm.cloned());
    }
    if let Some(sel) = any.downcast_ref::<SelectionPlan>() {
        return Ok(GpuSelectionPlan::create(ctx.clone(), rewrite(ctx, sel.input.clone())?, sel.expr.clone()));
    }
    if let Some(j) = any.downcast_ref::<HashJoin>() {
        return Ok(GpuHashJoin::create(ctx.clone(), rewrite(ctx, j.left.clone())?, rewrite(ctx, j.right.clone())?, j.on.clone(), j.schema.clone(),
                                      comm.map(|c| (c.clone(), false))));
    }
    if let Some(l) = any.downcast_ref::<PhysicalLimitPlan>() {
        return Ok(GpuLimitPlan::create_limit(ctx.clone(), rewrite(ctx, l.input.clone())?, l.n));
    }
    if let Some(o) = any.downcast_ref::<PhysicalOffsetPlan>() {
        return Ok(GpuLimitPlan::create_offset(ctx.clone(), rewrite(ctx, o.input.clone())?, o.n));
    }
    Ok(plan)
}

// ------------------------------------------------------------------ multi-GPU: one process per GPU, each holding its row range
// (nqe.h "sharded operators"; RCCL over xGMI inside the library, collectives on the context's stream)
extern "C" {
    fn nqe_comm_get_unique_id(id_out: *mut c_void /* 128 bytes */) -> i32;
    fn nqe_comm_create(ctx: *mut NqeCtx, unique_id: *const c_void, rank: i32, world: i32, out: *mut *mut NqeComm) -> i32;
    // (hosts without RCCL between their ranks: nqe_comm_create_custom takes all_gather / all_gather_v, nqe_comm_create_p2p takes
    // send / recv / group brackets with RCCL's matching rules — see nqe.h; a failed rank fails every rank, nobody blocks)
    fn nqe_comm_destroy(comm: *mut NqeComm) -> i32;
    fn nqe_sharded_aggregate_execute(comm: *mut NqeComm, input: *const NqeTable, pred: *const NqeExprNode, pred_nodes: i32, group: *const NqeExprNode,
                                     group_nodes: i32, aggs: *const NqeAggregate, num_aggs: i32, out: *mut *mut NqeTable, keys_out: *mut *mut NqeTable) -> i32;
    fn nqe_sharded_hash_join_probe(comm: *mut NqeComm, build: *const NqeJoinTable, right_local: *const NqeTable, right_key: i32, gather: i32,
                                   out: *mut *mut NqeTable) -> i32;
    fn nqe_sharded_selection_projection_execute(comm: *mut NqeComm, in_local: *const NqeTable, pred: *const NqeExprNode, pred_nodes: i32,
                                                nodes: *const NqeExprNode, expr_offsets: *const i32, num_exprs: i32, gather: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_all_gather(comm: *mut NqeComm, local: *const NqeTable, out: *mut *mut NqeTable) -> i32;
}
/// The host distributes the 128-byte id however it talks to its peers (MPI, a TCP store, a file): rank 0 draws it.
#[derive(Debug)]
pub struct GpuComm { raw: *mut NqeComm, pub rank: i32, pub world: i32 }
unsafe impl Send for GpuComm {}
unsafe impl Sync for GpuComm {}
impl GpuComm {
    pub fn unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        if unsafe { nqe_comm_get_unique_id(id.as_mut_ptr() as *mut c_void) } != 0 { return Err(ErrorCode::Others); }
        Ok(id)
    }
    pub fn create(ctx: &GpuCtx, id: &[u8; 128], rank: i32, world: i32) -> Result<Arc<Self>> {
        let mut c = std::ptr::null_mut();
        ctx.check(unsafe { nqe_comm_create(ctx.0, id.as_ptr() as *const c_void, rank, world, &mut c) })?;
        Ok(Arc::new(Self { raw: c, rank, world }))
    }
    /// every rank's result batch concatenated in rank order (= row order for row-range shards), on every rank
    pub fn all_gather(&self, ctx: &GpuCtx, local: &GpuBatch) -> Result<GpuBatch> {
        let mut out = std::ptr::null_mut();
        ctx.check(unsafe { nqe_table_all_gather(self.raw, local.table.0, &mut out) })?;
        Ok(GpuBatch::wrap(out))
    }
    /// fused Projection∘Selection over this rank's row range (gather: the whole result on every rank)
    pub fn selection_projection(&self, ctx: &GpuCtx, local: &GpuBatch, schema: &NaiveSchema, pred: &PhysicalExprRef, exprs: &[PhysicalExprRef], gather: bool) -> Result<GpuBatch> {
        let (mut p, mut keep) = (vec![], vec![]);
        flatten(pred, schema, &mut p, &mut keep)?;
        let (nodes, offs) = flatten_list(exprs, schema, &mut keep)?;
        let mut out = std::ptr::null_mut();
        ctx.check(unsafe { nqe_sharded_selection_projection_execute(self.raw, local.table.0, p.as_ptr(), p.len() as i32, nodes.as_ptr(), offs.as_ptr(),
                                                                    (offs.len() - 1) as i32, gather as i32, &mut out) })?;
        Ok(GpuBatch::wrap(out))
    }
}
impl Drop for GpuComm { fn drop(&mut self) { unsafe { nqe_comm_destroy(self.raw); } } }

// Quirks the device operators share with the mirrors (SURVEY §7 ledger): Q3 (predicate from batch 0) is reproduced above; Q8 (the
// aggregate reports its INPUT schema) is reproduced; Q9 / Q11 (un-grouped aggregate state and the join's hash table survive a
// second `execute()` of the SAME operator object) are not — `run_sql` builds a fresh tree per query (db.rs:24-37), so no caller of
// the reference can observe them; physical_plan.py reproduces them for completeness.
