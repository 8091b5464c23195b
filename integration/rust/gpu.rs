//! UNCOMPILED SOURCE — the build image has no Rust toolchain.
//!
//! A module to drop into the reference crate as `src/physical_plan/gpu.rs` (plus `mod gpu;` in
//! `src/physical_plan/mod.rs`, `build.rs` linking `nqe_hip`, and `pub(crate)` on the three fields of
//! `PhysicalBinaryExpr`, `expression/binary.rs:91-96`).  It implements the reference's own operator trait
//! (`PhysicalPlan`, `physical_plan/plan.rs:14-21`) on top of the C ABI of `include/nqe.h`; written against
//! arrow-rs 13 (`Cargo.lock`) and the crate's types as of the surveyed commit.  The tested callers of the same ABI are
//! `naive_query_engine_amd/capi.py` (ctypes) and `naive_query_engine_amd/host/naive_db.hpp` (C++), which mirror these
//! classes one to one; this file shows what the Rust side of that boundary looks like.
#![allow(dead_code)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};
use std::sync::Arc;

use arrow::array::{make_array, Array, ArrayData, ArrayRef};
use arrow::buffer::Buffer;
use arrow::datatypes::DataType;
use arrow::record_batch::RecordBatch;

use crate::error::{ErrorCode, Result};
use crate::logical_plan::expression::{AggregateFunc, Column, Operator, ScalarValue};
use crate::logical_plan::schema::NaiveSchema;
use crate::physical_plan::{ColumnExpr, PhysicalBinaryExpr, PhysicalExprRef, PhysicalLiteralExpr, PhysicalPlan, PhysicalPlanRef};

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

const NQE_BOOLEAN: i32 = 1; const NQE_INT64: i32 = 2; const NQE_UINT64: i32 = 3; const NQE_FLOAT64: i32 = 4; const NQE_UTF8: i32 = 5;

extern "C" {
    fn nqe_ctx_create(device: i32, stream: *mut c_void, out: *mut *mut NqeCtx) -> i32;
    fn nqe_ctx_destroy(ctx: *mut NqeCtx) -> i32;
    fn nqe_last_error(ctx: *const NqeCtx) -> *const c_char;
    fn nqe_table_create(ctx: *mut NqeCtx, cols: *const NqeColumn, n: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_release(t: *mut NqeTable) -> i32;
    fn nqe_table_num_rows(t: *const NqeTable) -> i64;
    fn nqe_table_num_columns(t: *const NqeTable) -> i32;
    fn nqe_table_column(t: *const NqeTable, i: i32, out: *mut NqeColumn) -> i32;
    fn nqe_table_download_column(t: *const NqeTable, i: i32, values: *mut c_void, validity: *mut u8, data: *mut c_void) -> i32;
    fn nqe_selection_execute(ctx: *mut NqeCtx, t: *const NqeTable, pred: *const NqeExprNode, n: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_projection_execute(ctx: *mut NqeCtx, t: *const NqeTable, nodes: *const NqeExprNode, offs: *const i32, ne: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_selection_projection_execute(ctx: *mut NqeCtx, t: *const NqeTable, pred: *const NqeExprNode, pn: i32,
                                        nodes: *const NqeExprNode, offs: *const i32, ne: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_aggregate_execute(ctx: *mut NqeCtx, t: *const NqeTable, pred: *const NqeExprNode, pn: i32, group: *const NqeExprNode, gn: i32,
                             aggs: *const NqeAggregate, na: i32, out: *mut *mut NqeTable, keys_out: *mut *mut NqeTable) -> i32;
    fn nqe_hash_join_execute(ctx: *mut NqeCtx, l: *const NqeTable, r: *const NqeTable, lk: i32, rk: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_concat(ctx: *mut NqeCtx, tables: *const *const NqeTable, n: i32, out: *mut *mut NqeTable) -> i32;
}

// ------------------------------------------------------------------ context, upload, download
#[derive(Debug)]
pub struct GpuCtx(*mut NqeCtx);
unsafe impl Send for GpuCtx {} // one host thread at a time (nqe.h conventions); the reference is single-threaded

/// owned device table handle
pub struct GpuTable(*mut NqeTable);
impl Drop for GpuTable { fn drop(&mut self) { unsafe { nqe_table_release(self.0); } } }

impl GpuCtx {
    pub fn new(device: i32) -> Result<Arc<Self>> {
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
            5 => ErrorCode::LogicalError(msg), 8 => ErrorCode::IntervalError(msg), 9 => ErrorCode::PlanError(msg),
            11 => ErrorCode::NotSupported(msg), 12 => ErrorCode::NotImplemented, _ => ErrorCode::Others,
        })
    }
    /// RecordBatch → device table: the Arrow buffers are handed over as they are (nqe_table_create copies them to HBM)
    pub fn upload(&self, batch: &RecordBatch) -> Result<GpuTable> {
        let cols: Vec<NqeColumn> = batch.columns().iter().map(|a| {
            let d = a.data();
            assert_eq!(d.offset(), 0, "sliced arrays must be copied first");
            let dtype = match d.data_type() {
                DataType::Boolean => NQE_BOOLEAN, DataType::Int64 => NQE_INT64, DataType::UInt64 => NQE_UINT64,
                DataType::Float64 => NQE_FLOAT64, DataType::Utf8 => NQE_UTF8, _ => 0, // 0 → NQE_ERR_NOT_SUPPORTED (selection.rs:98)
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
        Ok(GpuTable(t))
    }
    /// device table → RecordBatch with the given schema (column types come from the table itself)
    pub fn download(&self, t: &GpuTable, schema: &NaiveSchema) -> Result<RecordBatch> {
        let n_cols = unsafe { nqe_table_num_columns(t.0) };
        let mut arrays: Vec<ArrayRef> = vec![];
        for i in 0..n_cols {
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
            let mut b = ArrayData::builder(dt.clone()).len(n).add_buffer(Buffer::from(values));
            if dt == DataType::Utf8 { b = b.add_buffer(Buffer::from(data)); }
            if !validity.is_empty() { b = b.null_bit_buffer(Some(Buffer::from(validity))); }
            arrays.push(make_array(b.build()?));
        }
        Ok(RecordBatch::try_new(Arc::new(schema.clone().into()), arrays)?)
    }
}
impl Drop for GpuCtx { fn drop(&mut self) { unsafe { nqe_ctx_destroy(self.0); } } }

// ------------------------------------------------------------------ expressions → flat post-order nodes
fn zero_node() -> NqeExprNode { NqeExprNode { kind: 0, op: 0, column: 0, dtype: 0, is_null: 0, utf8_length: 0, value: NqeValue { i64_: 0 } } }

/// `keep` holds the bytes of Utf8 literals for the duration of the call (the ABI borrows them)
fn flatten(e: &PhysicalExprRef, schema: &NaiveSchema, out: &mut Vec<NqeExprNode>, keep: &mut Vec<Vec<u8>>) -> Result<()> {
    if let Some(c) = e.as_any().downcast_ref::<ColumnExpr>() {
        // prefer idx, else the FIRST field with that name (column.rs:39-57, quirk Q12)
        let idx = match (c.idx, &c.name) { (Some(i), _) => i, (None, Some(n)) => schema.index_of(n)?, _ => return Err(ErrorCode::LogicalError("ColumnExpr must has name or idx".into())) };
        out.push(NqeExprNode { kind: 0, column: idx as i32, ..zero_node() });
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
                if let Some(s) = v { keep.push(s.as_bytes().to_vec()); let b = keep.last().unwrap(); n.value.utf8 = b.as_ptr() as *const c_char; n.utf8_length = b.len() as i32; }
            }
        }
        out.push(n);
    } else if let Some(b) = e.as_any().downcast_ref::<PhysicalBinaryExpr>() {
        flatten(&b.left, schema, out, keep)?;
        flatten(&b.right, schema, out, keep)?;
        out.push(NqeExprNode { kind: 2, op: b.op.clone() as i32, ..zero_node() }); // Operator is declared in nqe_operator's order
    } else {
        return Err(ErrorCode::NotSupported("expression kind has no device implementation (cast/unary)".into()));
    }
    Ok(())
}

// ------------------------------------------------------------------ operators
/// SelectionPlan (selection.rs:24-107); with `project` set: Projection∘Selection fused into one call
#[derive(Debug)]
pub struct GpuSelectionPlan { input: PhysicalPlanRef, expr: PhysicalExprRef, project: Option<(NaiveSchema, Vec<PhysicalExprRef>)>, ctx: Arc<GpuCtx> }

impl GpuSelectionPlan {
    pub fn create(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, expr: PhysicalExprRef) -> PhysicalPlanRef { Arc::new(Self { input, expr, project: None, ctx }) }
    pub fn create_fused(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, expr: PhysicalExprRef, schema: NaiveSchema, exprs: Vec<PhysicalExprRef>) -> PhysicalPlanRef {
        Arc::new(Self { input, expr, project: Some((schema, exprs)), ctx })
    }
}
impl PhysicalPlan for GpuSelectionPlan {
    fn schema(&self) -> &NaiveSchema { self.project.as_ref().map_or(self.input.schema(), |p| &p.0) }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        let input = self.input.execute()?;
        let in_schema = self.input.schema();
        let (mut pred, mut keep) = (vec![], vec![]);
        flatten(&self.expr, in_schema, &mut pred, &mut keep)?;
        let (mut nodes, mut offs) = (vec![], vec![0i32]);
        if let Some((_, exprs)) = &self.project {
            for e in exprs { flatten(e, in_schema, &mut nodes, &mut keep)?; offs.push(nodes.len() as i32); }
        }
        // NOTE: the reference evaluates the predicate on input[0] only and zips it against every batch (quirk Q3);
        // per-batch evaluation below is what a multi-batch-correct engine does — use nqe_filter with batch 0's
        // predicate column to reproduce the quirk bit for bit, as physical_plan.py::SelectionPlan does.
        input.iter().map(|batch| {
            let t = self.ctx.upload(batch)?;
            let mut out = std::ptr::null_mut();
            let st = unsafe {
                if self.project.is_some() {
                    nqe_selection_projection_execute(self.ctx.0, t.0, pred.as_ptr(), pred.len() as i32, nodes.as_ptr(), offs.as_ptr(), (offs.len() - 1) as i32, &mut out)
                } else {
                    nqe_selection_execute(self.ctx.0, t.0, pred.as_ptr(), pred.len() as i32, &mut out)
                }
            };
            self.ctx.check(st)?;
            self.ctx.download(&GpuTable(out), self.schema())
        }).collect()
    }
}

/// ProjectionPlan (projection.rs:19-70)
#[derive(Debug)]
pub struct GpuProjectionPlan { input: PhysicalPlanRef, schema: NaiveSchema, exprs: Vec<PhysicalExprRef>, ctx: Arc<GpuCtx> }
impl GpuProjectionPlan {
    pub fn create(ctx: Arc<GpuCtx>, input: PhysicalPlanRef, schema: NaiveSchema, exprs: Vec<PhysicalExprRef>) -> PhysicalPlanRef { Arc::new(Self { input, schema, exprs, ctx }) }
}
impl PhysicalPlan for GpuProjectionPlan {
    fn schema(&self) -> &NaiveSchema { &self.schema }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        let input = self.input.execute()?;
        if self.schema.fields().is_empty() { return Ok(input); } // projection.rs:47-48
        let (mut nodes, mut offs, mut keep) = (vec![], vec![0i32], vec![]);
        for e in &self.exprs { flatten(e, self.input.schema(), &mut nodes, &mut keep)?; offs.push(nodes.len() as i32); }
        input.iter().map(|batch| {
            let t = self.ctx.upload(batch)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe { nqe_projection_execute(self.ctx.0, t.0, nodes.as_ptr(), offs.as_ptr(), (offs.len() - 1) as i32, &mut out) })?;
            self.ctx.download(&GpuTable(out), &self.schema)
        }).collect()
    }
}

/// PhysicalAggregatePlan (aggregate/mod.rs:28-222); `filter` = a SelectionPlan predicate fused below it
#[derive(Debug)]
pub struct GpuAggregatePlan {
    input: PhysicalPlanRef, group_expr: Vec<PhysicalExprRef>, aggs: Vec<(AggregateFunc, ColumnExpr)>, filter: Option<PhysicalExprRef>,
    out_schema: NaiveSchema, ctx: Arc<GpuCtx>,
}
impl PhysicalPlan for GpuAggregatePlan {
    fn schema(&self) -> &NaiveSchema { self.input.schema() } // the INPUT schema, as the reference returns (aggregate/mod.rs:44, quirk Q8)
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.input.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        let batches = self.input.execute()?;
        let in_schema = self.input.schema();
        // concat_batches (aggregate/mod.rs:144): one device table
        let parts = batches.iter().map(|b| self.ctx.upload(b)).collect::<Result<Vec<_>>>()?;
        let raw: Vec<*const NqeTable> = parts.iter().map(|t| t.0 as *const NqeTable).collect();
        let mut single = std::ptr::null_mut();
        self.ctx.check(unsafe { nqe_table_concat(self.ctx.0, raw.as_ptr(), raw.len() as i32, &mut single) })?;
        let single = GpuTable(single);
        let (mut pred, mut group, mut keep) = (vec![], vec![], vec![]);
        if let Some(f) = &self.filter { flatten(f, in_schema, &mut pred, &mut keep)?; }
        if let Some(g) = self.group_expr.first() { flatten(g, in_schema, &mut group, &mut keep)?; } // group_expr[0] only (:146)
        let aggs = self.aggs.iter().map(|(f, c)| Ok(NqeAggregate {
            func: f.clone() as i32, // AggregateFunc is declared in nqe_agg_func's order (expression.rs:491-502)
            column: match (c.idx, &c.name) { (Some(i), _) => i as i32, (None, Some(n)) => in_schema.index_of(n)? as i32, _ => -1 },
        })).collect::<Result<Vec<_>>>()?;
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { nqe_aggregate_execute(self.ctx.0, single.0, pred.as_ptr(), pred.len() as i32, group.as_ptr(), group.len() as i32,
                                                      aggs.as_ptr(), aggs.len() as i32, &mut out, std::ptr::null_mut()) })?;
        Ok(vec![self.ctx.download(&GpuTable(out), &self.out_schema)?]) // one row per group, sorted by key (reference: HashMap order)
    }
}

/// HashJoin (hash_join.rs:44-289): LEFT = build side, RIGHT = probe side, on[0] only, inner only
#[derive(Debug)]
pub struct GpuHashJoin { left: PhysicalPlanRef, right: PhysicalPlanRef, on: Vec<(Column, Column)>, schema: NaiveSchema, ctx: Arc<GpuCtx> }
impl PhysicalPlan for GpuHashJoin {
    fn schema(&self) -> &NaiveSchema { &self.schema }
    fn children(&self) -> Result<Vec<PhysicalPlanRef>> { Ok(vec![self.left.clone(), self.right.clone()]) }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        let (lc, rc) = self.on.first().ok_or_else(|| ErrorCode::PlanError("Inner Join on Conditions can't not be empty".to_string()))?;
        let lb = self.left.execute()?;
        let parts = lb.iter().map(|b| self.ctx.upload(b)).collect::<Result<Vec<_>>>()?;
        let raw: Vec<*const NqeTable> = parts.iter().map(|t| t.0 as *const NqeTable).collect();
        let mut single = std::ptr::null_mut();
        self.ctx.check(unsafe { nqe_table_concat(self.ctx.0, raw.as_ptr(), raw.len() as i32, &mut single) })?; // concat_batches (:132)
        let single = GpuTable(single);
        let lk = self.left.schema().index_of(&lc.name)? as i32;  // by NAME, first match (:134-136)
        let rk = self.right.schema().index_of(&rc.name)? as i32;
        self.right.execute()?.iter().map(|batch| { // one output batch per probe batch (:177-250)
            let r = self.ctx.upload(batch)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe { nqe_hash_join_execute(self.ctx.0, single.0, r.0, lk, rk, &mut out) })?;
            self.ctx.download(&GpuTable(out), &self.schema)
        }).collect()
    }
}

// ------------------------------------------------------------------ Arrow C Data Interface (nqe.h "Arrow C Data Interface")
// arrow-rs speaks the same interface (arrow::ffi): a RecordBatch goes over as ONE struct array, no per-buffer marshalling.
extern "C" {
    fn nqe_table_import_arrow(ctx: *mut NqeCtx, array: *mut arrow::ffi::FFI_ArrowArray, schema: *const arrow::ffi::FFI_ArrowSchema, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_export_arrow(t: *const NqeTable, names: *const *const c_char, out_array: *mut arrow::ffi::FFI_ArrowArray,
                              out_schema: *mut arrow::ffi::FFI_ArrowSchema) -> i32;
}
impl GpuCtx {
    /// RecordBatch → device table through the C Data Interface (the library copies the buffers to HBM and releases the array)
    pub fn upload_arrow(&self, batch: &RecordBatch) -> Result<GpuTable> {
        let sa: arrow::array::StructArray = batch.clone().into();
        let (mut array, schema) = (arrow::ffi::FFI_ArrowArray::new(sa.data()), arrow::ffi::FFI_ArrowSchema::try_from(sa.data_type())?);
        let mut t = std::ptr::null_mut();
        self.check(unsafe { nqe_table_import_arrow(self.0, &mut array, &schema, &mut t) })?;
        Ok(GpuTable(t))
    }
    /// device table → RecordBatch: the exported structs own host copies; arrow-rs calls their release callbacks on drop
    pub fn download_arrow(&self, t: &GpuTable) -> Result<RecordBatch> {
        let (mut array, mut schema) = (arrow::ffi::FFI_ArrowArray::empty(), arrow::ffi::FFI_ArrowSchema::empty());
        self.check(unsafe { nqe_table_export_arrow(t.0, std::ptr::null(), &mut array, &mut schema) })?;
        let data = arrow::ffi::ArrowArray::new(array, schema).to_data()?;
        Ok(RecordBatch::from(&arrow::array::StructArray::from(data)))
    }
}

// ------------------------------------------------------------------ the rewrite pass (what rewrite.py / naive_db.hpp `rewrite` do)
// The planner keeps building the plain tree (planner/mod.rs:42-182); this pass substitutes the subtrees the device runs in one
// go.  It needs `pub(crate)` on SelectionPlan / ProjectionPlan / PhysicalAggregatePlan's fields and an `as_any` on PhysicalPlan
// (the trait has none today; the expression trait does: expression/mod.rs:25-29).
pub fn rewrite(ctx: &Arc<GpuCtx>, plan: PhysicalPlanRef) -> Result<PhysicalPlanRef> {
    use crate::physical_plan::{PhysicalAggregatePlan, ProjectionPlan, SelectionPlan};
    if let Some(p) = plan.as_any().downcast_ref::<ProjectionPlan>() {
        if let Some(sel) = p.input.as_any().downcast_ref::<SelectionPlan>() {
            if !p.schema.fields().is_empty() {
                return Ok(GpuSelectionPlan::create_fused(ctx.clone(), rewrite(ctx, sel.input.clone())?, sel.expr.clone(), p.schema.clone(), p.expr.clone()));
            }
        }
        return Ok(GpuProjectionPlan::create(ctx.clone(), rewrite(ctx, p.input.clone())?, p.schema.clone(), p.expr.clone()));
    }
    if let Some(a) = plan.as_any().downcast_ref::<PhysicalAggregatePlan>() {
        let (input, filter) = match a.input.as_any().downcast_ref::<SelectionPlan>() {
            Some(sel) => (rewrite(ctx, sel.input.clone())?, Some(sel.expr.clone())),
            None => (rewrite(ctx, a.input.clone())?, None),
        };
        return Ok(GpuAggregatePlan::from_reference(ctx.clone(), a, input, filter)); // copies group_expr / (func, column) pairs / data_field()s
    }
    if let Some(sel) = plan.as_any().downcast_ref::<SelectionPlan>() {
        return Ok(GpuSelectionPlan::create(ctx.clone(), rewrite(ctx, sel.input.clone())?, sel.expr.clone()));
    }
    // HashJoin → GpuHashJoin with rewritten children; Limit / Offset / Scan keep their operators (children rewritten)
    Ok(plan)
}

// ------------------------------------------------------------------ multi-GPU: one process per GPU, each holding its row range
// (nqe.h "sharded operators"; RCCL over xGMI inside the library, collectives on the context's stream)
pub enum NqeComm {}
pub enum NqeJoinTable {}
extern "C" {
    fn nqe_comm_get_unique_id(id_out: *mut u8 /* 128 bytes */) -> i32;
    fn nqe_comm_create(ctx: *mut NqeCtx, unique_id: *const u8, rank: i32, world: i32, out: *mut *mut NqeComm) -> i32;
    // (hosts without RCCL between their ranks: nqe_comm_create_custom takes all_gather / all_gather_v, nqe_comm_create_p2p takes
    // send / recv / group brackets with RCCL's matching rules — see nqe.h; a failed rank fails every rank, nobody blocks)
    fn nqe_comm_destroy(comm: *mut NqeComm) -> i32;
    fn nqe_sharded_aggregate_execute(comm: *mut NqeComm, t: *const NqeTable, pred: *const NqeExprNode, pn: i32, group: *const NqeExprNode, gn: i32,
                                     aggs: *const NqeAggregate, na: i32, out: *mut *mut NqeTable, keys_out: *mut *mut NqeTable) -> i32;
    fn nqe_hash_join_build(ctx: *mut NqeCtx, left: *const NqeTable, left_key: i32, out: *mut *mut NqeJoinTable) -> i32;
    fn nqe_join_table_release(jt: *mut NqeJoinTable) -> i32;
    fn nqe_sharded_hash_join_probe(comm: *mut NqeComm, build: *const NqeJoinTable, right_local: *const NqeTable, right_key: i32, gather: i32,
                                   out: *mut *mut NqeTable) -> i32;
    fn nqe_sharded_selection_projection_execute(comm: *mut NqeComm, t: *const NqeTable, pred: *const NqeExprNode, pn: i32, nodes: *const NqeExprNode,
                                                offs: *const i32, ne: i32, gather: i32, out: *mut *mut NqeTable) -> i32;
    fn nqe_table_all_gather(comm: *mut NqeComm, local: *const NqeTable, out: *mut *mut NqeTable) -> i32;
}
/// The host distributes the 128-byte id however it talks to its peers (MPI, a TCP store, a file): rank 0 draws it.
pub struct GpuComm { raw: *mut NqeComm, pub rank: i32, pub world: i32 }
impl GpuComm {
    pub fn unique_id() -> Result<[u8; 128]> { let mut id = [0u8; 128]; if unsafe { nqe_comm_get_unique_id(id.as_mut_ptr()) } != 0 { return Err(ErrorCode::Others); } Ok(id) }
    pub fn create(ctx: &GpuCtx, id: &[u8; 128], rank: i32, world: i32) -> Result<Self> {
        let mut c = std::ptr::null_mut();
        ctx.check(unsafe { nqe_comm_create(ctx.0, id.as_ptr(), rank, world, &mut c) })?;
        Ok(Self { raw: c, rank, world })
    }
}
impl Drop for GpuComm { fn drop(&mut self) { unsafe { nqe_comm_destroy(self.raw); } } }
// GpuAggregatePlan / GpuHashJoin / GpuSelectionPlan take an Option<Arc<GpuComm>>: with one, `execute()` calls the sharded entry
// point on this process's shard of the input (rows [rank*n/world, (rank+1)*n/world) of every table) and gets the WHOLE result
// (aggregate; join / filter with gather = 1) or its own rows in rank order (gather = 0).

// ------------------------------------------------------------------ the planner edit (planner/mod.rs:42-182)
//
//     LogicalPlan::Filter(filter) => {
//         let predicate = Self::create_physical_expression(&filter.predicate, plan)?;
//         let input = Self::create_physical_plan(&filter.input)?;
//         match gpu::context() { Some(ctx) => Ok(GpuSelectionPlan::create(ctx, input, predicate)), None => Ok(SelectionPlan::create(input, predicate)) }
//     }
//
// and likewise for Projection (:48-62), Join (:71-89) and Aggregate (:95-170) — or, leaving the planner alone, run `gpu::rewrite`
// over the tree it returns (db.rs:34-36: between create_physical_plan and execute).  To keep intermediates in HBM between operators,
// carry `GpuTable` handles in a `GpuBatch` next to `RecordBatch` instead of downloading after every operator — what
// `naive_query_engine_amd/physical_plan.py` (`DeviceRecordBatch`) and `host/naive_db.hpp` do.
