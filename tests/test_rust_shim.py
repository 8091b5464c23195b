"""CPU-only: integration/rust/gpu.rs (uncompiled source: the image has no Rust toolchain) stays consistent with include/nqe.h and
with itself (tools/check_rust_shim.py), and the check is not vacuous: seeded defects of the kinds a compiler would reject are found."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools import check_rust_shim as crs  # noqa: E402

SHIM = os.path.join(ROOT, "integration", "rust", "gpu.rs")


def test_shim_is_consistent():
    problems, n_ffi, n_calls = crs.check(SHIM)
    assert problems == []
    assert n_ffi >= 30 and n_calls >= 150


def test_every_operator_entry_point_of_the_header_is_bound():
    """the operators of the hot path and the resident-table plumbing are all declared in the shim's extern blocks"""
    rust = crs.extern_functions(crs.strip_rust(open(SHIM).read()))
    for name in ["nqe_table_create", "nqe_table_project", "nqe_table_slice", "nqe_table_concat", "nqe_expr_evaluate", "nqe_filter", "nqe_selection_execute",
                 "nqe_projection_execute", "nqe_selection_projection_execute", "nqe_aggregate_execute", "nqe_hash_join_build", "nqe_hash_join_probe",
                 "nqe_sharded_aggregate_execute", "nqe_sharded_hash_join_probe", "nqe_sharded_selection_projection_execute", "nqe_table_all_gather",
                 "nqe_table_import_arrow", "nqe_table_export_arrow"]:
        assert name in rust, name


def mutated(tmp_path, old, new):
    src = open(SHIM).read()
    assert old in src, old
    p = tmp_path / "gpu.rs"
    p.write_text(src.replace(old, new, 1))
    return crs.check(str(p))[0]


def test_seeded_defects_are_found(tmp_path):
    # an argument dropped from an extern declaration
    assert any("nqe_hash_join_probe takes 5 arguments" in x for x in mutated(tmp_path, "right: *const NqeTable, right_key: i32, out: *mut *mut NqeTable) -> i32;\n    fn nqe_join_table_release",
                                                                            "right: *const NqeTable, out: *mut *mut NqeTable) -> i32;\n    fn nqe_join_table_release"))
    # a pointer's constness
    assert any("nqe_table_release argument 1" in x for x in mutated(tmp_path, "fn nqe_table_release(table: *mut NqeTable)", "fn nqe_table_release(table: *const NqeTable)"))
    # an integer width
    assert any("nqe_table_slice argument 3" in x for x in mutated(tmp_path, "offset: i64, length: i64, out", "offset: i32, length: i64, out"))
    # a symbol the header does not have
    assert any("nqe_table_frobnicate" in x for x in mutated(tmp_path, "fn nqe_table_release(", "fn nqe_table_frobnicate(t: i32) -> i32;\n    fn nqe_table_release("))
    # the defects VERDICT r03 found: an associated function nobody defines, a constructor that is missing
    assert any("GpuAggregatePlan::from_reference" in x for x in mutated(tmp_path, "pub fn from_reference(", "pub fn from_reference_renamed("))
    assert any("GpuHashJoin::create" in x for x in mutated(tmp_path, "pub fn create(ctx: Arc<GpuCtx>, left: PhysicalPlanRef", "pub fn build_it(ctx: Arc<GpuCtx>, left: PhysicalPlanRef"))
    # a method / a field that does not exist
    assert any("self.selekt()" in x for x in mutated(tmp_path, "None => self.select(&input),", "None => self.selekt(&input),"))
    assert any("self.proj " in x or "self.proj" in x for x in mutated(tmp_path, "match &self.project {", "match &self.proj {"))
    assert any("ctx.upload_it()" in x for x in mutated(tmp_path, "child.execute()?.iter().map(|b| ctx.upload(b))", "child.execute()?.iter().map(|b| ctx.upload_it(b))"))
    # a bracket
    assert mutated(tmp_path, "fn wrap(raw: *mut NqeTable) -> Self { GpuBatch { table: Arc::new(GpuTable(raw)) } }", "fn wrap(raw: *mut NqeTable) -> Self { GpuBatch { table: Arc::new(GpuTable(raw)) }")


def test_rewrite_covers_every_operator_the_planner_builds():
    """planner/mod.rs:42-182 builds Scan, Projection, Limit, Offset, Join (HashJoin), Filter (Selection), Aggregate: the shim's rewrite
    pass has an arm for each (HashJoin / Limit / Offset were left untouched)"""
    src = crs.strip_rust(open(SHIM).read())
    body = src[src.index("pub fn rewrite_sharded"):]
    for ty in ["ScanPlan", "ProjectionPlan", "SelectionPlan", "PhysicalAggregatePlan", "HashJoin", "PhysicalLimitPlan", "PhysicalOffsetPlan"]:
        assert f"downcast_ref::<{ty}>()" in body, ty
