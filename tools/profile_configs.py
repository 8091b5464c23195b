"""The kernels that make up one step of each bench config (shared by tools/summarize_profiles.py and tools/round_table.py)."""
# the kernels that make up one step of each config (substring of the rocprofv3 kernel name → launches per step), and the step's
# algorithmic bytes
CONFIGS = {
    "headline": ({"agg_grouped_fast_kernel": 1}, 16e9),
    "headline_random_keys": ({"agg_grouped_fast_kernel": 1}, 16e9),
    "c2": ({"keep_from_range_tile_kernel": 1, "scan_redundant_kernel": 1, "compact_staged_kernel": 1}, 2.0e9),
    "c3": ({"agg_grouped_fast_kernel": 1}, 16e9),
    "c3_random_keys": ({"agg_grouped_fast_kernel": 1}, 16e9),
    "c4": ({"join_sample_range_kernel": 1, "join_fused_write_kernel": 1}, 4.816e9),
    "c4_sparse_keys": ({"probe_packed_kernel": 1, "join_fused_write_kernel": 1}, 4.816e9),
    "agg_65536_groups": ({"agg_slab_scatter_soa_kernel": 1, "agg_range_segments_kernel": 1, "agg_range_emit_kernel": 1}, 1.6e9),
    "headline_single_column": ({"agg_grouped_fast_kernel": 1}, 8e9),
    "headline_int64_values": ({"agg_grouped_fast_kernel": 1}, 16e9),
    "agg_tree_predicate": ({"nqe_jit_agg": 1, "agg_merge_partials_kernel": 1}, 16e9),
    "agg_three_value_columns": ({"agg_tiny_groups_kernel": 1, "agg_fold_partials_kernel": 3}, 24e9),
    "agg_4096_groups": ({"agg_grouped_fast_kernel": 1, "agg_fold_partials_kernel": 1}, 1.6e9),
    "agg_5000_groups": ({"agg_grouped_fast_kernel": 1, "agg_fold_partials_kernel": 1, "agg_range_emit_kernel": 1}, 1.6e9),
    "agg_12000_groups_count_sum_avg": ({"agg_grouped_fast_kernel": 1, "agg_fold_partials_kernel": 1, "agg_range_emit_kernel": 1}, 1.6e9),
    "agg_6000_groups": ({"agg_grouped_fast_kernel": 1, "agg_fold_partials_kernel": 1, "agg_range_emit_kernel": 1}, 1.6e9),
    "agg_11000_groups": ({"agg_grouped_fast_kernel": 1, "agg_fold_partials_kernel": 1, "agg_range_emit_kernel": 1}, 1.6e9),
    "c2_random_ids": ({"keep_from_range_tile_kernel": 1, "scan_redundant_kernel": 1, "compact_staged_kernel": 1}, 2.0e9),
    "c2_expression_trees": ({"nqe_jit_selproj": 1}, 2.4e9),
    "agg_readme_shape": ({"agg_tiny_groups_kernel": 1, "agg_fold_partials_kernel": 3}, 24e9),
    "headline_nullable": ({"agg_grouped_fast_kernel": 1}, 16.125e9),
    "agg_1048576_groups": ({"agg_slab_scatter_soa_kernel": 1, "agg_range_segments_kernel": 1, "agg_range_emit_kernel": 1}, 1.6e9),
    # borrowed probe table: every output column written (one optimistic pass); immutable probe table: its columns are shared
    "c4_shared_probe_columns": ({"join_sample_range_kernel": 1, "join_fused_write_kernel": 1}, 4.816e9),
    "c4_wide_payload": ({"join_sample_range_kernel": 1, "join_fused_write_kernel": 1}, 4.816e9),
    "c4_dim_1e7": ({"join_sample_range_kernel": 1, "join_fused_write_kernel": 1}, 4.96e9),
    "c4_dup_keys": ({"probe_count_kernel": 1, "probe_write_kernel": 1}, 4.816e9),
    "c4_partial_match": ({"probe_presence_kernel": 1, "join_fused_write_kernel": 1}, 4.496e9),
    # 10^8-row build side: the probe kernels per step as for c4 (the kernel stats of this config also hold the partitioned build's kernels)
    "c4_dim_1e8": ({"join_sample_range_kernel": 1, "join_fused_write_kernel": 1}, 6.4e9),
}


