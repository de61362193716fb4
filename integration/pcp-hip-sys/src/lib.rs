//! Raw bindings of `include/pcp_hip.h` (ABI v6): one line per exported symbol, `#[repr(C)]` mirrors of its structs — plus the four HIP
//! runtime calls the device-resident store needs (hipMalloc / hipFree / hipMemcpy; `build.rs` links amdhip64).  UNCOMPILED here.
//! Every entry point cites the libpcp item it replaces in the header; the safe layer is `pcp-gpu-cstore`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const PCP_ABI_VERSION: u32 = 8;
pub const PCP_CONST: u32 = 0xFFFF_FFFF; // operand is a term::Constant; off[i] = its value
pub const PCP_NOVAR: u32 = 0xFFFF_FFFE; // operand slot unused
pub const PCP_SUM: u32 = 0xC000_0000; //   var[i] = PCP_SUM | t: term::Sum number t (pcp_model_push_sum)
pub const PCP_BOUND_MAX: i32 = 0x1FFF_FFFF;
pub const PCP_STATUS_HULL: u8 = 0xFE;

pub const PCP_OK: i32 = 0;
pub const PCP_ERR_ARG: i32 = -1;
pub const PCP_ERR_CONTRACT: i32 = -2; // a libpcp assert! would have fired
pub const PCP_ERR_HIP: i32 = -3;
pub const PCP_ERR_NOMEM: i32 = -4;
pub const PCP_ERR_UNSUPPORTED: i32 = -5;
pub const PCP_ERR_NODEVICE: i32 = -6; // there is no CPU fallback

// pcp_kind
pub const PCP_NEQ: u8 = 0;
pub const PCP_EQ: u8 = 1;
pub const PCP_LT: u8 = 2;
pub const PCP_LT3: u8 = 3;
pub const PCP_GT3: u8 = 4;
pub const PCP_EQ3: u8 = 5;
pub const PCP_MUL3: u8 = 6;
pub const PCP_BOOL: u8 = 7; //  logic::Boolean, "X = 1" over a 0/1 view (one operand)
pub const PCP_NBOOL: u8 = 8; // logic::BooleanNeg, "X = 0"
// pcp_status == trilean::SKleene
pub const PCP_FALSE: u8 = 0;
pub const PCP_TRUE: u8 = 1;
pub const PCP_UNKNOWN: u8 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pcp_prop {
    pub kind: u8,
    pub group_kind: u8, // 0 standalone, 1 Conjunction member, 2 Distinct member
    pub reserved: u16,
    pub group: u32,
    pub var: [u32; 3],
    pub off: [i32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pcp_stats {
    pub steps: u64,
    pub steps3: u64,
    pub narrowings: u64,
    pub waves: u64,
    pub failed_nodes: u64,
    pub nodes: u64,
    pub evaluated: u64,
    pub full_evals: u64,
}

#[repr(C)]
pub struct pcp_device_batch {
    pub lb_in: *const i32,
    pub ub_in: *const i32,
    pub lb_out: *mut i32,
    pub ub_out: *mut i32,
    pub active_in: *const u64, // NULL = every unit active = implicit-active nodes
    pub active_out: *mut u64,
    pub status: *mut u8,
    pub bits_in: *const u64, // set mode only
    pub bits_out: *mut u64,
    /// ABI v7, nullable: [n_nodes] the ONE variable in which node i differs from a fixpoint of this model (a child of a propagated node:
    /// `pcp_branch_device_hint` writes it), or any value >= n_vars = no promise.  Same results, less work (include/pcp_hip.h).
    pub dirty_var: *const u32,
    /// ABI v7: 0 = `PCP_CELLS_I32` (lb / ub int32 rows), 1 = `PCP_CELLS_PACKED16` (ONE row of 32-bit cells `(-lb & 0xffff) | ub << 16` per node in
    /// lb_in / lb_out; all-XNeqY models with a declared hull within +-16383 only).
    pub cell_format: u32,
    pub reserved: u32,
}
pub const PCP_CELLS_I32: u32 = 0;
pub const PCP_CELLS_PACKED16: u32 = 1;

/// One node of a formula unit (logic/conjunction.rs, logic/disjunction.rs): type 0 leaf (first = index into the leaves), 1 Conjunction,
/// 2 Disjunction (first = index of the first child, children consecutive).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pcp_fnode {
    pub type_: u8,
    pub reserved: u8,
    pub n_children: u16,
    pub first: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pcp_plan {
    pub nodes_per_block: u32,
    pub team: u32,
    pub packed: u32,
    pub word_level: u32,
    pub global_dom: u32,
    pub compact: u32,
    pub implicit_active: u32,
    pub set_mode: u32,
    pub grid: u32,
    pub block: u32,
    pub lds_bytes: u32,
    pub list_cap: u32,
    pub path: u32, // 0 generic sweep kernels (pcp_kernels.hip), 1 assignment-driven all-XNeqY kernel (pcp_neq.hip), 2 10-bit-cell kernel for
                   // large binary stores (pcp_big.hip), 3 formula kernel (pcp_formula.hip), 4 one wavefront per node for small stores (pcp_small.hip)
}

/// The device-resident DFS of `pcp_dfs_device`: a LIFO stack of implicit-active nodes and its 8 bytes of state.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct pcp_dfs_state {
    pub lb: *mut i32,
    pub ub: *mut i32,
    pub capacity: u32,
    pub sp: *mut u32,
    pub stop: *mut u32,
    pub status: *mut u8,
    pub counters: *mut u64, // nodes, solutions, failed, error, internal
    pub first_solution: *mut i32,
    pub dirty: *mut u32, // ABI v7, nullable: [capacity] per stack row, the variable the row was branched on (>= n_vars: none)
}

/// `pcp_dfs_forest_device_set` (ABI v5): the search loop over FDSpace on the device, one tree per workgroup, an undo trail per tree.
pub const PCP_DFS_FULL: u32 = 0xFFFF_FFFF;
#[repr(C)]
#[derive(Clone, Copy)]
pub struct pcp_forest_state {
    pub n_trees: u32,
    pub level_capacity: u32,
    pub trail_capacity: u32,
    pub reserved: u32,
    pub bits: *mut u64,        // [n_trees][n_vars][set_words]
    pub tree: *mut u32,        // [n_trees][4]: levels, trail length, pending variable, finished
    pub levels: *mut u32,      // [n_trees][level_capacity][4]
    pub trail: *mut u32,       // [n_trees][trail_capacity][4]
    pub counters: *mut u64,    // [n_trees][4]: nodes, solutions, failed, error
    pub total_nodes: *mut u64,
    pub stop: *mut u32,
    pub first_solution: *mut i32,
    pub solution_flag: *mut u32,
}

/// `pcp_debug_counters` slots (ABI v6): which code paths ran since the last `pcp_stats_reset`.
pub const PCP_DBG_BIG_DENSE: usize = 0;
pub const PCP_DBG_BIG_SPARSE: usize = 1;
pub const PCP_DBG_NEQ_TILES: usize = 2;
pub const PCP_DBG_NEQ_OVERLAP: usize = 3;
pub const PCP_DBG_SMALL_NODES: usize = 4;
pub const PCP_DBG_COUNT: usize = 16;

pub enum pcp_ctx {}

extern "C" {
    pub fn pcp_abi_version() -> u32;
    pub fn pcp_strerror(err: i32) -> *const c_char;
    pub fn pcp_ctx_create(hip_device: i32, out: *mut *mut pcp_ctx) -> i32;
    pub fn pcp_ctx_destroy(ctx: *mut pcp_ctx);
    pub fn pcp_last_error(ctx: *const pcp_ctx) -> *const c_char;
    pub fn pcp_model_reset(ctx: *mut pcp_ctx, n_vars: u32, set_words: u32) -> i32; // Store::empty
    pub fn pcp_model_push_props(ctx: *mut pcp_ctx, n: u32, props: *const pcp_prop) -> i32; // Store::alloc
    pub fn pcp_model_push_sum(ctx: *mut pcp_ctx, n_members: u32, vars: *const u32, term: *mut u32) -> i32; // Sum::new
    pub fn pcp_model_push_formula(ctx: *mut pcp_ctx, n_nodes: u32, nodes: *const pcp_fnode, n_leaves: u32, leaves: *const pcp_prop) -> i32; // one formula unit (logic/)
    pub fn pcp_model_truncate(ctx: *mut pcp_ctx, n_units: u32) -> i32; // FrozenStore::restore
    pub fn pcp_model_n_units(ctx: *const pcp_ctx, n_units: *mut u32, n_props: *mut u32) -> i32;
    pub fn pcp_model_set_hull(ctx: *mut pcp_ctx, lo: i32, hi: i32) -> i32; // hull of the VStore::alloc domains
    pub fn pcp_propagate(ctx: *mut pcp_ctx, n_nodes: u32, lb: *mut i32, ub: *mut i32, bits: *mut u64, active: *mut u64,
                         status: *mut u8, stats: *mut pcp_stats) -> i32; // Consistency::consistency
    pub fn pcp_propagate_device(ctx: *mut pcp_ctx, n_nodes: u32, batch: *const pcp_device_batch, hip_stream: *mut c_void) -> i32;
    /// ABI v8: the same for nodes that carry unary propagators of their own (Enumerate's `x != v`, search/branching/enumerate.rs:48-59):
    /// `node_unit_off` device u32 [n_nodes + 1], `node_units` device pcp_prop records (one variable against one Constant).
    pub fn pcp_propagate_device_units(ctx: *mut pcp_ctx, n_nodes: u32, batch: *const pcp_device_batch, node_unit_off: *const u32,
                                      node_units: *const pcp_prop, hip_stream: *mut c_void) -> i32;
    pub fn pcp_branch_device(ctx: *mut pcp_ctx, n_nodes: u32, lb: *const i32, ub: *const i32, active: *const u64, status: *const u8,
                             child_lb: *mut i32, child_ub: *mut i32, child_active: *mut u64, counts: *mut u32,
                             hip_stream: *mut c_void) -> i32; // Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter
    pub fn pcp_branch_device_hint(ctx: *mut pcp_ctx, n_nodes: u32, lb: *const i32, ub: *const i32, active: *const u64, status: *const u8,
                                  child_lb: *mut i32, child_ub: *mut i32, child_active: *mut u64, child_dirty: *mut u32, counts: *mut u32,
                                  hip_stream: *mut c_void) -> i32; // the same, and every child's pcp_device_batch.dirty_var entry
    pub fn pcp_pack_rows(ctx: *mut pcp_ctx, n_nodes: u32, lb: *const i32, ub: *const i32, cells: *mut u32, hip_stream: *mut c_void) -> i32;
    pub fn pcp_unpack_rows(ctx: *mut pcp_ctx, n_nodes: u32, cells: *const u32, lb: *mut i32, ub: *mut i32, hip_stream: *mut c_void) -> i32;
    pub fn pcp_branch_device_cells(ctx: *mut pcp_ctx, n_nodes: u32, cells: *const u32, status: *const u8, child_cells: *mut u32, child_dirty: *mut u32,
                                   counts: *mut u32, hip_stream: *mut c_void) -> i32;
    pub fn pcp_branch_device_set(ctx: *mut pcp_ctx, n_nodes: u32, bits: *const u64, lb: *const i32, ub: *const i32, active: *const u64,
                                 status: *const u8, child_bits: *mut u64, child_active: *mut u64, counts: *mut u32,
                                 hip_stream: *mut c_void) -> i32; // the same brancher over IntervalSet domains
    pub fn pcp_dfs_forest_device(ctx: *mut pcp_ctx, st: *const pcp_dfs_state, n_trees: u32, n_steps: u32, stop_on_solution: u32, node_limit: u64,
                                 hip_stream: *mut c_void) -> i32; // st's arrays are [n_trees]-strided: tree t is a pcp_dfs_device instance
    pub fn pcp_dfs_forest_device_set(ctx: *mut pcp_ctx, st: *const pcp_forest_state, n_steps: u32, stop_on_solution: u32, node_limit: u64,
                                     hip_stream: *mut c_void) -> i32;
    pub fn pcp_dfs_forest_split_set(ctx: *mut pcp_ctx, st: *const pcp_forest_state, n_pairs: u32, pairs: *const u32, done: *mut u32,
                                    hip_stream: *mut c_void) -> i32; // pairs = n_pairs x (donor, receiver), device memory
    pub fn pcp_dfs_device(ctx: *mut pcp_ctx, st: *const pcp_dfs_state, n_steps: u32, stop_on_solution: u32, node_limit: u64,
                          hip_stream: *mut c_void) -> i32; // OneSolution/AllSolution<Propagation<Brancher<..>>> under StopNode, n_steps nodes
    pub fn pcp_stats_reset(ctx: *mut pcp_ctx, hip_stream: *mut c_void) -> i32;
    pub fn pcp_stats_read(ctx: *mut pcp_ctx, out: *mut pcp_stats, hip_stream: *mut c_void) -> i32;
    pub fn pcp_debug_counters(ctx: *mut pcp_ctx, out: *mut u64, n: u32, hip_stream: *mut c_void) -> i32; // out[PCP_DBG_*]
    pub fn pcp_last_kernel_ms(ctx: *mut pcp_ctx, ms: *mut f32) -> i32;
    pub fn pcp_last_plan(ctx: *const pcp_ctx, out: *mut pcp_plan) -> i32;
    pub fn pcp_set_option(ctx: *mut pcp_ctx, key: *const c_char, value: i64) -> i32;
}

// ---- the HIP runtime, as far as pcp-gpu-cstore::resident needs it (hip_runtime_api.h) ---------------------------------------------
pub const HIP_MEMCPY_HOST_TO_DEVICE: i32 = 1; // hipMemcpyHostToDevice
pub const HIP_MEMCPY_DEVICE_TO_HOST: i32 = 2; // hipMemcpyDeviceToHost
extern "C" {
    pub fn hipMalloc(ptr: *mut *mut c_void, size: usize) -> i32;
    pub fn hipFree(ptr: *mut c_void) -> i32;
    pub fn hipMemcpy(dst: *mut c_void, src: *const c_void, size: usize, kind: i32) -> i32; // synchronises the null stream
}
