//! `GpuCStore<VStore>`: drop-in for `CStoreFD<VStore>` (src/libpcp/propagation/mod.rs:33-34) in
//! `Space<VStore, CStore, NoRecomputation<..>>` (src/libpcp/search/mod.rs:41-43).  Only `Consistency::consistency`
//! (src/libpcp/propagation/store.rs:247-257) changes: the propagation fixpoint runs on the GPU.  Alloc / Empty / Clone /
//! Freeze / Collection / DisplayStateful delegate to the stock store, which keeps the boxed propagators — every bound of
//! `IntCStore<VStore>` (src/libpcp/concept.rs:120-138) is implemented below.  `resident::ResidentCStore` is the variant that keeps
//! the node's domains in HBM between `consistency()` calls (pcp_propagate_device) and moves only what changed.
//! UNCOMPILED: never built where this repository is built (no cargo / rustc); the C++ twin pcp_amd/host/pcp_host.hpp — same
//! structure, GpuCStore and ResidentGpuCStore — is compiled and GPU-tested (tests/test_gpu_parity.py, tests/test_reified.py).
use gcollections::ops::*;
use interval::interval::Interval;
use interval::ops::Range;
use pcp::concept::*;
use pcp::kernel::*;
use pcp::model::Model;
use pcp::propagation::lower::{Lower, Operand, PropDesc, Tree}; // added by apply_lower_hook.sh
use pcp::propagation::CStoreFD;
use pcp::term::identity::Identity;
use pcp::variable::ops::*;
use pcp_hip_sys::*;
use std::ptr;
use trilean::SKleene;

/// The engine context: destroyed when the store (or its frozen form) goes away.  Kept apart from GpuCStore so that `freeze(self)`
/// can move the fields out (a type with a Drop impl cannot be destructured).
pub struct Device { ctx: *mut pcp_ctx }
impl Drop for Device {
    fn drop(&mut self) { unsafe { pcp_ctx_destroy(self.ctx) } }
}

pub struct GpuCStore<VStore> {
    cpu: CStoreFD<VStore>,  // the boxed propagators, `active`, label/restore
    dev: Device,
    n_vars: usize,          // variables the device model was reset with
    dev_units: Vec<(usize, u64)>,  // (host index, allocation stamp) of each device-side unit, ascending
    stamp_of: Vec<u64>,     // allocation stamp of each host unit: a restore followed by an alloc re-uses an INDEX, never a stamp
    next_stamp: u64,
    cpu_only: bool,         // a propagator without a lowering: this space stays on the CPU path
}

impl<VStore> GpuCStore<VStore> {
    pub fn new(hip_device: i32) -> Self
    where
        CStoreFD<VStore>: Empty,
    {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { pcp_ctx_create(hip_device, &mut ctx) };
        assert!(rc == PCP_OK, "pcp_ctx_create: {} (the engine has no CPU path)", rc);
        GpuCStore { cpu: CStoreFD::empty(), dev: Device { ctx }, n_vars: usize::MAX, dev_units: vec![], stamp_of: vec![], next_stamp: 0, cpu_only: false }
    }

    fn check(&self, rc: i32) {
        // PCP_ERR_CONTRACT <=> one of libpcp's own assert! would have fired: keep the reference's error convention
        assert!(rc == PCP_OK, "libpcp_hip: {}", unsafe { std::ffi::CStr::from_ptr(pcp_last_error(self.dev.ctx)) }.to_string_lossy());
    }

    fn rows_of(desc: &[PropDesc], gid: u32) -> Vec<pcp_prop> {
        let grouped = desc.len() > 1;
        desc.iter()
            .map(|d| {
                let mut p = pcp_prop { kind: d.kind, group_kind: if grouped { 1 } else { 0 }, reserved: 0, group: gid, var: [PCP_NOVAR; 3], off: [0; 3] };
                for (i, op) in d.ops.iter().enumerate() {
                    match *op {
                        Operand::Var(v, off) => { p.var[i] = v as u32; p.off[i] = off; }
                        Operand::Const(c) => { p.var[i] = PCP_CONST; p.off[i] = c; }
                    }
                }
                p
            })
            .collect()
    }

    /// Brings the device model in line with `cpu.propagators` (append-only since the last call, or truncated by a restore:
    /// store.rs:223-230, 319-323).  Unit u of the device is host unit dev_units[u].  The search loop restores to label L and then
    /// allocs the right-branch constraint at the same index L where the left branch's was (branch.rs:51-55): an index alone does
    /// not identify a unit, its allocation stamp does — a device unit is kept only while the host unit at its index still carries
    /// the stamp it was sent with.
    fn sync_model(&mut self, n_vars: usize, hull: (i32, i32)) {
        if n_vars != self.n_vars {
            self.check(unsafe { pcp_model_reset(self.dev.ctx, n_vars as u32, 0) });
            self.check(unsafe { pcp_model_set_hull(self.dev.ctx, hull.0, hull.1) }); // the root's hull bounds every later node
            self.n_vars = n_vars;
            self.dev_units.clear();
        }
        let len = self.cpu.propagators_len();                 // accessor added by apply_lower_hook.sh
        self.stamp_of.truncate(len);                          // a restore shortened the store since the last alloc
        let keep = self.dev_units.iter().take_while(|&&(u, st)| u < len && self.stamp_of[u] == st).count();
        if keep < self.dev_units.len() {
            self.check(unsafe { pcp_model_truncate(self.dev.ctx, keep as u32) });
            self.dev_units.truncate(keep);
        }
        let first_new = self.dev_units.last().map_or(0, |&(u, _)| u + 1);
        for u in first_new..len {
            match self.cpu.propagator(u).lower() {
                Some(desc) => {
                    let rows = Self::rows_of(&desc, u as u32);
                    self.check(unsafe { pcp_model_push_props(self.dev.ctx, rows.len() as u32, rows.as_ptr()) });
                    self.dev_units.push((u, self.stamp_of[u]));
                }
                None => match self.cpu.propagator(u).lower_tree() {
                    // a formula unit (logic/: Disjunction, a Conjunction over formulas): pcp_model_push_formula, breadth-first layout
                    Some(tree) => {
                        let (nodes, leaves) = Self::flatten(&tree);
                        self.check(unsafe { pcp_model_push_formula(self.dev.ctx, nodes.len() as u32, nodes.as_ptr(), leaves.len() as u32, leaves.as_ptr()) });
                        self.dev_units.push((u, self.stamp_of[u]));
                    }
                    None => { self.cpu_only = true; return; }
                },
            }
        }
    }

    /// nodes[0] is the root, the children of an inner node are consecutive (include/pcp_hip.h, pcp_model_push_formula).
    fn flatten(tree: &Tree) -> (Vec<pcp_fnode>, Vec<pcp_prop>) {
        let mut nodes = vec![pcp_fnode::default()];
        let mut leaves: Vec<pcp_prop> = vec![];
        let mut queue: Vec<(&Tree, usize)> = vec![(tree, 0)];
        let mut qi = 0;
        while qi < queue.len() {
            let (t, at) = queue[qi];
            qi += 1;
            match t {
                Tree::Leaf(d) => {
                    nodes[at] = pcp_fnode { type_: 0, reserved: 0, n_children: 0, first: leaves.len() as u32 };
                    leaves.push(Self::rows_of(std::slice::from_ref(d), 0)[0]);
                }
                Tree::And(fs) | Tree::Or(fs) => {
                    let ty = if let Tree::And(_) = t { 1 } else { 2 };
                    nodes[at] = pcp_fnode { type_: ty, reserved: 0, n_children: fs.len() as u16, first: nodes.len() as u32 };
                    for f in fs {
                        nodes.push(pcp_fnode::default());
                        queue.push((f, nodes.len() - 1));
                    }
                }
            }
        }
        (nodes, leaves)
    }
}

impl<VStore> Consistency<VStore> for GpuCStore<VStore>
where
    VStore: VStoreConcept<Item = Interval<i32>>,
    CStoreFD<VStore>: Consistency<VStore>,
{
    fn consistency(&mut self, vstore: &mut VStore) -> SKleene {
        let n = vstore.size();
        let mut lb: Vec<i32> = (0..n).map(|i| vstore[i].lower()).collect();
        let mut ub: Vec<i32> = (0..n).map(|i| vstore[i].upper()).collect();
        let hull = (lb.iter().copied().min().unwrap_or(0), ub.iter().copied().max().unwrap_or(0));
        self.sync_model(n, hull);
        if self.cpu_only {
            return self.cpu.consistency(vstore); // an unknown propagator kind: the stock engine
        }
        // node = (bounds of every variable, `active` of the device-side units as u64 words)
        let words = (self.dev_units.len() + 63) / 64;
        let mut active = vec![0u64; words];
        for (k, &(u, _)) in self.dev_units.iter().enumerate() {
            if self.cpu.is_active(u) { active[k >> 6] |= 1u64 << (k & 63); }
        }
        let mut status = 0u8;
        let rc = unsafe {
            pcp_propagate(self.dev.ctx, 1, lb.as_mut_ptr(), ub.as_mut_ptr(), ptr::null_mut(),
                          if words > 0 { active.as_mut_ptr() } else { ptr::null_mut() }, &mut status, ptr::null_mut())
        };
        self.check(rc);
        if status != PCP_FALSE {
            // post-conditions of Store::consistency: narrowed domains go through MonotonicUpdate::update so that the trail records
            // the old values (variable/memory/trail_memory.rs:100-104), the event delta is left drained, entailed units leave `active`
            for i in 0..n {
                let ok = vstore.update(&Identity::new(i), Interval::new(lb[i], ub[i]));
                debug_assert!(ok);
            }
            let _ = vstore.drain_delta().count();
            for (k, &(u, _)) in self.dev_units.iter().enumerate() {
                if (active[k >> 6] >> (k & 63)) & 1 == 0 { self.cpu.deactivate(u); }
            }
        }
        match status { PCP_FALSE => SKleene::False, PCP_TRUE => SKleene::True, _ => SKleene::Unknown }
    }
}

// Everything else of IntCStore is the stock store's (concept.rs:120-138).
impl<VStore> Collection for GpuCStore<VStore> where CStoreFD<VStore>: Collection { type Item = <CStoreFD<VStore> as Collection>::Item; }
impl<VStore> AssociativeCollection for GpuCStore<VStore> where CStoreFD<VStore>: AssociativeCollection {
    type Location = <CStoreFD<VStore> as AssociativeCollection>::Location;
}
impl<VStore> Alloc for GpuCStore<VStore> where CStoreFD<VStore>: Alloc {
    fn alloc(&mut self, p: Self::Item) -> Self::Location {
        // Store::alloc (store.rs:223-230) appends at index len; whatever the device holds at that index or beyond is stale
        let idx = self.cpu.propagators_len();
        self.stamp_of.truncate(idx);
        self.stamp_of.push(self.next_stamp);
        self.next_stamp += 1;
        self.cpu.alloc(p)
    }
}
impl<VStore> Empty for GpuCStore<VStore> where CStoreFD<VStore>: Empty {
    fn empty() -> Self { GpuCStore::new(0) }
}
// Clone (store.rs:260-272: deep-clones the propagators, drops reactor/scheduler): a fresh context, the model is re-sent lazily.
impl<VStore> Clone for GpuCStore<VStore> where CStoreFD<VStore>: Clone + Empty {
    fn clone(&self) -> Self {
        let mut c = GpuCStore::new(0);
        c.cpu = self.cpu.clone();
        c.stamp_of = self.stamp_of.clone();
        c.next_stamp = self.next_stamp;
        c
    }
}
// Freeze / Snapshot (store.rs:306-324): the label is the stock store's (propagators.len(), active.clone()); restoring truncates
// `cpu.propagators`, and the next consistency() truncates the device model to match (sync_model).
impl<VStore> Freeze for GpuCStore<VStore>
where
    CStoreFD<VStore>: Freeze,
{
    type FrozenState = FrozenGpuCStore<VStore>;
    fn freeze(self) -> Self::FrozenState {
        let GpuCStore { cpu, dev, n_vars, dev_units, stamp_of, next_stamp, cpu_only } = self;
        FrozenGpuCStore { cpu: cpu.freeze(), dev, n_vars, dev_units, stamp_of, next_stamp, cpu_only }
    }
}

/// The frozen store: the stock FrozenStore (store.rs:274-324) plus the device-side bookkeeping, which is immutable while frozen.
pub struct FrozenGpuCStore<VStore>
where
    CStoreFD<VStore>: Freeze,
{
    cpu: <CStoreFD<VStore> as Freeze>::FrozenState,
    dev: Device,
    n_vars: usize,
    dev_units: Vec<(usize, u64)>,
    stamp_of: Vec<u64>,
    next_stamp: u64,
    cpu_only: bool,
}

impl<VStore> Snapshot for FrozenGpuCStore<VStore>
where
    CStoreFD<VStore>: Freeze,
{
    /// (the stock label = (propagators.len(), active.clone()) of store.rs:315-317, stamp_of.len()): the stamps of the units that exist
    /// at the label survive a restore; later allocs get fresh ones, so the device never mistakes a re-used index for the old unit.
    type Label = (<<CStoreFD<VStore> as Freeze>::FrozenState as Snapshot>::Label, usize);
    type State = GpuCStore<VStore>;

    fn label(&mut self) -> Self::Label {
        (self.cpu.label(), self.stamp_of.len())
    }

    fn restore(self, label: Self::Label) -> Self::State {
        let FrozenGpuCStore { cpu, dev, n_vars, dev_units, mut stamp_of, next_stamp, cpu_only } = self;
        stamp_of.truncate(label.1);
        // dev_units is NOT cut here: sync_model (next consistency()) drops the device units whose host unit is gone or re-allocated
        // and calls pcp_model_truncate once, instead of one truncate per restore of a search that restores far more often than it propagates.
        // `cpu_only` is sticky by design: a store that ever held an unlowerable propagator may still hold it after this restore.
        GpuCStore { cpu: cpu.restore(label.0), dev, n_vars, dev_units, stamp_of, next_stamp, cpu_only }
    }
}

// DisplayStateful<Model> (store.rs:103-115) and DisplayStateful<(Model, VStore)> (store.rs:83-101): the propagators live in the stock store.
impl<VStore> DisplayStateful<Model> for GpuCStore<VStore>
where
    CStoreFD<VStore>: DisplayStateful<Model>,
{
    fn display(&self, model: &Model) { self.cpu.display(model) }
}
impl<VStore> DisplayStateful<(Model, VStore)> for GpuCStore<VStore>
where
    CStoreFD<VStore>: DisplayStateful<(Model, VStore)>,
{
    fn display(&self, state: &(Model, VStore)) { self.cpu.display(state) }
}

/// Compile-time check that the store is a drop-in: `Space<VStoreFD, GpuCStore<VStoreFD>, NoRecomputation<..>>` (search/mod.rs:41-43) needs
/// exactly `IntCStore<VStore>` (concept.rs:120-138).
#[allow(dead_code)]
fn assert_is_int_cstore<VStore>()
where
    VStore: VStoreConcept<Item = Interval<i32>> + 'static,
    GpuCStore<VStore>: IntCStore<VStore>,
{
}

pub mod resident;
