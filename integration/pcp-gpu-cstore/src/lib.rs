//! `GpuCStore<VStore>`: drop-in for `CStoreFD<VStore>` (src/libpcp/propagation/mod.rs:33-34) in
//! `Space<VStore, CStore, NoRecomputation<..>>` (src/libpcp/search/mod.rs:41-43).  Only `Consistency::consistency`
//! (src/libpcp/propagation/store.rs:247-257) changes: the propagation fixpoint runs on the GPU.  Alloc / Empty / Clone /
//! Freeze / Collection / DisplayStateful delegate to the stock store, which keeps the boxed propagators.
//! NEVER COMPILED where this repository is built (no cargo); the C++ twin pcp_amd/host/pcp_host.hpp is compiled and GPU-tested.
use gcollections::ops::*;
use interval::interval::Interval;
use interval::ops::Range;
use pcp::concept::*;
use pcp::kernel::*;
use pcp::propagation::lower::{Lower, Operand, PropDesc}; // added by apply_lower_hook.sh
use pcp::propagation::CStoreFD;
use pcp::term::identity::Identity;
use pcp::variable::ops::*;
use pcp_hip_sys::*;
use std::ptr;
use trilean::SKleene;

pub struct GpuCStore<VStore> {
    cpu: CStoreFD<VStore>,  // the boxed propagators, `active`, label/restore
    ctx: *mut pcp_ctx,
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
        GpuCStore { cpu: CStoreFD::empty(), ctx, n_vars: usize::MAX, dev_units: vec![], stamp_of: vec![], next_stamp: 0, cpu_only: false }
    }

    fn check(&self, rc: i32) {
        // PCP_ERR_CONTRACT <=> one of libpcp's own assert! would have fired: keep the reference's error convention
        assert!(rc == PCP_OK, "libpcp_hip: {}", unsafe { std::ffi::CStr::from_ptr(pcp_last_error(self.ctx)) }.to_string_lossy());
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
            self.check(unsafe { pcp_model_reset(self.ctx, n_vars as u32, 0) });
            self.check(unsafe { pcp_model_set_hull(self.ctx, hull.0, hull.1) }); // the root's hull bounds every later node
            self.n_vars = n_vars;
            self.dev_units.clear();
        }
        let len = self.cpu.propagators_len();                 // accessor added by apply_lower_hook.sh
        self.stamp_of.truncate(len);                          // a restore shortened the store since the last alloc
        let keep = self.dev_units.iter().take_while(|&&(u, st)| u < len && self.stamp_of[u] == st).count();
        if keep < self.dev_units.len() {
            self.check(unsafe { pcp_model_truncate(self.ctx, keep as u32) });
            self.dev_units.truncate(keep);
        }
        let first_new = self.dev_units.last().map_or(0, |&(u, _)| u + 1);
        for u in first_new..len {
            match self.cpu.propagator(u).lower() {
                Some(desc) => {
                    let rows = Self::rows_of(&desc, u as u32);
                    self.check(unsafe { pcp_model_push_props(self.ctx, rows.len() as u32, rows.as_ptr()) });
                    self.dev_units.push((u, self.stamp_of[u]));
                }
                None => { self.cpu_only = true; return; }
            }
        }
    }
}

impl<VStore> Drop for GpuCStore<VStore> {
    fn drop(&mut self) { unsafe { pcp_ctx_destroy(self.ctx) } }
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
            pcp_propagate(self.ctx, 1, lb.as_mut_ptr(), ub.as_mut_ptr(), ptr::null_mut(),
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
