//! `ResidentCStore<VStore>`: `GpuCStore` whose node stays in HBM between `consistency()` calls.  UNCOMPILED (no cargo / rustc where this
//! repository is built); its compiled, GPU-tested twin is `pcp_amd/host/pcp_host_resident.hpp` (tests/test_gpu_parity.py::
//! test_cpp_host_resident_store) — same fields, same three steps.
//!
//! `GpuCStore::consistency` hands the engine host buffers (`pcp_propagate`): two PCIe copies of the WHOLE node per call.  A search calls
//! `consistency()` once per node and changes almost nothing in between — `Branch::commit` restores a label and adds one branch constraint
//! (search/branching/branch.rs:51-55).  This store keeps the node's rows (lb, ub, `active` words, status byte) in device memory together
//! with a host mirror of what they hold:
//!   in : only the index range whose bounds differ from the mirror goes host-to-device;
//!   run: `pcp_propagate_device` from the current pair of rows into a second pair (out of place);
//!   out: the status byte; the rows only when the node did not fail, and then the output pair becomes the current one (a failed node's
//!        output rows are unspecified: the input rows and the mirror survive it); only the variables the fixpoint narrowed go through
//!        `MonotonicUpdate::update` (variable/store.rs:151-166), so the trail records them.
//! Everything else — Alloc, Empty, Clone, Freeze/Snapshot, Collection, DisplayStateful — is `GpuCStore`'s (lib.rs), reached through `inner`.
use super::*;
use std::os::raw::c_void;

struct Rows {
    lb: [*mut i32; 2],  // two pairs of rows: [cur] holds the node, the other takes the output
    ub: [*mut i32; 2],
    act: [*mut u64; 2],
    status: *mut u8,
    cur: usize,
    n: usize,
    words_cap: usize,
}
impl Rows {
    fn none() -> Rows {
        Rows { lb: [ptr::null_mut(); 2], ub: [ptr::null_mut(); 2], act: [ptr::null_mut(); 2], status: ptr::null_mut(), cur: 0, n: usize::MAX, words_cap: 0 }
    }
    fn hip(e: i32) { assert!(e == 0, "HIP error {}", e); }
    fn alloc(n: usize, words: usize) -> Rows {
        let mut r = Rows::none();
        r.n = n;
        r.words_cap = words.max(16);
        unsafe {
            for k in 0..2 {
                Self::hip(hipMalloc(&mut r.lb[k] as *mut *mut i32 as *mut *mut c_void, n.max(1) * 4));
                Self::hip(hipMalloc(&mut r.ub[k] as *mut *mut i32 as *mut *mut c_void, n.max(1) * 4));
                Self::hip(hipMalloc(&mut r.act[k] as *mut *mut u64 as *mut *mut c_void, r.words_cap * 8));
            }
            Self::hip(hipMalloc(&mut r.status as *mut *mut u8 as *mut *mut c_void, 8));
        }
        r
    }
}
impl Drop for Rows {
    fn drop(&mut self) {
        unsafe {
            for p in [self.lb[0] as *mut c_void, self.lb[1] as *mut c_void, self.ub[0] as *mut c_void, self.ub[1] as *mut c_void,
                      self.act[0] as *mut c_void, self.act[1] as *mut c_void, self.status as *mut c_void] {
                if !p.is_null() { hipFree(p); }
            }
        }
    }
}

pub struct ResidentCStore<VStore> {
    inner: GpuCStore<VStore>,
    rows: Rows,
    mirror: Option<(Vec<i32>, Vec<i32>, Vec<u64>)>, // what the device's current rows hold (lb, ub, active words); None = nothing resident
    pub bytes_in: u64,
    pub bytes_out: u64,
}

impl<VStore> ResidentCStore<VStore>
where
    CStoreFD<VStore>: Empty,
{
    pub fn new(hip_device: i32) -> Self {
        ResidentCStore { inner: GpuCStore::new(hip_device), rows: Rows::none(), mirror: None, bytes_in: 0, bytes_out: 0 }
    }
}

impl<VStore> Consistency<VStore> for ResidentCStore<VStore>
where
    VStore: VStoreConcept<Item = Interval<i32>>,
    CStoreFD<VStore>: Consistency<VStore>,
{
    fn consistency(&mut self, vstore: &mut VStore) -> SKleene {
        let n = vstore.size();
        let lb: Vec<i32> = (0..n).map(|i| vstore[i].lower()).collect();
        let ub: Vec<i32> = (0..n).map(|i| vstore[i].upper()).collect();
        let hull = (lb.iter().copied().min().unwrap_or(0), ub.iter().copied().max().unwrap_or(0));
        let g = &mut self.inner;
        g.sync_model(n, hull);
        if g.cpu_only {
            return g.cpu.consistency(vstore); // an unknown propagator kind: the stock engine
        }
        let words = (g.dev_units.len() + 63) / 64;
        let mut active = vec![0u64; words];
        for (k, &(u, _)) in g.dev_units.iter().enumerate() {
            if g.cpu.is_active(u) { active[k >> 6] |= 1u64 << (k & 63); }
        }
        if n != self.rows.n || words > self.rows.words_cap {
            self.rows = Rows::alloc(n, words);
            self.mirror = None;
        }
        let (cur, other) = (self.rows.cur, self.rows.cur ^ 1);
        // in: the range of variables whose bounds differ from what the device's current rows hold; the `active` words when they differ
        let (mut lo, mut hi) = (0usize, n);
        let mut act_differs = words > 0;
        if let Some((ml, mu, ma)) = &self.mirror {
            while lo < n && lb[lo] == ml[lo] && ub[lo] == mu[lo] { lo += 1; }
            while hi > lo && lb[hi - 1] == ml[hi - 1] && ub[hi - 1] == mu[hi - 1] { hi -= 1; }
            act_differs = words > 0 && *ma != active;
        }
        unsafe {
            if hi > lo {
                Rows::hip(hipMemcpy(self.rows.lb[cur].add(lo) as *mut c_void, lb.as_ptr().add(lo) as *const c_void, (hi - lo) * 4, HIP_MEMCPY_HOST_TO_DEVICE));
                Rows::hip(hipMemcpy(self.rows.ub[cur].add(lo) as *mut c_void, ub.as_ptr().add(lo) as *const c_void, (hi - lo) * 4, HIP_MEMCPY_HOST_TO_DEVICE));
                self.bytes_in += 8 * (hi - lo) as u64;
            }
            if act_differs {
                Rows::hip(hipMemcpy(self.rows.act[cur] as *mut c_void, active.as_ptr() as *const c_void, words * 8, HIP_MEMCPY_HOST_TO_DEVICE));
                self.bytes_in += 8 * words as u64;
            }
        }
        self.mirror = Some((lb.clone(), ub.clone(), active.clone())); // the current rows now hold exactly this node
        // run: out of place into the other pair of rows (null stream)
        let batch = pcp_device_batch {
            lb_in: self.rows.lb[cur], ub_in: self.rows.ub[cur], lb_out: self.rows.lb[other], ub_out: self.rows.ub[other],
            active_in: if words > 0 { self.rows.act[cur] } else { ptr::null() },
            active_out: if words > 0 { self.rows.act[other] } else { ptr::null_mut() },
            status: self.rows.status, bits_in: ptr::null(), bits_out: ptr::null_mut(), dirty_var: ptr::null(),
            cell_format: PCP_CELLS_I32, reserved: 0,
        };
        let rc = unsafe { pcp_propagate_device(g.dev.ctx, 1, &batch, ptr::null_mut()) };
        g.check(rc);
        // out
        let mut status = 0u8;
        unsafe { Rows::hip(hipMemcpy(&mut status as *mut u8 as *mut c_void, self.rows.status as *const c_void, 1, HIP_MEMCPY_DEVICE_TO_HOST)); }
        self.bytes_out += 1;
        assert!(status != PCP_STATUS_HULL, "a bound left the hull of the root's domains");
        if status == PCP_FALSE {
            return SKleene::False; // nothing to fetch; the current rows still hold this node as it came in
        }
        let (mut nl, mut nu) = (vec![0i32; n], vec![0i32; n]);
        unsafe {
            Rows::hip(hipMemcpy(nl.as_mut_ptr() as *mut c_void, self.rows.lb[other] as *const c_void, n * 4, HIP_MEMCPY_DEVICE_TO_HOST));
            Rows::hip(hipMemcpy(nu.as_mut_ptr() as *mut c_void, self.rows.ub[other] as *const c_void, n * 4, HIP_MEMCPY_DEVICE_TO_HOST));
            if words > 0 {
                Rows::hip(hipMemcpy(active.as_mut_ptr() as *mut c_void, self.rows.act[other] as *const c_void, words * 8, HIP_MEMCPY_DEVICE_TO_HOST));
            }
        }
        self.bytes_out += 8 * n as u64 + 8 * words as u64;
        self.rows.cur = other; // the output pair is the node now
        // post-conditions of Store::consistency: narrowed domains through MonotonicUpdate::update (the trail records the old values,
        // variable/memory/trail_memory.rs:100-104), the event delta left drained, entailed units out of `active`
        for i in 0..n {
            if nl[i] != lb[i] || nu[i] != ub[i] {
                let ok = vstore.update(&Identity::new(i), Interval::new(nl[i], nu[i]));
                debug_assert!(ok);
            }
        }
        let _ = vstore.drain_delta().count();
        for (k, &(u, _)) in g.dev_units.iter().enumerate() {
            if (active[k >> 6] >> (k & 63)) & 1 == 0 { g.cpu.deactivate(u); }
        }
        self.mirror = Some((nl, nu, active));
        match status { PCP_TRUE => SKleene::True, _ => SKleene::Unknown }
    }
}

// The rest of IntCStore (concept.rs:120-138) goes through `inner`; the device rows and the mirror are a cache, not state: a clone or a
// restored store starts without them and uploads its first node whole.
impl<VStore> Collection for ResidentCStore<VStore> where GpuCStore<VStore>: Collection { type Item = <GpuCStore<VStore> as Collection>::Item; }
impl<VStore> AssociativeCollection for ResidentCStore<VStore> where GpuCStore<VStore>: AssociativeCollection {
    type Location = <GpuCStore<VStore> as AssociativeCollection>::Location;
}
impl<VStore> Alloc for ResidentCStore<VStore> where GpuCStore<VStore>: Alloc {
    fn alloc(&mut self, p: Self::Item) -> Self::Location { self.inner.alloc(p) }
}
impl<VStore> Empty for ResidentCStore<VStore> where CStoreFD<VStore>: Empty {
    fn empty() -> Self { ResidentCStore::new(0) }
}
impl<VStore> Clone for ResidentCStore<VStore> where GpuCStore<VStore>: Clone {
    fn clone(&self) -> Self { ResidentCStore { inner: self.inner.clone(), rows: Rows::none(), mirror: None, bytes_in: 0, bytes_out: 0 } }
}
impl<VStore> Freeze for ResidentCStore<VStore> where CStoreFD<VStore>: Freeze {
    type FrozenState = FrozenResidentCStore<VStore>;
    fn freeze(self) -> Self::FrozenState {
        let ResidentCStore { inner, rows, mirror, bytes_in, bytes_out } = self;
        FrozenResidentCStore { inner: inner.freeze(), rows, mirror, bytes_in, bytes_out }
    }
}
pub struct FrozenResidentCStore<VStore> where CStoreFD<VStore>: Freeze {
    inner: FrozenGpuCStore<VStore>,
    rows: Rows,
    mirror: Option<(Vec<i32>, Vec<i32>, Vec<u64>)>,
    bytes_in: u64,
    bytes_out: u64,
}
impl<VStore> Snapshot for FrozenResidentCStore<VStore> where CStoreFD<VStore>: Freeze {
    type Label = <FrozenGpuCStore<VStore> as Snapshot>::Label;
    type State = ResidentCStore<VStore>;
    fn label(&mut self) -> Self::Label { self.inner.label() }
    fn restore(self, label: Self::Label) -> Self::State {
        // the vstore is restored separately (Space::restore, search/space.rs): the mirror stays valid as a description of the DEVICE
        // rows — the next consistency() diffs the restored vstore against it and uploads the difference
        let FrozenResidentCStore { inner, rows, mirror, bytes_in, bytes_out } = self;
        ResidentCStore { inner: inner.restore(label), rows, mirror, bytes_in, bytes_out }
    }
}
impl<VStore> DisplayStateful<Model> for ResidentCStore<VStore> where GpuCStore<VStore>: DisplayStateful<Model> {
    fn display(&self, model: &Model) { self.inner.display(model) }
}
impl<VStore> DisplayStateful<(Model, VStore)> for ResidentCStore<VStore> where GpuCStore<VStore>: DisplayStateful<(Model, VStore)> {
    fn display(&self, state: &(Model, VStore)) { self.inner.display(state) }
}
