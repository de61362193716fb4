#!/bin/sh
# The one in-crate change libpcp needs so that a foreign constraint store can see (kind, variable, offset) of a propagator:
# views and propagators keep their fields private (x_neq_y.rs:27-30, x_less_y.rs:28-31) and dependencies() only reveals
# variable indices.  Scripted edits of a checkout of ptal/pcp (no reference source is reproduced here):
#   1. new file  src/libpcp/propagation/lower.rs : the traits `Lower` / `LowerView` and the neutral description types
#   2. sed       `LowerView` becomes a SUPERTRAIT of IntVariable_ (concept.rs:79-99) and `Lower` of PropagatorConcept_
#                (propagation/concept.rs:21-41) — both the trait and its blanket impl — so that `self.x.lower_view()` resolves on
#                `Box<dyn IntVariable<VStore>>` by auto-deref exactly as `self.x.read(store)` does, and `p.lower()` on `Box<dyn PropagatorConcept>`
#   3. appended  impl blocks next to EVERY view and propagator type (inside their own module, where the fields are visible); types
#                without a lowering get the default (`None`: the space stays on the CPU path)
#   4. appended  four accessors on propagation::store::Store used by GpuCStore
# NEVER RUN where this repository is built (no cargo / rustc in the image): uncompiled.
# usage: sh apply_lower_hook.sh /path/to/pcp
set -e
ROOT="$1"
P="$ROOT/src/libpcp"
test -d "$P" || { echo "usage: $0 /path/to/pcp"; exit 1; }
grep -q 'pub mod lower;' "$P/propagation/mod.rs" && { echo "already applied"; exit 0; }

# ---- 1. the hook ------------------------------------------------------------------------------------------------------
cat > "$P/propagation/lower.rs" <<'RS'
//! Lowering hook for foreign constraint stores (pcp-gpu-cstore): a neutral description of what a propagator computes.
use std::any::Any;
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Operand { Var(usize, i32 /* Addition offset */), Const(i32) }
#[derive(Clone, Debug)]
pub struct PropDesc { pub kind: u8 /* pcp_kind of pcp_hip.h */, pub ops: Vec<Operand> }
/// A formula unit (logic/): a tree of Conjunction / Disjunction nodes over elementary leaves — what pcp_model_push_formula takes.
#[derive(Clone, Debug)]
pub enum Tree { Leaf(PropDesc), And(Vec<Tree>), Or(Vec<Tree>) }
/// `lower`: the unit as elementary members of ONE flat conjunction (one member = a plain propagator).  `lower_tree`: the unit as a
/// formula tree.  Both `None` = this propagator has no lowering: the store that asked keeps the space on the CPU path.
pub trait Lower {
    fn lower(&self) -> Option<Vec<PropDesc>> { None }
    fn lower_tree(&self) -> Option<Tree> {
        let mut d = self.lower()?;
        if d.len() == 1 { d.pop().map(Tree::Leaf) } else { Some(Tree::And(d.into_iter().map(Tree::Leaf).collect())) }
    }
}
/// Views lower to one operand (Identity -> Var(idx, 0); Addition(x, v) -> x shifted by v; Constant(c) -> Const(c)).
pub trait LowerView { fn lower_view(&self) -> Option<Operand> { None } }
pub fn shift(op: Operand, v: i32) -> Operand { match op { Operand::Var(i, o) => Operand::Var(i, o + v), Operand::Const(c) => Operand::Const(c + v) } }
/// Bounds are generic in libpcp (IntBound); the device engine computes in i32: any other bound type has no lowering.
pub fn as_i32<B: Any>(b: &B) -> Option<i32> { (b as &dyn Any).downcast_ref::<i32>().cloned() }
impl<R: LowerView + ?Sized> LowerView for Box<R> { fn lower_view(&self) -> Option<Operand> { (**self).lower_view() } }
impl<R: Lower + ?Sized> Lower for Box<R> {
    fn lower(&self) -> Option<Vec<PropDesc>> { (**self).lower() }
    fn lower_tree(&self) -> Option<Tree> { (**self).lower_tree() }
}
RS
echo 'pub mod lower;' >> "$P/propagation/mod.rs"

# ---- 2. supertrait bounds -----------------------------------------------------------------------------------------------
# concept.rs: `pub trait IntVariable_<VStore>: ... + DisplayStateful<Model>` and its blanket impl `R: Debug + DisplayStateful<Model>,`
sed -i '/^pub trait IntVariable_<VStore>:/,/^{/ s/^    + DisplayStateful<Model>$/    + DisplayStateful<Model>\n    + ::propagation::lower::LowerView/' "$P/concept.rs"
sed -i '/^impl<R, VStore> IntVariable_<VStore> for R/,/^{/ s/^    R: Debug + DisplayStateful<Model>,$/    R: Debug + DisplayStateful<Model>,\n    R: ::propagation::lower::LowerView,/' "$P/concept.rs"
# propagation/concept.rs: `pub trait PropagatorConcept_<VStore, Event>: ... + NotFormula<VStore>` and `R: NotFormula<VStore>,`
sed -i '/^pub trait PropagatorConcept_<VStore, Event>:/,/^{/ s/^    + NotFormula<VStore>$/    + NotFormula<VStore>\n    + ::propagation::lower::Lower/' "$P/propagation/concept.rs"
sed -i '/^impl<VStore, Event, R> PropagatorConcept_<VStore, Event> for R/,/^{/ s/^    R: NotFormula<VStore>,$/    R: NotFormula<VStore>,\n    R: ::propagation::lower::Lower,/' "$P/propagation/concept.rs"
grep -q 'lower::LowerView' "$P/concept.rs" || { echo "concept.rs: the IntVariable_ bounds were not found where expected (concept.rs:79-99)"; exit 1; }
grep -q 'lower::Lower' "$P/propagation/concept.rs" || { echo "propagation/concept.rs: the PropagatorConcept_ bounds were not found (concept.rs:21-41)"; exit 1; }

# ---- 3. views -----------------------------------------------------------------------------------------------------------
cat >> "$P/term/identity.rs" <<'RS'
impl<Domain> ::propagation::lower::LowerView for Identity<Domain> {
    fn lower_view(&self) -> Option<::propagation::lower::Operand> { Some(::propagation::lower::Operand::Var(self.index(), 0)) }
}
RS
cat >> "$P/term/constant.rs" <<'RS'
impl<V: 'static> ::propagation::lower::LowerView for Constant<V> {
    fn lower_view(&self) -> Option<::propagation::lower::Operand> { ::propagation::lower::as_i32(&self.value).map(::propagation::lower::Operand::Const) }
}
RS
cat >> "$P/term/addition.rs" <<'RS'
impl<VStore, Domain, Bound> ::propagation::lower::LowerView for Addition<VStore>
where
    VStore: VStoreConcept<Item = Domain>,
    Domain: Collection<Item = Bound>,
    Bound: 'static,
{
    // `self.x` is a Box<dyn IntVariable<VStore>>: lower_view() resolves through the supertrait added above
    fn lower_view(&self) -> Option<::propagation::lower::Operand> {
        let v = ::propagation::lower::as_i32(&self.v)?;
        self.x.lower_view().map(|o| ::propagation::lower::shift(o, v))
    }
}
RS
cat >> "$P/term/sum.rs" <<'RS'
// A Sum view is registered with the engine by pcp_model_push_sum; the shim keeps models with Sums on the CPU path until
// `Operand` grows a Sum case (the C++ twin pcp_host.hpp and the Python mirror already lower them).
impl<VStore> ::propagation::lower::LowerView for Sum<VStore> {}
RS

# ---- 3. propagators -----------------------------------------------------------------------------------------------------
# binary kinds: kind code, file, type
for spec in "0 x_neq_y XNeqY" "1 x_eq_y XEqY" "2 x_less_y XLessY"; do
  set -- $spec
  cat >> "$P/propagators/cmp/$2.rs" <<RS
impl<VStore: Collection> ::propagation::lower::Lower for $3<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        Some(vec![::propagation::lower::PropDesc { kind: $1, ops: vec![self.x.lower_view()?, self.y.lower_view()?] }])
    }
}
RS
done
for spec in "3 x_less_y_plus_z XLessYPlusZ" "4 x_greater_y_plus_z XGreaterYPlusZ" "6 x_eq_y_mul_z XEqYMulZ"; do
  set -- $spec
  cat >> "$P/propagators/cmp/$2.rs" <<RS
impl<VStore: Collection> ::propagation::lower::Lower for $3<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        Some(vec![::propagation::lower::PropDesc { kind: $1, ops: vec![self.x.lower_view()?, self.y.lower_view()?, self.z.lower_view()?] }])
    }
}
RS
done
cat >> "$P/propagators/cmp/x_eq_y_plus_z.rs" <<'RS'
impl<VStore: Collection> ::propagation::lower::Lower for XEqYPlusZ<VStore> {
    // geq = XGreaterYPlusZ(x + 1, y, z): undo the +1 to recover x (x_eq_y_plus_z.rs:36-41)
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        let g = self.geq.lower()?.pop()?;
        Some(vec![::propagation::lower::PropDesc { kind: 5, ops: vec![::propagation::lower::shift(g.ops[0], -1), g.ops[1], g.ops[2]] }])
    }
}
RS
cat >> "$P/logic/conjunction.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for Conjunction<VStore> {
    // one unit of several elementary members (only flat conjunctions of lowerable members; `f` is a Box<dyn PropagatorConcept>:
    // lower() resolves through the supertrait added above)
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        let mut out = vec![];
        for f in &self.fs { let mut d = f.lower()?; if d.len() != 1 { return None; } out.push(d.pop()?); }
        Some(out)
    }
    // a conjunction over arbitrary formulas (conjunction.rs:77-119): an AND node
    fn lower_tree(&self) -> Option<::propagation::lower::Tree> {
        let mut out = vec![];
        for f in &self.fs { out.push(f.lower_tree()?); }
        Some(::propagation::lower::Tree::And(out))
    }
}
RS
cat >> "$P/propagators/distinct.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for Distinct<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> { self.conj.lower() }
}
RS
cat >> "$P/propagators/all_equal.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for AllEqual<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> { self.conj.lower() }
}
RS
# the reified layer: formula units (pcp_model_push_formula); NotFormula::not was applied when the formula was built (logic/ops.rs:17-19)
cat >> "$P/logic/disjunction.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for Disjunction<VStore> {
    // disjunction.rs:78-141: an OR node (no flat form)
    fn lower_tree(&self) -> Option<::propagation::lower::Tree> {
        let mut out = vec![];
        for f in &self.fs { out.push(f.lower_tree()?); }
        Some(::propagation::lower::Tree::Or(out))
    }
}
RS
cat >> "$P/logic/boolean.rs" <<'RS'
impl<VStore: Collection> ::propagation::lower::LowerView for Boolean<VStore> {
    fn lower_view(&self) -> Option<::propagation::lower::Operand> { self.var.lower_view() }   // a Boolean reads as its 0/1 variable (boolean.rs:81-103)
}
impl<VStore: Collection> ::propagation::lower::Lower for Boolean<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {                            // the formula "var = 1" (boolean.rs:111-140): PCP_BOOL
        Some(vec![::propagation::lower::PropDesc { kind: 7, ops: vec![self.var.lower_view()?] }])
    }
}
RS
cat >> "$P/logic/boolean_neg.rs" <<'RS'
impl<VStore: Collection> ::propagation::lower::Lower for BooleanNeg<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {                            // "var = 0" (boolean_neg.rs:71-96): PCP_NBOOL
        use ::propagation::lower::LowerView;
        Some(vec![::propagation::lower::PropDesc { kind: 8, ops: vec![self.b.lower_view()?] }])
    }
}
RS
cat >> "$P/propagators/cumulative.rs" <<'RS'
// Cumulative is a model builder (join allocates propagators), not a PropagatorConcept: nothing to lower here.
RS

# ---- 4. store accessors -------------------------------------------------------------------------------------------------
cat >> "$P/propagation/store.rs" <<'RS'
// Accessors for foreign constraint stores (pcp-gpu-cstore): read-only views of `propagators` and `active`.
impl<VStore, Event, R, S> Store<VStore, Event, R, S> {
    pub fn propagators_len(&self) -> usize { self.propagators.len() }
    pub fn propagator(&self, idx: usize) -> &Box<dyn PropagatorConcept<VStore, Event>> { &self.propagators[idx] }
    pub fn is_active(&self, idx: usize) -> bool { self.active.contains(idx) }
    pub fn deactivate(&mut self, idx: usize) { self.active.remove(idx); }
}
RS
echo "applied: cargo build in $ROOT must still pass (the edits add trait bounds every in-tree type now satisfies)"
