#!/bin/sh
# The one in-crate change libpcp needs so that a foreign constraint store can see (kind, variable, offset) of a propagator:
# views and propagators keep their fields private (x_neq_y.rs:27-30, x_less_y.rs:28-31) and dependencies() only reveals
# variable indices.  Append-only edits of a checkout of ptal/pcp (no reference source is reproduced here):
#   * new file  src/libpcp/propagation/lower.rs : the trait `Lower` and the neutral description types
#   * appended  impl blocks next to each view and propagator (inside their own module, where the fields are visible)
#   * appended  four accessors on propagation::store::Store used by GpuCStore
# usage: sh apply_lower_hook.sh /path/to/pcp
set -e
P="$1/src/libpcp"
test -d "$P" || { echo "usage: $0 /path/to/pcp"; exit 1; }

cat > "$P/propagation/lower.rs" <<'RS'
//! Lowering hook for foreign constraint stores (pcp-gpu-cstore): a neutral description of what a propagator computes.
#[derive(Clone, Copy, Debug)]
pub enum Operand { Var(usize, i32 /* Addition offset */), Const(i32) }
#[derive(Clone, Debug)]
pub struct PropDesc { pub kind: u8 /* pcp_kind of pcp_hip.h */, pub ops: Vec<Operand> }
/// `None` = this propagator has no lowering: the store that asked keeps the space on the CPU path.
pub trait Lower { fn lower(&self) -> Option<Vec<PropDesc>> { None } }
/// Views lower to one operand (Identity -> Var(idx, 0); Addition(x, v) -> x shifted by v; Constant(c) -> Const(c)).
pub trait LowerView { fn lower_view(&self) -> Option<Operand> { None } }
pub fn shift(op: Operand, v: i32) -> Operand { match op { Operand::Var(i, o) => Operand::Var(i, o + v), Operand::Const(c) => Operand::Const(c + v) } }
RS
echo 'pub mod lower;' >> "$P/propagation/mod.rs"

cat >> "$P/term/identity.rs" <<'RS'
impl<Domain> ::propagation::lower::LowerView for Identity<Domain> {
    fn lower_view(&self) -> Option<::propagation::lower::Operand> { Some(::propagation::lower::Operand::Var(self.index(), 0)) }
}
RS
cat >> "$P/term/constant.rs" <<'RS'
impl ::propagation::lower::LowerView for Constant<i32> {
    fn lower_view(&self) -> Option<::propagation::lower::Operand> { Some(::propagation::lower::Operand::Const(self.value)) }
}
RS
cat >> "$P/term/addition.rs" <<'RS'
impl<VStore> ::propagation::lower::LowerView for Addition<VStore, i32> {
    fn lower_view(&self) -> Option<::propagation::lower::Operand> { self.x.lower_view().map(|o| ::propagation::lower::shift(o, self.v)) }
}
RS
# binary kinds: kind code, file, type
for spec in "0 x_neq_y XNeqY" "1 x_eq_y XEqY" "2 x_less_y XLessY"; do
  set -- $spec
  cat >> "$P/propagators/cmp/$2.rs" <<RS
impl<VStore> ::propagation::lower::Lower for $3<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        Some(vec![::propagation::lower::PropDesc { kind: $1, ops: vec![self.x.lower_view()?, self.y.lower_view()?] }])
    }
}
RS
done
for spec in "3 x_less_y_plus_z XLessYPlusZ" "4 x_greater_y_plus_z XGreaterYPlusZ" "6 x_eq_y_mul_z XEqYMulZ"; do
  set -- $spec
  cat >> "$P/propagators/cmp/$2.rs" <<RS
impl<VStore> ::propagation::lower::Lower for $3<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        Some(vec![::propagation::lower::PropDesc { kind: $1, ops: vec![self.x.lower_view()?, self.y.lower_view()?, self.z.lower_view()?] }])
    }
}
RS
done
cat >> "$P/propagators/cmp/x_eq_y_plus_z.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for XEqYPlusZ<VStore> {
    // geq = XGreaterYPlusZ(x + 1, y, z): undo the +1 to recover x (x_eq_y_plus_z.rs:36-41)
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        let g = self.geq.lower()?.pop()?;
        Some(vec![::propagation::lower::PropDesc { kind: 5, ops: vec![::propagation::lower::shift(g.ops[0], -1), g.ops[1], g.ops[2]] }])
    }
}
RS
cat >> "$P/logic/conjunction.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for Conjunction<VStore> {
    // one unit of several elementary members (only flat conjunctions of lowerable members)
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> {
        let mut out = vec![];
        for f in &self.fs { let mut d = f.lower()?; if d.len() != 1 { return None; } out.push(d.pop()?); }
        Some(out)
    }
}
RS
cat >> "$P/propagators/distinct.rs" <<'RS'
impl<VStore> ::propagation::lower::Lower for Distinct<VStore> {
    fn lower(&self) -> Option<Vec<::propagation::lower::PropDesc>> { self.conj.lower() }
}
RS
cat >> "$P/propagation/store.rs" <<'RS'
// Accessors for foreign constraint stores (pcp-gpu-cstore): read-only views of `propagators` and `active`.
impl<VStore, Event, R, S> Store<VStore, Event, R, S> {
    pub fn propagators_len(&self) -> usize { self.propagators.len() }
    pub fn propagator(&self, idx: usize) -> &Box<dyn PropagatorConcept<VStore, Event>> { &self.propagators[idx] }
    pub fn is_active(&self, idx: usize) -> bool { self.active.contains(idx) }
    pub fn deactivate(&mut self, idx: usize) { self.active.remove(idx); }
}
RS
echo "done: add  + ::propagation::lower::Lower  to the PropagatorConcept bounds (propagation/concept.rs:21-53) and"
echo "           + ::propagation::lower::LowerView  to the view concept (term/ops.rs), give Boolean/Disjunction/Sum the default impls"
