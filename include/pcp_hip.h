/* pcp_hip.h — C ABI of libpcp_hip.so, the MI355X (gfx950) propagation-fixpoint engine.
 *
 * Drop-in boundary.  The reference (ptal/pcp, libpcp 0.7.0, pure Rust) has no FFI;
 * the seam it offers is the trait
 *     kernel::Consistency<VStore>::consistency(&mut self, &mut VStore) -> SKleene
 *                                                    (src/libpcp/kernel/consistency.rs:17-19)
 * as implemented by propagation::store::Store       (src/libpcp/propagation/store.rs:240-258)
 * and required of any constraint store by IntCStore  (src/libpcp/concept.rs:120-138).
 * Every entry point below is what a Rust `GpuCStore: IntCStore<VStore>` would bind through
 * `extern "C"` (the binding stub is in INTEGRATION.md); each cites the reference item it replaces.
 *
 * Conventions: plain pointers and sizes; no exceptions or aborts cross the boundary; every call
 * returns PCP_OK (0) or a negative pcp_err.  Where the reference panics on a contract violation
 * (assert!), this library returns PCP_ERR_CONTRACT.  A pcp_ctx is used from one host thread at a
 * time; distinct contexts are independent.  There is NO CPU fallback: if no HIP device is present
 * pcp_ctx_create fails with PCP_ERR_NODEVICE.
 */
#ifndef PCP_HIP_H
#define PCP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCP_ABI_VERSION 8

/* Operand encodings for pcp_prop.var[i]. */
#define PCP_CONST 0xFFFFFFFFu /* operand is a term::Constant (term/constant.rs:43-68); off[i] = its value   */
#define PCP_NOVAR 0xFFFFFFFEu /* operand slot unused (var[2] of the binary kinds)                          */
#define PCP_SUM 0xC0000000u   /* var[i] = PCP_SUM | t: the operand is the term::Sum view number t (term/sum.rs:56-92,
                                 registered with pcp_model_push_sum); off[i] adds a constant (Addition over the Sum, and
                                 the constants / Addition offsets of its members folded together)                    */

/* Interval bounds and offsets must stay inside [-PCP_BOUND_MAX, PCP_BOUND_MAX] so that no filter can
 * overflow i32 (the reference wraps in release and panics in debug; SURVEY.md §7 "i32 overflow"). */
#define PCP_BOUND_MAX 0x1FFFFFFF

typedef enum {
  PCP_OK = 0,
  PCP_ERR_ARG = -1,         /* null pointer, bad size, bad enum                                         */
  PCP_ERR_CONTRACT = -2,    /* a reference assert! would fire: empty initial domain (variable/store.rs:136),
                               var >= n_vars (variable/store.rs:176-179), same var twice in one propagator
                               (reactors/indexed_deps.rs:69-77), bound/offset out of PCP_BOUND_MAX        */
  PCP_ERR_HIP = -3,         /* a HIP runtime call failed; pcp_last_error() has the text                   */
  PCP_ERR_NOMEM = -4,
  PCP_ERR_UNSUPPORTED = -5, /* e.g. XEqYMulZ over set-mode domains, a set-mode store larger than one CU's LDS    */
  PCP_ERR_NODEVICE = -6     /* no HIP device: the engine never falls back to a CPU path                   */
} pcp_err;

/* Elementary propagator kinds.  X, Y, Z are views: Identity(var) (term/identity.rs:47-70),
 * Addition(var, off) (term/addition.rs:80-110) or Constant(off) (term/constant.rs:43-68). */
typedef enum {
  PCP_NEQ = 0,  /* XNeqY            propagators/cmp/x_neq_y.rs:66-104                                   */
  PCP_EQ = 1,   /* XEqY             propagators/cmp/x_eq_y.rs:67-116                                    */
  PCP_LT = 2,   /* XLessY           propagators/cmp/x_less_y.rs:67-117 (x>y, x>=y, x<=y are ctor sugar,
                                    propagators/cmp/mod.rs:34-60)                                        */
  PCP_LT3 = 3,  /* XLessYPlusZ      propagators/cmp/x_less_y_plus_z.rs:75-128                           */
  PCP_GT3 = 4,  /* XGreaterYPlusZ   propagators/cmp/x_greater_y_plus_z.rs:75-128                        */
  PCP_EQ3 = 5,  /* XEqYPlusZ = GT3(x+1,y,z) && LT3(x-1,y,z), one propagator
                                    propagators/cmp/x_eq_y_plus_z.rs:31-105                              */
  PCP_MUL3 = 6, /* XEqYMulZ         propagators/cmp/x_eq_y_mul_z.rs:68-115 (narrows x only)             */
  PCP_BOOL = 7, /* logic::Boolean   logic/boolean.rs:111-140: the formula "X = 1" over a 0/1 view (one operand)   */
  PCP_NBOOL = 8 /* logic::BooleanNeg logic/boolean_neg.rs:71-96: "X = 0"                                          */
} pcp_kind;

/* The reified layer (logic/): ONE unit = a tree of Conjunction (logic/conjunction.rs:77-119) and Disjunction
 * (logic/disjunction.rs:78-141) nodes over elementary leaves.  NotFormula::not is applied by the caller when the tree is built, as
 * the reference does (implication f g = Disjunction[f, g.not()], logic/mod.rs:30-36; equivalence = Conjunction of both
 * implications, logic/mod.rs:38-45): the engine only sees positive trees. */
typedef enum { PCP_F_LEAF = 0, PCP_F_AND = 1, PCP_F_OR = 2 } pcp_fnode_type;
typedef struct {
  uint8_t type;        /* pcp_fnode_type */
  uint8_t reserved;    /* must be 0 */
  uint16_t n_children; /* inner nodes: >= 1 */
  uint32_t first;      /* leaf: index into `leaves`; inner node: index in `nodes` of its first child (children are consecutive) */
} pcp_fnode;

/* SKleene as returned by Consistency::consistency (trilean::SKleene; propagation/store.rs:247-257). */
typedef enum { PCP_FALSE = 0, PCP_TRUE = 1, PCP_UNKNOWN = 2 } pcp_status;

/* One elementary filter.  A *unit* is what the reference stores as ONE Box<dyn PropagatorConcept>
 * (one index in Store::propagators, one bit of Store::active, propagation/store.rs:33-38):
 *   group_kind == 0 : the prop is a unit by itself;
 *   group_kind == 1 : consecutive props OF ONE pcp_model_push_props CALL carrying the same `group` value form ONE unit — a
 *                     logic::Conjunction (logic/conjunction.rs:77-119);
 *   group_kind == 2 : the same, built by Distinct::new (propagators/distinct.rs:63-83); it differs from
 *                     1 only in the ORDER of its reactor subscriptions (distinct.rs:117-124), which no
 *                     fixpoint depends on.  join_distinct (distinct.rs:26-45) instead pushes its pairs
 *                     as standalone units.
 * Units are numbered in push order; bit u of an `active` row is unit u. */
typedef struct {
  uint8_t kind;       /* pcp_kind */
  uint8_t group_kind; /* 0 standalone, 1 Conjunction member, 2 Distinct member */
  uint16_t reserved;  /* must be 0 */
  uint32_t group;     /* Conjunction id (only compared for equality with the previous prop) */
  uint32_t var[3];    /* variable index, PCP_CONST, or PCP_NOVAR */
  int32_t off[3];     /* Addition offset (0 = Identity) or the Constant's value */
} pcp_prop;

/* Counters.  A *filter step* = one evaluation of one elementary propagator's propagate()+is_subsumed()
 * (one scheduler pop, propagation/store.rs:166-175).  The reference keeps no such
 * counter.  Step counts are schedule-dependent: compare rates, never counts.
 * `steps`/`steps3` are REFERENCE-EQUIVALENT steps: every (propagator, node) pair the reference's scheduler would pop —
 * each live propagator once in the initial sweep (init_scheduler, store.rs:144-149) plus every wake-up.  The engine
 * proves most of the sweep's pairs no-ops in bulk (whole 64-propagator words from range tables of the tile's
 * domains) without looking at them; `evaluated` counts the pairs that WERE looked at one by one — tested on the
 * node's own domains (sweep level 1 / level 2, full filter) or run by a wake-up round — and `full_evals` the pairs
 * that ran the full filter (propagate() + is_subsumed() literally, with its domain writes). */
typedef struct {
  uint64_t steps;        /* reference-equivalent filter steps on binary kinds              */
  uint64_t steps3;       /* the same on ternary kinds (LT3/GT3/EQ3/MUL3)                   */
  uint64_t narrowings;   /* domain writes that strictly shrank a domain                    */
  uint64_t waves;        /* sum over nodes of fixpoint waves (>=1 per node)                */
  uint64_t failed_nodes; /* nodes that ended PCP_FALSE                                     */
  uint64_t nodes;        /* nodes propagated                                               */
  uint64_t evaluated;    /* (propagator, node) pairs tested individually on the node's domains */
  uint64_t full_evals;   /* pairs that ran the full filter                                 */
} pcp_stats;

typedef struct pcp_ctx pcp_ctx;

/* ---- context ------------------------------------------------------------------------------------ */
/* One context per CStore (the reference builds one Store per Space, search/space.rs:21-36). */
int32_t pcp_ctx_create(int32_t hip_device, pcp_ctx** out);
void pcp_ctx_destroy(pcp_ctx* ctx);
const char* pcp_last_error(const pcp_ctx* ctx); /* text of the last failing call on ctx (never NULL) */
const char* pcp_strerror(int32_t err);
uint32_t pcp_abi_version(void);

/* ---- model (≡ the immutable part of Store: `propagators`) ------------------------------------------ */
/* ≡ Store::empty() (propagation/store.rs:40-54) over a VStore of n_vars variables.
 *   set_words == 0 : Interval<i32> domains (VStoreFD, variable/mod.rs:36): bounds reasoning only.
 *   set_words  > 0 : IntervalSet<i32> domains (VStoreSet, variable/mod.rs:38 — the reference's default FDSpace,
 *                    search/mod.rs:41-43): every domain is a SET, carried as `set_words` u64 words per variable; value v is
 *                    bit (v - lo) where [lo, hi] is the hull declared with pcp_model_set_hull — REQUIRED in set mode, with
 *                    hi - lo < 64 * set_words.  XNeqY removes interior values, XEqY intersects sets, `active`/True follow
 *                    set disjointness.
 *   WHAT SET MODE REFUSES (PCP_ERR_UNSUPPORTED from pcp_model_push_props / _push_sum / _push_formula), and why:
 *     - XEqYMulZ over sets: the reference computes y.read() * z.read() with IntervalSet's Mul (x_eq_y_mul_z.rs:68-115), which lives in the
 *       third-party crate intervallum (not vendored in the reference tree, version pinned only by Cargo.lock); no test of the reference
 *       exercises it over IntervalSet, so neither the oracle nor this engine has anything to pin a set-valued product against.  The
 *       Interval<i32> product (bounds only) IS pinned (5 vectors, x_eq_y_mul_z.rs tests) and is what interval mode runs.
 *     - Sum views over sets (term/sum.rs:56-92): Sum::read adds IntervalSets member by member — the same unpinned third-party algebra
 *       (IntervalSet + IntervalSet); the reference's only Sum users (cumulative.rs) are tested over Interval stores.
 *     - the reified layer over sets (Boolean / BooleanNeg / formula units): Disjunction::propagate and is_subsumed are domain-agnostic, but
 *       every reference test of logic/ runs over VStoreFD (Interval); formula leaves would reuse the set-mode leaf filters (XNeqY / XEqY /
 *       XLessY / LT3 / GT3 / EQ3 are offered over sets as plain units), the kernel that combines them (pcp_formula.hip) reads 8-byte
 *       interval cells only.  Refused rather than approximated: bounds-only leaves under a set-valued store would disagree with the
 *       reference on interior values.
 *   ENUMERATE (search/branching/enumerate.rs:33-60: children `x = v` and `x != v`) needs no engine support in set mode — both children are
 *   exact set operations the caller applies to `bits` before propagating (keep bit v / clear bit v), as pcp_branch_device_set does for
 *   BinarySplit.  In interval mode `x != v` with v inside (lb, ub) is not a domain operation: it stays a per-node propagator — a unit
 *   pushed with pcp_model_push_props before the node's call and removed with pcp_model_truncate after it (what the host twins do for
 *   every branch constraint), or, for a BATCH of nodes with different such propagators each, pcp_propagate_device_units (ABI v8). */
int32_t pcp_model_reset(pcp_ctx* ctx, uint32_t n_vars, uint32_t set_words);
/* ≡ Store::alloc, append-only (propagation/store.rs:223-230). */
int32_t pcp_model_push_props(pcp_ctx* ctx, uint32_t n, const pcp_prop* props);
/* Registers a term::Sum view over n_members variables (term/sum.rs:30-32) and returns its number in *term; a prop refers to
 * it as var[i] = PCP_SUM | term.  read = the interval sum of the members (sum.rs:76-81); an update through a Sum of several
 * variables never narrows anything — it fails iff the new domain does not overlap the sum (sum.rs:66-69); a Sum of ONE
 * variable forwards to it (sum.rs:63-64).  The propagator depends on every member (sum.rs:85-91), so naming a variable twice
 * in one propagator — directly or through its Sums — is PCP_ERR_CONTRACT like the reference's reactor panic.  Interval mode
 * only.  pcp_model_reset forgets the terms. */
int32_t pcp_model_push_sum(pcp_ctx* ctx, uint32_t n_members, const uint32_t* vars, uint32_t* term);
/* ≡ Store::alloc of ONE formula propagator: nodes[0] is the root, at most 8 levels deep; leaves are elementary props (group fields
 * ignored; Sum operands allowed).  Its dependencies are the sorted, de-duplicated union of its leaves' (conjunction.rs:107-118,
 * disjunction.rs:119-131), so a variable may occur in several leaves.  A pop of the unit runs Disjunction::propagate literally: a
 * child is propagated only when every other child is disentailed.  Where the reference would PANIC — Boolean::propagate on a domain
 * without 1 reached through a Conjunction (a non-monotonic update, variable/store.rs:153-156) — the node fails instead.
 * Interval mode only.  Models with formula units run the general formula kernel (pcp_formula.hip). */
int32_t pcp_model_push_formula(pcp_ctx* ctx, uint32_t n_nodes, const pcp_fnode* nodes, uint32_t n_leaves, const pcp_prop* leaves);
/* ≡ FrozenStore::restore's `propagators.truncate(label.0)` (propagation/store.rs:319-323); n_units counts
 * units, not elementary props. */
int32_t pcp_model_truncate(pcp_ctx* ctx, uint32_t n_units);
int32_t pcp_model_n_units(const pcp_ctx* ctx, uint32_t* n_units, uint32_t* n_props);
/* Optional: [lo, hi] = hull of the initial domains the variables were allocated with (VStore::alloc,
 * variable/store.rs:130-139).  Updates are monotone (variable/store.rs:156-170), so every domain of every node of the
 * search lies inside it.  With a hull within +-16383 the engine keeps the domains as 16-bit cells (twice the nodes per
 * workgroup) without the second, 32-bit launch it otherwise queues behind every such call for nodes that do not fit.
 * A node whose bounds leave the declared hull is a contract violation: pcp_propagate returns PCP_ERR_CONTRACT;
 * pcp_propagate_device leaves that node's outputs untouched, sets its status to PCP_STATUS_HULL and raises a sticky device
 * flag: the next pcp_stats_read returns PCP_ERR_CONTRACT however many launches happened in between (pcp_branch_device counts
 * such nodes in counts[4]).  pcp_model_reset forgets the hull.
 * The same refusal (status PCP_STATUS_HULL, outputs untouched, sticky flag) meets a node of pcp_propagate_device with a bound
 * beyond +-PCP_BOUND_MAX, hull or not: every tile checks the bounds it stages. */
int32_t pcp_model_set_hull(pcp_ctx* ctx, int32_t lo, int32_t hi);
#define PCP_STATUS_HULL 0xFE

/* ---- propagation (≡ Consistency::consistency, propagation/store.rs:247-257) ---------------------------- */
/* Host-buffer form: n_nodes independent spaces sharing the model.
 *   lb, ub  : [n_nodes][n_vars] i32, node-major, in/out (Interval<i32> bounds).  Set mode: OUT only (bounds of the sets).
 *   bits    : interval mode: must be NULL.  Set mode: [n_nodes][n_vars][set_words] u64 in/out, the domains.
 *   active  : [n_nodes][ceil(n_units/64)] u64 in/out; bit u == Store::active of unit u.  NULL = every unit
 *             active on entry, result not returned.
 *   status  : [n_nodes] out, pcp_status.
 *   stats   : nullable; counters of THIS call.
 * Post-condition: for status != PCP_FALSE, (lb,ub) are the reference's fixpoint domains and
 * `active` has lost exactly the entailed units; for PCP_FALSE nodes domains/active are unspecified. */
int32_t pcp_propagate(pcp_ctx* ctx, uint32_t n_nodes, int32_t* lb, int32_t* ub, uint64_t* bits,
                      uint64_t* active, uint8_t* status, pcp_stats* stats);

/* Device-resident form: all pointers are HIP device pointers on ctx's device; work is enqueued on
 * `hip_stream` (a hipStream_t, NULL = the null stream) and NOT synchronised.  *_out may alias *_in.
 * active_in NULL = all units active; active_out NULL = do not write the mask back.
 * IMPLICIT-ACTIVE NODES: with active_in == NULL a node is its domains only — no `active` rows are read or kept: a unit that
 * is entailed runs as a no-op, so liveness is derived (a unit is inactive iff it is entailed under the current domains,
 * SURVEY.md A.4); the status comes from an entailment scan of the final domains and active_out, when given, is
 * materialised from them.  Results are identical to passing all-ones rows; a search whose root has every unit active
 * (every driver) never needs rows at all (pcp_branch_device accepts active == child_active == NULL). */
typedef struct {
  const int32_t* lb_in;
  const int32_t* ub_in;
  int32_t* lb_out;
  int32_t* ub_out;
  const uint64_t* active_in;
  uint64_t* active_out;
  uint8_t* status;
  const uint64_t* bits_in; /* set mode only: [n_nodes][n_vars][set_words]; lb_in/ub_in are then ignored (may be NULL) */
  uint64_t* bits_out;      /* set mode only; may alias bits_in                                                        */
  /* ABI v7, optional (NULL = none): [n_nodes] — per node the ONE variable whose domain differs from a fixpoint of this same model, or
   * PCP_NOVAR (any value >= n_vars) = no promise, the node is propagated from scratch.  The caller PROMISES: take the node's row, give
   * that variable back the domain it had, and the row is the result of a consistency() over this model (a parent's fixpoint; a
   * child = the parent with one branch constraint folded in, which is what pcp_branch_device_hint writes).  Legal because a propagator
   * that is entailed or quiescent at a fixpoint is a no-op until one of ITS variables changes (propagation/store.rs:191-198 wakes exactly
   * the propagators of changed variables; the reference's prepare() re-schedules everything, store.rs:144-149, and finds the others
   * no-ops — DESIGN.md 2): the fixpoint, the status and the derived `active` rows are the same with and without the hint, only the
   * work differs.  Used by the all-XNeqY kernel (interval mode, implicit nodes); every other path ignores it.  A hint that breaks the
   * promise gives an unspecified (sound but possibly not fully propagated) result. */
  const uint32_t* dirty_var;
  /* ABI v7: the format of the bounds rows.  0 (PCP_CELLS_I32, the default): lb / ub are two int32 rows per node.  1 (PCP_CELLS_PACKED16): ONE
   * row of n_vars 32-bit CELLS per node, cell = (-lb & 0xffff) | ub << 16 — the format the all-XNeqY kernel keeps in LDS — in lb_in / lb_out;
   * ub_in / ub_out are ignored.  4 n_vars bytes per node instead of 8: staging is a copy, open-node stacks and records halve.  Needs a
   * declared hull within +-16383 (pcp_model_set_hull), an all-XNeqY model over implicit nodes (active_in == active_out == NULL), interval
   * mode; anything else: PCP_ERR_UNSUPPORTED.  pcp_pack_rows / pcp_unpack_rows convert on the device. */
  uint32_t cell_format;
  uint32_t reserved;
} pcp_device_batch;
#define PCP_CELLS_I32 0u
#define PCP_CELLS_PACKED16 1u
/* int32 rows <-> packed cells, [n_nodes][n_vars] each, device pointers, enqueued on hip_stream.  A bound outside +-16383 cannot be packed:
 * its cell is clamped and the context's sticky hull flag is raised (the next pcp_stats_read returns PCP_ERR_CONTRACT). */
int32_t pcp_pack_rows(pcp_ctx* ctx, uint32_t n_nodes, const int32_t* lb, const int32_t* ub, uint32_t* cells, void* hip_stream);
int32_t pcp_unpack_rows(pcp_ctx* ctx, uint32_t n_nodes, const uint32_t* cells, int32_t* lb, int32_t* ub, void* hip_stream);
/* pcp_branch_device_hint over rows of packed cells (implicit nodes): the same brancher, the same child order (option branch_reverse), the same
 * counts and hints; child_cells has room for 2 n_nodes rows.  A search whose open nodes stay cells between launches needs nothing else. */
int32_t pcp_branch_device_cells(pcp_ctx* ctx, uint32_t n_nodes, const uint32_t* cells, const uint8_t* status, uint32_t* child_cells,
                                uint32_t* child_dirty, uint32_t* counts, void* hip_stream);
int32_t pcp_propagate_device(pcp_ctx* ctx, uint32_t n_nodes, const pcp_device_batch* batch, void* hip_stream);
/* ABI v8 — nodes with propagators of their OWN.  In the reference every branch appends ONE unary propagator to ONE node's cstore
 * (Branch::distribute, search/branching/branch.rs:36-55); BinarySplit's  x <= v / x > v  and Enumerate's  x = v  narrow their variable once and are
 * then entailed, so a driver folds them into the child's bounds (what pcp_branch_device does) — but Enumerate's  x != v  with v INSIDE the
 * domain removes nothing on an Interval (x_neq_y.rs:82-93: only at a bound) and stays active until v reaches a bound
 * (search/branching/enumerate.rs:48-59).  Such propagators travel with their node:
 *   node_unit_off : device uint32 [n_nodes + 1], CSR offsets into node_units (NULL = no node has any: exactly pcp_propagate_device)
 *   node_units    : device pcp_prop [node_unit_off[n_nodes]] — kind PCP_NEQ / PCP_EQ / PCP_LT over ONE variable and ONE Constant in either
 *                   operand position (var[i] = PCP_CONST, off[i] = the value; the variable's off = its Addition offset); group fields ignored.
 * They are scheduled with the model's propagators (every one once, then again while anything narrows) and count in the node's status: a
 * node is PCP_TRUE only if its own propagators are entailed too.  Their liveness is not reported (`active` rows cover the model's units): a
 * caller drops a node unit when its constant has left the domain.  A malformed unit refuses its node (PCP_STATUS_HULL, sticky flag).
 * Interval mode, stores of at most 128 variables (interned constants included) and 2048 elementary filters — the one-wavefront-per-node
 * kernel (plan.path 4) —, int32 rows; anything else: PCP_ERR_UNSUPPORTED.  (Set mode needs none of this: there x != v is an exact set
 * operation on `bits`.) */
int32_t pcp_propagate_device_units(pcp_ctx* ctx, uint32_t n_nodes, const pcp_device_batch* batch, const uint32_t* node_unit_off,
                                   const pcp_prop* node_units, void* hip_stream);
/* One context = one queue of launches: the device-side scratch behind a launch (counters, team words, `active` scratch, the tile tickets of the
 * persistent kernels) belongs to the context, so the launches of one context must not overlap on the device — enqueue them on one stream, or order
 * the streams; concurrent launches take one context each (the model is uploaded per context).  The tile tickets are guarded at run time as well:
 * a ticketed launch enqueued on another stream than the context's last ticketed launch waits for that one (hipStreamWaitEvent), and after any HIP
 * error reported by this context the tickets are zeroed in front of the next launch that uses them. */

/* ---- branching on the device (the caller side of the path) -------------------------------------
 * ≡ Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter (search/branching/brancher.rs:52-71) applied to every
 * PCP_UNKNOWN node of a propagated batch:
 *   variable = first index among the variables of minimal size > 1   (first_smallest_var.rs:30-39)
 *   value    = (lb + ub) / 2 truncated toward zero                   (middle_val.rs:25-27)
 *   children = `x <= value` then `x > value`                          (binary_split.rs:46-57), folded into the bounds
 * (a var-vs-constant XLessY narrows its variable on its first run and is then entailed and unlinked, x_less_y.rs:87-109).
 * Children are written in tree order, two per Unknown node, each with a copy of its parent's `active` row — the
 * cstore label (len, active.clone()) of propagation/store.rs:315-317.  All pointers are device pointers.
 *   child_lb/child_ub : [2*n_nodes][n_vars] (capacity);  child_active : [2*n_nodes][ceil(n_units/64)]
 *   counts            : device uint32[5] out = { n_children, n_true, n_false, n_unknown, n_other } — n_other counts nodes whose
 *                       status is none of the three (PCP_STATUS_HULL: a node the engine refused); a driver should stop on it
 * Nothing is synchronised; read `counts` after synchronising hip_stream. */
int32_t pcp_branch_device(pcp_ctx* ctx, uint32_t n_nodes, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                          const uint8_t* status, int32_t* child_lb, int32_t* child_ub, uint64_t* child_active,
                          uint32_t* counts, void* hip_stream);
/* ABI v7: the same, and child_dirty [2*n_nodes] (capacity, nullable) receives for every child the variable it was branched on — the
 * pcp_device_batch.dirty_var entry of that child when its parent was a propagated (fixpoint) row, as it is in every search loop. */
int32_t pcp_branch_device_hint(pcp_ctx* ctx, uint32_t n_nodes, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                               const uint8_t* status, int32_t* child_lb, int32_t* child_ub, uint64_t* child_active,
                               uint32_t* child_dirty, uint32_t* counts, void* hip_stream);

/* The same over FDSpace (set mode): FirstSmallestVar compares CARDINALITIES (first_smallest_var.rs:30-39: Domain::size()), MiddleVal
 * is (lower + upper) / 2 of the set's bounds, the children keep the values <= value resp. > value of the variable's set.
 *   bits [n_nodes][n_vars][set_words], lb/ub [n_nodes][n_vars] = the outputs of pcp_propagate_device;
 *   child_bits [2*n_nodes][n_vars][set_words] (capacity); active / child_active as above (both NULL for implicit nodes). */
int32_t pcp_branch_device_set(pcp_ctx* ctx, uint32_t n_nodes, const uint64_t* bits, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                              const uint8_t* status, uint64_t* child_bits, uint64_t* child_active, uint32_t* counts, void* hip_stream);

/* ---- the reference's search loop, one node per step, without the host in the loop -----------------------------------------------
 * ≡ OneSolution / AllSolution<Propagation<Brancher<FirstSmallestVar, MiddleVal, BinarySplit>>> over a VectorStack
 * (search/mod.rs:45-52, search/engine/one_solution.rs:92-105), optionally under StopNode (search/stop_node.rs:47-62): each step
 * pops the node on top of a LIFO stack of implicit-active nodes, runs its propagation fixpoint (one node = the reference's
 * one-consistency-call-per-node pattern, all CUs on that node), and branches it in place — exactly the reference's left-first
 * order.  n_steps steps are ENQUEUED on hip_stream with no host synchronisation in between: the kernels read the stack pointer
 * from device memory.  Steps after the stack has emptied (or `stop` was raised: solution found / node limit / error) do nothing.
 *   lb, ub           : [capacity][n_vars] device rows; rows [0, *sp) are the open nodes, row *sp - 1 the top
 *   sp, stop         : device uint32 each (the caller initialises *sp = 1 with the root in row 0, *stop = 0)
 *   status           : [capacity] device scratch
 *   counters         : device uint64[5] = { nodes, solutions, failed nodes, error (1 stack overflow: the node stays on the stack, uncounted;
 *                      2 hull violation; 3 an Unknown node without a variable to branch on — the reference panics, first_smallest_var.rs:36), internal },
 *                      accumulated (the caller zeroes them)
 *   first_solution   : [n_vars] device or NULL: the first solution found
 *   node_limit       : != 0 = StopNode(limit) nested as the reference nests it, Monitor<Statistics, StopNode<..>> (stop_node.rs:90-97): the node that
 *                      brings `nodes` to the limit is counted as a node and NEITHER as a solution nor as a failure (its status reaches the monitor as
 *                      EndOfSearch, stop_node.rs:57-62), and `stop` is raised.  The same rule in pcp_dfs_forest_device (per tree) and
 *                      pcp_dfs_forest_device_set (on total_nodes).
 * Interval mode only. */
typedef struct {
  int32_t* lb;
  int32_t* ub;
  uint32_t capacity;
  uint32_t* sp;
  uint32_t* stop;
  uint8_t* status;
  uint64_t* counters;
  int32_t* first_solution;
  /* ABI v7, nullable: [capacity] uint32 (pcp_dfs_forest_device: [n_trees][capacity]), one word per stack row = the variable that row was
   * branched on (the pcp_device_batch.dirty_var of the open node in that row), or >= n_vars for a row the caller put there (a root): the
   * engine writes it when it pushes children and, when it pops a row, propagates from that variable alone instead of from every assigned
   * variable of the node.  A caller that moves rows between stacks moves their words along.  All-XNeqY in-kernel loop only. */
  uint32_t* dirty;
} pcp_dfs_state;
int32_t pcp_dfs_device(pcp_ctx* ctx, const pcp_dfs_state* st, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, void* hip_stream);
/* The same loop on n_trees independent stacks at once, one workgroup per tree (ABI v5): tree t is exactly a pcp_dfs_device instance on
 *   lb, ub [n_trees][capacity][n_vars] (tree t's rows start at t * capacity), sp / stop [n_trees], status [n_trees][capacity],
 *   counters [n_trees][5], first_solution [n_trees][n_vars] or NULL;  node_limit applies to each tree.
 * The trees are the subtrees below the open nodes of a frontier (pcp_propagate_device + pcp_branch_device produce one); the caller adds
 * up the counters and decides when to stop launching (all sp == 0, any stop != 0, its node budget).  All-XNeqY models over implicit
 * nodes only (PCP_ERR_UNSUPPORTED otherwise: pcp_dfs_device runs any interval model, one tree). */
int32_t pcp_dfs_forest_device(pcp_ctx* ctx, const pcp_dfs_state* st, uint32_t n_trees, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, void* hip_stream);

/* ---- the same loop over FDSpace (set mode), a FOREST of trees, one per workgroup (ABI v5) --------------------------------------
 * ≡ the engine above over VStoreSet = VStoreTrail<IntervalSet<i32>> (variable/mod.rs:38): like the reference, a node is restored
 * by undoing a TRAIL (variable/memory/trail_memory.rs:100-104) instead of being copied — a tree's current node stays in one CU's
 * LDS, every narrowing appends (word, removed bits) to the tree's trail, a backtrack ORs them back down to the level's mark and
 * takes the right branch.  Within a tree: exactly the reference's left-first order (FirstSmallestVar on CARDINALITIES, MiddleVal,
 * BinarySplit).  Several trees = the subtrees below the open nodes of a frontier (pcp_propagate_device + pcp_branch_device_set
 * produce one), searched concurrently; n_trees = 1 with the root in bits[0] IS the reference's search.  One call runs at most
 * n_steps nodes per tree; the caller repeats it until every tree is finished, `stop` is raised, or the budget is spent.
 *   bits        : [n_trees][n_vars][set_words] device: in = each tree's root, afterwards each tree's current node
 *   tree        : [n_trees][4] device uint32 = { levels, trail length, pending variable, finished | given << 8 }; the caller initialises every
 *                 tree to { 0, 0, PCP_DFS_FULL, 0 } (PCP_DFS_FULL: the current node has not been propagated at all)
 *   levels      : [n_trees][level_capacity][4] device uint32: the branch decisions whose right child is still open
 *   trail       : [n_trees][trail_capacity][4] device uint32
 *   counters    : [n_trees][4] device uint64 = { nodes, solutions, failed nodes, error (1 level stack full — the node stays current,
 *                 uncounted; 3 an Unknown node without a variable to branch on; 4 trail full: the tree cannot be restored; 5 internal round cap) }, accumulated
 *   total_nodes : device uint64: nodes of all trees; with node_limit != 0 no node beyond it is run (StopNode, stop_node.rs:57-62)
 *   stop        : device uint32: raised on a solution (stop_on_solution), at the node limit, on an error; the caller zeroes it
 *   first_solution / solution_flag : [n_vars] device int32 and a device uint32 (zeroed by the caller), or both NULL: the first solution
 *                 any tree reports (with several trees "first" is in time, not in the reference's order)
 * Set mode, implicit nodes. */
#define PCP_DFS_FULL 0xFFFFFFFFu
typedef struct {
  uint32_t n_trees, level_capacity, trail_capacity, reserved;
  uint64_t* bits;
  uint32_t* tree;
  uint32_t* levels;
  uint32_t* trail;
  uint64_t* counters;
  uint64_t* total_nodes;
  uint32_t* stop;
  int32_t* first_solution;
  uint32_t* solution_flag;
} pcp_forest_state;
int32_t pcp_dfs_forest_device_set(pcp_ctx* ctx, const pcp_forest_state* st, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, void* hip_stream);
/* Between two calls of pcp_dfs_forest_device_set: finished trees take over work from trees that still have some.  pairs = n_pairs x
 * (donor, receiver) tree indices (device uint32).  For every pair whose receiver is finished and whose donor has an open right branch
 * left, the donor's OLDEST open right branch (the subtree nearest its root) becomes the receiver's new root — built from the donor's
 * current node and its trail — and done[pair] = 1; otherwise done[pair] = 0 and nothing changes.  The donor skips that branch when it
 * backtracks to it; counters are untouched, so the sum over the trees stays the search's.  A tree may appear in one pair per call.
 * tree[t][3] = finished (bit 0) | number of levels given away (bits 8..). */
int32_t pcp_dfs_forest_split_set(pcp_ctx* ctx, const pcp_forest_state* st, uint32_t n_pairs, const uint32_t* pairs, uint32_t* done, void* hip_stream);

/* Counters accumulate on the device across pcp_propagate_device calls.  pcp_stats_reset also clears the sticky hull-violation word;
 * pcp_stats_read is the only call that reports (and then clears) it. */
int32_t pcp_stats_reset(pcp_ctx* ctx, void* hip_stream);
int32_t pcp_stats_read(pcp_ctx* ctx, pcp_stats* out, void* hip_stream); /* synchronises hip_stream */

/* Diagnostics (ABI v6): kernel-internal event counters accumulated since the last pcp_stats_reset.  They describe HOW a launch got
 * to its result, never the result (the reference has no counterpart); tests use them to prove that a code path was taken.
 * out[0 .. n) receives the first n of PCP_DBG_COUNT counters (n <= PCP_DBG_COUNT); synchronises hip_stream. */
typedef enum {
  PCP_DBG_BIG_DENSE = 0,   /* path 2: wake-up rounds that streamed the record table again (one per node and round)            */
  PCP_DBG_BIG_SPARSE = 1,  /* path 2: wake-up rounds that walked the adjacency lists of the changed variables                  */
  PCP_DBG_NEQ_TILES = 2,   /* path 1: tiles run (a persistent workgroup runs several)                                         */
  PCP_DBG_NEQ_OVERLAP = 3, /* path 1: tiles whose rows were already in flight while the previous tile's rounds ran            */
  PCP_DBG_SMALL_NODES = 4, /* path 4: nodes run by the small-store kernel                                                     */
  PCP_DBG_NEQ_LEAN = 8,           /* path 1: tiles whose round 0 took the lean form (full 16-node tiles with at most four assigned variables) */
  PCP_DBG_NEQ_LEAN_PASSES = 9,    /* path 1: passes of that form beyond a tile's first (tiles in which a node narrowed)                      */
  PCP_DBG_NEQ_LEAN_HANDOVER = 10, /* path 1: lean tiles handed to the general rounds (a narrowing assigned or emptied a variable)              */
  PCP_DBG_COUNT = 16
} pcp_dbg_counter;
int32_t pcp_debug_counters(pcp_ctx* ctx, uint64_t* out, uint32_t n, void* hip_stream);

/* Timing of the LAST pcp_propagate_device call's kernels, measured with HIP events recorded on the
 * stream the kernels were launched on (bench.py's roofline leg).  Synchronises on the stop event. */
int32_t pcp_last_kernel_ms(pcp_ctx* ctx, float* ms);

/* The launch geometry the LAST pcp_propagate_device / pcp_propagate call chose (tests pin the benchmarked path to
 * the oracle by asserting on it; bench.py reports it). */
typedef struct {
  uint32_t nodes_per_block; /* B: nodes whose domains one workgroup keeps in LDS                       */
  uint32_t team;            /* workgroups cooperating on one node (1 = batch geometry)                 */
  uint32_t packed;          /* 1 = 16-bit packed LDS cells                                             */
  uint32_t word_level;      /* 0 = chunked record sweep, 1/2 = word-group sweep (level -1 range test)  */
  uint32_t global_dom;      /* 1 = domains stay in HBM (variable store larger than LDS); 2 = that variant with the domains in
                               LDS after all as 10-bit cells (declared hull of at most 1024 values)    */
  uint32_t compact;         /* 1 = 8-byte record stream                                                */
  uint32_t implicit_active; /* 1 = no `active` rows: liveness derived from the domains                 */
  uint32_t set_mode;        /* 1 = IntervalSet (bitset) domains                                        */
  uint32_t grid, block, lds_bytes, list_cap;
  uint32_t path;            /* 0 = the generic kernels (every propagator tested in the initial sweep, in bulk where range tests allow);
                               1 = the assignment-driven kernel of all-XNeqY models over implicit nodes: the sweep is the adjacency
                               lists of the assigned variables (an XNeqY between two unassigned variables is a no-op, x_neq_y.rs:82-93);
                               2 = the 10-bit-cell kernel of binary models whose store does not fit LDS as pairs (implicit nodes, declared hull);
                               3 = the formula kernel: a store with formula propagators (pcp_model_push_formula) or Boolean leaves;
                               4 = the small-store kernel: at most 128 slots and 2048 elementary filters, one wavefront per node */
} pcp_plan;
int32_t pcp_last_plan(const pcp_ctx* ctx, pcp_plan* out);

/* Knobs (all optional; the defaults pick everything from the model and the batch).  key:
 *   "block_threads" 256/512/1024, "nodes_per_block" 0 = auto, "force_path" (0 auto, 1 batch LDS kernel, 2 team kernel),
 *   "team" workgroups per node, "list_cap", "global_dom" 1 = domains stay in HBM (2: 10-bit LDS cells allowed), "dom10" 0 = never use
 *   10-bit cells, "implicit_active" 0 = materialise rows for active_in NULL, "group_level" 0 = no group test, "packed" 0 = never use 16-bit cells,
 *   "word_level" 0 = never use the word-group sweep, "solo_cascade" 0 = a wake-up round with one changed variable is an ordinary round
 *   (1, the default: its records are re-run in place and a bound jumps over the values assigned neighbours forbid), "branch_reverse" 1 = pcp_branch_device writes child k of the batch
 *   to row n_children-1-k (a caller appending the rows to a LIFO stack then pops the first node's left child first).
 * "time_kernels" 0 = no HIP events around the fixpoint launches: a call enqueues the kernel and nothing else, pcp_last_kernel_ms then has
 *   nothing to report (1, the default: two event records per launch, a few microseconds of queue time each), "small_path" 0 = small stores
 *   use the generic kernels too (1, the default: path 4; any option that asks for a geometry of the generic kernels — "nodes_per_block",
 *   "force_path", "team", "global_dom" — keeps them as well), "big_path" 0 = never use the 10-bit-cell kernel (path 2), "big_round" /
 *   "big_dense_k" its wake-up rounds (0 auto, 1 dense, 2 sparse; dense iff k * list entries >= records), "neq_persist" 0 = one workgroup
 *   per tile instead of persistent workgroups, "neq_dfs_block" threads per tree of the in-kernel search loop (0 = auto: 512 for one tree,
 *   256 for a forest), "neq_dfs" 0 = pcp_dfs_device launches one step at a time on all-XNeqY models too (1, the default: the whole search
 *   loop runs in one workgroup, n_steps nodes per launch), "neq_path" 0 = all-XNeqY models use the generic kernels too (1, the default: the
 *   assignment-driven kernel when the nodes are implicit), "neq_block" threads per workgroup of that kernel (0 = auto).
 *   (Rounds 3-4 had "neq_wave" / "neq_prefetch": two measured-and-rejected launch forms of the all-XNeqY kernel; their code left the
 *   library in round 5 — tools/micro/neq_rejected_r4.patch)
 * Unknown key -> PCP_ERR_ARG. */
int32_t pcp_set_option(pcp_ctx* ctx, const char* key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* PCP_HIP_H */
