"""Host-side scheduler that turns a gate list into fused passes for ``dq_apply_fused_*``.

The reference executes one gate per ``Gate.forward`` call (circuit.py:261, operation.py:274-289), i.e.
at least two full read+write sweeps of the statevector per gate.  Here gates are grouped so that one
HBM read + one HBM write applies a whole group ("pass").  The kernel side is described in
``csrc/dq_wave.hip``; this module only decides *which* gates go together and emits the descriptor
structs of ``include/dq_hip.h``.

Vocabulary
  bit      amplitude-index bit position, LSB = 0; wire w of an n-qubit circuit is bit n-1-w.
  tile     the m index bits a wavefront owns in a pass: the low L bits plus h gathered high bits.
  round    a stretch of a pass during which R chosen tile bits are "register slots"; a non-diagonal
           gate needs its target bit(s) to be slots.  Changing rounds costs a trip through the wave's LDS buffer.
  action   how a gate acts on a qubit: 'D' if it is diagonal in that qubit (controls, targets of
           diagonal gates), 'N' otherwise.  Two gates commute when every shared qubit is 'D' in both,
           which is what lets the scheduler pull later gates forward.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence

from . import _lib


# a 'grad' PrimOp carries row + (variant << GRAD_VARIANT_SHIFT) in ``mode`` (include/dq_hip.h, DQ_FG_GRAD: 0 all sums,
# 1 a real matrix, 2 a I + i b X, 3 diagonal, 4 a unitary a I + i b X in a backward that records no graph)
GRAD_VARIANT_SHIFT = 24
GRAD_ROW_MASK = (1 << GRAD_VARIANT_SHIFT) - 1


@dataclass
class PrimOp:
    """One kernel-level gate: ``kind`` in {'gen', 'x', 'diag'}; ``targets`` in matrix order (MSB
    first); ``mat`` = offset (in complex numbers) of its 2^k x 2^k matrix in the flat matrix buffer.

    ``kind == 'grad'`` is not a gate but a reduction of the adjoint method's reverse sweep (DQ_FG_GRAD,
    include/dq_hip.h): ``targets = (q, s)`` -- q the trainable gate's target, s the index bit that tells psi from the
    cotangent -- ``controls`` the gate's controls, ``mode`` the row of the accumulator; no matrix.
    ``kind == 'expz'``: the expectation value of a Z string taken from the registers of whatever pass it lands in
    (DQ_FG_EXPZ): no targets, ``controls`` = the string's qubits, ``mode`` = the row, ``order`` = the qubits whose gates
    it waits for (all of them, for a value of the final state); no matrix."""

    kind: str
    targets: tuple[int, ...]
    controls: tuple[int, ...] = ()
    mat: int = 0              # offset in the CALLER's matrix buffer (source of gather_matrices)
    mode: int = 0             # matrix structure promised by the gate class: 0 general, 1 real, 2 Rx-like
    pos: int = 0              # offset in the kernel's matrix buffer, assigned by layout_matrices
    order: tuple[int, ...] = ()   # bits the gate does not touch but is ordered on like a control (the scheduler only)

    @property
    def k(self) -> int:
        return len(self.targets)


@dataclass
class FusedStep:
    desc: _lib.DqFusedPass
    ops: list[int]            # indices into the PrimOp list, in execution order
    nrounds: int
    ntranspose: int           # LDS round trips the kernel will do (incl. back to canonical)
    permutes: bool = False    # writes to other index bits than it reads (needs in != out)
    c64: bool = False         # a complex64 pass: uncontrolled Rx-like gates take the deferred form (defer_rx)


@dataclass
class SingleStep:
    op: int


@dataclass
class Geometry:
    """The tile of a pass: ONE wavefront owns 2^m amplitudes (csrc/dq_wave.hip) -- 2^slots of them per lane in registers,
    the other six tile bits on the lanes; the library derives slot order, lane order and handlers itself."""

    m: int            # tile bits
    slots: int        # register slots per lane (R)
    vb: int           # the I/O layouts keep tile bits [0, vb) as slots (1 for c64: 16 bytes per lane; 0 for c128)
    min_low: int      # minimum contiguous low bits of a tile (coalescing floor)
    max_gates: int = _lib.FUSED_MAX_GATES
    max_rounds: int = _lib.FUSED_MAX_ROUNDS - 1  # one spare so a trailing round never overflows
    plan_width: int = 4       # gathered bits of every pass from dry runs of the pass (_plan_tiles): beam width
    plan_branch: int = 3      # ... and tiles tried per beam state; plan_width = 0: first-come tiles (no dry runs)
    permute_store: bool = False   # passes may write to other index bits than they read (out-of-place; _place_writes)
    free_low: bool | str = True   # ... including the contiguous low bits: every pass picks ALL its tile qubits (_schedule)
    plan_restarts: int = 3    # beam searches with different random branches (states of >= 2^plan_restart_bits amplitudes)
    plan_restart_bits: int = 26
    plan_min_bits: int = 20   # states below 2^plan_min_bits amplitudes: greedy (width 1) -- planning time matters there
    far_bit: int = 19         # index bits >= far_bit are "far": every one gathered doubles the number of
    max_far: int | None = None  # distant address streams of a tile; None = no limit
    # positions (in the layout the LAST pass leaves, i.e. after ``final_perm``) whose qubits that pass keeps out of its tile
    # unless a gate needs them: the sharded state launches it in slices by exactly these bits (executor.run(slicing=...)),
    # and a bit that merely PADS the tile would cost a factor two in slices for nothing
    final_free: tuple = ()
    # a run behind |0..0> (or on a state with index bits known to be |0>): the LAST pass pads its tile with the index bits
    # that no pass has had in its tile so far, while it has room -- a bit left outside every tile makes `zero_state_masks`
    # give up on the whole schedule (the result would keep uninitialised memory), and which bits are left is a matter of
    # the plan: on three of eight ranks of the n = 34 job one of the three qubits that came from the rank bits was, and
    # their second stretch ran 8 full passes instead of 6.25 (44 ms against 34)
    pad_last_with_untouched: bool = False
    # the index bits the run's input is known to be |0> on (a shard behind its first exchange: the qubits that came from the
    # rank bits; None: a circuit's own |0..0>, i.e. all of them): among schedules of as many passes the planner keeps the
    # one that moves the least behind THAT state (`zero_state_cost`)
    known_zero: int | None = None

    @property
    def logt(self) -> int:
        return self.m - self.slots


def default_geometry(is_c128: bool) -> Geometry:
    """The wave tile of a precision: m = 12 and six register slots for complex64 (64 lanes x 64 amplitudes), m = 11 and
    five for complex128 (64 x 32).  Records of a pass: gates + 2 per layout change (twice that when it exchanges more
    bits than one trip moves) <= 112."""
    if is_c128:     # 32 amplitudes of 16 bytes per lane: five slots, an 11-bit tile, 128-byte runs = 3 low bits
        return Geometry(m=11, slots=5, vb=0, min_low=3, max_gates=72, max_rounds=8)
    return Geometry(m=12, slots=6, vb=1, min_low=4, max_gates=72, max_rounds=8)


def wave_supports(ops: Sequence['PrimOp'], is128: bool = False) -> bool:
    """Can the wave-tile kernel run all of ``ops``?  (One-target dense gates and X, diagonal gates on one or two
    targets, dense gates on two targets, any controls, the reductions: everything the scheduler fuses.)"""
    return all((op.kind in ('gen', 'x') and len(op.targets) == 1) or (op.kind == 'diag' and len(op.targets) <= 2)
               or (op.kind == 'gen' and len(op.targets) == 2)
               or op.kind in ('grad', 'expz') for op in ops)


# ---------------------------------------------------------------------------------------------------
def _action(op: PrimOp) -> dict[int, str]:
    """How ``op`` acts on each of its qubits: 'D' = as a function of Z (controls, targets of diagonal gates),
    'X' = as a function of X (target of an X or of an Rx-like matrix a*I + i*b*X, controlled or not),
    'G' = a reduction of the reverse sweep reading the psi / lambda bit, 'N' = anything else."""
    act = {q: 'D' for q in op.controls + op.order}
    if op.kind == 'expz':
        # <Z..Z> is invariant under unitaries on the other qubits and under diagonal gates on its own -- but the
        # reference's fixed matrices are unitary to float32 rounding only, so a value taken before the last
        # gate differs from the final state's by 1e-7: the caller lists every qubit in ``order`` and the record waits
        # for everything
        return {q: 'N' for q in op.controls + op.order}
    if op.kind == 'grad':
        # a snapshot on its target (ordered against everything there).  On the psi / lambda bit reductions do not
        # disturb each other (a type of their own, 'G'), but they are ordered against every gate that tells psi from
        # lambda -- the complex128 sweep's U^dagger U corrections, controlled by that bit: between an exact inverse and
        # its correction the sum lambda (x) conj(psi) is not the invariant it is before and after both
        act[op.targets[0]] = 'N'
        act[op.targets[1]] = 'G'
        return act
    if op.kind == 'diag':
        t_act = 'D'
    elif op.k == 1 and (op.kind == 'x' or (op.kind == 'gen' and op.mode == 2)):
        t_act = 'X'
    else:
        t_act = 'N'
    for q in op.targets:
        act[q] = t_act
    return act


class _Dag:
    """Dependency bookkeeping with the commutation rule: two gates commute when on every shared qubit both are
    'D' or both are 'X' (each is then a polynomial in the same commuting Paulis on the shared qubits), or both 'G'.  Per qubit
    the gate list therefore splits into groups -- maximal runs of one type, every 'N' gate a group of its own --
    and a gate depends on all members of the group before its own."""

    def __init__(self, ops: Sequence[PrimOp], n: int, x_commute: bool = True):
        self.ops = ops
        self.n_ops = len(ops)
        self.succ: list[list[int]] = [[] for _ in ops]
        self.indeg = [0] * len(ops)
        prev: list[list[int]] = [[] for _ in range(n)]   # the group every member of the current one waits for
        cur: list[list[int]] = [[] for _ in range(n)]    # the current group ...
        cur_t: list[str] = ['N'] * n                     # ... and its type
        for i, op in enumerate(ops):
            deps: set[int] = set()
            for q, t in _action(op).items():
                if t == 'X' and not x_commute:
                    t = 'N'
                if t == 'N' or t != cur_t[q]:
                    if cur[q]:
                        prev[q] = cur[q]
                    cur[q] = []
                    cur_t[q] = t
                deps.update(prev[q])
                cur[q].append(i)
                if t == 'N':                             # closes its own group at once
                    prev[q], cur[q] = cur[q], []
            for d in deps:
                self.succ[d].append(i)
            self.indeg[i] = len(deps)
        self.ready = sorted(i for i in range(self.n_ops) if self.indeg[i] == 0)
        self.done = 0
        self._native = None

    def native(self):
        """The DAG inside libdqhip (csrc/dq_plan.hip: the planner's dry runs as a loop over flat arrays), made on first
        use: (handle, scratch arrays)."""
        if self._native is None:
            import numpy as np

            lib = _lib.load()
            off = np.zeros(self.n_ops + 1, dtype=np.int32)
            for i, s_ in enumerate(self.succ):
                off[i + 1] = off[i] + len(s_)
            succ = np.array([x for s_ in self.succ for x in s_] or [0], dtype=np.int32)
            tmask = np.array([0 if op.kind == 'diag' else sum(1 << t for t in op.targets) for op in self.ops] or [0],
                             dtype=np.uint64)
            fus = np.array([1 if _fusable(op) else 0 for op in self.ops] or [0], dtype=np.uint8)
            h = lib.dq_dag_create(self.n_ops, off.ctypes.data, succ.ctypes.data, tmask.ctypes.data, fus.ctypes.data)
            if not h:
                raise RuntimeError('dq_dag_create failed: ' + lib.dq_last_error().decode())
            n1 = max(self.n_ops, 1)
            self._native = (lib, h, np.empty(n1, np.int32), np.empty(n1, np.int32), np.empty(n1, np.int32),
                            np.empty(64, np.int32))
        return self._native

    def __del__(self):
        nat = getattr(self, '_native', None)
        if nat is not None:
            try:
                nat[0].dq_dag_destroy(nat[1])
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass

    def retire(self, i: int) -> None:
        self.ready.remove(i)
        self.done += 1
        new = []
        for s in self.succ[i]:
            self.indeg[s] -= 1
            if self.indeg[s] == 0:
                new.append(s)
        if new:
            self.ready = sorted(self.ready + new)


def _closure(dag: '_Dag', tile: set[int], cap: int, indeg: list[int] | None = None,
             ready: list[int] | None = None) -> tuple[int, list[int], dict[int, int]]:
    """Dry run of a pass that owns ``tile`` from the front (``indeg``, ``ready``; default: the DAG's current one),
    ignoring the round / slot limits: retires every fusable diagonal gate and every other fusable gate whose
    targets lie in the tile, up to ``cap`` gates.  Returns (gates retired, the gates left ready but stuck,
    the in-degrees it changed)."""
    import ctypes as C

    import numpy as np

    lib, h, stuck, cidx, cval, _ = dag.native()
    base = np.ascontiguousarray(dag.indeg if indeg is None else indeg, dtype=np.int32)
    rd = np.array(dag.ready if ready is None else ready, dtype=np.int32)
    ns, nc = C.c_int(0), C.c_int(0)
    count = lib.dq_dag_closure(h, _mask(tile), cap, base.ctypes.data, rd.ctypes.data, len(rd), stuck.ctypes.data, C.byref(ns),
                               cidx.ctypes.data, cval.ctypes.data, C.byref(nc))
    if count < 0:
        raise RuntimeError('dq_dag_closure failed: ' + lib.dq_last_error().decode())
    return count, stuck[:ns.value].tolist(), dict(zip(cidx[:nc.value].tolist(), cval[:nc.value].tolist()))


def _mask(bits) -> int:
    m = 0
    for b in bits:
        m |= 1 << b
    return m


def _rank_candidates(dag: '_Dag', tile: set[int], cands: Sequence[int], cap: int, indeg, ready) -> list[int]:
    """Gates a pass would retire with ``tile`` + each of ``cands`` (one native call for all of them)."""
    import numpy as np

    lib, h, _s, _i, _v, counts = dag.native()
    base = np.ascontiguousarray(dag.indeg if indeg is None else indeg, dtype=np.int32)
    rd = np.array(dag.ready if ready is None else ready, dtype=np.int32)
    cd = np.array(list(cands), dtype=np.int32)
    rc = lib.dq_dag_rank(h, _mask(tile), cap, base.ctypes.data, rd.ctypes.data, len(rd), cd.ctypes.data, len(cd), counts.ctypes.data)
    if rc < 0:
        raise RuntimeError('dq_dag_rank failed: ' + lib.dq_last_error().decode())
    return counts[:len(cd)].tolist()


def _grow_tile(dag: '_Dag', low: set[int], hcap: int, cap: int, indeg: list[int] | None = None,
               ready: list[int] | None = None, pick=None, far: tuple[int, int] | None = None,
               prev: set[int] | None = None, need: int = 0) -> set[int]:
    """Gathered bits of one pass, grown one bit at a time: dry-run the pass with the bits chosen so far, look at
    the gates it leaves stuck at the front, and add the missing target bit that lets the pass retire the most
    gates (ties: the bit most stuck gates wait for, then the lowest).  ``pick(ranked)`` may choose another of the
    ranked candidates (the beam search's branching).

    ``prev`` / ``need`` (free low bits, ``low`` empty): the tile must share at least ``need`` qubits with the tile
    ``prev`` of the pass before -- they become its contiguous low bits, which that pass must be able to write as whole
    runs -- so once the room left equals what is still missing, only qubits of ``prev`` are candidates, and a tile
    that ends short is padded with qubits of ``prev`` that no gate asked for."""
    import ctypes as C

    import numpy as np

    lib, h = dag.native()[:2]
    base_arr = np.ascontiguousarray(dag.indeg if indeg is None else indeg, dtype=np.int32)
    rd = np.array(dag.ready if ready is None else ready, dtype=np.int32)
    cq, cw, cc = (C.c_int * 64)(), (C.c_int * 64)(), (C.c_int * 64)()
    nbase = C.c_int(0)
    chosen: set[int] = set()
    while len(chosen) < hcap:
        tile = low | chosen
        # one native call per step (csrc/dq_plan.hip): the dry run with the tile, the qubits the stuck gates wait for,
        # and the dry run with each of them added
        nq = lib.dq_dag_grow_step(h, _mask(tile), cap, base_arr.ctypes.data, rd.ctypes.data, len(rd), C.byref(nbase), cq, cw, cc)
        if nq < 0:
            raise RuntimeError('dq_dag_grow_step failed: ' + lib.dq_last_error().decode())
        if nbase.value >= cap:
            break
        cands = {cq[k]: (cw[k], cc[k]) for k in range(nq)}
        if far is not None and sum(1 for b_ in chosen if b_ >= far[0]) >= far[1]:
            cands = {q: w for q, w in cands.items() if q < far[0]}     # the budget of far-apart bits is spent
        if prev is not None and hcap - len(chosen) <= need - len(chosen & prev):
            cands = {q: w for q, w in cands.items() if q in prev}
        if not cands:
            break
        ranked = sorted(((c_, w_, -q) for q, (w_, c_) in cands.items()), reverse=True)
        best = ranked[0] if pick is None else pick(ranked)
        chosen.add(-best[2])
    if prev is not None:
        for q in sorted(prev - chosen):
            if len(chosen & prev) >= need:
                break
            if len(chosen) >= hcap:     # (cannot happen: the candidates were restricted in time)
                break
            chosen.add(q)
    return chosen


def _plan_tiles(dag: '_Dag', low: set[int], hcap: int, cap: int, width: int, branch: int,
                seed: int = 20250929, far: tuple[int, int] | None = None, free_low: int = 0) -> list[set[int] | None]:
    """Gathered-bit sets for ALL passes of a circuit by beam search over dry runs (`_closure`): every beam state
    (a front of the DAG) is extended by the greedy tile and by ``branch - 1`` randomised ones (one of the three best
    candidates at each growth step, fixed seed), the ``width`` states that have retired the most gates survive.
    ``None`` entries stand for a gate that cannot be fused and runs on its own.  width = 1: plain greedy.

    ``free_low`` = L > 0: the entries are WHOLE tiles (up to hcap + L qubits, none of them fixed), each sharing at
    least L qubits with the one before (the first with ``low``): see `_grow_tile`."""
    import random

    rng = random.Random(seed)

    def jitter(ranked):
        return ranked[rng.randrange(min(3, len(ranked)))]

    import numpy as np

    beam = [(0, np.array(dag.indeg, dtype=np.int32), list(dag.ready), [], set(low))]
    while True:
        nxt = []
        for done, indeg, ready, hist, prev in beam:
            if done >= dag.n_ops:
                return hist
            seen = set()
            for b_ in range(branch if width > 1 else 1):
                if free_low:
                    tile_bits = _grow_tile(dag, set(), hcap + free_low, cap, indeg, ready, jitter if b_ else None, far,
                                           prev=prev, need=free_low)
                    whole = tile_bits
                else:
                    tile_bits = _grow_tile(dag, low, hcap, cap, indeg, ready, jitter if b_ else None, far)
                    whole = low | tile_bits
                key = frozenset(tile_bits)
                if key in seen:
                    continue
                seen.add(key)
                count, stuck, changed = _closure(dag, whole, cap, indeg, ready)
                nindeg = indeg.copy()
                for k_, v in changed.items():
                    nindeg[k_] = v
                if count == 0:       # nothing fusable at the front: the lowest ready gate runs on its own
                    i = min(stuck)
                    stuck.remove(i)
                    for s_ in dag.succ[i]:
                        nindeg[s_] -= 1
                        if nindeg[s_] == 0:
                            stuck.append(s_)
                    nxt.append((done + 1, nindeg, stuck, hist + [None], prev))
                    break
                nxt.append((done + count, nindeg, stuck, hist + [tile_bits], whole))
        nxt.sort(key=lambda t: -t[0])
        beam = nxt[:width]


@dataclass
class _Round:
    slots: list[int] = field(default_factory=list)  # global bits that must be register slots
    ops: list[int] = field(default_factory=list)


def _fusable(op: PrimOp) -> bool:
    if op.kind == 'diag':
        return op.k <= 2
    return op.k <= 2


def schedule(ops: Sequence[PrimOp], n: int, geom: Geometry, fuse: bool = True,
             final_perm: Sequence[int] | None = None) -> list[FusedStep | SingleStep]:
    """List scheduling over the commutation DAG.  Returns steps in execution order.  The gathered bits of the
    passes come from the dry-run planner and, for comparison, from the first-come rule (a gate claims its bits while
    the tile has room) -- on circuits whose passes are bounded by the gate cap rather than by the tile the cheap
    rule can win; the schedule with fewer passes (then fewer LDS trips) is kept.

    ``final_perm`` (needs ``geom.permute_store``): the LAST pass leaves index bit b at position final_perm[b] instead
    of restoring the canonical order -- the re-labelling of the local qubits that a shard exchange needs
    (distributed._exchange_qubits) rides on a pass that has to be made anyway.  Callers check ``applied_final_perm``
    on the result: False when no pass could take it (last step not fused, or the permutation moves a bit below the
    contiguous run), and they then permute on their own."""
    if not fuse or n < geom.m:
        return Steps(SingleStep(i) for i in range(len(ops)))
    width = geom.plan_width if n >= geom.plan_min_bits else min(geom.plan_width, 1)
    best = _schedule(ops, n, geom, 0, final_perm)

    def cost(steps):
        return (not steps.applied_final_perm, len(steps), sum(s_.ntranspose for s_ in steps if isinstance(s_, FusedStep)))

    several = sum(isinstance(s_, FusedStep) for s_ in best) > 1
    if width and several:
        cand = _schedule(ops, n, geom, width, final_perm)
        if cost(cand) < cost(best):
            best = cand
    force = geom.free_low == 'force'          # (a testing aid: take the candidate below whenever it exists)
    if geom.free_low and geom.permute_store and several and (width > 1 or force):
        # every pass picks ALL its tile qubits (the stores also re-label the contiguous low bits): fewer passes
        # when the circuit does not keep coming back to the same low qubits; None: it could not restore the
        # canonical order with its last pass, or a gate had to run on its own
        cand = _schedule(ops, n, geom, max(width, 2), final_perm, free_low=True)
        if cand is not None and (cost(cand) < cost(best) or force):
            best = cand
    return best


class Steps(list):
    """The steps of a schedule; ``applied_final_perm`` = the last pass writes the requested final permutation."""

    applied_final_perm = False


def _schedule(ops: Sequence[PrimOp], n: int, geom: Geometry, width: int,
              final_perm: Sequence[int] | None = None, free_low: bool = False) -> 'Steps | None':
    """The planner's restarts give several tile sequences: tried shortest first, the first that can be carried out wins
    (with free low bits a sequence may turn out infeasible -- `_schedule_planned` -- and the next one often is not)."""
    if not width:
        return _schedule_planned(ops, n, geom, width, final_perm, free_low, None)
    dag = _Dag(ops, n)
    L = geom.min_low
    restarts = geom.plan_restarts if width > 1 and n >= geom.plan_restart_bits else 1
    far = (geom.far_bit, geom.max_far) if geom.max_far is not None else None
    plans = [_plan_tiles(dag, set(range(L)), geom.m - L, geom.max_gates, width, geom.plan_branch, 20250929 + r, far,
                         free_low=L if free_low else 0) for r in range(restarts)]
    seen = []
    best = None
    for plan in sorted(plans, key=len):
        if plan in seen:
            continue
        if best is not None and len(plan) > len(best[1]):
            break                       # (longer plans cannot give fewer passes)
        seen.append(plan)
        out = _schedule_planned(ops, n, geom, width, final_perm, free_low, list(plan))
        if out is not None:
            # among schedules of as many passes: the one that moves the least behind |0..0> (most circuits start there:
            # `zero_state_masks`), then the one with the fewest layout changes
            key = (len(out), zero_state_cost(out, n, geom.known_zero), sum(s_.ntranspose for s_ in out if isinstance(s_, FusedStep)))
            if best is None or key < best[0]:
                best = (key, out)
        if len(seen) >= 4 and best is not None or len(seen) >= 8:
            break
    return None if best is None else best[1]


def _schedule_planned(ops: Sequence[PrimOp], n: int, geom: Geometry, width: int, final_perm: Sequence[int] | None,
                      free_low: bool, planned: list | None) -> 'Steps | None':
    """``free_low`` (with permuted stores and a planner): the L contiguous low bits of a pass hold whichever qubits the
    pass before wrote there -- ``low_list``, position by position, chosen from the qubits that pass had in its tile and
    this one wants -- so all m tile qubits are picked per pass.  Returns None when that does not work out (a gate that
    runs on its own needs the canonical order, and so does the end of the circuit: qubits 0 .. L-1 must then be in
    the last tile)."""
    dag = _Dag(ops, n)
    steps: list = []                    # SingleStep | (geometry, low bits, gathered bits, rounds) of a fused pass, finalised below
    L = geom.min_low
    low_list = list(range(L))           # the qubits on index bits 0 .. L-1 when the pass starts
    low = set(low_list)
    hcap = geom.m - L
    far = (geom.far_bit, geom.max_far) if geom.max_far is not None else None
    planned = planned if planned is not None else []
    planned.reverse()                   # consumed from the end
    prev_tile: set[int] | None = None
    while dag.done < dag.n_ops:
        high: set[int] = set()          # tile bits beyond the low ones
        rounds: list[_Round] = [_Round()]
        count = 0
        if width == 0:
            allowed = None              # first come, first served: a gate claims its bits while the tile has room
        elif planned:
            allowed = planned.pop()
            if allowed is None:         # a gate the fused kernel does not take
                i = dag.ready[0]
                if not _fusable(ops[i]):
                    if free_low:
                        return None
                    steps.append(SingleStep(i))
                    dag.retire(i)
                    continue
                allowed = _grow_tile(dag, low, hcap, geom.max_gates, far=far)
            elif free_low and prev_tile is not None:
                # this pass's low qubits: L of those the planned tile shares with the previous pass's (busiest first:
                # position 0 is a register slot of the load layout), topped up with the old low qubits
                busy = {}
                for i in dag.ready:
                    for t in ops[i].targets:
                        busy[t] = busy.get(t, 0) + 1
                cand = sorted((b for b in allowed if b in prev_tile), key=lambda b: (-busy.get(b, 0), b))
                fill = [b for b in low_list if b not in cand] + sorted(b for b in prev_tile if b not in cand and b not in low)
                low_list = (cand + fill)[:L]
                low = set(low_list)
            if free_low:
                allowed = set(allowed) - low
        else:                           # the passes ran out of step with the dry runs (round / gate caps)
            allowed = _grow_tile(dag, low, hcap, geom.max_gates, far=far)

        def fits_tile(op: PrimOp) -> bool:
            need = {t for t in op.targets if t not in low} - high
            if allowed is not None and len(high | need | allowed) > hcap:
                return False    # room is kept for the planned bits; spare room goes first come, first served
            if geom.max_far is not None and sum(1 for b in high | need if b >= geom.far_bit) > geom.max_far:
                return False    # too many far-apart address streams per tile (DRAM row conflicts)
            return len(high) + len(need) <= hcap and len(high | need) <= min(hcap, n - L)

        def round_accepts(cur: _Round, tset: set[int], first: bool) -> bool:
            new = set(cur.slots) | tset
            if len(new) > geom.slots:
                return False
            if first and cur.ops:
                # keep the first round loadable straight from HBM: only index bit 0 (c64) and gathered bits
                # may be slots, and at most R - vb gathered ones
                if any(b in low_list[geom.vb:] for b in new):
                    return False
                if sum(1 for b in new if b not in low) > geom.slots - geom.vb:
                    return False
            return True

        for _attempt in (0, 1):
            high.clear()
            rounds[:] = [_Round()]
            count = 0
            progressed = True
            while progressed and count < geom.max_gates:
                progressed = False
                cur = rounds[-1]
                pick = None
                pick_rank = 99
                for i in dag.ready:
                    op = ops[i]
                    if not _fusable(op):
                        continue
                    if op.kind == 'diag':
                        rank = 0
                    else:
                        tset = set(op.targets)
                        in_tile = all(t in low or t in high for t in tset)
                        room = round_accepts(cur, tset, first=len(rounds) == 1)
                        more_rounds = len(rounds) < geom.max_rounds and len(tset) <= geom.slots
                        if tset <= set(cur.slots):
                            rank = 0
                        elif in_tile and room:
                            rank = 1
                        elif fits_tile(op) and room:
                            rank = 2
                        elif in_tile and more_rounds:
                            rank = 3
                        elif fits_tile(op) and more_rounds:
                            rank = 4
                        else:
                            continue
                    if rank < pick_rank:
                        pick, pick_rank = i, rank
                        if rank == 0:
                            break
                if pick is None:
                    break
                op = ops[pick]
                if op.kind != 'diag':
                    tset = set(op.targets)
                    for t in tset:
                        if t not in low:
                            high.add(t)
                    if pick_rank in (3, 4):
                        rounds.append(_Round())
                        cur = rounds[-1]
                    for t in op.targets:
                        if t not in cur.slots:
                            cur.slots.append(t)
                cur.ops.append(pick)
                dag.retire(pick)
                count += 1
                progressed = True

            if count or allowed is None:
                break
            allowed = None       # the planned bits left no room for the front gate: first come, first served

        if count == 0:
            # nothing fusable is ready: run the lowest-index ready gate on its own
            if free_low:
                return None
            i = dag.ready[0]
            steps.append(SingleStep(i))
            dag.retire(i)
            continue
        if free_low and dag.done >= dag.n_ops:
            # the last pass restores the canonical order: the qubits that belong on index bits 0 .. L-1 must be in its tile
            missing = set(range(L)) - low - high
            if len(high) + len(missing) > hcap:
                return None
            high |= missing
        steps.append((geom, list(low_list), high, rounds))
        prev_tile = low | high
    return _place_writes(ops, n, steps, geom.permute_store, final_perm)


class _Infeasible(Exception):
    """A schedule with free low bits asks a pass to write qubits to the contiguous low bits that are not in its tile."""


def _place_writes(ops: Sequence[PrimOp], n: int, pending: list, permute: bool,
                  final_perm: Sequence[int] | None = None) -> 'Steps | None':
    """Finalise the passes.  With ``permute`` a pass writes the qubits the NEXT pass gathers to the cheapest index
    bits (right above the contiguous run) and everybody else above them, in their current order -- gathered reads
    from far-apart addresses are what a pass pays for, scattered writes are nearly free (mb_scatter) --
    so from the second pass on every tile is read as one contiguous block; and it writes the qubits the next pass
    wants on its contiguous LOW bits there (``low_list`` of the next pass: they must be in this pass's tile).  ``phys``
    maps a qubit's index bit (logical) to where it currently lives; the last pass, and any pass followed by a gate
    that runs on its own, writes the canonical order back."""
    phys = list(range(n))
    out = Steps()
    final = list(range(n))
    touched: set[int] = set()           # logical index bits some pass has had in its tile so far
    if (permute and final_perm is not None and pending and not isinstance(pending[-1], SingleStep)
            and all(final_perm[b] == b for b in range(pending[-1][0].min_low))):
        final = list(final_perm)
        out.applied_final_perm = True
    for k, item in enumerate(pending):
        if isinstance(item, SingleStep):
            assert phys == list(range(n))
            out.append(item)
            continue
        geom, low_list, high, rounds = item
        assert all(phys[b] == i for i, b in enumerate(low_list)), 'the low qubits of a pass are not where it expects them'
        if phys == list(range(n)):
            tops, thigh, trounds = ops, high, rounds
        else:                               # the pass sees physical bits
            def tr(bits):
                return tuple(phys[b] for b in bits)
            tops = list(ops)
            trounds = []
            for rd in rounds:
                for oi in rd.ops:
                    tops[oi] = PrimOp(ops[oi].kind, tr(ops[oi].targets), tr(ops[oi].controls), ops[oi].mat, ops[oi].mode,
                                      ops[oi].pos, tr(ops[oi].order))
                trounds.append(_Round(slots=[phys[b] for b in rd.slots], ops=list(rd.ops)))
            thigh = {phys[b] for b in high}
        nxt = pending[k + 1] if permute and k + 1 < len(pending) and not isinstance(pending[k + 1], SingleStep) else None
        if nxt is None:
            wphys = final if k == len(pending) - 1 else list(range(n))
        else:
            ngeom, nlow, nhigh, _ = nxt
            near = list(range(ngeom.min_low, ngeom.min_low + len(nhigh)))
            wphys = [None] * n
            for i, b in enumerate(nlow):
                wphys[b] = i
            # the qubits this pass has in its own tile first: index bits L, L + 1, .. then extend the contiguous runs it
            # writes (a tile bit written to bit L doubles them to 256 bytes); which of its bits a tile reads where does
            # not matter to the reader
            mine = set(low_list) | set(high)
            for b, pos in zip(sorted(nhigh, key=lambda b: (b not in mine, phys[b])), near):
                wphys[b] = pos
            taken = {w for w in wphys if w is not None}
            rest = [p_ for p_ in range(n) if p_ not in taken]
            for b in sorted((b for b in range(n) if wphys[b] is None), key=lambda b: phys[b]):
                wphys[b] = rest.pop(0)
        inv = {phys[b]: b for b in range(n)}                          # physical (read side) -> logical
        avoid = ({phys[b] for b in range(n) if wphys[b] in geom.final_free}
                 if (geom.final_free and k == len(pending) - 1) else None)
        touched |= set(low_list) | set(high)
        prefer = None
        if geom.pad_last_with_untouched and k == len(pending) - 1:
            prefer = [phys[b] for b in range(n) if b not in touched]      # (read positions, ascending logical bit)
        try:
            step = _finalize(tops, n, geom, thigh, trounds, [wphys[inv[p_]] for p_ in range(n)], avoid, prefer)
        except _Infeasible:
            return None
        step.permutes = wphys != phys
        phys = wphys
        out.append(step)
    assert phys == final
    return out


def _finalize(ops: Sequence[PrimOp], n: int, geom: Geometry, high: set[int], rounds: list[_Round],
              wpos: Sequence[int] | None = None, avoid_pad: set[int] | None = None,
              prefer_pad: Sequence[int] | None = None) -> FusedStep:
    """``wpos[p]`` = index bit the pass WRITES what it reads at index bit p to (None: where it was).  ``avoid_pad``: read
    positions the tile is not PADDED with (``Geometry.final_free``) while others are left; ``prefer_pad``: read positions
    it is padded with FIRST (``Geometry.pad_last_with_untouched``)."""
    m, R, vb = geom.m, geom.slots, geom.vb
    rounds = [r for r in rounds if r.ops]
    L = geom.min_low
    h = m - L
    # gathered bits: the ones gates need, then the lowest free bits above the contiguous part
    highs = set(high)
    for q in (prefer_pad or ()):
        if len(highs) >= h:
            break
        if q >= L and q not in highs and not (avoid_pad and q in avoid_pad):
            highs.add(q)
    p = L
    if avoid_pad and n - L - len(avoid_pad - highs) >= h:      # (enough other bits to pad with)
        while len(highs) < h:
            if p not in highs and p not in avoid_pad:
                highs.add(p)
            p += 1
    while len(highs) < h:
        if p not in highs:
            highs.add(p)
        p += 1
    assert len(highs) == h and max(highs) < n and h <= _lib.FUSED_MAX_HIGH
    order = sorted(highs)                  # tile bit L+i <-> order[i]
    local = {b: b for b in range(L)}
    for i, b in enumerate(order):
        local[b] = L + i
    tile = set(range(L)) | highs

    def io_layout(need: list[int]) -> list[int] | None:
        """Slots of an I/O-capable layout covering ``need`` (tile-local), or None."""
        if any(vb <= q < L for q in need):
            return None
        hi = [q for q in need if q >= L]
        if len(hi) > R - vb:
            return None
        pad = [q for q in range(m - 1, L - 1, -1) if q not in hi]
        hi = hi + pad[: R - vb - len(hi)]
        return sorted(list(range(vb)) + hi)

    def ascending_tb(slots_l: list[int]) -> list[int]:
        return [b for b in range(m) if b not in slots_l]

    desc = _lib.DqFusedPass()
    desc.m, desc.L, desc.h, desc.slots = m, L, h, R
    for i, b in enumerate(order):
        desc.high_pos[i] = b
        desc.high_sorted[i] = b

    needs = [sorted(local[b] for b in rd.slots) for rd in rounds]

    # a layout per round: the register slots the round's gates need, topped up with the slots already there (then the top
    # tile bits); the other tile bits go to the lanes in ascending order -- the kernel's translator (csrc/dq_wave.hip)
    # picks the lane order of every trip itself
    layouts: list[tuple[tuple[int, ...], tuple[int, ...]]] = []
    prev: tuple[tuple[int, ...], tuple[int, ...]] | None = None
    for rd, need in zip(rounds, needs):
        assert len(need) <= R
        if prev is not None and set(need) <= set(prev[0]):
            lay = prev
        else:
            io = io_layout(need)
            if io is not None:
                lay = (tuple(io), tuple(ascending_tb(io)))
            else:
                slots_l = list(need)
                cand = (list(prev[0])[::-1] if prev else []) + list(range(m - 1, -1, -1))
                for c in cand:
                    if len(slots_l) >= R:
                        break
                    if c not in slots_l:
                        slots_l.append(c)
                slots_l.sort()
                lay = (tuple(slots_l), tuple(ascending_tb(slots_l)))
        layouts.append(lay)
        prev = lay

    def is_io(lay) -> bool:
        sl = list(lay[0])
        return io_layout(sl) == sl and list(lay[1]) == ascending_tb(sl)

    default_io = io_layout([])
    load_rb = list(layouts[0][0]) if is_io(layouts[0]) else default_io

    def read_pos(tl: int) -> int:                  # tile-local bit -> index bit on the read side
        return tl if tl < L else order[tl - L]

    if wpos is None:
        wpos = list(range(n))
    wtile = [wpos[read_pos(tl)] for tl in range(m)]            # tile-local bit -> index bit on the write side
    to_low = sorted((tl for tl in range(m) if wtile[tl] < L), key=lambda tl: wtile[tl])
    if len(to_low) != L:
        raise _Infeasible           # a qubit wanted on the contiguous low bits is not in this tile
    if to_low == list(range(L)):
        # the low bits stay: the store layout is an I/O layout like the load's (the last round's if it is one)
        store_rb = list(layouts[-1][0]) if is_io(layouts[-1]) else default_io
        store_tb = ascending_tb(store_rb)
    else:
        # the low bits are re-labelled on the way out: slot 0 = the tile bit written to index bit 0 (complex64: a lane
        # stores two adjacent amplitudes), the tile bits written to index bits vb .. L-1 on the lowest lane bits (128
        # contiguous bytes per 8 lanes), the other thread bits in the order of their write positions
        store_rb = to_low[:vb]
        # (a layout change costs next to nothing, so the slots are simply the tile bits written FARTHEST away and the
        # lanes of a store instruction cover the longest contiguous runs the write positions allow)
        prefer = sorted(range(m), key=lambda tl: -wtile[tl])
        for c in prefer:
            if len(store_rb) >= R:
                break
            if c not in store_rb and c not in to_low:
                store_rb.append(c)
        store_rb = store_rb[:vb] + sorted(store_rb[vb:])
        store_tb = to_low[vb:] + sorted((tl for tl in range(m) if tl not in store_rb and tl not in to_low),
                                        key=lambda tl: wtile[tl])
    for s in range(R):
        desc.load_rb[s] = load_rb[s]
        desc.store_rb[s] = store_rb[s]
        desc.load_slot_off[s] = 1 << read_pos(load_rb[s])
        desc.store_slot_off[s] = 1 << wtile[store_rb[s]]
    for i, t in enumerate(store_tb):
        desc.store_tb[i] = t
    for i in range(L):
        desc.store_low_pos[i] = wtile[i]
    for i in range(h):
        desc.store_high_pos[i] = wtile[L + i]
    tileset = set(order)
    blk = [p_ for p_ in range(L, n) if p_ not in tileset]         # read position of block-index bit j
    assert len(blk) <= _lib.FUSED_MAX_BLK
    for j, pos in enumerate(blk):
        desc.store_blk_pos[j] = wpos[pos]

    exec_order: list[int] = []
    gi = 0
    ntrans = 0
    cur = (tuple(load_rb), tuple(ascending_tb(load_rb)))
    for ri, (rd, lay) in enumerate(zip(rounds, layouts)):
        r = desc.rounds[ri]
        r.flags = 0
        first = gi
        if lay != cur:
            ntrans += 1
            r.flags |= _lib.ROUND_TRANSPOSE
            cur = lay
        slot_of = {tl: s for s, tl in enumerate(lay[0])}
        for s in range(R):
            r.rb[s] = lay[0][s]
        for i, t in enumerate(lay[1]):
            r.tb[i] = t
        for oi in rd.ops:
            _encode_gate(desc.gates[gi], ops[oi], local, slot_of, tile)
            exec_order.append(oi)
            gi += 1
        r.gate_begin = first
        r.gate_end = gi
    if cur != (tuple(store_rb), tuple(store_tb)):
        ntrans += 1
        desc.rounds[len(rounds) - 1].flags |= _lib.ROUND_TRANSPOSE_AFTER
    desc.nrounds = len(rounds)
    return FusedStep(desc=desc, ops=exec_order, nrounds=len(rounds), ntranspose=ntrans, c64=vb == 1)


def _encode_gate(g: _lib.DqFusedGate, op: PrimOp, local: dict[int, int], slot_of: dict[int, int], tile: set[int]) -> None:
    reg_c, thr_c, out_c = 0, 0, 0
    for c in op.controls:
        if c in tile:
            tl = local[c]
            if tl in slot_of:
                reg_c |= 1 << slot_of[tl]
            else:
                thr_c |= 1 << tl
        else:
            out_c |= 1 << c
    g.reg_cmask, g.thr_cmask, g.out_cmask = reg_c, thr_c, out_c
    g.mat = op.mat
    g.q = g.q2 = g.loc = g.loc2 = 0
    g.fast = _lib.FAST_NONE
    g.mat_advance = 0
    g.reserved = 0

    def locate(b: int) -> tuple[int, int]:
        if b in tile:
            tl = local[b]
            if tl in slot_of:
                return _lib.LOC_REG, slot_of[tl]
            return _lib.LOC_THR, tl
        return _lib.LOC_OUT, b

    if op.kind == 'diag':
        g.kind = _lib.FG_DIAG1 if op.k == 1 else _lib.FG_DIAG2
        g.loc, g.q = locate(op.targets[0])
        if op.k == 2:
            g.loc2, g.q2 = locate(op.targets[1])
        return
    if op.kind == 'expz':
        g.kind = _lib.FG_EXPZ
        g.reserved = op.mode
        return
    slots = [slot_of[local[t]] for t in op.targets]
    if op.kind == 'grad':
        g.kind = _lib.FG_GRAD
        g.q, g.q2 = slots
        g.reserved = op.mode & GRAD_ROW_MASK
        g.loc = op.mode >> GRAD_VARIANT_SHIFT        # which sums the gate's gradient needs (include/dq_hip.h)
        return
    if op.k == 1:
        g.kind = _lib.FG_X1 if op.kind == 'x' else _lib.FG_GEN1
        g.q = slots[0]
        g.loc = op.mode if op.kind == 'gen' else 0
    else:
        g.kind = _lib.FG_GEN2
        g.q, g.q2 = slots
        # promised real (1; 4: and X-shaped, DQ_MODE_XREAL): channel superoperators; 5: X-shaped with complex entries (Rxx ...)
        g.loc = op.mode if op.mode in (1, 4, 5) else 0


def zero_state_masks(steps: Sequence, n: int, known_zero: int | None = None) -> list[int] | None:
    """The passes of a schedule run on the initial state |0..0> (the reference's default, circuit.py:49): an index bit no
    pass has had in its tile yet still factors out as |0>, so the state is zero wherever such a bit is 1 -- nothing there
    has to be read, computed or written (include/dq_hip.h, dq_apply_fused_zext_*).  Returns, per step, the mask of
    those bits on the READ side of the step (0 once every bit has been in a tile: from then on the passes are ordinary);
    None when it does not work out: a gate that runs on its own (it reads the whole buffer) while such bits are left, or
    bits left at the end (the result would hold uninitialised memory).  The contiguous low bits of a pass never count:
    a lane loads them in one piece, and they are in every tile from the first pass on (whose input is a real state).

    For the 28-qubit headline circuit: pass 0 touches one tile per sample, pass 1 2^8 of the 2^16, pass 2 reads 2^-8 of
    the state and writes all of it -- three of nineteen passes for the price of one pass's stores.

    ``known_zero``: not |0..0> but a state in which THESE index bits (positions before the first step) are known to be
    |0> -- a shard right after its first exchange: the qubits that came from the rank bits (`distributed._remap`)."""
    full = (1 << n) - 1
    # index bits (positions on the read side of the next step) that may be non-zero
    live = 0 if known_zero is None else full & ~known_zero
    masks: list[int] = []
    for st in steps:
        if live == full:
            masks.append(0)
            continue
        if not isinstance(st, FusedStep):
            return None
        d = st.desc
        L, h = d.L, d.h
        tile = [p_ for p_ in range(L)] + [d.high_pos[i] for i in range(h)]
        wtile = [d.store_low_pos[i] for i in range(L)] + [d.store_high_pos[i] for i in range(h)]
        tset = set(tile)
        blk = [p_ for p_ in range(L, n) if p_ not in tset]
        wpos = {p_: w for p_, w in zip(tile, wtile)}
        wpos.update({p_: d.store_blk_pos[j] for j, p_ in enumerate(blk)})
        low = (1 << L) - 1
        masks.append(full & ~live & ~low)
        now = live | sum(1 << p_ for p_ in tile)
        live = sum(1 << wpos[p_] for p_ in range(n) if (now >> p_) & 1)
    if live != full:
        return None
    return masks if any(masks) else None


def zero_state_cost(steps: Sequence, n: int, known_zero: int | None = None) -> float:
    """What the passes of a schedule move when the circuit starts from |0..0> (``known_zero``: from a state with THESE index
    bits known to be |0>), in units of a full pass (one read + one write of the state): `zero_state_masks` says which index
    bits are still known to be zero at every pass."""
    masks = zero_state_masks(steps, n, known_zero)
    if masks is None:
        return float(len(steps))
    cost = 0.0
    for st, kz in zip(steps, masks):
        if not kz:
            cost += 1.0
            continue
        nz = bin(kz).count('1')
        inside = sum(1 for i in range(st.desc.h) if (kz >> st.desc.high_pos[i]) & 1)
        cost += (2.0 ** -nz + 2.0 ** -(nz - inside)) / 2
    return cost


def layout_matrices(steps: Sequence, ops: Sequence[PrimOp]) -> tuple[list[int], int]:
    """Assign the kernel-side matrix layout: the matrices of a fused pass lie back to back in gate order
    (the kernel fetches gate i's matrix from a running pointer, together with the gate record), single-gate
    steps in between.  Patches ``mat`` / ``mat_advance`` / ``mat_base`` of every descriptor and ``pos`` of
    every op; returns (op indices in buffer order, buffer length in complex numbers incl. the tail pad)."""
    order: list[int] = []
    off = 0
    for st in steps:
        if isinstance(st, FusedStep):
            st.desc.mat_base = off
            for gi, oi in enumerate(st.ops):
                g = st.desc.gates[gi]
                op = ops[oi]
                size = 0 if g.kind in (_lib.FG_X1, _lib.FG_GRAD, _lib.FG_EXPZ) else (1 << op.k) ** 2
                g.mat, g.mat_advance, op.pos = off, size, off
                if size:
                    order.append(oi)
                    off += size
        else:
            op = ops[st.op]
            op.pos = off
            order.append(st.op)
            off += (1 << op.k) ** 2
    return order, off + _lib.MAT_PAD


def rx_defer_positions(steps: Sequence, ops: Sequence[PrimOp]) -> list[int]:
    """Offsets (kernel matrix buffer, after ``layout_matrices``) of the gates that run on the deferred Rx handlers:
    uncontrolled Rx-like gates of complex64 passes."""
    pos = []
    for st in steps:
        if isinstance(st, FusedStep) and st.c64:
            for gi, oi in enumerate(st.ops):
                if deferred_rx(st.desc.gates[gi]):
                    pos.append(ops[oi].pos)
    return pos


def deferred_rx(g) -> bool:
    """Does this record of a complex64 pass read the deferred Rx block (include/dq_hip.h, DQ_MODE_RX)?  An Rx-like 2x2
    gate without a control of any kind."""
    return (g.kind == _lib.FG_GEN1 and g.loc == 2 and g.reg_cmask == 0 and g.thr_cmask == 0 and g.out_cmask == 0)


def defer_rx(flat, index):
    """Rewrite, in the kernel matrix buffer ``flat`` (Bm, total), the blocks of the gates at ``index`` (a LongTensor of
    ``rx_defer_positions``) from the matrix  a I + i b X  to what the deferred handlers read (include/dq_hip.h,
    DQ_MODE_RX):  f = a, t = b / a  where |a| >= |b|,  f = i b, t = -a / b  elsewhere; per sample.  The scalar f
    leaves the gate and is applied once per pass: three packed operations per amplitude pair instead of four."""
    import torch

    if index is None or index.numel() == 0:
        return flat
    a = flat[:, index].real
    b = flat[:, index + 1].imag
    form1 = a.abs() < b.abs()
    one, zero = torch.ones_like(a), torch.zeros_like(a)
    t = torch.where(form1, -a / torch.where(form1, b, one), b / torch.where(form1, one, a))
    flat[:, index] = torch.complex(torch.where(form1, zero, a), torch.where(form1, b, zero))
    flat[:, index + 1] = torch.complex(zero, t)
    flat[:, index + 3] = torch.complex(form1.to(a.dtype), zero)
    return flat


def gather_matrices(src, ops: Sequence[PrimOp], order: Sequence[int]):
    """Kernel-side matrix buffer (Bm, total) from a caller buffer ``src`` (Bm, *) indexed by ``op.mat``."""
    import torch

    segs = [src[:, ops[oi].mat : ops[oi].mat + (1 << ops[oi].k) ** 2] for oi in order]
    segs.append(src.new_zeros(src.shape[0], _lib.MAT_PAD))
    return torch.cat(segs, dim=1).contiguous()


def kernel_matrices(steps: Sequence, ops: Sequence[PrimOp], src):
    """layout_matrices + gather_matrices for callers that drive ``apply_fused`` themselves (tests, tools):
    ``src`` is (Bm, *) or 1-D, indexed by ``op.mat``; returns the (Bm, total) buffer the descriptors expect."""
    import torch

    order, _total = layout_matrices(steps, ops)
    km = gather_matrices(src if src.ndim == 2 else src.reshape(1, -1), ops, order)
    pos = rx_defer_positions(steps, ops)
    return defer_rx(km, torch.tensor(pos, dtype=torch.long, device=km.device)) if pos else km


def algorithmic_bytes(ops: Sequence[PrimOp], n: int, amp_bytes: int, batch: int) -> int:
    """every touched amplitude read once and written once per gate."""
    return sum(2 * (1 << (n - len(op.controls))) * amp_bytes * batch for op in ops)
