/*
 * dq_hip.h -- C ABI of libdqhip.so, the MI355X (gfx950) statevector backend.
 *
 * This is the drop-in boundary for the QubitCircuit hot path of TuringQ/deepquantum.  The
 * reference has no FFI of its own (it is pure Python/PyTorch); every entry point below replaces
 * one Python-level seam of the reference, cited as <file:line> relative to the reference tree
 * (src/deepquantum/...).  The Python host (deepquantum_amd/_lib.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *   - All data pointers are DEVICE pointers unless the parameter is documented "host".
 *   - A state is `batch` contiguous vectors of 2^n interleaved complex numbers
 *     (c64 = 2 x float, c128 = 2 x double), index bit p <-> wire (n-1-p): wire 0 is the most
 *     significant bit of the flat amplitude index (operation.py:45-55, qmath.py:497-505).
 *   - `targets`/`controls` are bit positions (LSB = 0), host arrays.  targets[0] is the MOST
 *     significant bit of the gate-matrix index (qmath.py:502-503, wires[0] = matrix MSB).
 *   - Gate matrices are row-major 2^k x 2^k complex in the state's precision, DEVICE memory,
 *     `mat_batch_stride` = elements (complex numbers) between the matrices of consecutive batch
 *     samples (0 = one matrix shared by the whole batch; this is the vmap case of
 *     circuit.py:232-240).
 *   - Every call enqueues on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     returns without synchronising.  The library keeps no device memory and no global mutable
 *     state besides a thread-local error string; it is re-entrant.  The one exception is an A/B MEASUREMENT
 *     knob, not part of the data path's contract: dq_set_dense_path sets a process-wide (atomic) integer that
 *     later launches of every thread read; results never depend on it.
 *   - Return value: DQ_OK (0) or a negative DqStatus; dq_last_error() describes the failure.
 *     No exception crosses the ABI.
 */
#ifndef DQ_HIP_H
#define DQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dq_stream_t; /* hipStream_t */

typedef enum {
    DQ_OK = 0,
    DQ_ERR_ARG = -1,         /* invalid argument (bad bit index, overlap, null pointer, ...) */
    DQ_ERR_LAUNCH = -2,      /* HIP runtime reported an error at launch */
    DQ_ERR_UNSUPPORTED = -3  /* valid request outside what this build implements */
} DqStatus;

#define DQ_ABI_VERSION 25

int dq_abi_version(void);
/* Thread-local, never NULL. */
const char* dq_last_error(void);
/* What the compiled library believes the descriptor structs below look like: sizeof(DqFusedGate), sizeof(DqFusedRound),
 * sizeof(DqFusedPass), then the offsets of DqFusedPass::rounds, gates, load_slot_off, store_high_pos, store_tb,
 * slots -- at most `max` values written to `out`; returns how many there are.  Lets a binding check its own struct
 * definitions against the library instead of against numbers typed in by hand. */
int dq_struct_layout(int* out, int max);
/* Fills CU count, LDS bytes per workgroup, total global memory of the current device. */
int dq_device_info(int* cu_count, int64_t* lds_per_block, int64_t* global_mem);

/* ------------------------------------------------------------------------------------------
 * 1. Single gate application.  Replaces qmath.evolve_state (qmath.py:485-506) and
 *    Gate.op_state_control (operation.py:203-219): psi' = (U on targets, conditioned on all
 *    controls = 1) psi.  in == out is allowed (each amplitude group is private to one thread)
 *    when k <= 4; for k > 4, in and out must not alias.  k <= 10.
 * ------------------------------------------------------------------------------------------ */
int dq_apply_gate_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                      const int* targets, int k, const int* controls, int nc, int64_t batch,
                      dq_stream_t stream);
int dq_apply_gate_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                       const int* targets, int k, const int* controls, int nc, int64_t batch,
                       dq_stream_t stream);

/* Dense gates on 5..10 targets run as a GEMM on the matrix cores (csrc/dq_dense.hip; exact f32 / f64 MFMA).  A/B knob:
 * 0 = the round-1 kernel (one thread per output amplitude, VALU), 1 = MFMA (default). */
int dq_set_dense_path(int mfma);

/* ------------------------------------------------------------------------------------------
 * 2. Fused pass.  Replaces a run of consecutive Gate.forward calls inside
 *    nn.Sequential(self.operators) (circuit.py:261, operation.py:274-289): one HBM read + one HBM
 *    write of the state applies every gate of the pass.  A pass owns `m` "tile" index bits: the
 *    low `L` bits (contiguous, for coalescing) plus `h = m - L` gathered high bits; each
 *    workgroup stages one 2^m tile (registers + LDS) and walks the pass's rounds.  A round names
 *    R register-slot bits (tile-local positions); gates of the round act on register slots.
 *    The host scheduler (deepquantum_amd/fusion.py) builds these descriptors.
 *
 *    Geometry (dq_fused_geometry): the WAVE TILE.  complex64: m = 12, 6 register slots; complex128: m = 11, 5 slots;
 *    64 threads: ONE wavefront owns a tile (64 lanes x 64 / 32 amplitudes in registers), a workgroup is four
 *    independent waves, and a pass has no workgroup barrier at all; a layout change between rounds goes through a small
 *    wave-private LDS buffer.  The library derives everything below the level of "which tile bits are register slots
 *    in which round" itself (csrc/dq_wave.hip translates the descriptor into the kernel's records): the order of a
 *    round's slots and of its thread bits is the host's to choose freely.  Every DqFusedKind below runs there, in both
 *    precisions.  (ABI <= 20 also had workgroup-tile geometries -- m = 13 / 12 with 4 slots, 512 / 256 threads, the
 *    tile staged in LDS between rounds -- with handler ids, LDS offset tables and in-wave exchange records in this
 *    descriptor; they went with ABI 21.)
 * ------------------------------------------------------------------------------------------ */
#define DQ_FUSED_MAX_HIGH 12
#define DQ_FUSED_MAX_LOW 8      /* contiguous low tile bits: L <= 8 */
#define DQ_FUSED_MAX_ROUNDS 24
#define DQ_FUSED_MAX_GATES 128   /* (ABI <= 23: 80) */
#define DQ_FUSED_MAX_SLOTS 6

typedef enum {
    DQ_FG_GEN1 = 0,  /* general 2x2 on register slot q                       */
    DQ_FG_X1 = 1,    /* Pauli-X / CNOT / Toffoli: swap the pair of slot q     */
    DQ_FG_DIAG1 = 2, /* diagonal 2x2 (Z,S,T,Rz,P,CZ...) target anywhere      */
    DQ_FG_GEN2 = 3,  /* general 4x4 on register slots q (matrix MSB) and q2 */
    DQ_FG_DIAG2 = 4, /* diagonal 4x4 (Rzz...) both targets anywhere           */
    DQ_FG_RESERVED5 = 5, /* (ABI <= 20: an in-wave exchange record of the workgroup-tile kernels; refused) */
    DQ_FG_GRAD = 6,  /* not a gate: a reduction for the reverse sweep of the adjoint method (dq_apply_fused_grad_c64 / _c128).
                        The state is psi and the cotangent lambda side by side along ONE extra index bit (register slot
                        q2: 0 = psi, 1 = lambda); the record adds  G[a][b] = sum lambda[target = a] conj(psi[target = b])
                        (target = register slot q; the sum runs over everything else, restricted to the controls being
                        1) to row `reserved` of the caller's accumulator.  No matrix, no handler id.
                        `loc` says which of the eight real sums the record forms:
                        0 all; 1 Re G only (the trainable gate's matrix is real); 2 Re (G00 + G11), left in Re G00, and
                        Im (G01 + G10), left in Im G01 (a matrix a I + i b X: a Pauli-X rotation); 3 G00 and G11 only
                        (a diagonal matrix); 4 Im (G01 + G10) alone, left in Im G01 (a UNITARY a I + i b X whose gradient
                        is only ever taken along the rotation: the trace part cancels).  The OTHER components of the row are left untouched -- nothing is added to
                        them, so a caller that zeroed the accumulator reads zeros there and may use the whole row in
                        linear algebra (executor._first_order multiplies the 2x2 row by U^-dagger) */
    DQ_FG_EXPZ = 7   /* not a gate: the expectation value of a Z string taken from the registers, so that a circuit's
                        <Z..Z> observables cost no extra read of the state (replaces qmath.expectation qmath.py:830-860
                        for Z-type observables): adds  sum_i (-1)^popc(i & zmask) |a_i|^2  to component 0 of row
                        `reserved` of the accumulator of a dq_apply_fused_grad_* call.  The Z bits are given like
                        controls: reg_cmask (register slots), thr_cmask (tile-local bits on lanes), out_cmask (index
                        bits outside the tile).  Wave-tile geometries only.  No matrix */
} DqFusedKind;

typedef enum { DQ_LOC_REG = 0, DQ_LOC_THR = 1, DQ_LOC_OUT = 2 } DqBitLoc;

/* Structure of a GEN1 matrix, promised by the host from the gate class (never from values: that would
 * need a device sync): the kernel skips the multiplications by the exact zeros. */
typedef enum {
    DQ_MODE_GENERAL = 0,
    DQ_MODE_REAL = 1, /* all four entries real: H, Ry, X, Z ... */
    DQ_MODE_RX = 2,   /* a I + i b X (a, b real): Rx and products of such.  For an UNCONTROLLED gate of this mode in a
                         complex64 pass (fast id 8..11) the caller hands over, in place of the matrix, the block
                         { (f_re, f_im), (0, t), (-, -), (flag, -) }:  f = a and t = b / a, flag = 0  if |a| >= |b|,
                         f = i b and t = -a / b, flag = 1  otherwise -- the kernel applies [[1, it], [it, 1]]
                         resp. [[it, 1], [1, it]] (three packed operations per amplitude pair instead of four) and
                         multiplies the pass's deferred scalar by f (fusion.defer_rx builds the block) */
    DQ_MODE_HAD = 3,  /* s * [[1, 1], [1, -1]], s real: Hadamard.  The kernel takes sums and differences and
                         applies the product of the factors s of a pass once, at its end */
    DQ_MODE_XREAL = 4, /* GEN2 only (ABI 24): a REAL 4x4 matrix that is non-zero only where the parities of the row and
                         the column index agree -- a 2x2 block on (00, 11) and one on (01, 10).  The superoperator
                         sum K (x) conj(K) of every non-diagonal channel of channel.py:16-383 (bit flip, bit-phase flip,
                         depolarizing, Pauli, amplitude damping, generalized amplitude damping) on the (row, column) bit
                         pair of its wire.  The other entries are never read */
    DQ_MODE_XCPLX = 5 /* GEN2 only (ABI 24): the same shape -- a 2x2 block on (00, 11), one on (01, 10) -- with COMPLEX
                         entries: exp(-i theta XX / 2), exp(-i theta YY / 2), exp(-i theta (XX + YY) / 4)
                         (gate.py:2085-2390) and their controlled forms */
} DqFusedMode;

typedef struct {
    uint8_t kind;       /* DqFusedKind */
    uint8_t q;          /* GEN/X: register slot of the (first) target; DIAG: position per loc */
    uint8_t q2;         /* GEN2: slot of the second target (matrix LSB); DIAG2: position per loc2 */
    uint8_t loc;        /* DIAG1/2: DqBitLoc of target 1 (REG: q = slot, THR: q = tile-local bit,
                           OUT: q = global bit position); GEN1: DqFusedMode of the matrix; GEN2: GENERAL or
                           REAL (real 4x4, exact zeros skipped: channel superoperators) */
    uint8_t loc2;       /* DIAG2: same for target 2 */
    uint8_t reg_cmask;  /* controls that are register slots (bit s = slot s) */
    uint16_t thr_cmask; /* controls that are thread bits (tile-local bit positions) */
    uint32_t mat;       /* offset (in complex numbers) of this gate's matrix inside `mats` */
    uint32_t fast;      /* must be DQ_FAST_NONE (ABI <= 20: handler id of the workgroup-tile kernels' jump table) */
    uint64_t out_cmask; /* controls outside the tile (global bit positions) */
    uint32_t mat_advance; /* complex numbers this gate occupies in `mats` (0 for X1): the matrices of a pass
                             lie back to back in gate order, mat(i+1) = mat(i) + mat_advance(i), so the kernel
                             fetches gate i's matrix together with its record instead of after decoding it */
    uint32_t reserved;  /* DQ_FG_GRAD, DQ_FG_EXPZ: row of the accumulator; otherwise 0 (pads the record to 32 bytes: one
                           s_load_dwordx8 per gate) */
} DqFusedGate;          /* 32 bytes */
#define DQ_FAST_NONE 0xFFFFFFFFu
/* `mats` must be readable for DQ_MAT_PAD complex numbers past the last matrix of a pass (prefetch). */
#define DQ_MAT_PAD 16

#define DQ_FUSED_MAX_TBITS 9
#define DQ_FUSED_MAX_BLK 24      /* block-index bits: n - m <= 24 */
typedef struct {
    uint8_t rb[DQ_FUSED_MAX_SLOTS]; /* tile-local bit position of register slot s (any order; the I/O layouts of the
                                       pass header are ascending) */
    uint8_t tb[DQ_FUSED_MAX_TBITS]; /* tile-local bit position of thread-index bit i (the other m - slots
                                       tile bits, in an order the host picks to avoid LDS bank conflicts) */
    uint8_t flags;                  /* DQ_ROUND_TRANSPOSE: the layout (rb, tb) differs from the one the registers are in
                                       when the round starts (the previous round's, or the load layout for round 0);
                                       DQ_ROUND_TRANSPOSE_AFTER (last round only): it differs from the store layout.
                                       dq_apply_fused_* recomputes and checks them. */
    uint8_t gate_begin, gate_end;   /* [begin, end) into gates[] */
} DqFusedRound;                     /* 18 bytes */
#define DQ_ROUND_TRANSPOSE 0x01u
#define DQ_ROUND_TRANSPOSE_AFTER 0x02u

typedef struct {
    uint8_t m, L, h, nrounds;
    uint8_t high_pos[DQ_FUSED_MAX_HIGH];        /* global bit of tile bit L+i, any order */
    uint8_t high_sorted[DQ_FUSED_MAX_HIGH];     /* the same positions ascending (tile -> base) */
    /* Register slots (tile-local, ascending) of the layouts used for the global load and the global
     * store.  An I/O layout keeps tile bit 0 as slot 0 for c64 (16 B per lane) and otherwise only
     * gathered bits (>= L) as slots; its thread bits are the remaining tile bits in ascending order,
     * so a wave instruction always moves >= 2^L contiguous amplitudes. */
    uint8_t load_rb[DQ_FUSED_MAX_SLOTS];
    uint8_t store_rb[DQ_FUSED_MAX_SLOTS];
    DqFusedRound rounds[DQ_FUSED_MAX_ROUNDS];
    uint32_t mat_base;                          /* = gates[0].mat (also aligns gates[] to 32 bytes) */
    DqFusedGate gates[DQ_FUSED_MAX_GATES];
    /* Host-precomputed addressing of the two I/O layouts: offset (in amplitudes, inside one state) that
     * register slot s contributes, i.e. 2^(global bit of tile bit load_rb[s]); saves the kernel a scalar
     * loop per slot per tile. */
    uint64_t load_slot_off[DQ_FUSED_MAX_SLOTS];
    uint64_t store_slot_off[DQ_FUSED_MAX_SLOTS];
    /* Where the pass WRITES.  A pass may store its tile -- and its block index -- to other index bits (>= L) than
     * it read them from: a bit permutation of the state on the way out, so that the qubits of the NEXT pass already
     * sit in cheap (near) positions when that pass gathers them (scattered writes are nearly free on this memory
     * system, gathered reads are not: DESIGN.md).  store_high_pos[i] = global bit that tile bit L + i is written
     * to; store_blk_pos[j] = global bit that bit j of the block index (= the j-th lowest non-tile bit on the READ
     * side) is written to.  Together a permutation of [L, n).  In-place passes (in == out) must use the read
     * positions (store_high_pos == high_pos, store_blk_pos ascending): anything else needs in != out.
     * store_slot_off is in write positions. */
    uint8_t store_high_pos[DQ_FUSED_MAX_HIGH];
    uint8_t store_blk_pos[DQ_FUSED_MAX_BLK];
    /* The low tile bits move too: tile bit i < L is written to global bit store_low_pos[i] (identity for an in-place
     * pass), so that the NEXT pass may find other qubits on its contiguous low bits -- every pass then chooses all m
     * tile qubits freely instead of sharing L fixed ones with every other pass, provided the qubits it wants on its low
     * bits were in the tile of the pass before (which must write them as contiguous runs).  store_low_pos,
     * store_high_pos and store_blk_pos together are a permutation of [0, n).  The store layout is explicit for the
     * same reason: register slots store_rb (any tile bits; complex64: the tile bit of slot 0 must be written to
     * global bit 0, a lane stores two adjacent amplitudes) and thread bits store_tb (the other tile bits; the host
     * puts the tile bits written to global bits 1 .. L-1 on the lowest lane bits: 128 contiguous bytes per 8 lanes). */
    uint8_t store_low_pos[DQ_FUSED_MAX_LOW];
    uint8_t store_tb[DQ_FUSED_MAX_TBITS];
    uint8_t slots;                              /* register slots per thread of the geometry the pass was planned for
                                                   (dq_fused_geometry); with m it selects the kernel */
} DqFusedPass;

/* The tile geometry of each precision (m = slots + log2(threads)): variant 0; DQ_ERR_ARG for any other. */
int dq_fused_geometry(int is_c128, int variant, int* m, int* slots, int* threads);
/* Host-side core of the pass planner (no device code; deepquantum_amd/fusion.py drives it): dry runs of a pass over the
 * commutation DAG of a gate list.  dq_dag_create copies the DAG: successors of gate i = succ[succ_off[i] .. succ_off[i+1]),
 * target_mask[i] = the qubits gate i needs inside the tile (0: it runs anywhere, e.g. a diagonal gate), fusable[i] = may it
 * enter a fused pass at all.  dq_dag_closure retires, depth-first from the `ready` gates, every fusable gate whose
 * targets lie in `tile` (bit mask over qubits) while fewer than `cap` have retired; returns how many did, and optionally
 * the gates left stuck at the front and the in-degrees it changed (index / value pairs; `indeg` itself is not written).
 * dq_dag_rank does the count for tile | (1 << cand[c]), every c.  A handle is not thread-safe (it carries scratch). */
void* dq_dag_create(int n_ops, const int* succ_off, const int* succ, const uint64_t* target_mask, const uint8_t* fusable);
void dq_dag_destroy(void* dag);
int dq_dag_closure(void* dag, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, int* stuck, int* nstuck,
                   int* changed_idx, int* changed_val, int* nchanged);
int dq_dag_rank(void* dag, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, const int* cand, int ncand,
                int* counts);
/* One growth step of a tile: *base = dq_dag_closure(tile); the qubits outside the tile that the stuck gates wait for
 * (cand_q, cand_w = how many gates wait for each; arrays of 64), and cand_count = dq_dag_closure(tile | qubit) for each.
 * Returns the number of candidates (0 when the pass is full: *base >= cap). */
int dq_dag_grow_step(void* dag, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, int* base, int* cand_q,
                     int* cand_w, int* cand_count);

/* The deferred form of the uncontrolled Rx-like gates of a complex64 pass (DQ_MODE_RX above), in place, in the matrix
 * buffer the passes will read: for every sample b < batch and every k < count the block of four complex numbers at
 * mats[b * mat_batch_stride + index[k]] -- { a, i b, i b, a } of the matrix a I + i b X -- becomes { f, i t, -, flag }.
 * `index`: DEVICE array of int64 offsets (complex numbers), e.g. fusion.rx_defer_positions of a schedule.  One launch;
 * host code that prepares matrix buffers for dq_apply_fused_c64 itself calls this instead of re-deriving the form
 * (no reference counterpart: the reference multiplies by the full matrix, gate.py:1443-1448). */
int dq_defer_rx_c64(void* mats, int64_t mat_batch_stride, const int64_t* index, int64_t count, int64_t batch,
                    dq_stream_t stream);

/* Test hook, no device needed: the kernel-side descriptor the library derives for a wave-tile pass (`pass`: HOST
 * pointer, complex64, m = 12, 6 slots) as raw bytes -- 80 bytes of slot offsets (load, store: 5 x 8 each), 6 + 6 + 6
 * words (byte shift of every lane bit on the load side / the store side, what it adds to the thread's tile-local base),
 * record bytes, matrix base, 24 + 24 index positions of the tile number's bits (read, write), one word for
 * dq_apply_fused_zext_* (`known_zero`; bits 0..5: how many bits the tile number has, 8..13 / 16..21: register slots /
 * lane bits that are not loaded), 2 + 2 words for dq_apply_fused_slice_* (index bits held fixed, OR-ed into every tile's
 * read / write base; zero here) + 3 reserved words, then the 32-byte records (word 0 = handler id, csrc/dq_wave_asm.inc).  At most `max_bytes` are copied to `out` (may be NULL); returns the size,
 * or a negative DqStatus.  tests/_wave_emulator.py executes such a descriptor on the CPU. */
int dq_wave_descriptor(const DqFusedPass* pass, int n, uint64_t known_zero, void* out, int max_bytes);
/* `pass` is a HOST pointer; it is copied into the kernel argument segment.  in == out allowed.
 * Requires n >= pass->m. */
int dq_apply_fused_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                       int64_t batch, const DqFusedPass* pass, dq_stream_t stream);
int dq_apply_fused_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                        int64_t batch, const DqFusedPass* pass, dq_stream_t stream);
/* Same, with `in` = ONE state of 2^n amplitudes shared by all `batch` outputs (in != out): the first pass of a
 * batched circuit reads the initial state directly, where the reference's vmap (circuit.py:238) materialises a
 * copy per sample. */
int dq_apply_fused_bcast_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                             int64_t batch, const DqFusedPass* pass, dq_stream_t stream);
int dq_apply_fused_bcast_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                              int64_t batch, const DqFusedPass* pass, dq_stream_t stream);

/* Same pass on an input in which the index bits of `known_zero` (bit mask, READ side of the pass) are known to be |0>:
 * the circuit's own initial state |0..0> (the reference's default, circuit.py:49 `init_state='zeros'`) and the passes
 * right behind it, while the qubits the circuit has not touched yet still factor out as |0>.
 *   - `in` is not read where one of these bits is 1; it need not be initialised there.  Inside the tile the registers of
 *     those halves are zero; tiles in which a known-zero bit OUTSIDE the tile is 1 are all zero and are skipped: neither
 *     read, computed nor written.
 *   - `out` is therefore left untouched where a known-zero bit outside the tile is 1 (at the position the pass writes that
 *     bit to): those bits are still known zero for the next pass; the tile's own bits are not any more.
 *   - none of the pass's L contiguous low bits may be named (a lane loads them in one piece), nor a bit >= n.
 * in_batch_stride: 2^n, or 0 = ONE input state shared by all `batch` outputs (as dq_apply_fused_bcast_*).
 * The first pass of the 28-qubit headline circuit then touches one tile per sample, the second 2^8, the third is
 * write-only: three of nineteen passes for the price of the third one's stores (deepquantum_amd/fusion.py,
 * zero_state_masks). */
int dq_apply_fused_zext_c64(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                            int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, dq_stream_t stream);
int dq_apply_fused_zext_c128(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                             int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, dq_stream_t stream);

/* ABI 25.  ONE SLICE of a pass: only the tiles whose index bits `slice_mask` (read side) equal `slice_value` run;
 * 2^popcount(slice_mask) such launches are the whole pass, bit for bit.  `slice_mask` names index bits < n OUTSIDE the
 * tile of the pass (not its contiguous low bits, not its gathered bits) and outside `known_zero`; `slice_value` is a subset
 * of it.  Gates controlled by such a bit see its value.  For the index-bit-sharded state (distributed.py:57-202 of the
 * reference exchanges whole shards gate by gate): the last pass in front of a k-qubit exchange is launched slice by
 * slice -- each slice's part of every chunk leaves for its peer while the next slice computes -- and the first pass
 * behind it starts with the first slice that has arrived. */
int dq_apply_fused_slice_c64(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                             int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, uint64_t slice_mask,
                             uint64_t slice_value, dq_stream_t stream);
int dq_apply_fused_slice_c128(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                              int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, uint64_t slice_mask,
                              uint64_t slice_value, dq_stream_t stream);

/* Reverse sweep of the adjoint method in fused passes (replaces the backward of autograd through circuit.py:261, one
 * matmul backward per gate -- qmath.py:504 -- with one saved state per gate).  `in` / `out` hold, per sample, psi and
 * the cotangent lambda as ONE state of n index bits in which one bit tells them apart; the pass un-applies gates from
 * both at once (the caller supplies the adjoint matrices) and its DQ_FG_GRAD records reduce, at the right moments of
 * the sweep, sum lambda (x) conj(psi) onto a trainable gate's target.  `grads` = DEVICE double [batch, ngrads, 8],
 * ADDED to (the caller zeroes it once per sweep): row r = Re, Im of G[0][0], G[0][1], G[1][0], G[1][1] of the record
 * with reserved = r.  Workgroups accumulate in LDS over their tiles (float32 sums for complex64, float64 for
 * complex128) and add to `grads` once.  complex128: wave-tile geometry only (m = 11, 5 slots). */
int dq_apply_fused_grad_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                            int64_t batch, const DqFusedPass* pass, double* grads, int64_t ngrads, dq_stream_t stream);
int dq_apply_fused_grad_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                             int64_t batch, const DqFusedPass* pass, double* grads, int64_t ngrads, dq_stream_t stream);

/* ABI 24: a pass with more records than the kernel-argument segment holds (112: gates and reductions, two or four per
 * layout change).  A reverse sweep has a reduction in front of every trainable gate, and what bounded its passes was
 * the NUMBER of records, not the tile: the 28-qubit training step takes 23 passes of up to 104 instead of 32 of up to 72.
 * The kernel reads such a pass's records from DEVICE memory:
 *   dq_wave_records   (host, no GPU needed) writes the records of `pass` as the kernel reads them -- 32 bytes each -- to
 *                     the HOST buffer `out` (at most max_bytes; out = NULL: none) and returns their size in bytes, or a
 *                     negative DqStatus;
 *   dq_apply_fused_grad_ext_*   run the pass with those records at the DEVICE address `records` (the caller copied them
 *                     there; `records_bytes` must be what dq_wave_records returned for this pass and n).  They must stay
 *                     valid until the pass has run -- captured in a HIP graph: until its last replay.
 * Everything else as dq_apply_fused_grad_*. */
int64_t dq_wave_records(const DqFusedPass* pass, int n, void* out, int64_t max_bytes);
int dq_apply_fused_grad_ext_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n, int64_t batch,
                                const DqFusedPass* pass, const void* records, int64_t records_bytes, double* grads,
                                int64_t ngrads, dq_stream_t stream);
int dq_apply_fused_grad_ext_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n, int64_t batch,
                                 const DqFusedPass* pass, const void* records, int64_t records_bytes, double* grads,
                                 int64_t ngrads, dq_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3. Reductions.  Results are written to DEVICE memory in double precision.
 *    `ws` is a caller-owned device workspace of at least dq_reduce_ws_bytes(batch) bytes.
 * ------------------------------------------------------------------------------------------ */
int64_t dq_reduce_ws_bytes(int64_t batch);

/* out[b] = Re <psi_b| P |psi_b>, P = Pauli string given as bit masks (a Y sets its bit in both
 * xmask and zmask).  Replaces qmath.expectation (qmath.py:830-860) + Observable.forward
 * (layer.py:127-165): one read of the state instead of one gate pass per Pauli factor + bmm. */
int dq_expect_pauli_c64(const void* psi, uint64_t xmask, uint64_t zmask, int n, int64_t batch,
                        double* out, void* ws, dq_stream_t stream);
int dq_expect_pauli_c128(const void* psi, uint64_t xmask, uint64_t zmask, int n, int64_t batch,
                         double* out, void* ws, dq_stream_t stream);

/* k <= 32 Z-type strings (xmask = 0) in ONE read of the state: `zmasks` is a HOST array; `nblocks` workgroups stride
 * over the state and each writes one row of partial sums: out = DEVICE double [batch, nblocks, k], fully overwritten;
 * the caller adds the nblocks rows.  A Hamiltonian of many ZZ terms (examples/qaoa.py:31-44: one observable per
 * edge, each a separate pass in qmath.expectation) costs one pass. */
int dq_expect_zmulti_c64(const void* psi, const uint64_t* zmasks, int k, int n, int64_t batch, double* out, int nblocks,
                         dq_stream_t stream);
int dq_expect_zmulti_c128(const void* psi, const uint64_t* zmasks, int k, int n, int64_t batch, double* out, int nblocks,
                          dq_stream_t stream);
/* out[b, i] = psi[b, i] * sum_j coef[b, j] (-1)^{popc(i & zmasks[j])}  (coef: DEVICE double [batch, k]): the
 * gradient of the above, i.e. (sum_j coef_j Z-string_j) |psi>, in one read + one write.  out may be psi. */
int dq_scale_zsigns_c64(const void* psi, void* out, const uint64_t* zmasks, int k, const double* coef, int n,
                        int64_t batch, dq_stream_t stream);
int dq_scale_zsigns_c128(const void* psi, void* out, const uint64_t* zmasks, int k, const double* coef, int n,
                         int64_t batch, dq_stream_t stream);

/* out[2b], out[2b+1] = Re, Im of <bra_b|ket_b> over `count` amplitudes.
 * Replaces inner_product_dist's local part (distributed.py:288-291) and `state.mH @ x`. */
int dq_inner_c64(const void* bra, const void* ket, int64_t count, int64_t batch, double* out, void* ws,
                 dq_stream_t stream);
int dq_inner_c128(const void* bra, const void* ket, int64_t count, int64_t batch, double* out, void* ws,
                  dq_stream_t stream);

/* probs[b, i] = |psi[b, i]|^2 in the state's real precision (qmath.py:624). */
int dq_probs_c64(const void* psi, void* probs, int64_t count, dq_stream_t stream);
int dq_probs_c128(const void* psi, void* probs, int64_t count, dq_stream_t stream);

/* Marginal distribution over `nw` bit positions (host array, bits[0] = MSB of the outcome index):
 * out[b, o] = sum over the other bits of |psi|^2, double precision, out must be zeroed by the
 * caller.  1 <= nw <= n <= 40, batch <= 65535, psi 16-byte aligned.  A workgroup owns 2^12 amplitudes (the low index
 * bits plus the lowest unmeasured ones) with a histogram in LDS over the measured bits among them: reads are coalesced
 * whatever the measured bits are.  Replaces the permute/reshape/sum of qmath.py:626. */
int dq_marginal_c64(const void* psi, int n, const int* bits, int nw, int64_t batch, double* out,
                    dq_stream_t stream);
int dq_marginal_c128(const void* psi, int n, const int* bits, int nw, int64_t batch, double* out,
                     dq_stream_t stream);

/* Gradient of a gate matrix: gU[b, i, j] += sum_groups gy[i] * conj(x[j]) over the amplitude groups
 * whose controls are all 1 (the matmul backward of qmath.py:504 / operation.py:216).  k <= 2.
 * gU is DEVICE double-precision complex [batch, 2^k, 2^k], zeroed by the caller. */
int dq_gate_grad_c64(const void* x, const void* gy, int n, const int* targets, int k, const int* controls,
                     int nc, int64_t batch, double* gU, dq_stream_t stream);
int dq_gate_grad_c128(const void* x, const void* gy, int n, const int* targets, int k, const int* controls,
                      int nc, int64_t batch, double* gU, dq_stream_t stream);

/* The same quantity for SEVERAL single-target gates on one pair of states, in one read of both (the reverse sweep of
 * the adjoint autograd node asks for all trainable gates of a circuit layer at once): gate g has target
 * targets[g] and controls ctrl_bits[ctrl_begin[g] .. ctrl_begin[g+1]).  Per call at most 8 gates (c64) / 4 (c128)
 * whose targets above bit 3 (c64) / 2 (c128) number at most 7, and n >= 11 (c64) / 10 (c128): DQ_ERR_UNSUPPORTED
 * otherwise (callers fall back to dq_gate_grad_*).  `nblocks` workgroups stride over the 2^(n-11) (c64) / 2^(n-10)
 * (c128) tiles; each writes ONE row of partial sums: out = DEVICE double complex [batch, nblocks, ngates, 2, 2], fully
 * overwritten; the caller adds the nblocks rows. */
int dq_gate_grad_multi_c64(const void* x, const void* gy, int n, int ngates, const int* targets, const int* ctrl_begin,
                           const int* ctrl_bits, int64_t batch, double* out, int nblocks, dq_stream_t stream);
int dq_gate_grad_multi_c128(const void* x, const void* gy, int n, int ngates, const int* targets, const int* ctrl_begin,
                            const int* ctrl_bits, int64_t batch, double* out, int nblocks, dq_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 4. Shard exchange helpers for the index-bit-partitioned state (distributed.py:57-202).
 *    `nl` = number of local qubits (shard = 2^nl amplitudes per batch sample).
 * ------------------------------------------------------------------------------------------ */
/* packed[b, c] = amps[b, expand(c)] where expand() inserts, at the bit positions set in `mask`,
 * the corresponding bits of `value` (mask/value over local bit positions).  Replaces the
 * arange/boolean-mask gather of distributed.py:109-117, 150-155. */
int dq_pack_c64(const void* amps, void* packed, int nl, uint64_t mask, uint64_t value, int64_t batch,
                dq_stream_t stream);
int dq_pack_c128(const void* amps, void* packed, int nl, uint64_t mask, uint64_t value, int64_t batch,
                 dq_stream_t stream);
/* amps[b, expand(c)] = a * x[b, c] + bcoef * y[b, c]; coef = DEVICE pointer to 2 complex numbers
 * {a, bcoef} per batch sample (coef_batch_stride in complex numbers, 0 = shared) in the state's
 * precision.  With y == NULL: amps[b, expand(c)] = x[b, c] (pure scatter, SWAP local<->global).
 * Replaces distributed.py:70, 126, 157. */
int dq_unpack_axpby_c64(void* amps, const void* x, const void* y, const void* coef, int64_t coef_batch_stride,
                        int nl, uint64_t mask, uint64_t value, int64_t batch, dq_stream_t stream);
int dq_unpack_axpby_c128(void* amps, const void* x, const void* y, const void* coef, int64_t coef_batch_stride,
                         int nl, uint64_t mask, uint64_t value, int64_t batch, dq_stream_t stream);

/* out[b, i] = in[b, sigma(i)] with sigma(i) = sum_p bit_p(i) << src_of_dst[p] (host array of nl entries, a
 * permutation of 0..nl-1): re-labels the local qubits of a shard in one read + one write.  in != out.
 * Used by the all-to-all qubit remap (swap k global qubits with k local ones in ONE exchange step over all
 * xGMI links instead of the reference's one pairwise exchange per gate, distributed.py:57-202) and to restore
 * the canonical order (the reshape/transpose copy of local_swap_gate, distributed.py:48-54). */
int dq_permute_bits_c64(const void* in, void* out, int nl, const int* src_of_dst, int64_t batch,
                        dq_stream_t stream);
int dq_permute_bits_c128(const void* in, void* out, int nl, const int* src_of_dst, int64_t batch,
                         dq_stream_t stream);

/* ABI 24.  Two arrays of `count` amplitudes side by side along a NEW index bit 0: out[2 i] = a[i], out[2 i + 1] = b[i] --
 * how a fused reverse sweep (dq_apply_fused_grad_*) wants psi and the cotangent lambda, i.e. the final state of
 * circuit.py:261 and the gradient autograd hands back for it -- and the way back: out[i] = in[2 i + which].  `count`
 * even (all samples of a batch at once: count = batch * 2^n); out of place; full coalesced lines both ways.  All
 * buffers 16-byte aligned (a complex64 view at an odd element offset is not: DQ_ERR_ARG). */
int dq_interleave_c64(const void* a, const void* b, void* out, int64_t count, dq_stream_t stream);
int dq_interleave_c128(const void* a, const void* b, void* out, int64_t count, dq_stream_t stream);
int dq_deinterleave_c64(const void* in, void* out, int64_t count, int which, dq_stream_t stream);
int dq_deinterleave_c128(const void* in, void* out, int64_t count, int which, dq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DQ_HIP_H */
