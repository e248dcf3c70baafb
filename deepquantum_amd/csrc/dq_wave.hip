// Wave-tile fused pass for gfx950 (complex64): ONE wavefront owns a tile of 2^12 amplitudes -- 64 lanes x 64
// amplitudes in 128 VGPRs -- so a pass has no workgroup barrier at all: every wave loads its tile (32 x 16 bytes per
// lane), walks the pass's records (gates on the six register-slot bits, layout changes through a small wave-private LDS
// buffer), and stores.  Replaces a run of consecutive Gate.forward calls of the reference (circuit.py:261 ->
// operation.py:274-289 -> qmath.py:485-506 / operation.py:203-219).
//
// Why: the workgroup-tile kernel of rounds 1-2 (512 threads, 64 KiB of LDS, two workgroups per CU; removed) spent a third of a
// pass waiting -- its eight waves meet at two barriers per layout change and only two such workgroups fit a CU, so VALU
// (65 % busy) and HBM (78 %) never overlap fully.  With wave-private tiles there is nothing to wait for: measured on
// the headline state, 96 Hadamard-sized gates + 4 layout changes per tile run at the speed of the bare load / store
// skeleton (tools/experiments/mb_wavetile.hip: 13.3 vs 13.2 ms per pass; the workgroup-tile kernel: 18.7 ms).
//
// Layout changes ("trips").  Registers hold the slot bits, lanes the other six tile bits.  When k slot bits trade
// places with k lane bits, the tile falls apart into 2^(6-k) sub-tiles (one per value of the slots that stay), each
// 2^k registers x 64 lanes, which are transposed one after the other through the same 2^k x (64 + pad) x 8-byte buffer:
// DS operations of one wave execute in order, so neither a barrier nor a wait separates the groups.  All addresses are
// a per-lane base (a few VALU operations per trip) plus an immediate.  k <= 4 (8.4 KiB per wave, twelve waves per CU);
// bigger changes take two trips.
//
// The host describes a pass by rounds (which tile bits are register slots when); translate() below turns that into the
// flat record list the kernel walks: it tracks which PHYSICAL slot (register-index bit) holds which tile bit, so the
// host's slot order never costs a register move, picks the lane order of every layout (bank-conflict-free where it
// can), and computes the per-trip address contributions.
#include <utility>

#include "dq_common.hpp"
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

namespace dq {

#include "dq_wave_asm.inc"
#include "dq_wave_asm64.inc"

constexpr int WAVE_LANES = 6;
constexpr int WAVE_MAX_REC = 112;      // records that travel in the kernel-argument segment (4 KiB)
constexpr int WAVE_EXT_REC = 256;      // ... of a pass whose records the caller keeps in DEVICE memory (dq_apply_fused_grad_ext_*)

// The two precisions: complex64 -- 64 amplitudes per lane, six slots, a 12-bit tile, slot 0 = index bit 0 inside the
// 16-byte piece a lane loads -- and complex128 -- 32 amplitudes per lane, five slots, an 11-bit tile, every slot a
// gathered bit.  Handler ids and the trip / swap tables come from the generated files.
struct WaveC64 {
    using real = float;
    using acc_t = float;        // the reverse sweep's accumulators in LDS
    static constexpr int M = 12, R = 6, NA = 64, VB = 1, MAXK = DQ_WAVE_MAXK, ELEM = 8, GRAD_VARIANTS = DQ_WAVE_GRAD_VARIANTS;
    static constexpr unsigned LDS_PER_WAVE = 8448;      // (15 * 66 + 64) * 8 bytes: the k = 4 sub-tile buffer
    static constexpr int ID_GEN_U = DQ_WID_GEN_U, ID_GEN_C = DQ_WID_GEN_C, ID_GEN_R = DQ_WID_GEN_R, ID_X_U = DQ_WID_X_U,
                         ID_X_C = DQ_WID_X_C, ID_X_R = DQ_WID_X_R, ID_X_R1 = DQ_WID_X_R1, ID_TRIP0 = DQ_WID_TRIP0,
                         ID_DIAG1 = DQ_WID_DIAG1, ID_DIAG2 = DQ_WID_DIAG2, ID_GRAD = DQ_WID_GRAD, ID_EXPZ = DQ_WID_EXPZ, ID_GEN2 = DQ_WID_GEN2, ID_GEN2R = DQ_WID_GEN2R, ID_GEN2X = DQ_WID_GEN2X, ID_GEN2XC = DQ_WID_GEN2XC, ID_SWAP = DQ_WID_SWAP;
    static int trip_id(unsigned mask) { return kWaveTripId[mask]; }
    static int swap_id(int i, int j) { return kWaveSwapId[i][j]; }
    __device__ static __forceinline__ void body(uint64_t kg, uint32_t gend, uint64_t mb, uint32_t moff, uint64_t tg, uint64_t ks,
                                                uint64_t inb, uint64_t outb, uint32_t ldsb, uint32_t tid, uint32_t flags) {
        wave_tile_body_f32(kg, gend, mb, moff, tg, ks, inb, outb, ldsb, tid, flags);
    }
};
struct WaveC128 {
    using real = double;
    using acc_t = double;
    static constexpr int M = 11, R = 5, NA = 32, VB = 0, MAXK = DQ_WAVE64_MAXK, ELEM = 16, GRAD_VARIANTS = DQ_WAVE64_GRAD_VARIANTS;
    static constexpr unsigned LDS_PER_WAVE = 8704;      // (7 * 68 + 64) * 16 bytes: the k = 3 sub-tile buffer
    static constexpr int ID_GEN_U = DQ_WID64_GEN_U, ID_GEN_C = DQ_WID64_GEN_C, ID_GEN_R = DQ_WID64_GEN_R, ID_X_U = DQ_WID64_X_U,
                         ID_X_C = DQ_WID64_X_C, ID_X_R = DQ_WID64_X_R, ID_X_R1 = DQ_WID64_X_R1, ID_TRIP0 = DQ_WID64_TRIP0,
                         ID_DIAG1 = DQ_WID64_DIAG1, ID_DIAG2 = DQ_WID64_DIAG2, ID_GRAD = DQ_WID64_GRAD, ID_EXPZ = DQ_WID64_EXPZ, ID_GEN2 = DQ_WID64_GEN2, ID_GEN2R = DQ_WID64_GEN2R, ID_GEN2X = -1, ID_GEN2XC = -1, ID_SWAP = DQ_WID64_SWAP;      // (two-target dense gates: the 64 dwords of matrix pass through the scalar registers two rows at a time)
    static int trip_id(unsigned mask) { return kWave64TripId[mask]; }
    static int swap_id(int i, int j) { return kWave64SwapId[i][j]; }
    __device__ static __forceinline__ void body(uint64_t kg, uint32_t gend, uint64_t mb, uint32_t moff, uint64_t tg, uint64_t ks,
                                                uint64_t inb, uint64_t outb, uint32_t ldsb, uint32_t tid, uint32_t flags) {
        wave_tile_body_f64(kg, gend, mb, moff, tg, ks, inb, outb, ldsb, tid, flags);
    }
};

template <typename T> using vec2 = T __attribute__((ext_vector_type(2)));

struct WaveRec {
    uint32_t w[8];
};

// Kernel-side descriptor (by value in the kernel-argument segment).
struct WaveKernPass {
    uint64_t load_off[5];           // bytes that physical slots 1..5 add on the read side
    uint64_t store_off[5];          // ... on the write side
    uint32_t load_lane_shift[6];    // lane bit b adds 1 << load_lane_shift[b] bytes to the load address
    uint32_t store_lane_shift[6];   // ... to the store address
    uint32_t tb_contrib[6];         // ... and this to the thread's tile-local base in the load layout
    uint32_t nrec_bytes, mat_base_bytes;
    uint8_t read_blk_pos[DQ_FUSED_MAX_BLK];    // index bit (read side) of bit j of the tile number; unused entries: any
    uint8_t store_blk_pos[DQ_FUSED_MAX_BLK];   // ... on the write side
    // bits 0..5: how many bits the tile number has (n - m, less the index bits outside the tile that are known to be |0>
    // in the input: dq_apply_fused_zext_*); bits 8..13 / 16..21: the physical register slots / lane bits of the load
    // layout that hold such bits (nothing is loaded where one of them is 1: the registers are zero)
    uint32_t zext;
    // dq_apply_fused_slice_*: index bits outside the tile that are NOT bits of the tile number either but held at a value --
    // OR-ed into every tile's read / write base (64 bits each, low word first); zero for a whole pass
    uint32_t fix_read[2], fix_write[2];
    uint32_t reserved[3];
    WaveRec rec[WAVE_MAX_REC];
};
static_assert(offsetof(WaveKernPass, store_off) == 40 && offsetof(WaveKernPass, load_lane_shift) == 80 &&
                  offsetof(WaveKernPass, tb_contrib) == 128 && offsetof(WaveKernPass, zext) == 208 &&
                  offsetof(WaveKernPass, rec) == 240, "descriptor layout");

struct WaveKernArgs {
    const void* in;
    void* out;
    const void* mats;
    int64_t mat_bstride;
    int64_t in_bstride;
    int n;
    int tpw;
    WaveKernPass p;
    double* grads;
    int64_t grad_bstride;
    const void* ext_rec;        // the records in device memory (a pass with more than WAVE_MAX_REC of them), or null
};
static_assert(sizeof(WaveKernArgs) <= 4096 && (offsetof(WaveKernArgs, p) + offsetof(WaveKernPass, rec)) % 32 == 0, "kernel-argument segment");

// GRAD: a pass of the adjoint method's reverse sweep (dq_apply_fused_grad_c64).  Its DQ_FG_GRAD records add their sums to
// accumulators in LDS (32 bytes per record, behind the staging buffers); a wave walks `tpw` tiles so that a workgroup
// adds to the caller's float64 `grads` only once: one atomic per record and component and 4 * tpw tiles.
template <class W, bool GRAD>
__global__ __launch_bounds__(256) void wave_pass_kernel(const vec2<typename W::real>* in, vec2<typename W::real>* out,
                                                        const vec2<typename W::real>* mats, int64_t mat_bstride,
                                                        int64_t in_bstride, int n, int tpw_flags, const WaveKernPass p,
                                                        double* grads, int64_t grad_bstride, const void* ext_rec) {
    // (low byte: log2 of the tiles a wave walks; bits 16, 17: streaming loads / stores)
    (void)in, (void)out, (void)mats, (void)mat_bstride;
    extern __shared__ __attribute__((aligned(16))) unsigned char dq_wave_smem[];
    (void)dq_wave_smem;
    const unsigned tid = threadIdx.x;
    if constexpr (GRAD) {
        for (unsigned i = tid; i < p.nrec_bytes / 4u; i += 256u)        // (eight accumulators per record)
            *(__attribute__((address_space(3))) typename W::acc_t*)(uintptr_t)(4u * W::LDS_PER_WAVE + (unsigned)sizeof(typename W::acc_t) * i) = 0;
        __syncthreads();
    }
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    unsigned grp = blockIdx.x, sample = blockIdx.y;
    // all samples read ONE input state (the first pass of a batched circuit): the B workgroups of a tile group become
    // neighbours in dispatch order on the same XCD, so the tiles come from HBM once and from that XCD's L2 otherwise
    if (in_bstride == 0 && (gridDim.x & 7u) == 0) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.y;
        const unsigned group = lin / (8u * nb), r = lin % (8u * nb);
        sample = r >> 3;
        grp = group * 8u + (r & 7u);
    }
  // a wave walks `tpw` tiles: its next loads leave right behind its stores (no launch gap, no prologue in between).  The
  // tiles of one wave lie a whole grid apart, so that the tiles in flight at any time stay neighbours (write locality)
  // (tpw is a power of two and the grid ceil(tiles / (4 tpw)) workgroups: stride and count are re-derived per tile from the
  // kernel arguments instead of living in SGPRs across the assembly)
    if (const unsigned xv = ((unsigned)tpw_flags >> 18) & 31u) {
        // workgroups go to the eight XCDs round robin; here XCD j takes the j-th run of 2^(xv-1) consecutive tile groups out
        // of every eight runs (xv = 31: a contiguous eighth of the whole pass), so that neighbouring 512-byte pieces of
        // the written state leave through the same L2
        const unsigned q = blockIdx.x >> 3, j = blockIdx.x & 7u;
        if (xv == 31u) grp = j * (gridDim.x >> 3) + q;
        else grp = ((q >> (xv - 1u)) << (xv + 2u)) + (j << (xv - 1u)) + (q & ((1u << (xv - 1u)) - 1u));
    }
  uint32_t tile32 = grp * 4u + wave;
  for (;;) {      // (ends by the tile count: after `tpw` strides the number is past it)
    const uint64_t tile_id = tile32;
    // where the tile lies: bit j of the tile number goes to index bit read_blk_pos[j] / store_blk_pos[j] (the descriptor
    // is read as words through the constant address space: scalar loads, constant byte positions)
    typedef const __attribute__((address_space(4))) uint32_t* KWords;
    uint64_t karg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    // (laundered: what the tile needs from the descriptor is re-read per tile -- hoisted out of the tile loop it would stay
    // live across the assembly, which owns s40..s99: 50 SGPR spills and a 169th VGPR, i.e. two waves per SIMD instead of three)
    asm volatile("" : "+s"(karg));
    const KWords hw = (KWords)(karg + offsetof(WaveKernArgs, p));
    {
        const uint32_t a_tf0 = ((KWords)karg)[offsetof(WaveKernArgs, tpw) / 4];
        const uint32_t ntiles = 1u << (hw[offsetof(WaveKernPass, zext) / 4] & 63u), lper = 2u + (a_tf0 & 0xffu);     // log2(4 tpw)
        if (tile32 >= ntiles) break;
        tile32 += ((ntiles + (1u << lper) - 1u) >> lper) * 4u;        // (for the next round; `tile_id` holds this one's)
    }
    uint64_t tg = 0, tw = 0;
#pragma unroll
    for (int w = 0; w < DQ_FUSED_MAX_BLK / 4; ++w) {
        const uint32_t rw = hw[offsetof(WaveKernPass, read_blk_pos) / 4 + w], sw = hw[offsetof(WaveKernPass, store_blk_pos) / 4 + w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t bit = (tile_id >> (4 * w + k)) & 1ull;
            tg |= bit << ((rw >> (8 * k)) & 0x3fu);
            tw |= bit << ((sw >> (8 * k)) & 0x3fu);
        }
    }
    {   // a slice of a pass (dq_apply_fused_slice_*): the index bits it holds fixed
        const uint32_t fo = offsetof(WaveKernPass, fix_read) / 4;
        tg |= (uint64_t)hw[fo] | ((uint64_t)hw[fo + 1] << 32);
        tw |= (uint64_t)hw[fo + 2] | ((uint64_t)hw[fo + 3] << 32);
    }
    // (the kernel's own arguments too, through the laundered pointer: re-read per tile instead of held across the assembly)
    typedef const __attribute__((address_space(4))) uint64_t* KQuads;
    const KQuads kq = (KQuads)karg;
    const uint64_t a_in = kq[0], a_out = kq[1], a_mats = kq[2], a_mbs = kq[3], a_ibs = kq[4];
    const uint64_t a_ext = kq[offsetof(WaveKernArgs, ext_rec) / 8];
    const uint64_t rec_base = a_ext ? a_ext : karg + offsetof(WaveKernArgs, p) + offsetof(WaveKernPass, rec);
    const uint32_t a_n = ((KWords)karg)[offsetof(WaveKernArgs, n) / 4], a_tf = ((KWords)karg)[offsetof(WaveKernArgs, tpw) / 4];
    constexpr uint64_t ES = W::ELEM;
    const uint64_t inb = a_in + ((uint64_t)sample * a_ibs + tg) * ES;
    const uint64_t outb = a_out + (((uint64_t)sample << a_n) + tw) * ES;
    const uint64_t mb = a_mats + (uint64_t)((int64_t)sample * (int64_t)a_mbs) * ES;
    // bit 0 / 1: streaming loads / stores (see wave_launch); a pass whose samples share ONE input keeps it in the L2
    // bits 8..13 / 16..21: register slots / lane bits whose index bit is known to be |0> in the input (not loaded)
    const unsigned flags = ((a_ibs == 0 ? (a_tf >> 16) & ~1u : a_tf >> 16) & 3u) | (hw[offsetof(WaveKernPass, zext) / 4] & 0x003f3f00u);
    W::body(rec_base, hw[offsetof(WaveKernPass, nrec_bytes) / 4], mb,
            hw[offsetof(WaveKernPass, mat_base_bytes) / 4], tg,
            karg + offsetof(WaveKernArgs, p) + offsetof(WaveKernPass, load_off), inb, outb, wave * W::LDS_PER_WAVE, tid,
            flags);
  }
    if constexpr (GRAD) {
        __syncthreads();
        typedef const __attribute__((address_space(4))) uint32_t* KW;
        const KW rw = ext_rec ? (KW)(uint64_t)ext_rec
                              : (KW)((uint64_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WaveKernArgs, p) + offsetof(WaveKernPass, rec));
        const unsigned nrec = p.nrec_bytes / 32u;
        double* const grow = grads + (uint64_t)sample * (uint64_t)grad_bstride;
        for (unsigned i = tid; i < nrec * 8u; i += 256u) {
            const unsigned id = rw[8u * (i >> 3)];
            // (an expectation value fills component 0 only: one atomic per workgroup instead of eight)
            if ((id >= (unsigned)W::ID_GRAD && id < (unsigned)W::ID_EXPZ) || (id == (unsigned)W::ID_EXPZ && (i & 7u) == 0u)) {
                const typename W::acc_t v = *(__attribute__((address_space(3))) typename W::acc_t*)(uintptr_t)(4u * W::LDS_PER_WAVE + (unsigned)sizeof(typename W::acc_t) * i);
                atomicAdd(grow + (uint64_t)rw[8u * (i >> 3) + 6] * 8u + (i & 7u), (double)v);
            }
        }
    }
}

// ---- host: DqFusedPass (rounds) -> records ------------------------------------------------------------------------
namespace {

template <class W>
struct Xlate {
    const DqFusedPass* p;
    WaveKernPass* k;
    int phys[W::R];        // tile-local bit held by physical slot s (register-index bit s)
    int lanes[WAVE_LANES];   // tile-local bit on lane bit b
    int nrec = 0;
    WaveRec* recs = nullptr;   // where the records go: k->rec (WAVE_MAX_REC of them) or the caller's array
    int cap = WAVE_MAX_REC;

    bool push(const WaveRec& r) {
        if (nrec >= cap) return false;
        recs[nrec++] = r;
        return true;
    }
    int slot_of(int tile_bit) const {
        for (int s = 0; s < W::R; ++s)
            if (phys[s] == tile_bit) return s;
        return -1;
    }

    // k slots (mask) <-> the lane bits inc_lanes[0..k) (ascending); `want` = the lane order afterwards (tile bits per
    // lane bit) or nullptr: chosen here
    bool trip(unsigned mask, const int* inc_lanes, int kk, const int* want) {
        const int a = 5 - kk, S = 64 + (1 << a);
        int moving[W::R], out_bits[W::R], inc_bits[W::R], stay_lanes[WAVE_LANES], stay_bits[WAVE_LANES], fpos[WAVE_LANES];
        int nm = 0, ns = 0;
        for (int s = 0; s < W::R; ++s)
            if ((mask >> s) & 1u) {
                moving[nm] = s;
                out_bits[nm] = phys[s];
                ++nm;
            }
        for (int i = 0; i < kk; ++i) inc_bits[i] = lanes[inc_lanes[i]];
        for (int b = 0; b < WAVE_LANES; ++b) {
            bool inc = false;
            for (int i = 0; i < kk; ++i) inc = inc || inc_lanes[i] == b;
            if (!inc) {
                stay_lanes[ns] = b;
                stay_bits[ns] = lanes[b];
                ++ns;
            }
        }
        // slot (8-byte units) of element (x, y, z) = x * S + (y << a) + F(z): the lane bits that stay fill the positions
        // below the field [a, a + k) in their old order, the highest of them position 5 (so the 32 lanes of a read
        // group and the 16 of a write group differ in the low positions only: no bank conflicts where that is possible)
        if (kk == 0) {
            for (int t = 0; t < ns; ++t) fpos[t] = t;
        } else {
            for (int t = 0; t < ns; ++t) fpos[t] = t < a ? t : 5;
        }
        uint32_t cw[WAVE_LANES] = {0}, cr[WAVE_LANES] = {0}, tbc[WAVE_LANES] = {0};
        for (int i = 0; i < kk; ++i) cw[inc_lanes[i]] = (unsigned)W::ELEM << (a + i);
        for (int t = 0; t < ns; ++t) cw[stay_lanes[t]] = (unsigned)W::ELEM << fpos[t];
        int nl[WAVE_LANES];
        if (want) {
            for (int b = 0; b < WAVE_LANES; ++b) nl[b] = want[b];
        } else if (kk == 0) {
            for (int b = 0; b < WAVE_LANES; ++b) nl[b] = lanes[b];
        } else {
            for (int b = 0; b < WAVE_LANES; ++b) nl[b] = b < a ? stay_bits[b] : (b < 5 ? out_bits[b - a] : stay_bits[a]);
        }
        for (int b = 0; b < WAVE_LANES; ++b) {
            bool found = false;
            for (int i = 0; i < kk && !found; ++i)
                if (nl[b] == out_bits[i]) {
                    cr[b] = ((unsigned)W::ELEM * (unsigned)S) << i;
                    found = true;
                }
            for (int t = 0; t < ns && !found; ++t)
                if (nl[b] == stay_bits[t]) {
                    cr[b] = (unsigned)W::ELEM << fpos[t];
                    found = true;
                }
            if (!found) return false;      // `want` is not made of the bits that are on the lanes afterwards
            tbc[b] = 1u << nl[b];
        }
        WaveRec ra{}, rb{};
        ra.w[0] = kk == 0 ? (uint32_t)W::ID_TRIP0 : (uint32_t)W::trip_id(mask);
        ra.w[1] = tbc[0], ra.w[2] = tbc[1], ra.w[3] = tbc[2], ra.w[5] = tbc[3], ra.w[6] = tbc[4], ra.w[7] = tbc[5];
        for (int b = 0; b < WAVE_LANES; ++b) rb.w[b] = cw[b] | (cr[b] << 16);
        for (int i = 0; i < kk; ++i) phys[moving[i]] = inc_bits[i];
        for (int b = 0; b < WAVE_LANES; ++b) lanes[b] = nl[b];
        return push(ra) && push(rb);
    }

    // make `slotset` (bit mask over tile-local bits) the register slots; `want` as above (for the last trip)
    bool go(unsigned slotset, const int* want) {
        for (;;) {
            unsigned mask = 0;
            int inc[WAVE_LANES], ninc = 0, nout = 0;
            for (int s = 0; s < W::R; ++s)
                if (!((slotset >> phys[s]) & 1u)) {
                    mask |= 1u << s;
                    ++nout;
                }
            for (int b = 0; b < WAVE_LANES; ++b)
                if ((slotset >> lanes[b]) & 1u) inc[ninc++] = b;
            if (nout != ninc) return false;
            if (nout == 0) return true;
            int kk = nout;
            if (kk > W::MAXK) kk = nout > 2 * W::MAXK ? W::MAXK : (nout + 1) / 2;      // e.g. 5 -> 3 + 2, 6 -> 3 + 3
            if (kk < nout) {                                  // keep the first kk outgoing slots / incoming lanes
                unsigned m2 = 0;
                int c = 0;
                for (int s = 0; s < W::R && c < kk; ++s)
                    if ((mask >> s) & 1u) {
                        m2 |= 1u << s;
                        ++c;
                    }
                mask = m2;
            }
            if (!trip(mask, inc, kk, kk == nout ? want : nullptr)) return false;
        }
    }
};

}  // namespace

// `dead`: index bits (read side) known to be |0> in the input (dq_apply_fused_zext_*; 0 = none).  Outside the tile they
// drop out of the tile number -- the tiles in which one of them is 1 are all zero: neither read nor written --, inside
// the tile the loads leave the registers of their 1-halves zero.
// `fixm` / `fixv` (dq_apply_fused_slice_*): index bits outside the tile (read side) held at the value `fixv` gives them: they
// drop out of the tile number too, and every tile's read and write bases get them OR-ed in (fix_read / fix_write).
template <class W>
static int wave_translate(const DqFusedPass* p, int n, WaveKernPass* k, uint64_t dead = 0, WaveRec* ext = nullptr, int ext_cap = 0,
                          uint64_t fixm = 0, uint64_t fixv = 0) {
    memset(k, 0, sizeof(*k));
    const int L = p->L, h = p->h;
    auto rpos = [&](int tl) { return tl < L ? tl : (int)p->high_pos[tl - L]; };
    auto wpos = [&](int tl) { return tl < L ? (int)p->store_low_pos[tl] : (int)p->store_high_pos[tl - L]; };
    Xlate<W> x;
    x.p = p;
    x.k = k;
    x.recs = ext ? ext : k->rec;
    x.cap = ext ? ext_cap : WAVE_MAX_REC;
    unsigned slotmask = 0;
    for (int s = 0; s < W::R; ++s) {
        x.phys[s] = p->load_rb[s];
        slotmask |= 1u << p->load_rb[s];
    }
    for (int b = 0, q = 0; b < WAVE_LANES; ++b, ++q) {
        while ((slotmask >> q) & 1u) ++q;
        x.lanes[b] = q;
    }
    {   // read side: the tile number's bits fill the index bits outside the tile, in ascending order
        uint64_t tilemask = (1ull << L) - 1ull;
        for (int i = 0; i < h; ++i) tilemask |= 1ull << p->high_pos[i];
        int nb = 0;
        uint64_t fix_r = 0, fix_w = 0;
        for (int j = 0, q = 0; j < n - W::M; ++j, ++q) {
            while (q < 64 && ((tilemask >> q) & 1ull)) ++q;
            if ((dead >> q) & 1ull) continue;       // (known |0>: not a bit of the tile number)
            if ((fixm >> q) & 1ull) {               // (held fixed by this slice of the pass)
                if ((fixv >> q) & 1ull) {
                    fix_r |= 1ull << q;
                    fix_w |= 1ull << p->store_blk_pos[j];
                }
                continue;
            }
            k->read_blk_pos[nb] = (uint8_t)q;
            k->store_blk_pos[nb] = p->store_blk_pos[j];
            ++nb;
        }
        for (int j = nb; j < DQ_FUSED_MAX_BLK; ++j) k->read_blk_pos[j] = k->store_blk_pos[j] = 63;
        k->zext = (uint32_t)nb;
        k->fix_read[0] = (uint32_t)fix_r, k->fix_read[1] = (uint32_t)(fix_r >> 32);
        k->fix_write[0] = (uint32_t)fix_w, k->fix_write[1] = (uint32_t)(fix_w >> 32);
        // ... re-ordered by where they land on the WRITE side: tiles that run at the same time (neighbours in the tile
        // number) then write neighbouring runs, i.e. whole DRAM pages between them, while the read side does not care
        // (a tile reads one contiguous 32 KiB block wherever it lies).  DQ_WAVE_TILE_ORDER=read keeps the read order.
        static const bool by_store = [] { const char* e = getenv("DQ_WAVE_TILE_ORDER"); return !(e && e[0] == 'r'); }();
        if (by_store)
            for (int i = 1; i < nb; ++i)        // insertion sort of (read, store) pairs by store position
                for (int j = i; j > 0 && k->store_blk_pos[j] < k->store_blk_pos[j - 1]; --j) {
                    std::swap(k->store_blk_pos[j], k->store_blk_pos[j - 1]);
                    std::swap(k->read_blk_pos[j], k->read_blk_pos[j - 1]);
                }
    }
    for (int s = W::VB; s < W::R; ++s) {
        k->load_off[s - W::VB] = (uint64_t)W::ELEM << rpos(x.phys[s]);
        if ((dead >> rpos(x.phys[s])) & 1ull) k->zext |= 1u << (8 + s);
    }
    for (int b = 0; b < WAVE_LANES; ++b) {
        k->load_lane_shift[b] = (W::ELEM == 8 ? 3u : 4u) + (uint32_t)rpos(x.lanes[b]);
        k->tb_contrib[b] = 1u << x.lanes[b];
        if ((dead >> rpos(x.lanes[b])) & 1ull) k->zext |= 1u << (16 + b);
    }
    k->mat_base_bytes = p->mat_base * (uint32_t)W::ELEM;

    const char* why = "too many records for one pass (gates + layout changes)";
    for (int r = 0; r < p->nrounds; ++r) {
        const DqFusedRound& rd = p->rounds[r];
        unsigned want = 0;
        for (int s = 0; s < W::R; ++s) want |= 1u << rd.rb[s];
        if (!x.go(want, nullptr)) goto fail;
        for (int gi = rd.gate_begin; gi < rd.gate_end; ++gi) {
            const DqFusedGate& g = p->gates[gi];
            if (g.kind == DQ_FG_GEN2 && W::ID_GEN2 >= 0) {
                // dense gate on two slots: the handler of the slot pair a < b reads the matrix index as 2 * bit(b) + bit(a);
                // a first target on the lower slot makes the kernel swap the two index bits of the matrix (w6)
                const int q1 = x.slot_of(rd.rb[g.q]), q2 = x.slot_of(rd.rb[g.q2]);
                const int a = q1 < q2 ? q1 : q2, b = q1 < q2 ? q2 : q1;
                unsigned pc = 0;
                for (int s = 0; s < W::R; ++s)
                    if ((g.reg_cmask >> s) & 1u) pc |= 1u << x.slot_of(rd.rb[s]);
                WaveRec rec{};
                // (loc = DQ_MODE_REAL: a matrix promised real -- channel superoperators -- takes the bodies that spend one
                // operation per entry instead of two)
                // loc = DQ_MODE_XREAL: real AND non-zero only on the two 2x2 blocks (00, 11) / (01, 10) -- the superoperators of
                // the reference's channels: half the operations again (complex64; complex128 takes the real bodies)
                // loc = DQ_MODE_XCPLX: the same shape with complex entries (Rxx, Ryy, Rxy): complex64 bodies of their own
                const int body = g.loc == DQ_MODE_XCPLX && W::ID_GEN2XC >= 0 ? W::ID_GEN2XC
                                 : g.loc == DQ_MODE_XREAL && W::ID_GEN2X >= 0 ? W::ID_GEN2X
                                 : (g.loc == DQ_MODE_REAL || g.loc == DQ_MODE_XREAL) && W::ID_GEN2R >= 0 ? W::ID_GEN2R : W::ID_GEN2;
                rec.w[0] = (uint32_t)(body + (W::swap_id(a, b) - W::ID_SWAP));
                rec.w[1] = g.thr_cmask;
                rec.w[2] = (uint32_t)g.out_cmask, rec.w[3] = (uint32_t)(g.out_cmask >> 32);
                rec.w[4] = g.mat_advance;
                for (int j = 0, i = 0; j < W::NA; ++j) {         // group i = the i-th pattern with bits a and b clear
                    if (((j >> a) & 1) || ((j >> b) & 1)) continue;
                    if (((unsigned)j & pc) == pc) rec.w[5] |= 1u << i;
                    ++i;
                }
                rec.w[6] = q1 == a ? 1u : 0u;
                if (!x.push(rec)) goto fail;
                continue;
            }
            if (g.kind == DQ_FG_EXPZ) {
                // <Z..Z> from the registers: the sign of a register from the Z bits that are register slots right now
                unsigned pm = 0;
                for (int s = 0; s < W::R; ++s)
                    if ((g.reg_cmask >> s) & 1u) pm |= 1u << x.slot_of(rd.rb[s]);
                WaveRec rec{};
                rec.w[0] = (uint32_t)W::ID_EXPZ;
                rec.w[1] = g.thr_cmask;
                rec.w[2] = (uint32_t)g.out_cmask, rec.w[3] = (uint32_t)(g.out_cmask >> 32);
                for (int j = 0; j < W::NA; ++j)
                    if (__builtin_popcount((unsigned)j & pm) & 1) rec.w[j < 32 ? 5 : 7] |= 1u << (j & 31);
                rec.w[6] = g.reserved;
                if (!x.push(rec)) goto fail;
                continue;
            }
            if (g.kind == DQ_FG_GRAD && W::ID_GRAD >= 0) {
                // reduction of the reverse sweep: the handlers want psi / lambda on physical slot 0 (where the load layout
                // puts index bit 0 anyway); a register swap brings it back there if a trip moved it
                int ps = x.slot_of(rd.rb[g.q2]);
                if (ps != 0) {
                    WaveRec sw{};
                    sw.w[0] = (uint32_t)W::swap_id(0, ps);
                    if (!x.push(sw)) goto fail;
                    const int t_ = x.phys[0];
                    x.phys[0] = x.phys[ps];
                    x.phys[ps] = t_;
                }
                const int q = x.slot_of(rd.rb[g.q]);
                unsigned pc = 0;
                for (int s = 0; s < W::R; ++s)
                    if ((g.reg_cmask >> s) & 1u) pc |= 1u << x.slot_of(rd.rb[s]);
                WaveRec rec{};
                // (DqFusedGate::loc = which of the sums the gate's gradient needs: see include/dq_hip.h, DQ_FG_GRAD)
                const int variant = (int)g.loc < W::GRAD_VARIANTS ? (int)g.loc : 0;
                rec.w[0] = (uint32_t)(W::ID_GRAD + (W::R - 1) * variant + q - 1);
                rec.w[1] = g.thr_cmask;
                rec.w[2] = (uint32_t)g.out_cmask, rec.w[3] = (uint32_t)(g.out_cmask >> 32);
                for (int j = 0, i = 0; j < W::NA; ++j) {         // group i = the i-th pattern with bits q and 0 clear
                    if (((j >> q) & 1) || (j & 1)) continue;
                    if (((unsigned)j & pc) == pc) rec.w[5] |= 1u << i;
                    ++i;
                }
                rec.w[6] = g.reserved;
                if (!x.push(rec)) goto fail;
                continue;
            }
            if (g.kind != DQ_FG_GEN1 && g.kind != DQ_FG_X1 && g.kind != DQ_FG_DIAG1 && g.kind != DQ_FG_DIAG2) {
                set_error("dq_apply_fused: the pass kernel takes dense gates on one or two targets, X and diagonal gates (record %d has kind %d); "
                          "such a gate runs on its own (dq_apply_gate_*)", gi, (int)g.kind);
                return DQ_ERR_UNSUPPORTED;
            }
            unsigned pc = 0;
            int nc = 0, onec = 0;
            for (int s = 0; s < W::R; ++s)
                if ((g.reg_cmask >> s) & 1u) {
                    onec = x.slot_of(rd.rb[s]);
                    pc |= 1u << onec;
                    ++nc;
                }
            const bool ctl = g.thr_cmask != 0 || g.out_cmask != 0;
            WaveRec rec{};
            rec.w[1] = g.thr_cmask;
            rec.w[2] = (uint32_t)g.out_cmask, rec.w[3] = (uint32_t)(g.out_cmask >> 32);
            rec.w[4] = g.mat_advance;
            if (g.kind == DQ_FG_DIAG1 || g.kind == DQ_FG_DIAG2) {
                // a phase per amplitude: PH0 / PH1 by the bit of one register slot (or PH0 for all), each picked per lane
                // from the gate's diagonal by up to two selectors (targets on thread bits / outside the tile) --
                // csrc/dq_wave_asm.inc, diag_code; controls on register slots become a mask over the 64 registers
                auto regmask = [&](unsigned must_set, unsigned must_clear) {
                    uint64_t mk = 0;
                    for (unsigned j = 0; j < (unsigned)W::NA; ++j)
                        if ((j & must_set) == must_set && (j & must_clear) == 0) mk |= 1ull << j;
                    return mk;
                };
                auto selector = [&](int loc, int q_) -> uint32_t {
                    return loc == DQ_LOC_THR ? ((1u << 6) | (uint32_t)q_) : loc == DQ_LOC_OUT ? ((2u << 6) | (uint32_t)q_) : 0u;
                };
                auto put = [&](int base, int variant, bool masked, uint32_t selA, uint32_t selB, uint32_t idx0, uint32_t idx1,
                               uint64_t mk, uint32_t advance) {
                    WaveRec r2 = rec;
                    r2.w[0] = (uint32_t)(base + variant + (masked ? W::R + 1 : 0));
                    r2.w[4] = advance;
                    r2.w[5] = selA | (selB << 8) | (idx0 << 16) | (idx1 << 24);
                    r2.w[6] = (uint32_t)mk, r2.w[7] = (uint32_t)(mk >> 32);
                    return x.push(r2);
                };
                const bool masked = nc > 0;
                const uint64_t ctlmask = regmask(pc, 0);
                bool ok;
                if (g.kind == DQ_FG_DIAG1) {
                    if (g.loc == DQ_LOC_REG) ok = put(W::ID_DIAG1, 1 + x.slot_of(rd.rb[g.q]), masked, 0, 0, 0x00, 0x55, ctlmask, 4);
                    else ok = put(W::ID_DIAG1, 0, masked, 0, selector(g.loc, g.q), 0x44, 0, ctlmask, 4);
                } else {
                    const bool r1 = g.loc == DQ_LOC_REG, r2_ = g.loc2 == DQ_LOC_REG;
                    const int p1 = r1 ? x.slot_of(rd.rb[g.q]) : -1, p2 = r2_ ? x.slot_of(rd.rb[g.q2]) : -1;
                    if (r1 && r2_) {    // both targets on register slots: the halves of slot p1, by the bit of slot p2
                        ok = put(W::ID_DIAG2, 1 + p2, true, 0, 0, 0x00, 0x55, regmask(pc, 1u << p1), 16) &&
                             put(W::ID_DIAG2, 1 + p2, true, 0, 0, 0xAA, 0xFF, regmask(pc | (1u << p1), 0), 0);
                    } else if (r1) {
                        ok = put(W::ID_DIAG2, 1 + p1, masked, selector(g.loc2, g.q2), 0, 0x10, 0x32, ctlmask, 16);
                    } else if (r2_) {
                        ok = put(W::ID_DIAG2, 1 + p2, masked, selector(g.loc, g.q), 0, 0x20, 0x31, ctlmask, 16);
                    } else {
                        ok = put(W::ID_DIAG2, 0, masked, selector(g.loc, g.q), selector(g.loc2, g.q2), 0xE4, 0, ctlmask, 16);
                    }
                }
                if (!ok) goto fail;
                continue;
            }
            const int q = x.slot_of(rd.rb[g.q]);
            if (nc > 0) {       // pair i of slot q = the i-th register pattern with bit q clear
                uint32_t pm = 0;
                for (int j = 0, i = 0; j < W::NA; ++j) {
                    if ((j >> q) & 1) continue;
                    if (((unsigned)j & pc) == pc) pm |= 1u << i;
                    ++i;
                }
                rec.w[5] = pm;
            }
            if (g.kind == DQ_FG_X1) {
                if (nc == 0) rec.w[0] = (ctl ? W::ID_X_C : W::ID_X_U) + q;
                else if (nc == 1) rec.w[0] = W::ID_X_R1 + (W::R - 1) * q + (onec < q ? onec : onec - 1);
                else rec.w[0] = W::ID_X_R + q;
            } else {
                if (nc > 0) rec.w[0] = W::ID_GEN_R + q;
                else if (ctl) rec.w[0] = W::ID_GEN_C + q;
                else rec.w[0] = W::ID_GEN_U + W::R * g.loc + q;
            }
            if (!x.push(rec)) goto fail;
        }
    }
    {   // into the store layout: slot 0 = the tile bit written to index bit 0, lanes exactly as the host ordered them
        unsigned want = 0;
        int want_lanes[WAVE_LANES];
        for (int s = 0; s < W::R; ++s) want |= 1u << p->store_rb[s];
        for (int b = 0; b < WAVE_LANES; ++b) want_lanes[b] = p->store_tb[b];
        if (!x.go(want, want_lanes)) goto fail;
        bool same = true;
        for (int b = 0; b < WAVE_LANES; ++b) same = same && x.lanes[b] == want_lanes[b];
        if (!same && !x.trip(0, nullptr, 0, want_lanes)) goto fail;
        const int s0 = W::VB ? x.slot_of(p->store_rb[0]) : 0;       // (complex64: slot 0 = the tile bit written to index bit 0)
        if (s0 != 0) {
            WaveRec rec{};
            rec.w[0] = (uint32_t)W::swap_id(0, s0);
            if (!x.push(rec)) goto fail;
            const int t = x.phys[0];
            x.phys[0] = x.phys[s0];
            x.phys[s0] = t;
        }
    }
    for (int s = W::VB; s < W::R; ++s) k->store_off[s - W::VB] = (uint64_t)W::ELEM << wpos(x.phys[s]);
    for (int b = 0; b < WAVE_LANES; ++b) k->store_lane_shift[b] = (W::ELEM == 8 ? 3u : 4u) + (uint32_t)wpos(x.lanes[b]);
    k->nrec_bytes = 32u * (unsigned)x.nrec;
    return DQ_OK;
fail:
    set_error("dq_apply_fused (wave tile): %s", why);
    return DQ_ERR_UNSUPPORTED;
}

template <class W, bool GRAD = false>
static int wave_launch(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                       const DqFusedPass* pass, hipStream_t s, double* grads = nullptr, int64_t ngrads = 0, uint64_t dead = 0,
                       const void* ext_rec = nullptr, int64_t ext_bytes = 0, uint64_t fixm = 0, uint64_t fixv = 0) {
    WaveKernPass kp;
    int rc;
    if (ext_rec) {      // the records lie in device memory (dq_wave_records wrote them, the caller copied them there): translate
        // again for the header, and hold the caller to the size this pass has
        static thread_local WaveRec scratch[WAVE_EXT_REC];
        rc = wave_translate<W>(pass, n, &kp, dead, scratch, WAVE_EXT_REC);
        if (!rc && (int64_t)kp.nrec_bytes != ext_bytes) {
            set_error("dq_apply_fused_grad_ext: %lld bytes of records in device memory, this pass has %u", (long long)ext_bytes, kp.nrec_bytes);
            return DQ_ERR_ARG;
        }
    } else {
        rc = wave_translate<W>(pass, n, &kp, dead, nullptr, 0, fixm, fixv);
    }
    if (rc) return rc;
    const uint64_t tiles = 1ull << (kp.zext & 63u);
    int tpw = 1;
    if (GRAD) {     // tiles per wave: up to 2 while >= 2048 workgroups per sample batch are left.  More tiles per wave save atomics
        // on the caller's sums but leave a longer tail: training step n = 28 (32 sweep passes) 89.3 ms at a cap of 64 (rounds
        // 3-4), 87.8 at 8, 87.4 at 2, 86.5 at 1 (profiles/r05/exp_grad_tpw.txt).  DQ_WAVE_GRAD_TPW overrides the cap
        static const int gcap = [] { const char* e = getenv("DQ_WAVE_GRAD_TPW"); return e ? atoi(e) : 2; }();
        // a forward pass that only takes <Z..Z> from its registers (DQ_FG_EXPZ records, no DQ_FG_GRAD): its own cap
        static const int zcap = [] { const char* e = getenv("DQ_WAVE_EXPZ_TPW"); return e ? atoi(e) : 2; }();
        bool only_expz = !ext_rec;
        for (unsigned r = 0; only_expz && r < kp.nrec_bytes / 32u; ++r)
            only_expz = !(kp.rec[r].w[0] >= (uint32_t)W::ID_GRAD && kp.rec[r].w[0] < (uint32_t)W::ID_EXPZ);
        const int cap = only_expz ? zcap : gcap;
        while (tpw < cap && (tiles * (uint64_t)batch) / (8ull * (uint64_t)tpw) >= 2048) tpw *= 2;
    } else {
        static const int tpw_env = [] { const char* e = getenv("DQ_WAVE_TPW"); return e ? atoi(e) : 1; }();
        while (tpw < tpw_env && (tiles * (uint64_t)batch) / (8ull * (uint64_t)tpw) >= 2048) tpw *= 2;
    }
    dim3 grid((unsigned)((tiles + 4ull * tpw - 1) / (4ull * tpw)), (unsigned)batch);
    // (the accumulators of the reductions: eight per record, behind the four waves' staging buffers)
    const size_t acc_rec = ext_rec ? (size_t)kp.nrec_bytes / 32u : (size_t)WAVE_MAX_REC;
    size_t lds = 4 * W::LDS_PER_WAVE + (GRAD ? acc_rec * 8 * sizeof(typename W::acc_t) : 0);
    if (const char* kb = getenv("DQ_WAVE_LDS_KB")) {      // occupancy experiments: workgroups per CU = 160 KiB / this
        lds = (size_t)atoi(kb) << 10;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wave_pass_kernel<W, GRAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    // Streaming (non-temporal) loads and stores for states far beyond the 256 MiB Infinity Cache: nothing a pass writes
    // survives until the next pass reads it, and not allocating the lines is worth 10-15 % of the memory side
    // (profiles/r03/mb_wavetile_nt.txt).  Smaller states stay cacheable.  DQ_WAVE_NT=0..3 overrides (experiments).
    static const int nt_env = [] { const char* e = getenv("DQ_WAVE_NT"); return e ? atoi(e) & 3 : -1; }();
    const uint64_t state_bytes = ((uint64_t)batch << n) * (uint64_t)W::ELEM;
    const int nt = nt_env >= 0 ? nt_env : (state_bytes >= (1ull << 30) ? 3 : 0);
    static const int xcd_env = [] { const char* e = getenv("DQ_WAVE_XCD"); return e ? atoi(e) : 1; }();
    // XCD-aware tile numbers (default on; measured on the headline, two boxes: 240.3 -> 234.7 ms and 248.3 -> 244.7 ms;
    // runs of 2 .. 512 groups per XCD give the same within 0.2 %).  DQ_WAVE_XCD: 0 off, 1 contiguous eighths, C = 2, 4, ..:
    // runs of C tile groups per XCD
    int xcd = 0;
    // (several tiles per wave -- the reducing instantiation: the tile numbers of a wave lie a whole grid apart and the mapping
    // of the workgroup holds for each of them.  Round 6: the last pass of the headline, <Z0> + canonical restore, 15.35 ->
    // 13.98 ms with it, profiles/r06/last_pass_knobs.txt; DQ_WAVE_XCD_TPW=0 restores round 5's behaviour)
    static const int xcd_tpw = [] { const char* e = getenv("DQ_WAVE_XCD_TPW"); return e ? atoi(e) : 1; }();
    if (xcd_env && in_bstride != 0 && (tpw == 1 || xcd_tpw)) {
        if (xcd_env == 1 && (grid.x & 7u) == 0) xcd = 31;
        else if (xcd_env > 1 && (xcd_env & (xcd_env - 1)) == 0 && grid.x % (8u * (unsigned)xcd_env) == 0)
            xcd = 32 - __builtin_clz((unsigned)xcd_env);       // log2(C) + 1
    }
    using V = vec2<typename W::real>;
    hipLaunchKernelGGL((wave_pass_kernel<W, GRAD>), grid, dim3(256), lds, s, static_cast<const V*>(in), static_cast<V*>(out),
                       static_cast<const V*>(mats), mat_bstride, in_bstride, n, (31 - __builtin_clz((unsigned)tpw)) | (nt << 16) | (xcd << 18), kp, grads, ngrads * 8,
                       ext_rec);
    return check_launch("dq_apply_fused (wave tile)");
}

int wave_launch_c64(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                    const DqFusedPass* pass, hipStream_t s, uint64_t dead, uint64_t fixm, uint64_t fixv) {
    return wave_launch<WaveC64>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, nullptr, 0, dead, nullptr, 0, fixm, fixv);
}
int wave_launch_grad_c64(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                         const DqFusedPass* pass, hipStream_t s, double* grads, int64_t ngrads, const void* ext_rec, int64_t ext_bytes) {
    return wave_launch<WaveC64, true>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads, 0, ext_rec, ext_bytes);
}
int wave_launch_grad_c128(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                          const DqFusedPass* pass, hipStream_t s, double* grads, int64_t ngrads, const void* ext_rec, int64_t ext_bytes) {
    return wave_launch<WaveC128, true>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads, 0, ext_rec, ext_bytes);
}
int wave_launch_c128(const void* in, void* out, const void* mats, int64_t mat_bstride, int64_t in_bstride, int n, int64_t batch,
                     const DqFusedPass* pass, hipStream_t s, uint64_t dead, uint64_t fixm, uint64_t fixv) {
    return wave_launch<WaveC128>(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, nullptr, 0, dead, nullptr, 0, fixm, fixv);
}

}  // namespace dq

// The records of a pass as the kernel reads them (32 bytes each), for a caller that keeps them in DEVICE memory: a pass with
// more records than the kernel-argument segment holds (dq_apply_fused_grad_ext_*).  Returns their size in bytes (or a
// negative error code); writes min(size, max_bytes) bytes to `out` (may be null: just the size).  No GPU needed.
extern "C" int64_t dq_wave_records(const DqFusedPass* pass, int n, void* out, int64_t max_bytes) {
    if (!pass) {
        dq::set_error("dq_wave_records: null pointer");
        return DQ_ERR_ARG;
    }
    static thread_local dq::WaveRec scratch[dq::WAVE_EXT_REC];
    dq::WaveKernPass kp;
    int rc;
    if (pass->m == 12 && pass->slots == 6) rc = dq::wave_translate<dq::WaveC64>(pass, n, &kp, 0, scratch, dq::WAVE_EXT_REC);
    else if (pass->m == 11 && pass->slots == 5) rc = dq::wave_translate<dq::WaveC128>(pass, n, &kp, 0, scratch, dq::WAVE_EXT_REC);
    else {
        dq::set_error("dq_wave_records: not a wave-tile pass (m = %d, %d slots)", pass->m, pass->slots);
        return DQ_ERR_ARG;
    }
    if (rc) return rc;
    const int64_t bytes = kp.nrec_bytes;
    if (out && max_bytes > 0) memcpy(out, scratch, (size_t)(bytes < max_bytes ? bytes : max_bytes));
    return bytes;
}

// Test hook (no GPU needed): the kernel-side descriptor the library would hand to the wave-tile kernel for `pass` --
// slot offsets, lane shifts, tile-number positions, records (struct WaveKernPass above) -- as raw bytes.  The precision
// follows the geometry: m = 12 / 6 slots = complex64, m = 11 / 5 slots = complex128.
extern "C" int dq_wave_descriptor(const DqFusedPass* pass, int n, uint64_t known_zero, void* out, int max_bytes) {
    if (!pass) {
        dq::set_error("dq_wave_descriptor: null pointer");
        return DQ_ERR_ARG;
    }
    // (header + records back to back, as in the kernel-argument segment -- also for a pass whose records would travel
    // through device memory: up to WAVE_EXT_REC of them)
    static thread_local dq::WaveRec scratch[dq::WAVE_EXT_REC];
    dq::WaveKernPass kp;
    int rc;
    if (pass->m == 12 && pass->slots == 6) rc = dq::wave_translate<dq::WaveC64>(pass, n, &kp, known_zero, scratch, dq::WAVE_EXT_REC);
    else if (pass->m == 11 && pass->slots == 5) rc = dq::wave_translate<dq::WaveC128>(pass, n, &kp, known_zero, scratch, dq::WAVE_EXT_REC);
    else {
        dq::set_error("dq_wave_descriptor: not a wave-tile pass (m = %d, %d slots)", pass->m, pass->slots);
        return DQ_ERR_ARG;
    }
    if (rc) return rc;
    const int head = (int)offsetof(dq::WaveKernPass, rec);
    const int bytes = head + (int)kp.nrec_bytes;
    if (out && max_bytes > 0) {
        memcpy(out, &kp, head < max_bytes ? head : max_bytes);
        if (max_bytes > head) memcpy((char*)out + head, scratch, (size_t)((bytes < max_bytes ? bytes : max_bytes) - head));
    }
    return bytes;
}
