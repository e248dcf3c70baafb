// Fused pass, host side: the C entry points dq_apply_fused_* / dq_apply_fused_grad_* / dq_apply_fused_bcast_* check a
// pass descriptor (DqFusedPass, include/dq_hip.h -- the host scheduler deepquantum_amd/fusion.py builds it) and hand it to
// the wave-tile kernel (csrc/dq_wave.hip), the ONE pass kernel of the library since ABI 21: one wavefront owns a tile,
// one HBM read + one HBM write of the state applies every gate of the pass.  Replaces a run of consecutive Gate.forward
// calls of the reference (circuit.py:261 -> operation.py:274-289 -> qmath.py:485-506 / operation.py:203-219), each of
// which costs the reference >= 2 full read + write passes.
//
// (Rounds 1-3 also had workgroup-tile kernels here -- 512 threads sharing a 13-bit tile through LDS, two barriers per
// layout change; 0.46-0.51 of the HBM peak against the wave tile's 0.67 -- kept for complex128 two-target dense gates
// until the wave-tile kernel learned them.  DESIGN.md 4.1 keeps their measurements.)
#include "dq_common.hpp"
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

namespace dq {

struct FusedVariant {
    int m, slots, logt;
};
// the wave tile of each precision (csrc/dq_wave.hip): 64 lanes x 64 complex64 / 32 complex128 amplitudes
static const FusedVariant kVariantsC64[] = {{12, 6, 6}};
static const FusedVariant kVariantsC128[] = {{11, 5, 6}};
static const int kNumVariantsC64 = 1, kNumVariantsC128 = 1;

// ngrads: rows of the caller's accumulator (dq_apply_fused_grad_*), -1 = a plain pass (DQ_FG_GRAD records refused)
template <typename T>
static int validate_pass(const DqFusedPass* p, int n, int slots, int logt, int64_t ngrads = -1) {
    const int m = slots + logt;
    if (p->m != m || p->L + p->h != m || p->h > DQ_FUSED_MAX_HIGH || p->L < 1) {
        set_error("dq_apply_fused: inconsistent geometry (m=%d L=%d h=%d)", p->m, p->L, p->h);
        return DQ_ERR_ARG;
    }
    if (n < m) {
        set_error("dq_apply_fused: n=%d smaller than tile m=%d", n, m);
        return DQ_ERR_ARG;
    }
    if (p->nrounds == 0 || p->nrounds > DQ_FUSED_MAX_ROUNDS) {
        set_error("dq_apply_fused: a pass has 1..%d rounds", DQ_FUSED_MAX_ROUNDS);
        return DQ_ERR_ARG;
    }
    uint64_t seen = 0;
    for (int i = 0; i < p->h; ++i) {
        const int hp = p->high_pos[i];
        if (hp < p->L || hp >= n || ((seen >> hp) & 1ull)) {
            set_error("dq_apply_fused: bad high bit %d", hp);
            return DQ_ERR_ARG;
        }
        seen |= 1ull << hp;
        if (i > 0 && p->high_sorted[i] <= p->high_sorted[i - 1]) {
            set_error("dq_apply_fused: high_sorted not ascending");
            return DQ_ERR_ARG;
        }
    }
    uint64_t seen2 = 0;
    for (int i = 0; i < p->h; ++i) seen2 |= 1ull << p->high_sorted[i];
    if (seen != seen2) {
        set_error("dq_apply_fused: high_sorted is not a permutation of high_pos");
        return DQ_ERR_ARG;
    }
    {   // load layout: tile bit 0 on slot 0 for complex64 (16 bytes per lane), otherwise gathered bits, ascending
        constexpr int vb = sizeof(T) == 4 ? 1 : 0;
        for (int s = 0; s < slots; ++s) {
            const bool ok = (s < vb) ? (p->load_rb[s] == s) : (p->load_rb[s] >= p->L && p->load_rb[s] < m);
            if (!ok || (s > 0 && p->load_rb[s] <= p->load_rb[s - 1])) {
                set_error("dq_apply_fused: load layout invalid at slot %d", s);
                return DQ_ERR_ARG;
            }
        }
    }
    if (p->L > DQ_FUSED_MAX_LOW) {
        set_error("dq_apply_fused: L = %d contiguous low bits, at most %d", p->L, DQ_FUSED_MAX_LOW);
        return DQ_ERR_UNSUPPORTED;
    }
    {   // write positions: a permutation of [0, n)
        const int nblk = n - m;
        if (nblk > DQ_FUSED_MAX_BLK) {
            set_error("dq_apply_fused: n - m = %d block bits, at most %d", nblk, DQ_FUSED_MAX_BLK);
            return DQ_ERR_UNSUPPORTED;
        }
        uint64_t wseen = 0;
        for (int i = 0; i < m + nblk; ++i) {
            const int pos = i < p->L ? p->store_low_pos[i]
                                     : (i < m ? p->store_high_pos[i - p->L] : p->store_blk_pos[i - m]);
            if (pos >= n || ((wseen >> pos) & 1ull)) {
                set_error("dq_apply_fused: write positions are not a permutation of [0, n) (entry %d = %d)", i, pos);
                return DQ_ERR_ARG;
            }
            wseen |= 1ull << pos;
        }
    }
    {   // store layout: explicit slots and thread bits covering the tile; complex64 stores two adjacent amplitudes per
        // lane, so the tile bit of slot 0 must be written to global bit 0
        unsigned used = 0;
        for (int s = 0; s < slots; ++s) {
            if (p->store_rb[s] >= m || ((used >> p->store_rb[s]) & 1u)) {
                set_error("dq_apply_fused: store layout invalid at slot %d", s);
                return DQ_ERR_ARG;
            }
            used |= 1u << p->store_rb[s];
        }
        for (int i = 0; i < logt; ++i) {
            if (p->store_tb[i] >= m || ((used >> p->store_tb[i]) & 1u)) {
                set_error("dq_apply_fused: store layout invalid at thread bit %d", i);
                return DQ_ERR_ARG;
            }
            used |= 1u << p->store_tb[i];
        }
        if (sizeof(T) == 4) {
            const int tb0 = p->store_rb[0];
            const int pos = tb0 < p->L ? p->store_low_pos[tb0] : p->store_high_pos[tb0 - p->L];
            if (pos != 0) {
                set_error("dq_apply_fused: the tile bit of store slot 0 (%d) is written to bit %d, not bit 0", tb0, pos);
                return DQ_ERR_ARG;
            }
        }
    }
    int next_gate = 0;
    uint32_t next_mat = p->mat_base;
    for (int r = 0; r < p->nrounds; ++r) {
        const DqFusedRound& rd = p->rounds[r];
        unsigned used = 0;
        for (int s = 0; s < slots; ++s) {
            if (rd.rb[s] >= m || ((used >> rd.rb[s]) & 1u)) {
                set_error("dq_apply_fused: round %d slot list invalid", r);
                return DQ_ERR_ARG;
            }
            used |= 1u << rd.rb[s];
        }
        for (int i = 0; i < logt; ++i) {
            if (rd.tb[i] >= m || ((used >> rd.tb[i]) & 1u)) {
                set_error("dq_apply_fused: round %d thread-bit list invalid", r);
                return DQ_ERR_ARG;
            }
            used |= 1u << rd.tb[i];
        }
        {   // the kernel trusts the layout flags: recompute them from the layouts
            const uint8_t* prb = r == 0 ? p->load_rb : p->rounds[r - 1].rb;
            // thread bits of an I/O layout: the tile bits that are not slots, ascending
            auto io_tb = [&](const uint8_t* iorb, uint8_t* tbo) {
                unsigned slotmask = 0;
                for (int s = 0; s < slots; ++s) slotmask |= 1u << iorb[s];
                for (int i = 0, q = 0; i < logt; ++i, ++q) {
                    while ((slotmask >> q) & 1u) ++q;
                    tbo[i] = (uint8_t)q;
                }
            };
            uint8_t ptb[DQ_FUSED_MAX_TBITS], stb[DQ_FUSED_MAX_TBITS];
            if (r == 0) io_tb(p->load_rb, ptb);
            else for (int i = 0; i < logt; ++i) ptb[i] = p->rounds[r - 1].tb[i];
            bool differs = false;
            for (int s = 0; s < slots; ++s) differs = differs || prb[s] != rd.rb[s];
            for (int i = 0; i < logt; ++i) differs = differs || ptb[i] != rd.tb[i];
            bool after = false;
            if (r == p->nrounds - 1) {
                for (int i = 0; i < logt; ++i) stb[i] = p->store_tb[i];
                for (int s = 0; s < slots; ++s) after = after || p->store_rb[s] != rd.rb[s];
                for (int i = 0; i < logt; ++i) {
                    after = after || rd.tb[i] != stb[i];
                }
            }
            const unsigned want = (differs ? DQ_ROUND_TRANSPOSE : 0u) | (after ? DQ_ROUND_TRANSPOSE_AFTER : 0u);
            if (rd.flags != want) {
                set_error("dq_apply_fused: round %d has layout flags %u, expected %u", r, rd.flags, want);
                return DQ_ERR_ARG;
            }
        }
        const int gate_begin = rd.gate_begin;
        if (gate_begin > rd.gate_end || rd.gate_end > DQ_FUSED_MAX_GATES) {
            set_error("dq_apply_fused: round %d gate range invalid", r);
            return DQ_ERR_ARG;
        }
        if (gate_begin != next_gate) {
            set_error("dq_apply_fused: round %d does not continue the gate list", r);
            return DQ_ERR_ARG;
        }
        next_gate = rd.gate_end;
        for (int gi = gate_begin; gi < rd.gate_end; ++gi) {
            const DqFusedGate& g = p->gates[gi];
            const bool slot_kind = g.kind == DQ_FG_GEN1 || g.kind == DQ_FG_X1 || g.kind == DQ_FG_GEN2;
            if (g.kind == DQ_FG_RESERVED5 || g.fast != DQ_FAST_NONE) {
                set_error("dq_apply_fused: record %d uses a handler id or the exchange record of the workgroup-tile kernels "
                          "(removed with ABI 21)", gi);
                return DQ_ERR_ARG;
            }
            if (g.kind == DQ_FG_GRAD) {
                if (ngrads < 0 || g.q >= slots || g.q2 >= slots || g.q == g.q2 || (g.reg_cmask >> slots) || g.loc > 4 ||
                    ((g.reg_cmask >> g.q) & 1u) || ((g.reg_cmask >> g.q2) & 1u) || g.mat != next_mat || g.mat_advance != 0 ||
                    (int64_t)g.reserved >= ngrads) {
                    set_error("dq_apply_fused: reduction record %d malformed, or not a dq_apply_fused_grad_* call", gi);
                    return DQ_ERR_ARG;
                }
                continue;
            }
            if (g.kind == DQ_FG_EXPZ) {
                if (ngrads < 0 || (g.reg_cmask >> slots) || g.mat != next_mat || g.mat_advance != 0 || (int64_t)g.reserved >= ngrads) {
                    set_error("dq_apply_fused: expectation record %d malformed, or not a dq_apply_fused_grad_* call", gi);
                    return DQ_ERR_ARG;
                }
                continue;
            }
            if (g.kind > DQ_FG_DIAG2 || (slot_kind && g.q >= slots) || ((g.kind == DQ_FG_GEN1 && g.loc > 3) || (g.kind == DQ_FG_GEN2 && g.loc > 1 && g.loc != DQ_MODE_XREAL && g.loc != DQ_MODE_XCPLX)) ||
                (g.kind == DQ_FG_GEN2 && (g.q2 >= slots || g.q2 == g.q)) || (g.reg_cmask >> slots)) {
                set_error("dq_apply_fused: gate %d malformed", gi);
                return DQ_ERR_ARG;
            }
            // the kernel walks the matrices with a running pointer: they must lie back to back in gate order
            static const uint32_t kSize[5] = {4, 0, 4, 16, 16};
            if (g.mat != next_mat || g.mat_advance != kSize[g.kind]) {
                set_error("dq_apply_fused: gate %d breaks the sequential matrix layout (mat=%u, expected %u)", gi,
                          g.mat, next_mat);
                return DQ_ERR_ARG;
            }
            next_mat += g.mat_advance;
        }
    }
    return DQ_OK;
}

template <typename T>
static int fused_impl(const void* in, void* out, const void* mats, int64_t mat_bstride, int n, int64_t batch,
                      const DqFusedPass* pass, dq_stream_t stream, bool broadcast_in = false, double* grads = nullptr,
                      int64_t ngrads = -1, uint64_t known_zero = 0, const void* ext_rec = nullptr, int64_t ext_bytes = 0,
                      uint64_t slice_mask = 0, uint64_t slice_value = 0) {
    const int64_t in_bstride = broadcast_in ? 0 : (int64_t)1 << n;
    if (broadcast_in && in == out) {
        set_error("dq_apply_fused_bcast: the shared input state cannot be the output buffer");
        return DQ_ERR_ARG;
    }
    if (!in || !out || !mats || !pass) {
        set_error("dq_apply_fused: null pointer");
        return DQ_ERR_ARG;
    }
    if (batch < 1 || batch > 65535) {
        set_error("dq_apply_fused: batch %lld out of range [1, 65535]", (long long)batch);
        return DQ_ERR_ARG;
    }
    constexpr bool is128 = sizeof(T) == 8;
    const FusedVariant* vars = is128 ? kVariantsC128 : kVariantsC64;
    int vi = -1;
    for (int i = 0; i < (is128 ? kNumVariantsC128 : kNumVariantsC64); ++i)
        if (vars[i].m == pass->m && vars[i].slots == pass->slots) vi = i;
    if (vi < 0) {
        set_error("dq_apply_fused: no kernel with m=%d and %d register slots (wave tile: m = %d, %d slots)", pass->m, pass->slots,
                  vars[0].m, vars[0].slots);
        return DQ_ERR_UNSUPPORTED;
    }
    const FusedVariant v = vars[vi];
    int rc = validate_pass<T>(pass, n, v.slots, v.logt, ngrads);
    if (rc) return rc;
    if (in == out) {   // in place: every amplitude must be written where it was read
        bool same = true;
        for (int i = 0; i < pass->L; ++i) same = same && pass->store_low_pos[i] == i;
        for (int i = 0; i < pass->h; ++i) same = same && pass->store_high_pos[i] == pass->high_pos[i];
        uint64_t tilemask = 0;
        for (int i = 0; i < pass->h; ++i) tilemask |= 1ull << pass->high_pos[i];
        for (int j = 0, q = pass->L; j < n - v.m; ++j, ++q) {
            while ((tilemask >> q) & 1ull) ++q;
            same = same && pass->store_blk_pos[j] == q;
        }
        if (!same) {
            set_error("dq_apply_fused: a pass that writes to other index bits than it reads needs in != out");
            return DQ_ERR_ARG;
        }
    }
    if (n - v.m > 31) {
        set_error("dq_apply_fused: grid too large (n=%d)", n);
        return DQ_ERR_UNSUPPORTED;
    }
    if (known_zero) {
        // index bits known to be |0>: below 2^n, and none of the contiguous low bits (a lane loads them in one piece)
        if ((n < 64 && (known_zero >> n)) || (known_zero & ((1ull << pass->L) - 1ull)) || ngrads >= 0) {
            set_error("dq_apply_fused_zext: known_zero = 0x%llx names an index bit >= n = %d or one of the %d contiguous low bits "
                      "(or the pass is a reverse-sweep pass)", (unsigned long long)known_zero, n, (int)pass->L);
            return DQ_ERR_ARG;
        }
    }
    if (slice_mask) {
        // a slice of the pass: index bits (read side) outside the tile, not known zero, held at slice_value
        uint64_t tilemask = (1ull << pass->L) - 1ull;
        for (int i = 0; i < pass->h; ++i) tilemask |= 1ull << pass->high_pos[i];
        if ((n < 64 && (slice_mask >> n)) || (slice_mask & tilemask) || (slice_mask & known_zero) || (slice_value & ~slice_mask) ||
            ngrads >= 0) {
            set_error("dq_apply_fused_slice: slice_mask = 0x%llx must name index bits < n = %d outside the tile of the pass and outside "
                      "known_zero, slice_value = 0x%llx a subset of it (and the pass no reverse-sweep pass)",
                      (unsigned long long)slice_mask, n, (unsigned long long)slice_value);
            return DQ_ERR_ARG;
        }
    }
    hipStream_t s = as_stream(stream);
    // one wavefront per tile: csrc/dq_wave.hip
    if (ngrads >= 0) {
        if constexpr (is128) return wave_launch_grad_c128(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads, ext_rec, ext_bytes);
        else return wave_launch_grad_c64(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, grads, ngrads, ext_rec, ext_bytes);
    }
    if constexpr (is128) return wave_launch_c128(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, known_zero, slice_mask, slice_value);
    else return wave_launch_c64(in, out, mats, mat_bstride, in_bstride, n, batch, pass, s, known_zero, slice_mask, slice_value);
}

}  // namespace dq

extern "C" int dq_fused_geometry(int is_c128, int variant, int* m, int* slots, int* threads) {
    if (variant < 0 || variant >= (is_c128 ? dq::kNumVariantsC128 : dq::kNumVariantsC64)) {
        dq::set_error("dq_fused_geometry: variant %d out of range", variant);
        return DQ_ERR_ARG;
    }
    const dq::FusedVariant v = is_c128 ? dq::kVariantsC128[variant] : dq::kVariantsC64[variant];
    if (m) *m = v.m;
    if (slots) *slots = v.slots;
    if (threads) *threads = 1 << v.logt;
    return DQ_OK;
}

extern "C" int dq_apply_fused_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                  int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream);
}
extern "C" int dq_apply_fused_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                   int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream);
}

extern "C" int dq_apply_fused_grad_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                       int64_t batch, const DqFusedPass* pass, double* grads, int64_t ngrads,
                                       dq_stream_t stream) {
    if (!grads || ngrads < 1) {
        dq::set_error("dq_apply_fused_grad_c64: no accumulator (grads = %p, ngrads = %lld)", (void*)grads, (long long)ngrads);
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, false, grads, ngrads);
}

extern "C" int dq_apply_fused_grad_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                        int64_t batch, const DqFusedPass* pass, double* grads, int64_t ngrads,
                                        dq_stream_t stream) {
    if (!grads || ngrads < 1) {
        dq::set_error("dq_apply_fused_grad_c128: no accumulator (grads = %p, ngrads = %lld)", (void*)grads, (long long)ngrads);
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, false, grads, ngrads);
}

// A reverse-sweep pass whose records (dq_wave_records) the caller keeps in DEVICE memory: more of them than the
// kernel-argument segment holds.  `records` must stay valid until the pass has run (a HIP graph: until its last replay).
extern "C" int dq_apply_fused_grad_ext_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                           int64_t batch, const DqFusedPass* pass, const void* records, int64_t records_bytes,
                                           double* grads, int64_t ngrads, dq_stream_t stream) {
    if (!grads || ngrads < 1 || !records || records_bytes < 32 || (records_bytes & 31)) {
        dq::set_error("dq_apply_fused_grad_ext_c64: no accumulator or no records (grads = %p, ngrads = %lld, records = %p, %lld bytes)",
                      (void*)grads, (long long)ngrads, records, (long long)records_bytes);
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, false, grads, ngrads, 0, records, records_bytes);
}

extern "C" int dq_apply_fused_grad_ext_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                            int64_t batch, const DqFusedPass* pass, const void* records, int64_t records_bytes,
                                            double* grads, int64_t ngrads, dq_stream_t stream) {
    if (!grads || ngrads < 1 || !records || records_bytes < 32 || (records_bytes & 31)) {
        dq::set_error("dq_apply_fused_grad_ext_c128: no accumulator or no records (grads = %p, ngrads = %lld, records = %p, %lld bytes)",
                      (void*)grads, (long long)ngrads, records, (long long)records_bytes);
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, false, grads, ngrads, 0, records, records_bytes);
}

// Same pass, but `in` is ONE state (2^n amplitudes) shared by all `batch` outputs: the first pass of a batched
// circuit reads the circuit's initial state directly instead of `batch` materialised copies of it.
extern "C" int dq_apply_fused_bcast_c64(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                        int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, true);
}
extern "C" int dq_apply_fused_bcast_c128(const void* in, void* out, const void* mats, int64_t mat_batch_stride, int n,
                                         int64_t batch, const DqFusedPass* pass, dq_stream_t stream) {
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, true);
}

// The same pass on an input in which the index bits of `known_zero` are known to be |0> -- the circuit's own initial
// state |0..0> and the passes right behind it: the input is not read where one of these bits is 1 (it need not even be
// initialised there), tiles in which one of them is 1 are skipped altogether, and the output is left untouched where a
// known-zero bit OUTSIDE the tile is 1 (at its write position: still known zero for the next pass).
extern "C" int dq_apply_fused_zext_c64(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                                       int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, dq_stream_t stream) {
    if (in_batch_stride != 0 && (n < 0 || n > 62 || in_batch_stride != (int64_t)1 << n)) {
        dq::set_error("dq_apply_fused_zext_c64: in_batch_stride is 0 (one shared input state) or 2^n");
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, in_batch_stride == 0, nullptr, -1, known_zero);
}
extern "C" int dq_apply_fused_zext_c128(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                                        int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, dq_stream_t stream) {
    if (in_batch_stride != 0 && (n < 0 || n > 62 || in_batch_stride != (int64_t)1 << n)) {
        dq::set_error("dq_apply_fused_zext_c128: in_batch_stride is 0 (one shared input state) or 2^n");
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, in_batch_stride == 0, nullptr, -1, known_zero);
}

// ABI 25.  ONE SLICE of a pass: only the tiles whose index bits `slice_mask` (read side; outside the tile, not in
// known_zero) equal `slice_value` run -- 2^popcount(slice_mask) such launches are the whole pass.  The sharded state
// launches the last pass in front of an exchange and the first pass behind it slice by slice, so that the wire can start
// after the first slice and the next stretch with the first slice that has arrived.
extern "C" int dq_apply_fused_slice_c64(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                                        int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, uint64_t slice_mask,
                                        uint64_t slice_value, dq_stream_t stream) {
    if (in_batch_stride != 0 && (n < 0 || n > 62 || in_batch_stride != (int64_t)1 << n)) {
        dq::set_error("dq_apply_fused_slice_c64: in_batch_stride is 0 (one shared input state) or 2^n");
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<float>(in, out, mats, mat_batch_stride, n, batch, pass, stream, in_batch_stride == 0, nullptr, -1, known_zero,
                                 nullptr, 0, slice_mask, slice_value);
}
extern "C" int dq_apply_fused_slice_c128(const void* in, int64_t in_batch_stride, void* out, const void* mats, int64_t mat_batch_stride,
                                         int n, int64_t batch, const DqFusedPass* pass, uint64_t known_zero, uint64_t slice_mask,
                                         uint64_t slice_value, dq_stream_t stream) {
    if (in_batch_stride != 0 && (n < 0 || n > 62 || in_batch_stride != (int64_t)1 << n)) {
        dq::set_error("dq_apply_fused_slice_c128: in_batch_stride is 0 (one shared input state) or 2^n");
        return DQ_ERR_ARG;
    }
    return dq::fused_impl<double>(in, out, mats, mat_batch_stride, n, batch, pass, stream, in_batch_stride == 0, nullptr, -1, known_zero,
                                  nullptr, 0, slice_mask, slice_value);
}

// ---- the deferred-Rx form of a pass's matrix buffer (DQ_MODE_RX, include/dq_hip.h) ----------------------------------
// One thread per (sample, gate): the block { a, i b, -, - } of a matrix a I + i b X becomes { f, i t, -, flag }.  At
// launch-bound sizes the same rewrite in tensor operations was 25 launches per pass sequence (50 of a gradient's 163).
namespace dq {
__global__ void defer_rx_kernel(float2* __restrict__ mats, int64_t bstride, const int64_t* __restrict__ index, int64_t count) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float2* blk = mats + (int64_t)blockIdx.y * bstride + index[k];
    const float a = blk[0].x, b = blk[1].y;
    const bool form1 = fabsf(a) < fabsf(b);
    // (IEEE divisions: the same quotients, bit for bit, as the element-wise tensor formulation this replaces)
    const float t = form1 ? __fdiv_rn(-a, b) : __fdiv_rn(b, a);
    blk[0] = form1 ? make_float2(0.0f, b) : make_float2(a, 0.0f);
    blk[1] = make_float2(0.0f, t);
    blk[3] = make_float2(form1 ? 1.0f : 0.0f, 0.0f);
}
}  // namespace dq

extern "C" int dq_defer_rx_c64(void* mats, int64_t mat_batch_stride, const int64_t* index, int64_t count, int64_t batch,
                               dq_stream_t stream) {
    if (count == 0) return DQ_OK;
    if (!mats || !index || count < 0 || batch < 1 || batch > 65535 || mat_batch_stride < 0) {
        dq::set_error("dq_defer_rx_c64: bad argument");
        return DQ_ERR_ARG;
    }
    hipLaunchKernelGGL(dq::defer_rx_kernel, dim3((unsigned)((count + 127) / 128), (unsigned)batch), dim3(128), 0,
                       dq::as_stream(stream), static_cast<float2*>(mats), mat_batch_stride, index, count);
    return dq::check_launch("dq_defer_rx_c64");
}
