// Host-side core of the pass planner (deepquantum_amd/fusion.py): the dry run of a pass over the commutation DAG of a
// gate list.  The planner asks "how many gates would a pass that owns this set of qubits retire from this front?"
// a few hundred thousand times per circuit (beam search over tiles, one candidate qubit at a time); in Python that was
// 10-12 s for the headline circuit, here it is a tight loop over flat arrays.  No device code; the reference has no
// counterpart (it applies gates one by one, circuit.py:261).
#include "dq_common.hpp"
#include <vector>
#include <string.h>

namespace {

struct Dag {
    int n;
    std::vector<int> succ_off, succ;
    std::vector<uint64_t> targets;      // qubits the gate needs inside the tile (0 for diagonal gates: they run anywhere)
    std::vector<uint8_t> fusable;
    std::vector<int> cur, touched, stack;      // scratch
};

// Exactly fusion._closure: depth-first from `ready` (last first), a gate retires while fewer than `cap` have, it is
// fusable and its targets lie in `tile`; successors whose in-degree reaches 0 are pushed in list order.
int closure(Dag& d, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, int* stuck, int* nstuck,
            int* changed_idx, int* changed_val, int* nchanged) {
    d.stack.assign(ready, ready + nready);
    d.touched.clear();
    int count = 0, ns = 0;
    while (!d.stack.empty()) {
        const int i = d.stack.back();
        d.stack.pop_back();
        if (count >= cap || !d.fusable[i] || (d.targets[i] & ~tile)) {
            if (stuck) stuck[ns] = i;
            ++ns;
            continue;
        }
        ++count;
        for (int e = d.succ_off[i]; e < d.succ_off[i + 1]; ++e) {
            const int s = d.succ[e];
            if (d.cur[s] < 0) {                // first touch: take the caller's value
                d.cur[s] = indeg[s];
                d.touched.push_back(s);
            }
            if (--d.cur[s] == 0) d.stack.push_back(s);
        }
    }
    if (nstuck) *nstuck = ns;
    int nc = 0;
    for (int s : d.touched) {
        if (changed_idx) {
            changed_idx[nc] = s;
            changed_val[nc] = d.cur[s];
        }
        ++nc;
        d.cur[s] = -1;
    }
    if (nchanged) *nchanged = nc;
    return count;
}

}  // namespace

extern "C" void* dq_dag_create(int n_ops, const int* succ_off, const int* succ, const uint64_t* target_mask,
                               const uint8_t* fusable) {
    if (n_ops < 0 || !succ_off || (!succ && succ_off[n_ops] > 0) || !target_mask || !fusable) {
        dq::set_error("dq_dag_create: null pointer");
        return nullptr;
    }
    Dag* d = new Dag;
    d->n = n_ops;
    d->succ_off.assign(succ_off, succ_off + n_ops + 1);
    d->succ.assign(succ, succ + succ_off[n_ops]);
    d->targets.assign(target_mask, target_mask + n_ops);
    d->fusable.assign(fusable, fusable + n_ops);
    d->cur.assign(n_ops, -1);
    return d;
}

extern "C" void dq_dag_destroy(void* dag) { delete static_cast<Dag*>(dag); }

extern "C" int dq_dag_closure(void* dag, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, int* stuck,
                              int* nstuck, int* changed_idx, int* changed_val, int* nchanged) {
    if (!dag || !indeg || (nready > 0 && !ready)) {
        dq::set_error("dq_dag_closure: null pointer");
        return DQ_ERR_ARG;
    }
    return closure(*static_cast<Dag*>(dag), tile, cap, indeg, ready, nready, stuck, nstuck, changed_idx, changed_val, nchanged);
}

extern "C" int dq_dag_rank(void* dag, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, const int* cand,
                           int ncand, int* counts) {
    if (!dag || !indeg || (nready > 0 && !ready) || (ncand > 0 && (!cand || !counts))) {
        dq::set_error("dq_dag_rank: null pointer");
        return DQ_ERR_ARG;
    }
    Dag& d = *static_cast<Dag*>(dag);
    for (int c = 0; c < ncand; ++c)
        counts[c] = closure(d, tile | (1ull << cand[c]), cap, indeg, ready, nready, nullptr, nullptr, nullptr, nullptr, nullptr);
    return DQ_OK;
}

// One growth step of fusion._grow_tile in one call: the dry run with `tile`, the qubits outside it that the gates left
// stuck at the front are waiting for (cand_q, with how many gates wait for each: cand_w), and for every such qubit the
// dry run with the tile plus that qubit (cand_count).  Returns the number of candidates; *base = gates retired with `tile`.
extern "C" int dq_dag_grow_step(void* dag, uint64_t tile, int cap, const int* indeg, const int* ready, int nready, int* base,
                                int* cand_q, int* cand_w, int* cand_count) {
    if (!dag || !indeg || (nready > 0 && !ready) || !base || !cand_q || !cand_w || !cand_count) {
        dq::set_error("dq_dag_grow_step: null pointer");
        return DQ_ERR_ARG;
    }
    Dag& d = *static_cast<Dag*>(dag);
    std::vector<int> stuck((size_t)d.n + 1);
    int ns = 0;
    *base = closure(d, tile, cap, indeg, ready, nready, stuck.data(), &ns, nullptr, nullptr, nullptr);
    int w[64] = {0}, order[64], nq = 0;
    for (int k = 0; k < ns; ++k) {
        const int i = stuck[k];
        if (!d.fusable[i]) continue;
        uint64_t miss = d.targets[i] & ~tile;
        while (miss) {
            const int q = __builtin_ctzll(miss);
            miss &= miss - 1;
            if (w[q]++ == 0) order[nq++] = q;
        }
    }
    if (*base >= cap) return 0;
    for (int c = 0; c < nq; ++c) {
        cand_q[c] = order[c];
        cand_w[c] = w[order[c]];
        cand_count[c] = closure(d, tile | (1ull << order[c]), cap, indeg, ready, nready, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    return nq;
}
