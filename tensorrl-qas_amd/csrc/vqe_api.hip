// vqe_api.hip - host side of libvqe_hip.so: handle management, Hamiltonian layout, batch
// upload and kernel dispatch behind the C ABI of include/vqe_hip.h.
#include "../../include/vqe_hip.h"
#include "vqe_device.h"
#include "vqe_stream.h"
#include "vqe_dm.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace vqe;

namespace {

std::string g_create_error;

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = n + n / 4 + 16;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
};

}  // namespace

struct vqe_handle {
  int n = 0, dev = 0;
  bool lds_path = true;
  int cu_count = 0, lds_per_cu = 0, last_wg_per_cu = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float last_ms = 0.f;
  std::string err;

  DevBuf<double2> init;
  // Hamiltonian (host copy, grouped by X mask)
  std::vector<uint32_t> gx_all;
  std::vector<std::vector<int>> group_terms;
  std::vector<uint64_t> hx, hz;
  std::vector<double> hcr, hci;
  int shard_rank = 0, shard_world = 1;
  int amp_rank = 0, amp_world = 1;
  bool ham_set = false;
  DevBuf<uint32_t> d_gx, d_term_z;
  DevBuf<double> d_term_cr, d_term_ci, d_tables;
  DevBuf<int32_t> d_tab_r, d_tab_i;

  DevBuf<int32_t> d_term_off;
  DevBuf<uint32_t> d_urec;   // unit path (HamDev::urec / utab)
  DevBuf<double> d_utab;
  HamDev ham{};
  NoiseCfg noise{0.0, 0.0, 0ull, 0ull, 0.0};

  // single circuit
  std::vector<GateRec> circ;
  int circ_params = -1;

  // resident batch
  int batch = 0;
  int64_t total_params = 0;
  int max_ops = 0, max_params = 0, max_pair = 0;
  std::vector<int64_t> h_par_begin;
  std::vector<int32_t> h_par_count;
  DevBuf<GateRec> d_gates;
  DevBuf<int64_t> d_gate_begin, d_par_begin, d_scratch_begin;
  DevBuf<int32_t> d_gate_count, d_par_count, d_nfev, d_new_gate, d_order;
  bool has_new_gate = false;
  std::vector<int32_t> h_gate_count;
  // host copy of the resident batch (the streaming path builds the pre-action circuits of an env-step from it)
  std::vector<GateRec> h_gates;
  std::vector<int64_t> h_gate_begin;
  std::vector<double> h_theta;
  std::vector<int32_t> h_new_gate;
  // streaming env-step: the pre-action batch (device)
  DevBuf<GateRec> d_gates2;
  DevBuf<int64_t> d_gate_begin2, d_par_begin2;
  DevBuf<int32_t> d_gate_count2, d_par_count2;
  DevBuf<double> d_theta, d_x, d_xraw, d_f, d_scratch;
  DevBuf<double2> d_state;
  DevBuf<unsigned long long> d_dbg;
  DevBuf<double> d_trace;
  bool trace_on = false;
  int trace_maxfun = 0, trace_stride = 0, trace_batch = 0;
  // exact channel mode of the noisy path (vqe_set_noise_mode, vqe_dm.h): density matrix and its Hamiltonian terms
  int noise_mode = 0;
  DevBuf<double2> dm_rho;
  DevBuf<double> dm_S, dm_partial, dm_cr, dm_ci;
  DevBuf<uint32_t> dm_gx, dm_tz;
  DevBuf<int32_t> dm_toff;
  int dm_groups = 0, dm_blocks_last = 0;
  float dm_gpu_ms = 0.f;       // device time of the last exact-mode run (init + block sweeps + tr(rho H)), summed over its evaluations
  bool last_run_dm = false;
  hipEvent_t dm_ev0 = nullptr, dm_ev1 = nullptr;
  uint64_t dm_ham_gen = ~0ull;
  DevBuf<double> d_cob_x0, d_cob_xres, d_cob_f;      // device-resident lock-step COBYLA of the streaming path
  DevBuf<int32_t> d_cob_nfev, d_cob_active;
  void* comm = nullptr;      // RCCL communicator of vqe_comm_init (ncclComm_t)
  int comm_world = 0;
  StreamWork sw;  // streaming-path work buffers
  uint64_t gen = 0;   // bumped whenever a resident batch / Hamiltonian shard / noise setting changes (plans of vqe_tile.h)
};

#ifdef VQE_STAMPS
static unsigned long long cby_out_[8];
#endif
namespace {

int fail(vqe_t* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
#define HIP_TRY(h, expr)                                                               \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return fail(h, _e == hipErrorOutOfMemory ? VQE_ENOMEM : VQE_EHIP,                \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                  \
  } while (0)

template <class T>
int upload(vqe_t* h, DevBuf<T>& b, const T* src, size_t n) {
  HIP_TRY(h, b.reserve(n ? n : 1));
  if (n) HIP_TRY(h, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return VQE_OK;
}

// Pauli-term sharding: greedy bin packing of X-mask groups over ranks by cost (table length
// on the LDS path, partner sweep + terms on the streaming path); deterministic, so every
// rank computes the same partition.
std::vector<int> assign_groups(const std::vector<uint32_t>& gx, const std::vector<std::vector<int>>& terms,
                               bool lds_path, int world) {
  std::vector<int> order(gx.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  auto cost = [&](int g) -> double {
    return lds_path ? (gx[g] == 0 ? 2.0 : 1.0) : 1.0 + 0.25 * (double)terms[g].size();
  };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
  std::vector<double> load(world, 0.0);
  std::vector<int> owner(gx.size(), 0);
  for (int g : order) {
    int best = 0;
    for (int r = 1; r < world; ++r) if (load[r] < load[best]) best = r;
    load[best] += cost(g);
    owner[g] = best;
  }
  return owner;
}

// ---- canonical index map of the register path -----------------------------------------------
// p' = M p over GF(2).  The top R rows of M are functionals chosen greedily so that as many X
// masks as possible have a non-zero image in the register bits (every one of them in the
// molecular Hamiltonians tried); the lower rows complete M to an invertible matrix with unit
// vectors.  Pauli masks transform as x' = M x, z' = M^-T z.
struct IndexMap {
  int n = 0;
  uint32_t row[32] = {0};      // rows of M (identity beyond the register path's n <= 13)
  uint32_t inv_col[32] = {0};  // columns of M^-1 (as bit masks over its rows)
  uint32_t map_x(uint32_t x) const {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= (uint32_t)(__builtin_popcount(row[i] & x) & 1) << i;
    return r;
  }
  uint32_t map_z(uint32_t z) const {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= (uint32_t)(__builtin_popcount(inv_col[i] & z) & 1) << i;
    return r;
  }
};

IndexMap identity_map(int n) {
  IndexMap m;
  m.n = n;
  for (int i = 0; i < n; ++i) m.row[i] = m.inv_col[i] = 1u << i;
  return m;
}

// M^-1 by Gauss-Jordan on [M | I] (rows as bit masks) -> inv_col
void finish_inverse(IndexMap& m) {
  const int n = m.n;
  uint32_t a[32], inv[32];
  for (int i = 0; i < n; ++i) { a[i] = m.row[i]; inv[i] = 1u << i; }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    while (piv < n && !((a[piv] >> c) & 1u)) ++piv;
    std::swap(a[c], a[piv]);
    std::swap(inv[c], inv[piv]);
    for (int r = 0; r < n; ++r)
      if (r != c && ((a[r] >> c) & 1u)) { a[r] ^= a[c]; inv[r] ^= inv[c]; }
  }
  for (int i = 0; i < n; ++i) {
    uint32_t col = 0;
    for (int j = 0; j < n; ++j) col |= ((inv[j] >> i) & 1u) << j;
    m.inv_col[i] = col;
  }
}

IndexMap choose_index_map(int n, int lt, const std::vector<uint32_t>& xs) {
  IndexMap m;
  m.n = n;
  const int R = n - lt;
  std::vector<uint32_t> rem(xs), rows;     // masks not yet hit / chosen functionals (any order)
  uint32_t ech[32] = {0};                  // echelon basis of the chosen rows, by highest bit
  auto independent = [&](uint32_t v) {
    for (int bit = n - 1; bit >= 0 && v; --bit)
      if (((v >> bit) & 1u) && ech[bit]) v ^= ech[bit];
    return v;
  };
  auto add_row = [&](uint32_t f) {
    const uint32_t red = independent(f);
    ech[31 - __builtin_clz(red)] = red;
    rows.push_back(f);
  };
  for (int i = 0; i < R; ++i) {
    uint32_t best = 0;
    int best_hits = -1;
    if (!rem.empty()) {
      for (uint32_t f = 1; f < (1u << n); ++f) {
        int hits = 0;
        for (uint32_t x : rem) hits += __builtin_popcount(f & x) & 1;
        if (hits > best_hits && independent(f)) { best_hits = hits; best = f; }
      }
    }
    if (best_hits <= 0) {   // nothing left to hit: any independent unit functional
      for (int bit = n - 1; bit >= 0; --bit) if (independent(1u << bit)) { best = 1u << bit; break; }
    }
    add_row(best);
    std::vector<uint32_t> keep;
    for (uint32_t x : rem) if (!(__builtin_popcount(best & x) & 1)) keep.push_back(x);
    rem.swap(keep);
  }
  for (int i = 0; i < R; ++i) m.row[lt + i] = rows[i];
  int filled = 0;
  for (int bit = 0; bit < n && filled < lt; ++bit)
    if (independent(1u << bit)) { add_row(1u << bit); m.row[filled++] = 1u << bit; }
  finish_inverse(m);
  return m;
}

// ---- unit path: X-mask groups with mostly-zero sign-sum tables -------------------------------------
// The sign-sum table D_x(p) = sum_k c_k (-1)^{popc(p & z_k)} of a fermionic excitation operator vanishes EXACTLY on
// every pair {p, p^x} whose occupation pattern the operator does not connect (a hopping pair XZ..ZX + YZ..ZY acts on
// 01 <-> 10 only, a double-excitation octet on one pattern pair in eight): 76 % of the entries of the bench
// Hamiltonian, 69 % of the shipped H2O one.  A group is cut into sub-cubes of NT pairs (fix F = n-1-LT index bits
// besides the selector bit); only sub-cubes on which D does not vanish become *units* (HamDev::urec).
// Entries below kUnitZeroTol x sum_k |c_k| are rounding residues of sums that cancel exactly in real arithmetic
// (3w - w - w - w is not 0 in floating point) and count as zero.
constexpr double kUnitZeroTol = 0x1p-44;       // 5.7e-14 relative: far above the residues (~1e-16), far below any term

// a permutation of the qubits as canonical index map: position lt.. (the register bits of the class path) cover
// the masks of the dense groups, the other qubits are placed by how often they are a fixed / selector bit of a
// unit - the most frequent ones highest, so that the lowest index bits (consecutive lanes, LDS banks) stay free
IndexMap choose_permutation(int n, int lt, const std::vector<uint32_t>& dense_xs, const std::vector<int>& hole_freq) {
  IndexMap m;
  m.n = n;
  std::vector<int> at(n, -1);            // qubit at canonical position i
  std::vector<bool> used(n, false);
  std::vector<uint32_t> rem(dense_xs);
  for (int pos = n - 1; pos >= 0; --pos) {
    int best = -1;
    if (pos >= lt && !rem.empty()) {
      int best_hits = 0;
      for (int q = 0; q < n; ++q) {
        if (used[q]) continue;
        int hits = 0;
        for (uint32_t x : rem) hits += (x >> q) & 1u;
        if (hits > best_hits) { best_hits = hits; best = q; }
      }
      if (best >= 0) {
        std::vector<uint32_t> keep;
        for (uint32_t x : rem) if (!((x >> best) & 1u)) keep.push_back(x);
        rem.swap(keep);
      }
    }
    if (best < 0)
      for (int q = n - 1; q >= 0; --q)
        if (!used[q] && (best < 0 || hole_freq[q] > hole_freq[best])) best = q;
    used[best] = true;
    at[pos] = best;
  }
  for (int i = 0; i < n; ++i) m.row[i] = m.inv_col[i] = 1u << at[i];
  return m;
}

// sign-sum table of a real group over pair representatives p0 = insert0(q, sel) in the index space of `im`
// (factor 2 of the pair symmetry included, as in the pair tables of the group lists)
void pair_table(const vqe_t* h, int g, const IndexMap& im, int n, int sel, std::vector<double>& D, double* scale) {
  const size_t len = (size_t)1 << (n - 1);
  D.assign(len, 0.0);
  *scale = 0.0;
  for (int k : h->group_terms[g]) {
    const uint32_t z = im.map_z((uint32_t)h->hz[k]);
    const double c = 2.0 * h->hcr[k];
    *scale += std::fabs(c);
    for (size_t q = 0; q < len; ++q) {
      const uint32_t p0 = (uint32_t)(((q >> sel) << (sel + 1)) | (q & (((size_t)1 << sel) - 1)));
      D[q] += (__builtin_popcount(p0 & z) & 1) ? -c : c;
    }
  }
}

// Fixed bits of a group's units: greedily the index bits (not the selector) on which the active pairs agree most;
// stops when no bit helps any more (every remaining one doubles the number of active patterns).  Returns the
// number of active patterns on `fixed`; the units of the group are patterns x 2^(F - |fixed|) (filler bits).
int choose_fixed_bits(int n, int sel, int F, const std::vector<uint32_t>& act, std::vector<int>& fixed) {
  fixed.clear();
  if (act.empty()) return 0;
  int patterns = 1;
  auto count = [&](int extra) {
    uint32_t seen = 0;
    for (uint32_t p0 : act) {
      uint32_t key = 0;
      for (size_t i = 0; i < fixed.size(); ++i) key |= ((p0 >> fixed[i]) & 1u) << i;
      key |= ((p0 >> extra) & 1u) << fixed.size();
      seen |= 1u << key;
    }
    return __builtin_popcount(seen);
  };
  while ((int)fixed.size() < F) {
    int best = -1, best_cnt = 1 << 30;
    for (int b = n - 1; b >= 0; --b) {
      if (b == sel || std::find(fixed.begin(), fixed.end(), b) != fixed.end()) continue;
      const int c = count(b);
      if (c < best_cnt) { best_cnt = c; best = b; }
    }
    if (best < 0 || best_cnt >= 2 * patterns) break;
    fixed.push_back(best);
    patterns = best_cnt;
  }
  return patterns;
}

// Build (or rebuild after re-sharding) the device Hamiltonian.
int build_hamiltonian(vqe_t* h) {
  const int n = h->n;
  ++h->gen;
  const std::vector<int> owner = assign_groups(h->gx_all, h->group_terms, h->lds_path, h->shard_world);
  std::vector<int> mine;
  for (size_t g = 0; g < owner.size(); ++g) if (owner[g] == h->shard_rank) mine.push_back((int)g);
  std::sort(mine.begin(), mine.end());

  std::vector<uint32_t> gx;
  std::vector<int32_t> term_off{0};
  std::vector<uint32_t> term_z;
  std::vector<double> term_cr, term_ci;
  const size_t dim = (size_t)1 << n;
  std::vector<int32_t> tab_r, tab_i;
  std::vector<double> tables;
  // LDS-path order: diagonal group first, then real-table groups (padded with zero-table
  // dummy groups to a multiple of energy_pd(n)), then groups that also need an imaginary table.
  auto group_has_im = [&](int g) {
    for (int k : h->group_terms[g]) if (h->hci[k] != 0.0) return true;
    return false;
  };
  // register path: canonical index p' = M p (see IndexMap); all masks below are in p'
  const bool reg_path = h->lds_path && n >= kRegMinQubits;
  const int lt = geo_lt(n);                        // Geo<N>::LT of the register path
  IndexMap im = identity_map(n), pm = identity_map(n);      // pm: the qubit permutation under the units' bank shear
  // unit path (8 <= n <= 13): pass 1 in the qubit order as given - which groups are sparse, which qubits are
  // their fixed / selector bits
  const int unit_F = n - 1 - lt;
  static const bool units_on = [] { const char* e = std::getenv("VQE_UNITS"); return !(e && e[0] == '0'); }();   // A/B knob
  std::vector<char> sparse(h->gx_all.size(), 0);
  bool any_sparse = false;
  if (h->lds_path && n >= kUnitMinQubits && units_on && unit_F >= 1) {
    std::vector<int> hole_freq(n, 0);
    std::vector<uint32_t> dense_xs;
    std::vector<double> D;
    std::vector<uint32_t> act;
    std::vector<int> fixed;
    for (int g : mine) {
      const uint32_t x = h->gx_all[g];
      if (!x || group_has_im(g)) continue;
      const int sel = 31 - __builtin_clz(x);
      double scale;
      pair_table(h, g, im, n, sel, D, &scale);
      act.clear();
      for (size_t q = 0; q < D.size(); ++q)
        if (std::fabs(D[q]) > kUnitZeroTol * scale)
          act.push_back((uint32_t)(((q >> sel) << (sel + 1)) | (q & (((size_t)1 << sel) - 1))));
      const int patterns = choose_fixed_bits(n, sel, unit_F, act, fixed);
      const int units = patterns << (unit_F - (int)fixed.size());
      // a unit costs 2 LDS reads, a group of the class path 2^(F+1) / 2: sparse when no more than half of its
      // sub-cubes are active
      if (2 * units <= (1 << unit_F)) {
        sparse[g] = 1;
        any_sparse = true;
        for (int b : fixed) ++hole_freq[b];
        ++hole_freq[sel];
      } else {
        dense_xs.push_back(x);
      }
    }
    if (any_sparse && reg_path) {      // (below the register path the state stays in logical order)
      pm = choose_permutation(n, lt, dense_xs, hole_freq);
      // the canonical map of the handle: the permutation followed by the bank shear of the unit path (kSwzCode)
      im = pm;
      if (kUnitShear) {
        for (int i = 0; i < 4 && i < n; ++i)
          for (int j = 4; j < 8 && j < n; ++j)
            if ((kSwzCode[j - 4] >> i) & 1u) im.row[i] ^= pm.row[j];
        finish_inverse(im);
      }
    }
  }
  if (reg_path && !any_sparse) {
    std::vector<uint32_t> xs;
    for (int g : mine) if (h->gx_all[g] && !group_has_im(g)) xs.push_back(h->gx_all[g]);
    im = choose_index_map(n, lt, xs);
  }
  // pass 2 in the canonical index space: the units themselves
  std::vector<uint32_t> urec;
  std::vector<double> utab;
  if (any_sparse) {
    const size_t NT = (size_t)1 << lt;
    std::vector<double> D;
    std::vector<uint32_t> act;
    std::vector<int> fixed;
    for (int g : mine) {
      if (!sparse[g]) continue;
      const uint32_t x = pm.map_x(h->gx_all[g]);      // the cubes are axis aligned in the permuted index, before the shear
      const int sel = 31 - __builtin_clz(x);
      double scale;
      pair_table(h, g, pm, n, sel, D, &scale);
      auto rep = [&](size_t q) { return (uint32_t)(((q >> sel) << (sel + 1)) | (q & (((size_t)1 << sel) - 1))); };
      act.clear();
      for (size_t q = 0; q < D.size(); ++q) if (std::fabs(D[q]) > kUnitZeroTol * scale) act.push_back(rep(q));
      choose_fixed_bits(n, sel, unit_F, act, fixed);
      // filler bits: the highest positions that are neither fixed nor the selector
      for (int b = n - 1; b >= 0 && (int)fixed.size() < unit_F; --b)
        if (b != sel && std::find(fixed.begin(), fixed.end(), b) == fixed.end()) fixed.push_back(b);
      uint32_t fmask = 0;
      for (int b : fixed) fmask |= 1u << b;
      uint32_t seen_keys[8];
      int n_keys = 0;
      for (uint32_t p0 : act) {       // distinct patterns of the fixed bits among the active pairs
        const uint32_t key = p0 & fmask;
        bool dup = false;
        for (int i = 0; i < n_keys; ++i) dup |= seen_keys[i] == key;
        if (!dup && n_keys < 8) seen_keys[n_keys++] = key;
      }
      std::sort(seen_keys, seen_keys + n_keys);
      // deposit masks: the free positions between consecutive holes (holes = fixed bits + selector), byte addresses
      std::vector<int> holes(fixed);
      holes.push_back(sel);
      std::sort(holes.begin(), holes.end());
      uint32_t m[5] = {0, 0, 0, 0, 0};
      for (size_t i = 0; i <= holes.size(); ++i) {
        const int lo = i == 0 ? 0 : holes[i - 1] + 1, hi = i == holes.size() ? n : holes[i];
        m[i] = (uint32_t)((((uint64_t)1 << hi) - ((uint64_t)1 << lo)) << 4);
      }
      for (int ki = 0; ki < n_keys; ++ki) {
        const uint32_t s = seen_keys[ki];
        const uint32_t toff = (uint32_t)(utab.size() * sizeof(double));
        const uint32_t rec[8] = {m[0], m[1], m[2], m[3], m[4], (kUnitShear ? swz_index(s) : s) << 4, (kUnitShear ? swz_index(x) : x) << 4, toff};
        urec.insert(urec.end(), rec, rec + 8);
        for (size_t t = 0; t < NT; ++t) {
          uint32_t p0 = s, tb = 0;       // deposit the bits of t into the free positions, ascending
          for (int b = 0; b < n; ++b)
            if (!((fmask >> b) & 1u) && b != sel) { p0 |= (uint32_t)((t >> tb) & 1u) << b; ++tb; }
          const size_t q = ((size_t)(p0 >> (sel + 1)) << sel) | (p0 & (((size_t)1 << sel) - 1));
          const double d = D[q];
          utab.push_back(std::fabs(d) > kUnitZeroTol * scale ? d : 0.0);
        }
      }
    }
    if (utab.size() * sizeof(double) + (size_t)kUnitUnroll * NT * sizeof(double) > 0x7FFFFFFFu)
      return fail(h, VQE_EINVAL, "Hamiltonian too large for the LDS-resident path");
    // padding to a multiple of kUnitUnroll: units with a table of zeros
    while ((urec.size() / 8) % kUnitUnroll) {
      const uint32_t rec[8] = {0, 0, 0, 0, 0, 0, 0, (uint32_t)(utab.size() * sizeof(double))};
      urec.insert(urec.end(), rec, rec + 8);
      utab.resize(utab.size() + NT, 0.0);
    }
    // table layout the unit loop reads: per trip of kUnitTrip units [thread][unit of the trip] - a thread's values of a
    // trip are 32 contiguous bytes (two 16-byte loads instead of four 8-byte ones, one offset computation per trip:
    // 335.7 -> 331.2 ms on one box)
    {
      std::vector<double> t(utab.size());
      const size_t n_trips = urec.size() / 8 / kUnitTrip;
      for (size_t T = 0; T < n_trips; ++T)
        for (size_t j = 0; j < (size_t)kUnitTrip; ++j)
          for (size_t th = 0; th < NT; ++th)
            t[(T * NT + th) * kUnitTrip + j] = utab[(T * kUnitTrip + j) * NT + th];
      utab.swap(t);
    }
  }
  auto gxm = [&](int g) { return im.map_x(h->gx_all[g]); };
  // section of a group: 0 diagonal, 1 real with a register bit in x' (register path only),
  // 2 other real groups, 3 groups that also need an imaginary table
  auto rank_of = [&](int g) {
    if (group_has_im(g)) return 3;
    const uint32_t x = gxm(g);
    if (x == 0) return 0;
    return reg_path && (x >> lt) ? 1 : 2;
  };
  // ... and inside a section by the top bit of x'
  auto top_bit = [&](int g) { const uint32_t x = gxm(g); return x ? 31 - __builtin_clz(x) : -1; };
  if (h->lds_path)
    std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) {
      return rank_of(a) != rank_of(b) ? rank_of(a) < rank_of(b) : top_bit(a) > top_bit(b);
    });
  int has_diag = 0, n_real = 0, n_cls = 0;
  auto add_dummy = [&](uint32_t xd) {   // zero table: pads a section to a multiple of energy_pd(n)
    gx.push_back(xd);
    term_off.push_back((int32_t)term_z.size());
    tab_r.push_back((int32_t)tables.size());
    tab_i.push_back(-1);
    tables.resize(tables.size() + dim / 2, 0.0);
  };
  int cur_rank = 0;
  auto enter_section = [&](int rank) {   // rank 4 = end of the list
    if (h->lds_path) {
      if (cur_rank <= 1 && rank >= 2) while (n_cls % energy_pd(n)) { add_dummy(1u << lt); ++n_cls; ++n_real; }
      if (cur_rank <= 2 && rank >= 3) while ((n_real - n_cls) % energy_pd(n)) { add_dummy(1u); ++n_real; }
    }
    cur_rank = rank;
  };
  for (int g : mine) {
    if (sparse[g]) continue;         // lives in the unit list
    const uint32_t x = gxm(g);
    const bool has_im = group_has_im(g);
    const int rank = rank_of(g);
    enter_section(rank);
    gx.push_back(x);
    for (int k : h->group_terms[g]) {
      term_z.push_back(im.map_z((uint32_t)h->hz[k]));
      term_cr.push_back(h->hcr[k]);
      term_ci.push_back(h->hci[k]);
    }
    term_off.push_back((int32_t)term_z.size());
    if (h->lds_path) {
      if (rank == 0) has_diag = 1; else if (!has_im) ++n_real;
      if (rank == 1) ++n_cls;
      const size_t len = x == 0 ? dim : dim / 2;
      const int hb = x == 0 ? 0 : 31 - __builtin_clz(x);
      if (tables.size() + 2 * len > 0x7FFFFFFFu) return fail(h, VQE_EINVAL, "Hamiltonian too large for the LDS-resident path");
      tab_r.push_back((int32_t)tables.size());
      tables.resize(tables.size() + len, 0.0);
      if (has_im) { tab_i.push_back((int32_t)tables.size()); tables.resize(tables.size() + len, 0.0); }
      else tab_i.push_back(-1);
      double* tr = tables.data() + tab_r.back();
      double* ti = has_im ? tables.data() + tab_i.back() : nullptr;
      // index of the pair member that entry q of the table belongs to
      std::vector<uint32_t> pidx(len);
      for (size_t q = 0; q < len; ++q) {
        if (x == 0) pidx[q] = (uint32_t)q;
        else if (rank == 1) {   // [j/2][tid][j&1] with r = insert0(j, cls), p' = tid | r << lt
          const int cls = hb - lt;
          const uint32_t t = (uint32_t)((q >> 1) & (((size_t)1 << lt) - 1));
          const uint32_t j = (uint32_t)(((q >> (lt + 1)) << 1) | (q & 1));
          const uint32_t r = ((j >> cls) << (cls + 1)) | (j & ((1u << cls) - 1u));
          pidx[q] = t | (r << lt);
        } else {
          pidx[q] = (uint32_t)(((q >> hb) << (hb + 1)) | (q & (((size_t)1 << hb) - 1)));
        }
      }
      for (int k : h->group_terms[g]) {
        const uint32_t z = im.map_z((uint32_t)h->hz[k]);
        for (size_t q = 0; q < len; ++q) {
          // pair tables carry the factor 2 of the p <-> p^x symmetry
          const double sgn = ((__builtin_popcount(pidx[q] & z) & 1) ? -1.0 : 1.0) * (x == 0 ? 1.0 : 2.0);
          tr[q] += sgn * h->hcr[k];
          if (ti) ti[q] += sgn * h->hci[k];
        }
      }
    } else {
      tab_r.push_back(0);
      tab_i.push_back(has_im ? 0 : -1);
    }
  }
  enter_section(4);
  int rc;
  if ((rc = upload(h, h->d_gx, gx.data(), gx.size()))) return rc;
  if ((rc = upload(h, h->d_tab_r, tab_r.data(), tab_r.size()))) return rc;
  if ((rc = upload(h, h->d_tab_i, tab_i.data(), tab_i.size()))) return rc;
  if ((rc = upload(h, h->d_tables, tables.data(), tables.size()))) return rc;
  if ((rc = upload(h, h->d_term_off, term_off.data(), term_off.size()))) return rc;
  if ((rc = upload(h, h->d_term_z, term_z.data(), term_z.size()))) return rc;
  if ((rc = upload(h, h->d_term_cr, term_cr.data(), term_cr.size()))) return rc;
  if ((rc = upload(h, h->d_term_ci, term_ci.data(), term_ci.size()))) return rc;
  if ((rc = upload(h, h->d_urec, urec.data(), urec.size()))) return rc;
  if ((rc = upload(h, h->d_utab, utab.data(), utab.size()))) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // host vectors go out of scope
  h->ham.n_groups = (int)gx.size();
  h->ham.n_terms = (int)term_z.size();
  h->ham.gx = h->d_gx.p;
  h->ham.tab_r = h->d_tab_r.p;
  h->ham.tab_i = h->d_tab_i.p;
  h->ham.tables = h->d_tables.p;
  h->ham.has_diag = has_diag;
  h->ham.n_real = n_real;
  h->ham.n_cls = n_cls;
  for (int i = 0; i < 16; ++i) h->ham.mrow[i] = im.row[i];
  h->ham.n_units = (int)(urec.size() / 8);
  h->ham.urec = h->d_urec.p;
  h->ham.utab = h->d_utab.p;
  h->ham.term_off = h->d_term_off.p;
  h->ham.term_z = h->d_term_z.p;
  h->ham.term_cr = h->d_term_cr.p;
  h->ham.term_ci = h->d_term_ci.p;
  return VQE_OK;
}

int check_gates(vqe_t* h, int64_t n_gates, const int32_t* kind, const int32_t* q0,
                const int32_t* q1, const int32_t* pidx, int n_params) {
  for (int64_t i = 0; i < n_gates; ++i) {
    const int k = kind[i];
    if (k < 0 || k > VQE_GATE_DEPOL2) return fail(h, VQE_EINVAL, "unknown gate kind");
    if (q0[i] < 0 || q0[i] >= h->n) return fail(h, VQE_EINVAL, "gate qubit out of range");
    if (k == VQE_GATE_CNOT || k == VQE_GATE_DEPOL2) {
      if (q1[i] < 0 || q1[i] >= h->n || q1[i] == q0[i])
        return fail(h, VQE_EINVAL, "two-qubit gate needs two distinct qubits in range");
    }
    if (k >= VQE_GATE_RX && k <= VQE_GATE_RZ) {
      if (pidx[i] < 0 || pidx[i] >= n_params)
        return fail(h, VQE_EINVAL, "rotation parameter index out of range");
    }
  }
  return VQE_OK;
}

template <int N>
int launch_lds(vqe_t* h, int which, const BatchArgs& A) {
  size_t lds = lds_bytes(N, A.max_ops, A.max_params, A.ham.n_groups, A.max_pair);
  // measurement knob: VQE_LDS_PAD=bytes of unused LDS per workgroup lowers the workgroups per CU
  static const long lds_pad = [] { const char* e = std::getenv("VQE_LDS_PAD"); return e ? std::atol(e) : 0L; }();
  if (lds_pad > 0) lds += (size_t)lds_pad;
  if (lds > (size_t)h->lds_per_cu)
    return fail(h, VQE_EINVAL, "circuit too large for the LDS-resident path (gates + parameters)");
  // register path: the raw ops are staged in the (idle) state region, 2^n records at most
  if (N >= kRegMinQubits && (size_t)A.max_ops > ((size_t)1 << N))
    return fail(h, VQE_EINVAL, "circuit too large for the LDS-resident path (more than 2^n rotations)");
  // More than 64 parameters in a circuit (the trainable regime): the WIDE variant of the kernel - on 256-thread
  // workgroups the optimiser update runs on the whole workgroup (BlockCtx), on one-wave workgroups (n <= 9) the lanes
  // walk their rows of the matrices side by side (WaveRowsCtx).  The plain variant keeps neither (registers).
  constexpr bool kHasWide = N >= 6;
  static const bool wide_on = [] { const char* e = std::getenv("VQE_WIDE_UPDATE"); return !(e && e[0] == '0'); }();   // A/B knob
  const bool wide = kHasWide && wide_on && which == 1 && A.max_params > 64;
  const bool noisy = A.noise.p1 > 0.0 || A.noise.p2 > 0.0;
  constexpr bool kW = false;     // up to 64 parameters per circuit: the instantiation without the workgroup-wide update
  const void* fn = which == 0 ? (const void*)k_lds_energy<N>
                   : which == 1 ? (wide ? (noisy ? (const void*)k_lds_minimize<N, kHasWide, true> : (const void*)k_lds_minimize<N, kHasWide, false>)
                                        : (noisy ? (const void*)k_lds_minimize<N, kW, true> : (const void*)k_lds_minimize<N, kW, false>))
                                : (const void*)k_lds_state<N>;
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {   // the unit loop addresses the state region absolutely (lds_load_abs): the dynamic LDS must start at 0
    hipFuncAttributes fa;
    HIP_TRY(h, hipFuncGetAttributes(&fa, fn));
    if (fa.sharedSizeBytes != 0) return fail(h, VQE_ESTATE, "LDS-resident kernel was built with static LDS");
  }
  h->last_wg_per_cu = std::max(1, std::min(8, (int)(h->lds_per_cu / lds)));
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  const dim3 grid(which == 2 ? 1 : A.batch), block(Geo<N>::NT);
  if (which == 0) hipLaunchKernelGGL(k_lds_energy<N>, grid, block, lds, h->stream, A);
  else if (which == 1 && wide && noisy) hipLaunchKernelGGL((k_lds_minimize<N, kHasWide, true>), grid, block, lds, h->stream, A);
  else if (which == 1 && wide) hipLaunchKernelGGL((k_lds_minimize<N, kHasWide, false>), grid, block, lds, h->stream, A);
  else if (which == 1 && noisy) hipLaunchKernelGGL((k_lds_minimize<N, kW, true>), grid, block, lds, h->stream, A);
  else if (which == 1) hipLaunchKernelGGL((k_lds_minimize<N, kW, false>), grid, block, lds, h->stream, A);
  else hipLaunchKernelGGL(k_lds_state<N>, grid, block, lds, h->stream, A);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  return VQE_OK;
}

int dispatch_lds(vqe_t* h, int which, const BatchArgs& A) {
  switch (h->n) {
#define C(N) case N: return launch_lds<N>(h, which, A);
#ifdef VQE_ONLY_N      // kernel experiments (tools/build_only_n.sh): one size, seconds to build
    C(VQE_ONLY_N)
#else
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13)
#endif
#undef C
  }
  return fail(h, VQE_EINVAL, "n_qubits outside the LDS-resident range");
}

BatchArgs make_args(vqe_t* h) {
  BatchArgs A{};
  A.n = h->n;
  A.batch = h->batch;
  A.gates = h->d_gates.p;
  A.gate_begin = h->d_gate_begin.p;
  A.gate_count = h->d_gate_count.p;
  A.par_begin = h->d_par_begin.p;
  A.par_count = h->d_par_count.p;
  A.order = h->d_order.p;
  A.theta = h->d_theta.p;
  A.xout = h->d_x.p;
  A.xraw = h->d_xraw.p;
  A.new_gate = nullptr;
  A.env_step = 0;
  A.fout = h->d_f.p;
  A.nfev = h->d_nfev.p;
  A.scratch = h->d_scratch.p;
  A.scratch_begin = h->d_scratch_begin.p;
  A.init = h->init.p;
  A.ham = h->ham;
  A.noise = h->noise;
  A.max_ops = h->max_ops;
  A.max_pair = h->max_pair;
  A.max_params = h->max_params;
  A.state_out = h->d_state.p;
  A.dbg = h->d_dbg.p;
  A.trace = nullptr;
  A.amp_rank = h->amp_rank;
  A.amp_world = h->amp_world;
  return A;
}

// Load a batch given begin/count arrays (circuits may alias the same gate range).
int load_batch(vqe_t* h, int batch, const std::vector<GateRec>& gates,
               const std::vector<int64_t>& gbeg, const std::vector<int32_t>& gcnt,
               const std::vector<int64_t>& pbeg, const std::vector<int32_t>& pcnt,
               const double* theta0, int64_t total_params) {
  int rc;
  std::vector<int64_t> sbeg(batch);
  int64_t stot = 0;
  int max_ops = 1, max_par = 1, max_pair = 0;
  std::vector<double> cost(batch);
  for (int b = 0; b < batch; ++b) {
    sbeg[b] = stot;
    stot += (int64_t)cby::scratch_doubles(pcnt[b], 16);   // the larger of the device contexts' paddings
    stot = (stot + 15) & ~(int64_t)15;  // 128-byte alignment: rows of the optimiser's global arrays are whole sectors / lines
    max_par = std::max(max_par, (int)pcnt[b]);
    int ops = 0, pair = 0;
    for (int64_t i = gbeg[b]; i < gbeg[b] + gcnt[b]; ++i) {
      const int k = gates[i].kind;
      ops += (k == G_CNOT) ? 0 : (k == G_DEPOL2 ? 2 : 1);
      pair += (k == G_RX || k == G_RY) ? 1 : 0;
    }
    max_ops = std::max(max_ops, ops);
    max_pair = std::max(max_pair, pair);
    // expected cycles of one evaluation of the fused kernel beyond the constant energy step:
    // ~1.3 k per simulated op, ~22 per squared parameter for the optimiser update
    double ca = 1300.0, cb = 22.0, cc = 0.0;
    if (const char* e = getenv("VQE_LPT_COEF")) sscanf(e, "%lf,%lf,%lf", &ca, &cb, &cc);   // experiments
    cost[b] = ca * ops + cb * (double)pcnt[b] * (double)pcnt[b] + cc * (double)pcnt[b];
  }
  // workgroup i runs circuit order[i], longest first: the hardware hands workgroups to free CU
  // slots in index order, so the short circuits fill the end of the launch
  std::vector<int32_t> order(batch);
  for (int b = 0; b < batch; ++b) order[b] = b;
  if (!getenv("VQE_NO_LPT"))   // experiments: launch in caller order
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
  if ((rc = upload(h, h->d_gates, gates.data(), gates.size()))) return rc;
  if ((rc = upload(h, h->d_gate_begin, gbeg.data(), gbeg.size()))) return rc;
  if ((rc = upload(h, h->d_gate_count, gcnt.data(), gcnt.size()))) return rc;
  if ((rc = upload(h, h->d_par_begin, pbeg.data(), pbeg.size()))) return rc;
  if ((rc = upload(h, h->d_par_count, pcnt.data(), pcnt.size()))) return rc;
  if ((rc = upload(h, h->d_order, order.data(), order.size()))) return rc;
  if ((rc = upload(h, h->d_scratch_begin, sbeg.data(), sbeg.size()))) return rc;
  if ((rc = upload(h, h->d_theta, theta0, (size_t)total_params))) return rc;
  HIP_TRY(h, h->d_scratch.reserve((size_t)stot + 2));
  HIP_TRY(h, h->d_x.reserve((size_t)total_params + 1));
  HIP_TRY(h, h->d_xraw.reserve((size_t)total_params + 1));
  HIP_TRY(h, h->d_f.reserve(batch));
  HIP_TRY(h, h->d_nfev.reserve(batch));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->batch = batch;
  h->total_params = total_params;
  h->max_ops = (max_ops + 3) & ~3;
  h->max_pair = std::min(h->max_ops, (max_pair + 3) & ~3);
  h->max_params = (max_par + 3) & ~3;
  h->h_par_begin = pbeg;
  h->h_par_count = pcnt;
  h->h_gate_count = gcnt;
  h->h_gates = gates;
  h->h_gate_begin = gbeg;
  h->h_theta.assign(theta0, theta0 + total_params);
  h->has_new_gate = false;
  ++h->gen;
  return VQE_OK;
}

int load_single(vqe_t* h, int batch, const double* theta) {
  if (h->circ_params < 0) return fail(h, VQE_ESTATE, "vqe_set_circuit has not been called");
  const int P = h->circ_params;
  std::vector<int64_t> gbeg(batch, 0), pbeg(batch);
  std::vector<int32_t> gcnt(batch, (int32_t)h->circ.size()), pcnt(batch, P);
  for (int b = 0; b < batch; ++b) pbeg[b] = (int64_t)b * P;
  return load_batch(h, batch, h->circ, gbeg, gcnt, pbeg, pcnt, theta, (int64_t)batch * P);
}

int ready(vqe_t* h) {
  if (!h) return VQE_EINVAL;
  if (!h->ham_set) return fail(h, VQE_ESTATE, "vqe_set_hamiltonian_pauli has not been called");
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no circuits loaded");
  return VQE_OK;
}

// COBYLA of all resident streams in lock-step (one batched evaluation per iteration): the streaming path's form of
// scipy.optimize.minimize(..., method='COBYLA') (environment_qulacs_TN_notin_agent.py:478).  The optimiser state of
// every stream lives on the device (k_s_cobyla: the host build's arithmetic, one thread per stream); the host only
// queues launches and looks at the number of running streams every kStreamPoll iterations - no copy of trial points
// or energies and no synchronisation per evaluation (VQE_STREAM_HOST_COBYLA=1: the round-2 host-driven loop, kept
// for A/B runs).  x: in x0, out the result (layout pbeg / pcnt); trial points travel through d_x.
constexpr int kStreamPoll = 8;

int stream_cobyla_host(vqe_t* h, BatchArgs& A, const std::vector<int64_t>& pbeg, const std::vector<int32_t>& pcnt,
                       std::vector<double>& x, std::vector<double>& f, std::vector<int32_t>& nfev) {
  const int B = h->batch;
  std::vector<vqe_cobyla_t*> cob(B, nullptr);
  struct Guard { std::vector<vqe_cobyla_t*>& v; ~Guard() { for (auto* c : v) vqe_cobyla_destroy(c); } } guard{cob};
  for (int b = 0; b < B; ++b)
    if (vqe_cobyla_create(pcnt[b], x.data() + pbeg[b], A.rhobeg, A.rhoend, A.maxfun, &cob[b]))
      return fail(h, VQE_ENOMEM, "host COBYLA allocation failed");
  HIP_TRY(h, h->d_x.reserve(x.size() + 1));
  uint64_t it = 0;
  int rc = 0;
  for (;;) {
    int active = 0;
    for (int b = 0; b < B; ++b) active += vqe_cobyla_ask(cob[b], x.data() + pbeg[b]) == 1;
    if (!active) break;
    if (!x.empty())
      HIP_TRY(h, hipMemcpyAsync(h->d_x.p, x.data(), x.size() * 8, hipMemcpyHostToDevice, h->stream));
    A.theta = h->d_x.p;  // trial points live in the output buffer; x0 stays untouched
    rc = stream_evaluate(h->sw, A, A.ham.n_terms, h->stream, h->noise.eval_base + (++it), true, h->err, true, h->gen);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(f.data(), h->d_f.p, (size_t)B * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < B; ++b)
      if (vqe_cobyla_ask(cob[b], nullptr) == 1) vqe_cobyla_tell(cob[b], f[b]);
  }
  for (int b = 0; b < B; ++b) vqe_cobyla_result(cob[b], x.data() + pbeg[b], &f[b], &nfev[b], nullptr);
  return VQE_OK;
}

// d_pbeg / d_pcnt: the device copies of pbeg / pcnt (the batch that A describes)
int stream_cobyla(vqe_t* h, BatchArgs& A, const std::vector<int64_t>& pbeg, const std::vector<int32_t>& pcnt,
                  std::vector<double>& x, std::vector<double>& f, std::vector<int32_t>& nfev,
                  const int64_t* d_pbeg, const int32_t* d_pcnt) {
  static const bool host_loop = [] { const char* e = std::getenv("VQE_STREAM_HOST_COBYLA"); return e && e[0] == '1'; }();
  if (host_loop) return stream_cobyla_host(h, A, pbeg, pcnt, x, f, nfev);
  const int B = h->batch;
  const size_t PT = x.size();
  HIP_TRY(h, h->d_x.reserve(PT + 1));
  HIP_TRY(h, h->d_cob_x0.reserve(PT + 1));
  HIP_TRY(h, h->d_cob_xres.reserve(PT + 1));
  HIP_TRY(h, h->d_cob_f.reserve(B));
  HIP_TRY(h, h->d_cob_nfev.reserve(B));
  HIP_TRY(h, h->d_cob_active.reserve((size_t)B + 1));
  if (PT) HIP_TRY(h, hipMemcpyAsync(h->d_cob_x0.p, x.data(), PT * 8, hipMemcpyHostToDevice, h->stream));
  // (trial points of streams that finish early stay where they are: the evaluations go on in lock-step over all streams)
  if (PT) HIP_TRY(h, hipMemcpyAsync(h->d_x.p, x.data(), PT * 8, hipMemcpyHostToDevice, h->stream));
  int32_t* n_active = h->d_cob_active.p + B;
  const dim3 grid((unsigned)((B + 63) / 64)), block(64);
  HIP_TRY(h, hipMemsetAsync(n_active, 0, 4, h->stream));
  hipLaunchKernelGGL(k_s_cobyla<true>, grid, block, 0, h->stream, B, d_pbeg, d_pcnt, (const int64_t*)h->d_scratch_begin.p,
                     h->d_scratch.p, (const double*)h->d_cob_x0.p, h->d_x.p, (const double*)h->d_f.p, A.rhobeg, A.rhoend,
                     A.maxfun, h->d_cob_active.p, n_active, h->d_cob_xres.p, h->d_cob_f.p, h->d_cob_nfev.p);
  A.theta = h->d_x.p;
  uint64_t it = 0;
  int32_t running = 1;
  while (running > 0) {
    int rc = stream_evaluate(h->sw, A, A.ham.n_terms, h->stream, h->noise.eval_base + (++it), true, h->err, true, h->gen);
    if (rc) return rc;
    HIP_TRY(h, hipMemsetAsync(n_active, 0, 4, h->stream));
    hipLaunchKernelGGL(k_s_cobyla<false>, grid, block, 0, h->stream, B, d_pbeg, d_pcnt, (const int64_t*)h->d_scratch_begin.p,
                       h->d_scratch.p, (const double*)h->d_cob_x0.p, h->d_x.p, (const double*)h->d_f.p, A.rhobeg, A.rhoend,
                       A.maxfun, h->d_cob_active.p, n_active, h->d_cob_xres.p, h->d_cob_f.p, h->d_cob_nfev.p);
    if (it % kStreamPoll == 0 || it >= (uint64_t)A.maxfun) {
      HIP_TRY(h, hipMemcpyAsync(&running, n_active, 4, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (it > (uint64_t)A.maxfun + kStreamPoll) return fail(h, VQE_ESTATE, "device COBYLA did not terminate");
  }
  HIP_TRY(h, hipGetLastError());
  if (PT) HIP_TRY(h, hipMemcpyAsync(x.data(), h->d_cob_xres.p, PT * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(f.data(), h->d_cob_f.p, (size_t)B * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(nfev.data(), h->d_cob_nfev.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

// Streaming path (n >= 14): kernels per op; the COBYLA loop runs all streams in lock-step (one batched
// evaluation per iteration), its state on the device (stream_cobyla).
int stream_run(vqe_t* h, int which, BatchArgs& A) {
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  int rc = 0;
  if (which == 0) {
    rc = stream_evaluate(h->sw, A, A.ham.n_terms, h->stream, h->noise.eval_base, true, h->err, true, h->gen);
  } else if (which == 4) {   // Pauli-term reduction only, on the states of the previous run
    if (h->sw.states_cap < ((size_t)h->batch << h->n)) return fail(h, VQE_ESTATE, "no states: run the energy first");
    rc = stream_evaluate(h->sw, A, A.ham.n_terms, h->stream, h->noise.eval_base, true, h->err, false, h->gen);
  } else if (which == 2) {
    rc = stream_evaluate(h->sw, A, A.ham.n_terms, h->stream, h->noise.eval_base, false, h->err, true, h->gen);
    if (!rc) {
      const size_t dim = (size_t)1 << h->n;
      hipLaunchKernelGGL(k_s_state_out, dim3((unsigned)(dim / kThreads)), dim3(kThreads), 0, h->stream, A,
                         h->sw.states, h->sw.masks, h->sw.meta);
      HIP_TRY(h, hipGetLastError());
    }
  } else if (which == 1) {
    const int B = h->batch;
    std::vector<double> x(h->h_theta), f(B, 0.0);
    std::vector<int32_t> nfev(B);
    rc = stream_cobyla(h, A, h->h_par_begin, h->h_par_count, x, f, nfev, h->d_par_begin.p, h->d_par_count.p);
    if (rc) return rc;
    if (h->total_params)
      HIP_TRY(h, hipMemcpyAsync(h->d_x.p, x.data(), x.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_f.p, f.data(), (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_nfev.p, nfev.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else {
    // which == 3: one CircuitEnv.step() per stream on the streaming path
    // (environment_qulacs_TN_notin_agent.py:283-291): host-driven COBYLA on the PRE-action circuits (the new
    // gate and the noise channel attached to it left out, its angle not a variable), float32 round trip,
    // then the full circuits.
    const int B = h->batch;
    std::vector<GateRec> g2;
    std::vector<int64_t> gbeg2(B), pbeg2(B);
    std::vector<int32_t> gcnt2(B), pcnt2(B), hole(B, -1);
    std::vector<double> x0;
    g2.reserve(h->h_gates.size());
    for (int b = 0; b < B; ++b) {
      const int64_t g0 = h->h_gate_begin[b];
      const int G = h->h_gate_count[b];
      const int skip = h->has_new_gate ? h->h_new_gate[b] : -1;
      int skip_end = skip + 1;
      if (skip >= 0) {
        const GateRec r = h->h_gates[g0 + skip];
        if (r.kind >= G_RX && r.kind <= G_RZ) hole[b] = r.pidx;
        if (skip + 1 < G) {
          const GateRec fo = h->h_gates[g0 + skip + 1];
          if ((fo.kind == G_DEPOL1 && r.kind >= G_RX && r.kind <= G_RZ && fo.q0 == r.q0) ||
              (fo.kind == G_DEPOL2 && r.kind == G_CNOT && fo.q0 == r.q0 && fo.q1 == r.q1))
            skip_end = skip + 2;
        }
      }
      gbeg2[b] = (int64_t)g2.size();
      pbeg2[b] = (int64_t)x0.size();
      for (int i = 0; i < G; ++i) {
        if (i >= skip && i < skip_end) continue;
        GateRec r = h->h_gates[g0 + i];
        if (r.kind >= G_RX && r.kind <= G_RZ && hole[b] >= 0 && r.pidx > hole[b]) r.pidx -= 1;
        g2.push_back(r);
      }
      gcnt2[b] = (int32_t)((int64_t)g2.size() - gbeg2[b]);
      for (int j = 0; j < h->h_par_count[b]; ++j)
        if (j != hole[b]) x0.push_back(h->h_theta[h->h_par_begin[b] + j]);
      pcnt2[b] = (int32_t)((int64_t)x0.size() - pbeg2[b]);
    }
    int rc2;
    ++h->gen;      // d_gates2 changes content: plans made for it are stale
    if ((rc2 = upload(h, h->d_gates2, g2.data(), g2.size()))) return rc2;
    if ((rc2 = upload(h, h->d_gate_begin2, gbeg2.data(), gbeg2.size()))) return rc2;
    if ((rc2 = upload(h, h->d_gate_count2, gcnt2.data(), gcnt2.size()))) return rc2;
    if ((rc2 = upload(h, h->d_par_begin2, pbeg2.data(), pbeg2.size()))) return rc2;
    if ((rc2 = upload(h, h->d_par_count2, pcnt2.data(), pcnt2.size()))) return rc2;
    BatchArgs A2 = A;
    A2.gates = h->d_gates2.p; A2.gate_begin = h->d_gate_begin2.p; A2.gate_count = h->d_gate_count2.p;
    A2.par_begin = h->d_par_begin2.p; A2.par_count = h->d_par_count2.p;
    std::vector<double> f(B, 0.0);
    std::vector<int32_t> nfev(B);
    rc = stream_cobyla(h, A2, pbeg2, pcnt2, x0, f, nfev, h->d_par_begin2.p, h->d_par_count2.p);
    if (rc) return rc;
    std::vector<double> xraw(h->h_theta), xr32(h->h_theta);
    for (int b = 0; b < B; ++b) {
      int k = 0;
      for (int j = 0; j < h->h_par_count[b]; ++j) {
        const double v = j == hole[b] ? h->h_theta[h->h_par_begin[b] + j] : x0[pbeg2[b] + k++];
        xraw[h->h_par_begin[b] + j] = v;
        xr32[h->h_par_begin[b] + j] = (double)(float)v;
      }
    }
    if (h->total_params) {
      HIP_TRY(h, hipMemcpyAsync(h->d_x.p, xr32.data(), xr32.size() * 8, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipMemcpyAsync(h->d_xraw.p, xraw.data(), xraw.size() * 8, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_nfev.p, nfev.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));      // host vectors stay alive until the copies are done
    A.theta = h->d_x.p;
    rc = stream_evaluate(h->sw, A, A.ham.n_terms, h->stream, h->noise.eval_base + (uint64_t)A.maxfun + 1, true, h->err, true, h->gen);
  }
  if (rc) return rc;
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  return VQE_OK;
}

// ---- exact channel mode (vqe_dm.h) -----------------------------------------------------------------
typedef std::complex<double> cplx;
struct Sup { cplx m[16][16]; };      // superoperator on the window: entry index e = i + 4 j, i = ket bits (a, b), j = bra bits

void sup_identity(Sup& S) {
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) S.m[r][c] = r == c ? 1.0 : 0.0;
}
// rho -> U rho U^+ :  S[(i, j), (i', j')] = U[i][i'] conj(U[j][j'])
void sup_conj(const cplx U[4][4], Sup& S) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int ip = 0; ip < 4; ++ip) for (int jp = 0; jp < 4; ++jp)
    S.m[i + 4 * j][ip + 4 * jp] = U[i][ip] * std::conj(U[j][jp]);
}
void sup_apply(Sup& acc, const Sup& G) {      // acc <- G acc
  Sup t;
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
    cplx v = 0.0;
    for (int k = 0; k < 16; ++k) v += G.m[r][k] * acc.m[k][c];
    t.m[r][c] = v;
  }
  acc = t;
}
// one-qubit operator on window position pos (0: qubit a = bit 0 of the 2-bit index, 1: qubit b)
void embed_1q(const cplx R[2][2], int pos, cplx U[4][4]) {
  for (int i = 0; i < 4; ++i) for (int ip = 0; ip < 4; ++ip) {
    const int other = pos ^ 1;
    U[i][ip] = (((i >> other) & 1) == ((ip >> other) & 1)) ? R[(i >> pos) & 1][(ip >> pos) & 1] : cplx(0.0);
  }
}
void pauli_1q(int p, cplx R[2][2]) {          // 0 I, 1 X, 2 Y, 3 Z
  R[0][0] = R[0][1] = R[1][0] = R[1][1] = 0.0;
  if (p == 0) { R[0][0] = R[1][1] = 1.0; }
  else if (p == 1) { R[0][1] = R[1][0] = 1.0; }
  else if (p == 2) { R[0][1] = cplx(0.0, -1.0); R[1][0] = cplx(0.0, 1.0); }
  else { R[0][0] = 1.0; R[1][1] = -1.0; }
}
// (1 - p) id + p / (4^k - 1) sum over the non-identity Paulis on the qubits of `mask` (bit 0: a, bit 1: b)
void sup_depol(int mask, double p, Sup& S) {
  const int k = (mask & 1) + ((mask >> 1) & 1);
  const double w = p / (k == 2 ? 15.0 : 3.0);
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) S.m[r][c] = r == c ? 1.0 - p : 0.0;
  for (int pa = 0; pa < 4; ++pa) for (int pb = 0; pb < 4; ++pb) {
    if ((pa && !(mask & 1)) || (pb && !(mask & 2)) || (!pa && !pb)) continue;
    cplx Ra[2][2], Rb[2][2], Ua[4][4], Ub[4][4], U[4][4];
    pauli_1q(pa, Ra); pauli_1q(pb, Rb);
    embed_1q(Ra, 0, Ua); embed_1q(Rb, 1, Ub);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      cplx v = 0.0;
      for (int m = 0; m < 4; ++m) v += Ua[i][m] * Ub[m][j];
      U[i][j] = v;
    }
    Sup P;
    sup_conj(U, P);
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) S.m[r][c] += w * P.m[r][c];
  }
}

struct DmBlockHost { int a, b; Sup S; };

// Gate list -> superoperator blocks.  A block collects the gates / channels that stay inside its two-qubit window; blocks
// on DISJOINT windows commute (they act on different index bits of rho), so several blocks are open at a time and a
// gate joins the open block that holds all of its qubits wherever that block was opened; a gate that touches an open
// window without fitting into it closes that block first (blocks are emitted in the order they are closed, which keeps
// every qubit's own sequence of operations intact).  Gate semantics as in vqe_device.h (qulacs: R = exp(+i theta/2 P),
// CNOT(control, target)).  Bench circuits (63 gates + 63 channels on 12 qubits): 57 blocks with consecutive fusion
// only, ~40 with this one.
void dm_make_blocks(int n, const GateRec* g, int G, const double* theta, double p1, double p2, std::vector<DmBlockHost>& out) {
  out.clear();
  std::vector<DmBlockHost> open;            // pairwise disjoint windows
  auto owner = [&](int q) { for (size_t k = 0; k < open.size(); ++k) if (open[k].a == q || open[k].b == q) return (int)k; return -1; };
  auto close = [&](int k) { out.push_back(open[k]); open.erase(open.begin() + k); };
  for (int i = 0; i < G; ++i) {
    const GateRec r = g[i];
    const bool two = r.kind == G_CNOT || r.kind == G_DEPOL2;
    const int qa = r.q0, qb = two ? r.q1 : -1;
    int k = owner(qa);
    const int k2 = two ? owner(qb) : k;
    if (k < 0 || k2 != k) {
      // no open block holds all qubits of the gate: close the ones it touches (the higher index first), open a new one
      const int c1 = k, c2 = two ? k2 : -1;
      if (c1 >= 0 && c2 >= 0 && c1 != c2) { close(std::max(c1, c2)); close(std::min(c1, c2)); }
      else if (c1 >= 0) close(c1);
      else if (c2 >= 0) close(c2);
      DmBlockHost nb{};
      nb.a = qa;
      nb.b = qb;
      if (nb.b < 0) {
        // a one-qubit gate opens the window: its partner is the other qubit of the next two-qubit gate that touches it,
        // if that qubit is free; else any free qubit; if every other qubit sits in an open window, the oldest block goes
        int want = -1;
        for (int j = i + 1; j < G && want < 0; ++j) {
          const bool t2 = g[j].kind == G_CNOT || g[j].kind == G_DEPOL2;
          if (t2 && g[j].q0 == qa) want = g[j].q1;
          else if (t2 && g[j].q1 == qa) want = g[j].q0;
        }
        if (want >= 0 && owner(want) < 0) nb.b = want;
        for (int q = 0; q < n && nb.b < 0; ++q) if (q != qa && owner(q) < 0) nb.b = q;
        if (nb.b < 0) { nb.b = open[0].a; close(0); }
      }
      sup_identity(nb.S);
      open.push_back(nb);
      k = (int)open.size() - 1;
    }
    DmBlockHost& cur = open[k];
    const int wa = cur.a;
    Sup Gs;
    if (r.kind == G_CNOT) {
      const int pc = r.q0 == wa ? 0 : 1, pt = pc ^ 1;
      cplx U[4][4];
      for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) U[x][y] = (x == (y ^ (((y >> pc) & 1) << pt))) ? 1.0 : 0.0;
      sup_conj(U, Gs);
    } else if (r.kind >= G_RX && r.kind <= G_RZ) {
      const double c = std::cos(0.5 * theta[r.pidx]), sn = std::sin(0.5 * theta[r.pidx]);
      cplx R[2][2], U[4][4];
      if (r.kind == G_RX) { R[0][0] = R[1][1] = c; R[0][1] = R[1][0] = cplx(0.0, sn); }
      else if (r.kind == G_RY) { R[0][0] = R[1][1] = c; R[0][1] = sn; R[1][0] = -sn; }
      else { R[0][0] = cplx(c, sn); R[1][1] = cplx(c, -sn); R[0][1] = R[1][0] = 0.0; }
      embed_1q(R, r.q0 == wa ? 0 : 1, U);
      sup_conj(U, Gs);
    } else if (r.kind == G_DEPOL1) {
      sup_depol(r.q0 == wa ? 1 : 2, p1, Gs);
    } else {
      sup_depol(3, p2, Gs);
    }
    sup_apply(cur.S, Gs);
  }
  while (!open.empty()) close(0);
}

// this handle's Hamiltonian terms (all of its share under term sharding) for k_dm_energy
int dm_prepare_ham(vqe_t* h) {
  if (h->dm_ham_gen == h->gen) return VQE_OK;
  const std::vector<int> owner = assign_groups(h->gx_all, h->group_terms, h->lds_path, h->shard_world);
  std::vector<uint32_t> gx, tz;
  std::vector<int32_t> toff{0};
  std::vector<double> cr, ci;
  for (size_t g = 0; g < owner.size(); ++g) {
    if (owner[g] != h->shard_rank) continue;
    gx.push_back(h->gx_all[g]);
    for (int k : h->group_terms[g]) { tz.push_back((uint32_t)h->hz[k]); cr.push_back(h->hcr[k]); ci.push_back(h->hci[k]); }
    toff.push_back((int32_t)tz.size());
  }
  int rc;
  if ((rc = upload(h, h->dm_gx, gx.data(), gx.size()))) return rc;
  if ((rc = upload(h, h->dm_tz, tz.data(), tz.size()))) return rc;
  if ((rc = upload(h, h->dm_toff, toff.data(), toff.size()))) return rc;
  if ((rc = upload(h, h->dm_cr, cr.data(), cr.size()))) return rc;
  if ((rc = upload(h, h->dm_ci, ci.data(), ci.size()))) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->dm_groups = (int)gx.size();
  h->dm_ham_gen = h->gen;
  return VQE_OK;
}

// tr(rho H) of one circuit into *e_host (blocking)
int dm_energy_one(vqe_t* h, const GateRec* g, int G, const double* theta, double* e_host) {
  const int n = h->n;
  std::vector<DmBlockHost> blocks;
  dm_make_blocks(n, g, G, theta, h->noise.p1, h->noise.p2, blocks);
  std::vector<double> S(blocks.size() * 512);
  for (size_t k = 0; k < blocks.size(); ++k)
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      S[k * 512 + r * 16 + c] = blocks[k].S.m[r][c].real();
      S[k * 512 + 256 + r * 16 + c] = blocks[k].S.m[r][c].imag();
    }
  const size_t total = (size_t)1 << (2 * n);
  const int eb = (int)((((size_t)1 << n) + 255) / 256);
  HIP_TRY(h, h->dm_rho.reserve(total));
  HIP_TRY(h, h->dm_partial.reserve((size_t)eb + 1));
  int rc;
  if ((rc = upload(h, h->dm_S, S.data(), S.size()))) return rc;
  const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)h->cu_count * 16);
  if (!h->dm_ev0) { HIP_TRY(h, hipEventCreate(&h->dm_ev0)); HIP_TRY(h, hipEventCreate(&h->dm_ev1)); }
  HIP_TRY(h, hipEventRecord(h->dm_ev0, h->stream));
  hipLaunchKernelGGL(k_dm_init, dim3(grid), dim3(256), 0, h->stream, h->dm_rho.p, (const double2*)h->init.p, n);
  for (size_t k = 0; k < blocks.size(); ++k) {
    DmBlockArgs A{};
    A.rho = h->dm_rho.p; A.S = h->dm_S.p + k * 512; A.n = n; A.a = blocks[k].a; A.b = blocks[k].b;
    int hb[4] = {A.a, A.b, A.a + n, A.b + n};
    std::sort(hb, hb + 4);
    for (int i = 0; i < 4; ++i) A.hole[i] = hb[i];
    A.n_groups = (uint32_t)(total / 16);
    const size_t tiles = (A.n_groups + 15) / 16;
    const unsigned gb = (unsigned)std::max<size_t>(1, std::min<size_t>((tiles + 3) / 4, (size_t)h->cu_count * 8));
    hipLaunchKernelGGL(k_dm_block, dim3(gb), dim3(256), 0, h->stream, A);
  }
  hipLaunchKernelGGL(k_dm_energy, dim3(eb), dim3(256), 0, h->stream, (const double2*)h->dm_rho.p, n, h->dm_groups,
                     (const uint32_t*)h->dm_gx.p, (const int32_t*)h->dm_toff.p, (const uint32_t*)h->dm_tz.p,
                     (const double*)h->dm_cr.p, (const double*)h->dm_ci.p, h->dm_partial.p);
  hipLaunchKernelGGL(k_dm_sum, dim3(1), dim3(64), 0, h->stream, (const double*)h->dm_partial.p, eb, h->dm_partial.p + eb);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(h->dm_ev1, h->stream));
  HIP_TRY(h, hipMemcpyAsync(e_host, h->dm_partial.p + eb, 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));      // (S and the blocks go out of scope)
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->dm_ev0, h->dm_ev1));
  h->dm_gpu_ms += ms;
  h->dm_blocks_last = (int)blocks.size();
  return VQE_OK;
}

// The resident batch in exact channel mode: which = 0 energies, 1 COBYLA (host-driven on exact energies; with
// A.env_step the pre-action circuit, float32 round trip and the energy of the full circuit, as the fused kernel does).
int dm_run(vqe_t* h, int which, const BatchArgs& A) {
  if (h->n < 2 || h->n > 13) return fail(h, VQE_EINVAL, "the exact channel mode (density matrix) serves 2 <= n_qubits <= 13");
  if (h->amp_world > 1) return fail(h, VQE_ESTATE, "the exact channel mode has no amplitude sharding");
  int rc = dm_prepare_ham(h);
  if (rc) return rc;
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  h->dm_gpu_ms = 0.f;
  h->last_run_dm = true;
  const int B = h->batch;
  std::vector<double> f(B, 0.0), x(h->h_theta), xraw(h->h_theta);
  std::vector<int32_t> nfev(B, 1);
  for (int b = 0; b < B; ++b) {
    const GateRec* g = h->h_gates.data() + h->h_gate_begin[b];
    const int G = h->h_gate_count[b], P = h->h_par_count[b];
    double* xb = x.data() + h->h_par_begin[b];
    if (which == 0) {
      if ((rc = dm_energy_one(h, g, G, xb, &f[b]))) return rc;
      continue;
    }
    // the circuit COBYLA sees: without the new gate (and the channel attached to it) when this is an env-step
    const int skip = (A.env_step && h->has_new_gate) ? h->h_new_gate[b] : -1;
    int skip_end = skip + 1, hole = -1;
    if (skip >= 0) {
      const GateRec r = g[skip];
      if (r.kind >= G_RX && r.kind <= G_RZ) hole = r.pidx;
      if (skip + 1 < G) {
        const GateRec fo = g[skip + 1];
        if ((fo.kind == G_DEPOL1 && r.kind >= G_RX && r.kind <= G_RZ && fo.q0 == r.q0) ||
            (fo.kind == G_DEPOL2 && r.kind == G_CNOT && fo.q0 == r.q0 && fo.q1 == r.q1))
          skip_end = skip + 2;
      }
    }
    std::vector<GateRec> g2;
    for (int i = 0; i < G; ++i) {
      if (i >= skip && i < skip_end) continue;
      GateRec r = g[i];
      if (r.kind >= G_RX && r.kind <= G_RZ && hole >= 0 && r.pidx > hole) r.pidx -= 1;
      g2.push_back(r);
    }
    std::vector<double> xo;
    for (int j = 0; j < P; ++j) if (j != hole) xo.push_back(xb[j]);
    double fo = 0.0;
    if (xo.empty()) {      // scipy returns after one evaluation
      if ((rc = dm_energy_one(h, g2.data(), (int)g2.size(), xo.data(), &fo))) return rc;
    } else {
      vqe_cobyla_t* cob = nullptr;
      if (vqe_cobyla_create((int)xo.size(), xo.data(), A.rhobeg, A.rhoend, A.maxfun, &cob)) return fail(h, VQE_ENOMEM, "host COBYLA allocation failed");
      std::vector<double> xt(xo.size());
      while (vqe_cobyla_ask(cob, xt.data()) == 1) {
        double e;
        if ((rc = dm_energy_one(h, g2.data(), (int)g2.size(), xt.data(), &e))) { vqe_cobyla_destroy(cob); return rc; }
        vqe_cobyla_tell(cob, e);
      }
      vqe_cobyla_result(cob, xo.data(), &fo, &nfev[b], nullptr);
      vqe_cobyla_destroy(cob);
    }
    double* xr = xraw.data() + h->h_par_begin[b];
    for (int j = 0, k = 0; j < P; ++j) {
      const double v = j == hole ? xb[j] : xo[k++];
      xr[j] = v;
      xb[j] = A.env_step ? (double)(float)v : v;
    }
    f[b] = fo;
    if (A.env_step && (rc = dm_energy_one(h, g, G, xb, &f[b]))) return rc;
  }
  if (which != 0 && h->total_params) {
    HIP_TRY(h, hipMemcpyAsync(h->d_x.p, x.data(), x.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_xraw.p, xraw.data(), xraw.size() * 8, hipMemcpyHostToDevice, h->stream));
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_f.p, f.data(), (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_nfev.p, nfev.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int run(vqe_t* h, int which, double rhobeg, double rhoend, int maxfun) {
  HIP_TRY(h, hipSetDevice(h->dev));
  BatchArgs A = make_args(h);
  A.rhobeg = rhobeg; A.rhoend = rhoend; A.maxfun = maxfun;
  int rc;
  if (which == 3) {
    which = 1;
    A.env_step = 1;
    A.new_gate = h->has_new_gate ? h->d_new_gate.p : nullptr;
    if (!h->lds_path) which = 3;      // streaming path: host-driven loop in stream_run
  }
  if ((which == 1 || which == 3) && (h->shard_world > 1 || h->amp_world > 1))
    return fail(h, VQE_ESTATE, "term-sharded handles hold partial energies: drive COBYLA with "
                               "vqe_cobyla_ask/tell and sum the partial energies of all ranks");
  if (which == 4 && h->lds_path)
    return fail(h, VQE_ESTATE, "the reduction-only launch exists on the streaming path (n >= 14) only");
  if (h->trace_on && which == 1 && h->lds_path) {
    const size_t stride = (size_t)1 + (size_t)h->max_params, words = (size_t)h->batch * (size_t)maxfun * stride;
    HIP_TRY(h, h->d_trace.reserve(words));
    HIP_TRY(h, hipMemsetAsync(h->d_trace.p, 0, words * sizeof(double), h->stream));
    A.trace = h->d_trace.p;
    h->trace_maxfun = maxfun; h->trace_stride = (int)stride; h->trace_batch = h->batch;
  }
  h->last_run_dm = false;
  if (h->noise_mode == 1 && (which == 0 || which == 1 || which == 3)) return dm_run(h, which == 0 ? 0 : 1, A);
  if (h->lds_path) rc = dispatch_lds(h, which, A);
  else rc = stream_run(h, which, A);
  if (rc) return rc;
  // every evaluation of a stochastic run consumes fresh trajectory numbers
  if (which != 4) h->noise.eval_base += ((which == 1 || which == 3) ? (uint64_t)maxfun + 1 : 1);
  return VQE_OK;
}

}  // namespace

// =========================================================================================
extern "C" {

int vqe_create(int n_qubits, int device_id, vqe_t** out) {
  if (!out) return fail(nullptr, VQE_EINVAL, "out is NULL");
  *out = nullptr;
  if (n_qubits < 1 || n_qubits > 30) return fail(nullptr, VQE_EINVAL, "n_qubits must be in [1, 30]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, VQE_ENODEV, "no HIP device available (the VQE engine has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, VQE_EINVAL, "device_id out of range");
  vqe_t* h = new (std::nothrow) vqe_t;
  if (!h) return fail(nullptr, VQE_ENOMEM, "out of host memory");
  h->n = n_qubits;
  h->dev = device_id;
  h->lds_path = n_qubits <= 13;
  hipDeviceProp_t prop;
  if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    delete h;
    return fail(nullptr, VQE_EHIP, "HIP device initialisation failed");
  }
  h->stream = h->own_stream;
  h->cu_count = prop.multiProcessorCount;
  h->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (h->lds_per_cu <= 0) h->lds_per_cu = 65536;
  if (h->d_dbg.reserve(8) != hipSuccess || hipMemset(h->d_dbg.p, 0, 64) != hipSuccess) {
    delete h;
    return fail(nullptr, VQE_ENOMEM, "device allocation failed");
  }
  *out = h;
  int rc = vqe_set_init_state(h, nullptr);
  if (rc) { g_create_error = h->err; vqe_destroy(h); *out = nullptr; return rc; }
  return VQE_OK;
}

void vqe_destroy(vqe_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->dev);
  (void)hipStreamSynchronize(h->stream);
  if (h->comm) (void)vqe_comm_destroy(h);
  if (h->dm_ev0) (void)hipEventDestroy(h->dm_ev0);
  if (h->dm_ev1) (void)hipEventDestroy(h->dm_ev1);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

const char* vqe_last_error(const vqe_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int vqe_set_stream(vqe_t* h, void* s) {
  if (!h) return VQE_EINVAL;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->stream = s ? (hipStream_t)s : h->own_stream;
  return VQE_OK;
}

int vqe_get_stream(vqe_t* h, void** hip_stream) {
  if (!h || !hip_stream) return VQE_EINVAL;
  *hip_stream = (void*)h->stream;
  return VQE_OK;
}

int vqe_sync(vqe_t* h) {
  if (!h) return VQE_EINVAL;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_device_info(vqe_t* h, int64_t info[4]) {
  if (!h || !info) return VQE_EINVAL;
  info[0] = h->cu_count; info[1] = h->lds_per_cu; info[2] = h->last_wg_per_cu; info[3] = h->lds_path;
  return VQE_OK;
}

int vqe_set_init_state(vqe_t* h, const double* amps) {
  if (!h) return VQE_EINVAL;
  HIP_TRY(h, hipSetDevice(h->dev));
  const size_t dim = (size_t)1 << h->n;
  HIP_TRY(h, h->init.reserve(dim));
  if (amps) {
    HIP_TRY(h, hipMemcpyAsync(h->init.p, amps, dim * 16, hipMemcpyHostToDevice, h->stream));
  } else {
    const double one[2] = {1.0, 0.0};
    HIP_TRY(h, hipMemsetAsync(h->init.p, 0, dim * 16, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->init.p, one, 16, hipMemcpyHostToDevice, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_set_init_state_dev(vqe_t* h, const void* dev_amps) {
  if (!h || !dev_amps) return VQE_EINVAL;
  HIP_TRY(h, hipSetDevice(h->dev));
  const size_t dim = (size_t)1 << h->n;
  HIP_TRY(h, h->init.reserve(dim));
  HIP_TRY(h, hipMemcpyAsync(h->init.p, dev_amps, dim * 16, hipMemcpyDeviceToDevice, h->stream));
  return VQE_OK;
}

int vqe_get_state_dev(vqe_t* h, const double* theta, void* dev_amps) {
  if (!h) return VQE_EINVAL;
  if (!dev_amps || (h->circ_params > 0 && !theta)) return fail(h, VQE_EINVAL, "bad arguments");
  HIP_TRY(h, hipSetDevice(h->dev));
  int rc = load_single(h, 1, theta);
  if (rc) return rc;
  const size_t dim = (size_t)1 << h->n;
  HIP_TRY(h, h->d_state.reserve(dim));
  if ((rc = run(h, 2, 0, 0, 0))) return rc;
  HIP_TRY(h, hipMemcpyAsync(dev_amps, h->d_state.p, dim * 16, hipMemcpyDeviceToDevice, h->stream));
  return VQE_OK;
}

int vqe_set_hamiltonian_pauli(vqe_t* h, int n_terms, const uint64_t* xmask, const uint64_t* zmask,
                              const double* coeff) {
  if (!h) return VQE_EINVAL;
  if (n_terms < 0 || (n_terms > 0 && (!xmask || !zmask || !coeff)))
    return fail(h, VQE_EINVAL, "bad Hamiltonian arguments");
  HIP_TRY(h, hipSetDevice(h->dev));
  const uint64_t lim = h->n >= 64 ? ~0ull : (((uint64_t)1 << h->n) - 1);
  h->hx.assign(xmask, xmask + n_terms);
  h->hz.assign(zmask, zmask + n_terms);
  h->hcr.assign(n_terms, 0.0);
  h->hci.assign(n_terms, 0.0);
  std::map<uint32_t, int> index;
  h->gx_all.clear();
  h->group_terms.clear();
  for (int k = 0; k < n_terms; ++k) {
    if ((xmask[k] | zmask[k]) & ~lim) return fail(h, VQE_EINVAL, "Pauli mask uses a qubit >= n_qubits");
    const int ny = __builtin_popcountll(xmask[k] & zmask[k]) & 3;  // i^{#Y}
    const double w = coeff[k];
    h->hcr[k] = ny == 0 ? w : (ny == 2 ? -w : 0.0);
    h->hci[k] = ny == 1 ? w : (ny == 3 ? -w : 0.0);
    const uint32_t x = (uint32_t)xmask[k];
    auto it = index.find(x);
    if (it == index.end()) {
      it = index.emplace(x, (int)h->gx_all.size()).first;
      h->gx_all.push_back(x);
      h->group_terms.emplace_back();
    }
    h->group_terms[it->second].push_back(k);
  }
  h->ham_set = true;
  return build_hamiltonian(h);
}

int vqe_set_hamiltonian_dense(vqe_t* h, const double* op_re_im, double tol) {
  if (!h) return VQE_EINVAL;
  if (!op_re_im || !(tol >= 0.0)) return fail(h, VQE_EINVAL, "bad Hamiltonian arguments");
  if (h->n > 13) return fail(h, VQE_EINVAL, "a dense 2^n x 2^n operator is accepted up to 13 qubits (17 TB at n = 20): use vqe_set_hamiltonian_pauli");
  // H = sum_{x,z} c(x,z) P(x,z), P|i> = i^{#Y} (-1)^{popc(i & z)} |i ^ x>  =>  for fixed x the coefficients are the
  // Walsh-Hadamard transform over i of the generalised diagonal H[i ^ x, i], divided by 2^n i^{#Y}
  const size_t dim = (size_t)1 << h->n;
  std::vector<uint64_t> xs, zs;
  std::vector<double> cs;
  std::vector<double> re(dim), im(dim);
  double scale = 0.0;
  for (size_t k = 0; k < 2 * dim * dim; ++k) scale = std::max(scale, std::fabs(op_re_im[k]));
  const double cut = tol * (scale > 0 ? scale : 1.0);
  for (size_t x = 0; x < dim; ++x) {
    bool any = false;
    for (size_t i = 0; i < dim; ++i) {
      const double* e = op_re_im + 2 * ((i ^ x) * dim + i);
      re[i] = e[0]; im[i] = e[1];
      any = any || e[0] != 0.0 || e[1] != 0.0;
    }
    if (!any) continue;
    for (size_t step = 1; step < dim; step <<= 1)
      for (size_t i = 0; i < dim; i += 2 * step)
        for (size_t j = i; j < i + step; ++j) {
          const double ar = re[j], ai = im[j], br = re[j + step], bi = im[j + step];
          re[j] = ar + br; im[j] = ai + bi; re[j + step] = ar - br; im[j + step] = ai - bi;
        }
    for (size_t z = 0; z < dim; ++z) {
      double cr = re[z] / (double)dim, ci = im[z] / (double)dim;
      const int ny = __builtin_popcountll(x & z) & 3;      // divide by i^ny
      double vr, vi;
      if (ny == 0) { vr = cr; vi = ci; } else if (ny == 1) { vr = ci; vi = -cr; } else if (ny == 2) { vr = -cr; vi = -ci; } else { vr = -ci; vi = cr; }
      if (std::fabs(vr) <= cut && std::fabs(vi) <= cut) continue;
      if (std::fabs(vi) > 1e-9 * (scale > 0 ? scale : 1.0) && std::fabs(vi) > cut)
        return fail(h, VQE_EINVAL, "operator is not Hermitian (complex Pauli coefficient)");
      xs.push_back(x); zs.push_back(z); cs.push_back(vr);
    }
  }
  return vqe_set_hamiltonian_pauli(h, (int)cs.size(), xs.data(), zs.data(), cs.data());
}

int vqe_hamiltonian_terms(vqe_t* h, int32_t* n_terms, int32_t* n_xgroups) {
  if (!h || !n_terms || !n_xgroups) return VQE_EINVAL;
  if (!h->ham_set) return fail(h, VQE_ESTATE, "no Hamiltonian set");
  *n_terms = (int32_t)h->hx.size();
  *n_xgroups = (int32_t)h->gx_all.size();
  return VQE_OK;
}

// ---- RCCL behind the C ABI ----------------------------------------------------------------------------------------
// The collective of the term-sharded expectation sum as a library call: one ncclAllReduce(SUM, float64, count = batch) of
// the handle's energy array, in place, on the handle's stream.  librccl is opened lazily (no link-time dependency: a
// process that never shards never loads it; under PyTorch the already-loaded copy of the same SONAME is reused).
extern "C++" {      // (helpers with C++ types inside the C-ABI block)
namespace {
struct Id128 { char b[128]; };      // ncclUniqueId, passed by value
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  if (!r.lib) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
      r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    }
  }
  return r;
}
}  // namespace
}  // extern "C++"

int vqe_comm_unique_id(void* id128) {
  if (!id128) return VQE_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return fail(nullptr, VQE_ENODEV, "librccl not found");
  const int rc = r.GetUniqueId(id128);
  return rc ? fail(nullptr, VQE_EHIP, std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error")) : VQE_OK;
}

int vqe_comm_init(vqe_t* h, int rank, int world, const void* id128) {
  if (!h || !id128) return VQE_EINVAL;
  if (world < 1 || rank < 0 || rank >= world) return fail(h, VQE_EINVAL, "bad rank / world");
  if (h->comm) return fail(h, VQE_ESTATE, "communicator exists already (vqe_comm_destroy first)");
  Rccl& r = rccl();
  if (!r.ok) return fail(h, VQE_ENODEV, "librccl not found");
  HIP_TRY(h, hipSetDevice(h->dev));
  Id128 id;
  std::memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, world, id, rank);
  if (rc) return fail(h, VQE_EHIP, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
  h->comm = comm;
  h->comm_world = world;
  return VQE_OK;
}

int vqe_comm_allreduce_energy(vqe_t* h) {
  if (!h) return VQE_EINVAL;
  if (!h->comm) return fail(h, VQE_ESTATE, "vqe_comm_init has not been called");
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no batch loaded");
  Rccl& r = rccl();
  const int rc = r.AllReduce(h->d_f.p, h->d_f.p, (size_t)h->batch, /* ncclFloat64 */ 8, /* ncclSum */ 0, h->comm, h->stream);
  return rc ? fail(h, VQE_EHIP, std::string("ncclAllReduce: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error")) : VQE_OK;
}

int vqe_comm_destroy(vqe_t* h) {
  if (!h) return VQE_EINVAL;
  if (h->comm) {
    (void)hipStreamSynchronize(h->stream);
    (void)rccl().CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_world = 0;
  }
  return VQE_OK;
}

int vqe_hamiltonian_layout(vqe_t* h, int32_t out[4]) {
  if (!h || !out) return VQE_EINVAL;
  if (!h->ham_set) return fail(h, VQE_ESTATE, "no Hamiltonian set");
  out[0] = h->ham.n_groups; out[1] = h->ham.n_units; out[2] = h->ham.n_cls; out[3] = h->ham.has_diag;
  return VQE_OK;
}

int vqe_set_term_shard(vqe_t* h, int rank, int world) {
  if (!h) return VQE_EINVAL;
  if (world < 1 || rank < 0 || rank >= world) return fail(h, VQE_EINVAL, "bad shard rank/world");
  h->shard_rank = rank;
  h->shard_world = world;
  if (h->ham_set) { HIP_TRY(h, hipSetDevice(h->dev)); return build_hamiltonian(h); }
  return VQE_OK;
}

int vqe_set_amplitude_shard(vqe_t* h, int rank, int world) {
  if (!h) return VQE_EINVAL;
  if (world < 1 || rank < 0 || rank >= world) return fail(h, VQE_EINVAL, "bad shard rank/world");
  if (h->lds_path && world > 1)
    return fail(h, VQE_ESTATE, "amplitude sharding of the energy sweep exists on the streaming path (n >= 14); use vqe_set_term_shard");
  const size_t blocks = ((size_t)1 << h->n) >> kETileBits;     // tiles of the energy sweep (vqe_tile.h)
  if (!h->lds_path && (blocks % (size_t)world) != 0) return fail(h, VQE_EINVAL, "world must divide the number of sweep tiles of the energy reduction");
  h->amp_rank = rank;
  h->amp_world = world;
  ++h->gen;
  return VQE_OK;
}

int vqe_set_noise(vqe_t* h, double p1, double p2, uint64_t seed) {
  if (!h) return VQE_EINVAL;
  if (!(p1 >= 0.0 && p1 <= 1.0 && p2 >= 0.0 && p2 <= 1.0)) return fail(h, VQE_EINVAL, "noise probability outside [0,1]");
  h->noise = NoiseCfg{p1, p2, seed, 0ull, h->noise.shot_sigma};
  ++h->gen;
  return VQE_OK;
}

int vqe_set_noise_mode(vqe_t* h, int mode) {
  if (!h) return VQE_EINVAL;
  if (mode != 0 && mode != 1) return fail(h, VQE_EINVAL, "noise mode: 0 (Pauli trajectories) or 1 (exact channel)");
  if (mode == 1 && (h->n < 2 || h->n > 13)) return fail(h, VQE_EINVAL, "the exact channel mode (density matrix) serves 2 <= n_qubits <= 13");
  h->noise_mode = mode;
  return VQE_OK;
}

int vqe_dm_plan(int n_qubits, int n_gates, const int32_t* kind, const int32_t* q0, const int32_t* q1, const int32_t* pidx,
                const double* theta, double p1, double p2, int cap_blocks, int32_t* n_blocks, int32_t* windows, double* S) {
  if (n_qubits < 2 || n_gates < 0 || !n_blocks || (n_gates > 0 && (!kind || !q0 || !q1 || !pidx))) return VQE_EINVAL;
  std::vector<GateRec> g((size_t)n_gates);
  for (int i = 0; i < n_gates; ++i) {
    if (kind[i] < 0 || kind[i] > VQE_GATE_DEPOL2 || q0[i] < 0 || q0[i] >= n_qubits) return VQE_EINVAL;
    const bool two = kind[i] == VQE_GATE_CNOT || kind[i] == VQE_GATE_DEPOL2;
    if (two && (q1[i] < 0 || q1[i] >= n_qubits || q1[i] == q0[i])) return VQE_EINVAL;
    if (kind[i] >= VQE_GATE_RX && kind[i] <= VQE_GATE_RZ && (pidx[i] < 0 || !theta)) return VQE_EINVAL;
    g[i] = GateRec{kind[i], q0[i], q1[i], pidx[i]};
  }
  std::vector<DmBlockHost> blocks;
  dm_make_blocks(n_qubits, g.data(), n_gates, theta, p1, p2, blocks);
  *n_blocks = (int32_t)blocks.size();
  if (!windows || !S) return VQE_OK;
  if ((int)blocks.size() > cap_blocks) return VQE_EINVAL;
  for (size_t k = 0; k < blocks.size(); ++k) {
    windows[2 * k] = blocks[k].a;
    windows[2 * k + 1] = blocks[k].b;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      S[k * 512 + r * 16 + c] = blocks[k].S.m[r][c].real();
      S[k * 512 + 256 + r * 16 + c] = blocks[k].S.m[r][c].imag();
    }
  }
  return VQE_OK;
}

int vqe_noise_mode_info(vqe_t* h, int32_t out[2]) {
  if (!h || !out) return VQE_EINVAL;
  out[0] = h->noise_mode;
  out[1] = h->dm_blocks_last;
  return VQE_OK;
}

int vqe_set_shot_noise(vqe_t* h, double sigma_total, uint64_t seed) {
  if (!h) return VQE_EINVAL;
  if (!(sigma_total >= 0.0)) return fail(h, VQE_EINVAL, "sigma_total must be >= 0");
  h->noise.shot_sigma = sigma_total;
  h->noise.seed = seed;
  return VQE_OK;
}

int vqe_set_circuit(vqe_t* h, int n_gates, const int32_t* kind, const int32_t* q0, const int32_t* q1,
                    const int32_t* pidx, int n_params) {
  if (!h) return VQE_EINVAL;
  if (n_gates < 0 || n_params < 0 || (n_gates > 0 && (!kind || !q0 || !q1 || !pidx)))
    return fail(h, VQE_EINVAL, "bad circuit arguments");
  int rc = check_gates(h, n_gates, kind, q0, q1, pidx, n_params);
  if (rc) return rc;
  h->circ.resize(n_gates);
  for (int i = 0; i < n_gates; ++i) h->circ[i] = GateRec{kind[i], q0[i], q1[i], pidx[i]};
  h->circ_params = n_params;
  return VQE_OK;
}

int vqe_energy_batch(vqe_t* h, int batch, const double* theta, double* energy) {
  if (!h) return VQE_EINVAL;
  if (batch < 1 || !energy || (h->circ_params > 0 && !theta)) return fail(h, VQE_EINVAL, "bad arguments");
  HIP_TRY(h, hipSetDevice(h->dev));
  int rc = load_single(h, batch, theta);
  if (rc) return rc;
  if ((rc = ready(h))) return rc;
  if ((rc = run(h, 0, 0, 0, 0))) return rc;
  HIP_TRY(h, hipMemcpyAsync(energy, h->d_f.p, (size_t)batch * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_energy(vqe_t* h, const double* theta, double* energy) { return vqe_energy_batch(h, 1, theta, energy); }

int vqe_get_state(vqe_t* h, const double* theta, double* amps) {
  if (!h) return VQE_EINVAL;
  if (!amps || (h->circ_params > 0 && !theta)) return fail(h, VQE_EINVAL, "bad arguments");
  HIP_TRY(h, hipSetDevice(h->dev));
  int rc = load_single(h, 1, theta);
  if (rc) return rc;
  const size_t dim = (size_t)1 << h->n;
  HIP_TRY(h, h->d_state.reserve(dim));
  if ((rc = run(h, 2, 0, 0, 0))) return rc;
  HIP_TRY(h, hipMemcpyAsync(amps, h->d_state.p, dim * 16, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_minimize_cobyla(vqe_t* h, const double* x0, double rhobeg, double rhoend, int maxfun, double* x,
                        double* f, int32_t* nfev) {
  if (!h) return VQE_EINVAL;
  if (maxfun < 1 || !(rhobeg > 0) || !(rhoend > 0)) return fail(h, VQE_EINVAL, "bad COBYLA arguments");
  if (h->circ_params > 0 && (!x0 || !x)) return fail(h, VQE_EINVAL, "x0/x is NULL");
  HIP_TRY(h, hipSetDevice(h->dev));
  int rc = load_single(h, 1, x0);
  if (rc) return rc;
  if ((rc = ready(h))) return rc;
  if ((rc = run(h, 1, rhobeg, rhoend, maxfun))) return rc;
  return vqe_batch_fetch(h, x, f, nfev);
}

int vqe_batch_load(vqe_t* h, int batch, const int64_t* gate_off, const int32_t* kind, const int32_t* q0,
                   const int32_t* q1, const int32_t* pidx, const int64_t* par_off, const double* theta0) {
  if (!h) return VQE_EINVAL;
  if (batch < 1 || !gate_off || !par_off) return fail(h, VQE_EINVAL, "bad batch arguments");
  HIP_TRY(h, hipSetDevice(h->dev));
  const int64_t G = gate_off[batch], PT = par_off[batch];
  if (gate_off[0] != 0 || par_off[0] != 0 || G < 0 || PT < 0) return fail(h, VQE_EINVAL, "offsets must start at 0");
  if ((G > 0 && (!kind || !q0 || !q1 || !pidx)) || (PT > 0 && !theta0)) return fail(h, VQE_EINVAL, "NULL array");
  std::vector<GateRec> gates((size_t)G);
  std::vector<int64_t> gbeg(batch), pbeg(batch);
  std::vector<int32_t> gcnt(batch), pcnt(batch);
  for (int b = 0; b < batch; ++b) {
    const int64_t g0 = gate_off[b], g1 = gate_off[b + 1], p0 = par_off[b], p1 = par_off[b + 1];
    if (g1 < g0 || p1 < p0 || g1 > G || p1 > PT) return fail(h, VQE_EINVAL, "offsets not monotone");
    int rc = check_gates(h, g1 - g0, kind + g0, q0 + g0, q1 + g0, pidx + g0, (int)(p1 - p0));
    if (rc) return rc;
    gbeg[b] = g0; gcnt[b] = (int32_t)(g1 - g0); pbeg[b] = p0; pcnt[b] = (int32_t)(p1 - p0);
  }
  for (int64_t i = 0; i < G; ++i) gates[i] = GateRec{kind[i], q0[i], q1[i], pidx[i]};
  return load_batch(h, batch, gates, gbeg, gcnt, pbeg, pcnt, theta0, PT);
}

int vqe_batch_run_energy(vqe_t* h) {
  int rc = ready(h);
  if (rc) return rc;
  return run(h, 0, 0, 0, 0);
}

int vqe_batch_run_reduction(vqe_t* h) {
  int rc = ready(h);
  if (rc) return rc;
  return run(h, 4, 0, 0, 0);
}

int vqe_batch_run_minimize(vqe_t* h, double rhobeg, double rhoend, int maxfun) {
  int rc = ready(h);
  if (rc) return rc;
  if (maxfun < 1 || !(rhobeg > 0) || !(rhoend > 0)) return fail(h, VQE_EINVAL, "bad COBYLA arguments");
  return run(h, 1, rhobeg, rhoend, maxfun);
}

int vqe_batch_set_new_gate(vqe_t* h, const int32_t* new_gate) {
  if (!h) return VQE_EINVAL;
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no batch loaded");
  if (!new_gate) { h->has_new_gate = false; return VQE_OK; }
  for (int b = 0; b < h->batch; ++b)
    if (new_gate[b] < -1 || new_gate[b] >= h->h_gate_count[b]) return fail(h, VQE_EINVAL, "new_gate index out of range");
  HIP_TRY(h, hipSetDevice(h->dev));
  int rc = upload(h, h->d_new_gate, new_gate, (size_t)h->batch);
  if (rc) return rc;
  h->h_new_gate.assign(new_gate, new_gate + h->batch);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->has_new_gate = true;
  return VQE_OK;
}

int vqe_batch_run_env_step(vqe_t* h, double rhobeg, double rhoend, int maxfun) {
  int rc = ready(h);
  if (rc) return rc;
  if (maxfun < 1 || !(rhobeg > 0) || !(rhoend > 0)) return fail(h, VQE_EINVAL, "bad COBYLA arguments");
  return run(h, 3, rhobeg, rhoend, maxfun);
}

int vqe_batch_fetch(vqe_t* h, double* x, double* f, int32_t* nfev) {
  if (!h) return VQE_EINVAL;
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no batch loaded");
  HIP_TRY(h, hipSetDevice(h->dev));
  if (x && h->total_params)
    HIP_TRY(h, hipMemcpyAsync(x, h->d_x.p, (size_t)h->total_params * 8, hipMemcpyDeviceToHost, h->stream));
  if (f) HIP_TRY(h, hipMemcpyAsync(f, h->d_f.p, (size_t)h->batch * 8, hipMemcpyDeviceToHost, h->stream));
  if (nfev) HIP_TRY(h, hipMemcpyAsync(nfev, h->d_nfev.p, (size_t)h->batch * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_batch_fetch_xopt(vqe_t* h, double* x) {
  if (!h) return VQE_EINVAL;
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no batch loaded");
  if (!x && h->total_params) return fail(h, VQE_EINVAL, "x is NULL");
  HIP_TRY(h, hipSetDevice(h->dev));
  if (h->total_params)
    HIP_TRY(h, hipMemcpyAsync(x, h->d_xraw.p, (size_t)h->total_params * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_batch_energy_devptr(vqe_t* h, void** p) {
  if (!h || !p) return VQE_EINVAL;
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no batch loaded");
  *p = h->d_f.p;
  return VQE_OK;
}

int vqe_batch_copy_energy(vqe_t* h, void* dst_dev) {
  if (!h || !dst_dev) return VQE_EINVAL;
  if (h->batch <= 0) return fail(h, VQE_ESTATE, "no batch loaded");
  HIP_TRY(h, hipSetDevice(h->dev));
  HIP_TRY(h, hipMemcpyAsync(dst_dev, h->d_f.p, (size_t)h->batch * 8, hipMemcpyDeviceToDevice, h->stream));
  return VQE_OK;
}

int vqe_batch_set_trace(vqe_t* h, int enable) {
  if (!h) return VQE_EINVAL;
  if (enable && !h->lds_path) return fail(h, VQE_ESTATE, "evaluation traces are recorded by the fused device loop (n <= 13) only");
  h->trace_on = enable != 0;
  return VQE_OK;
}

int vqe_batch_fetch_trace(vqe_t* h, int circuit, double* out, int32_t* maxfun, int32_t* stride) {
  if (!h || !maxfun || !stride) return VQE_EINVAL;
  if (h->trace_maxfun <= 0) return fail(h, VQE_ESTATE, "no trace recorded: vqe_batch_set_trace(1), then a minimize / env-step run");
  if (circuit < 0 || circuit >= h->trace_batch) return fail(h, VQE_EINVAL, "circuit index out of range");
  *maxfun = h->trace_maxfun;
  *stride = h->trace_stride;
  if (!out) return VQE_OK;      // size query
  HIP_TRY(h, hipSetDevice(h->dev));
  const size_t words = (size_t)h->trace_maxfun * (size_t)h->trace_stride;
  HIP_TRY(h, hipMemcpyAsync(out, h->d_trace.p + (size_t)circuit * words, words * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VQE_OK;
}

int vqe_debug_counters(vqe_t* h, uint64_t out[8]) {
  if (!h || !out) return VQE_EINVAL;
  HIP_TRY(h, hipSetDevice(h->dev));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, h->d_dbg.p, 64, hipMemcpyDeviceToHost));
#ifdef VQE_STAMPS   // slots 8..15: device-side COBYLA sections
  {
    static unsigned long long zero[8] = {0};
    HIP_TRY(h, hipMemcpyFromSymbol(cby_out_, HIP_SYMBOL(g_cby_dbg), 64));
    HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_cby_dbg), zero, 64));
    std::fprintf(stderr, "cobyla sections:");
    for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %llu", cby_out_[i]);
    unsigned long long io[2];
    HIP_TRY(h, hipMemcpyFromSymbol(io, HIP_SYMBOL(g_cby_in_out), 16));
    HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_cby_in_out), zero, 16));
    std::fprintf(stderr, " | in %llu out %llu\n", io[0], io[1]);
  }
#endif
  HIP_TRY(h, hipMemset(h->d_dbg.p, 0, 64));
  return VQE_OK;
}

int vqe_last_kernel_ms(vqe_t* h, float* ms) {
  if (!h || !ms) return VQE_EINVAL;
  if (h->last_run_dm) { *ms = h->dm_gpu_ms; return VQE_OK; }      // exact channel mode: device time only (the host builds the blocks in between)
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  HIP_TRY(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return VQE_OK;
}

int vqe_term_owner(int n_qubits, int n_terms, const uint64_t* xmask, int world, int32_t* owner) {
  if (n_qubits < 1 || n_qubits > 30 || n_terms < 0 || world < 1 || (n_terms > 0 && (!xmask || !owner))) return VQE_EINVAL;
  std::map<uint32_t, int> index;
  std::vector<uint32_t> gx;
  std::vector<std::vector<int>> terms;
  for (int k = 0; k < n_terms; ++k) {
    const uint32_t x = (uint32_t)xmask[k];
    auto it = index.find(x);
    if (it == index.end()) { it = index.emplace(x, (int)gx.size()).first; gx.push_back(x); terms.emplace_back(); }
    terms[it->second].push_back(k);
  }
  const std::vector<int> go = assign_groups(gx, terms, n_qubits <= 13, world);
  for (size_t g = 0; g < gx.size(); ++g) for (int k : terms[g]) owner[k] = go[g];
  return VQE_OK;
}

// ---- host COBYLA ---------------------------------------------------------------------------
struct vqe_cobyla {
  cby::CobylaM0<cby::HostCtx, true> c;
  std::vector<double> mem;
  int n = 0, want = 0, nfev = 0;
  double flast = 0.0;
  bool finished = false;
};

int vqe_cobyla_create(int n, const double* x0, double rhobeg, double rhoend, int maxfun, vqe_cobyla_t** out) {
  if (!out || n < 0 || (n > 0 && !x0) || maxfun < 1 || !(rhobeg > 0) || !(rhoend > 0)) return VQE_EINVAL;
  vqe_cobyla* c = new (std::nothrow) vqe_cobyla;
  if (!c) return VQE_ENOMEM;
  c->n = n;
  c->mem.assign(cby::scratch_doubles(n, 1) + 8, 0.0);
  c->c.bind(c->mem.data(), n);
  for (int i = 0; i < n; ++i) c->c.x[i] = x0[i];
  if (n == 0) { c->want = 1; c->c.nfvals = 1; c->c.status = cby::RUNNING; }
  else c->want = c->c.start(rhobeg, rhoend, maxfun);
  *out = c;
  return VQE_OK;
}

int vqe_cobyla_ask(vqe_cobyla_t* c, double* x) {
  if (!c) return VQE_EINVAL;
  if (!c->want) return 0;
  if (x) for (int i = 0; i < c->n; ++i) x[i] = c->c.x[i];
  return 1;
}

int vqe_cobyla_tell(vqe_cobyla_t* c, double f) {
  if (!c || !c->want) return VQE_ESTATE;
  c->flast = f;
  if (c->n == 0) { c->want = 0; c->c.status = cby::DONE_RHOEND; c->c.ifull = 1; return 0; }
  c->want = c->c.tell(f);
  return c->want;
}

int vqe_cobyla_result(vqe_cobyla_t* c, double* x, double* f, int32_t* nfev, int32_t* status) {
  if (!c) return VQE_EINVAL;
  if (x) for (int i = 0; i < c->n; ++i) x[i] = c->c.x[i];
  if (f) *f = (c->c.status == cby::DONE_RHOEND && c->c.ifull == 1) ? c->flast : c->c.fbest_ret;
  if (nfev) *nfev = c->c.nfvals;
  if (status) *status = c->c.status;
  return VQE_OK;
}

void vqe_cobyla_destroy(vqe_cobyla_t* c) { delete c; }

}  // extern "C"
