// vqe_stream.h - HBM-streaming path for n >= 14 qubits (state vector does not fit LDS).
//
// Every evaluation stream keeps its 2^n complex128 amplitudes in HBM (16 MiB at n = 20).
// The same affine GF(2) bookkeeping as the LDS path removes every CNOT; each remaining
// rotations are applied kOpsPerSweep at a time per coalesced read-modify-write sweep (32 * 2^n bytes
// per sweep).  The energy kernel
// transforms the Pauli masks into the physical layout instead of permuting the state, reads
// each amplitude once plus one partner amplitude per X-mask group, evaluates the sign sums
// on the fly and reduces per block; a second tiny kernel sums the block partials in a fixed
// order (bitwise reproducible, no atomics).  The COBYLA loop of this path is host driven
// (vqe_cobyla ask/tell) because with Pauli-term sharding every evaluation ends in a
// collective.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include "vqe_device.h"

namespace vqe {

struct StreamWork {
  double2* states = nullptr;  size_t states_cap = 0;   // [batch][2^n]
  Op* ops = nullptr;          size_t ops_cap = 0;      // [batch][max_ops]
  uint32_t* masks = nullptr;  size_t masks_cap = 0;    // [batch][64]: xm[32], zm[32]
  int32_t* meta = nullptr;    size_t meta_cap = 0;     // [batch][8]
  double2* cs = nullptr;      size_t cs_cap = 0;       // [batch][max_params]
  uint32_t* gxp = nullptr;    size_t gxp_cap = 0;      // [batch][n_groups] physical X masks
  uint32_t* tzp = nullptr;    size_t tzp_cap = 0;      // [batch][n_terms] physical Z masks
  double* tsg = nullptr;      size_t tsg_cap = 0;      // [batch][n_terms] (-1)^{z.c}
  double* partial = nullptr;  size_t partial_cap = 0;  // [batch][blocks]
  bool plan_fused = false;    // the energy plan's pass 0 is the tile of the last circuit pass (vqe_tile.h: fused pass)
  // LDS-tiled kernels (vqe_tile.h)
  void* passes = nullptr;     size_t passes_cap = 0;   // TilePass [batch][max_pass]
  void* opc = nullptr;        size_t opc_cap = 0;      // OpCoord [batch][max_ops]
  void* chunks = nullptr;     size_t chunks_cap = 0;   // ChunkRec [batch][max_ops]
  void* egrp = nullptr;       size_t egrp_cap = 0;     // EGroupRec [batch][n_groups]
  void* eterm = nullptr;      size_t eterm_cap = 0;    // ETermRec [batch][n_terms]
  double* ewi = nullptr;      size_t ewi_cap = 0;      // imaginary weights [batch][n_terms]
  double2* csop = nullptr;    size_t csop_cap = 0;     // (cos, sin) in op order [batch][max_ops]
  int32_t* npass = nullptr;   size_t npass_cap = 0;    // [batch] op passes, [batch] energy passes
  void* epasses = nullptr;    size_t epasses_cap = 0;  // TilePass [batch][kMaxEnergyPasses]
  int32_t* eorder = nullptr;  size_t eorder_cap = 0;   // [batch][n_groups] group ids pass by pass, then [batch][n_groups] pass of a group
  uint32_t* gcx = nullptr;    size_t gcx_cap = 0;      // [batch][n_groups]
  // plans depend on the gate lists, the Hamiltonian shard and (through the drawn Paulis) on the noise, not on theta:
  // a noiseless re-evaluation of the same resident batch (COBYLA iterations, repeated runs) keeps them
  const void* plan_src = nullptr; uint64_t plan_gen = ~0ull; bool plan_ops_ok = false, plan_energy_ok = false;
  void* trec = nullptr;       size_t trec_cap = 0;     // TermRec [batch][n_terms]
  int32_t* grec = nullptr;    size_t grec_cap = 0;     // [batch][n_groups]
  ~StreamWork() {
    (void)hipFree(states); (void)hipFree(ops); (void)hipFree(masks); (void)hipFree(meta);
    (void)hipFree(cs); (void)hipFree(gxp); (void)hipFree(tzp); (void)hipFree(tsg); (void)hipFree(partial);
    (void)hipFree(passes); (void)hipFree(opc); (void)hipFree(chunks); (void)hipFree(egrp); (void)hipFree(eterm); (void)hipFree(ewi); (void)hipFree(csop); (void)hipFree(npass); (void)hipFree(epasses); (void)hipFree(eorder);
    (void)hipFree(gcx); (void)hipFree(trec); (void)hipFree(grec);
  }
};

template <class T>
inline hipError_t sw_reserve(T*& p, size_t& cap, size_t n) {
  if (n <= cap) return hipSuccess;
  if (p) (void)hipFree(p);
  p = nullptr; cap = 0;
  hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
  if (e == hipSuccess) cap = n;
  return e;
}

// one thread per stream: gate list -> ops (+ zm rows, needed to move the Pauli masks)
__global__ void k_s_compile(BatchArgs A, Op* ops, uint32_t* masks, int32_t* meta, uint64_t eval_id) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.batch) return;
  const int n = A.n;
  uint32_t* xm = masks + (size_t)b * 64;
  uint32_t* zm = xm + 32;
  Op* out = ops + (size_t)b * A.max_ops;
  for (int q = 0; q < n; ++q) { xm[q] = 1u << q; zm[q] = 1u << q; }
  uint32_t c = 0;
  int phase = 0, nops = 0;
  const GateRec* g = A.gates + A.gate_begin[b];
  const int G = A.gate_count[b];
  auto pauli = [&](int q, int p) {
    if (p == 0) return;
    if (p == 1 || p == 2) c ^= 1u << q;
    if (p == 2 || p == 3) {
      if (nops < A.max_ops) out[nops] = Op{0u, zm[q], -1, OP_PZ | (int)(((c >> q) & 1u) << 8)};
      ++nops;
    }
    if (p == 2) phase = (phase + 3) & 3;
  };
  for (int i = 0; i < G; ++i) {
    const GateRec r = g[i];
    if (r.kind == G_CNOT) {
      zm[r.q1] ^= zm[r.q0];
      xm[r.q0] ^= xm[r.q1];
      if ((c >> r.q0) & 1u) c ^= 1u << r.q1;
    } else if (r.kind >= G_RX && r.kind <= G_RZ) {
      if (nops < A.max_ops)
        out[nops] = Op{xm[r.q0], zm[r.q0], r.pidx, r.kind | (int)(((c >> r.q0) & 1u) << 8)};
      ++nops;
    } else if (r.kind == G_DEPOL1) {
      const double u = noise_uniform(A.noise.seed, (uint64_t)b, eval_id, (uint64_t)i);
      if (u < A.noise.p1) pauli(r.q0, 1 + (int)(u / A.noise.p1 * 3.0));
    } else if (r.kind == G_DEPOL2) {
      const double u = noise_uniform(A.noise.seed, (uint64_t)b, eval_id, (uint64_t)i);
      if (u < A.noise.p2) {
        const int idx = 1 + (int)(u / A.noise.p2 * 15.0);
        pauli(r.q0, idx & 3);
        pauli(r.q1, idx >> 2);
      }
    }
  }
  int32_t* m = meta + (size_t)b * 8;
  m[0] = nops < A.max_ops ? nops : A.max_ops;
  m[1] = (int32_t)c;
  m[2] = phase;
}

__global__ void k_s_sincos(BatchArgs A, double2* cs) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.par_count[b]) return;
  double s, c;
  sincos(0.5 * A.theta[A.par_begin[b] + j], &s, &c);
  cs[(size_t)b * A.max_params + j] = make_double2(c, s);
}

__global__ void __launch_bounds__(kThreads) k_s_init(BatchArgs A, double2* states) {
  const size_t dim = (size_t)1 << A.n;
  const size_t p = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (p < dim) states[(size_t)blockIdx.y * dim + p] = A.init[p];
}

// ---- K ops per sweep ---------------------------------------------------------------------------
// A thread owns the 2^K amplitudes p0 ^ span(g[0..K-1]).  Slot j of the basis belongs to op j of
// the group: g[j] = its partner mask when that is independent of the earlier slots (the normal
// case: pairs e <-> e ^ (1 << j), static register indices), otherwise a filler unit vector; an op
// whose mask depends on the other slots (rare) takes the generic path with a computed flip code.
// p0 runs over the coset representatives with the pivot bits of the reduced basis cleared.  Every
// update is written per element (new[e] = c a[e] + s(e) a[partner]): parity(xm & zm) = 1 makes
// s(partner) = -s(e), so the two members of a pair need no case distinction.
template <int K>
__device__ __forceinline__ void s_apply_k(double2 (&v)[1 << K], const uint32_t (&idx)[1 << K], const Op op,
                                          const double2* csb, const int slot, const int flip) {
  constexpr int E = 1 << K;
  const int kind = op.kind & 0xff, inv = (op.kind >> 8) & 1;
  if (kind == OP_RX || kind == OP_RY) {
    const double2 c = csb[op.pidx];
    double2 w[E];
    bool done = false;
#pragma unroll
    for (int i = 0; i < K; ++i)      // own slot, or the mask of another slot (two rotations on one qubit)
      if (flip == (1 << i)) {
        done = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double2 a = v[e], bq = v[e ^ (1 << i)];
          if (kind == OP_RX) w[e] = make_double2(c.x * a.x - c.y * bq.y, c.x * a.y + c.y * bq.x);
          else {
            const double sg = (parity32(idx[e] & op.zm) ^ inv) ? -c.y : c.y;
            w[e] = make_double2(c.x * a.x + sg * bq.x, c.x * a.y + sg * bq.y);
          }
        }
      }
    if (!done) {   // dependent mask: dynamic partner index (scratch), wave-uniform and rare
      double2 tmp[E];
#pragma unroll
      for (int e = 0; e < E; ++e) tmp[e] = v[e];
      for (int e = 0; e < E; ++e) {
        const double2 a = tmp[e], bq = tmp[e ^ flip];
        if (kind == OP_RX) w[e] = make_double2(c.x * a.x - c.y * bq.y, c.x * a.y + c.y * bq.x);
        else {
          const double sg = (parity32(idx[e] & op.zm) ^ inv) ? -c.y : c.y;
          w[e] = make_double2(c.x * a.x + sg * bq.x, c.x * a.y + sg * bq.y);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = w[e];
  } else if (kind == OP_RZ) {
    const double2 c = csb[op.pidx];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double sg = (parity32(idx[e] & op.zm) ^ inv) ? -c.y : c.y;
      const double2 a = v[e];
      v[e] = make_double2(c.x * a.x - sg * a.y, c.x * a.y + sg * a.x);
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (parity32(idx[e] & op.zm) ^ inv) v[e] = make_double2(-v[e].x, -v[e].y);
  }
}

template <int K>
__global__ void __launch_bounds__(kThreads) k_s_opk(BatchArgs A, double2* states, const Op* ops,
                                                    const int32_t* meta, const double2* cs, int o) {
  constexpr int E = 1 << K;
  const int b = blockIdx.y;
  const int nops = meta[(size_t)b * 8];
  if (o >= nops) return;
  const int cnt = nops - o < K ? nops - o : K;
  const size_t dim = (size_t)1 << A.n;
  double2* psi = states + (size_t)b * dim;
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= dim >> K) return;
  Op op[K];
  uint32_t g[K], red[K];       // slot masks; the same span in reduced echelon form
  int hbit[K], flip[K];
  uint32_t pivots = 0;
  int nred = 0;
  auto reduce = [&](uint32_t x) {   // x modulo the span collected so far
#pragma unroll
    for (int i = 0; i < K; ++i) if (i < nred && ((x >> hbit[i]) & 1u)) x ^= red[i];
    return x;
  };
  auto push = [&](uint32_t x) {     // x != 0 reduced: new basis vector, keep the others reduced
    const int h = 31 - __clz((int)x);
#pragma unroll
    for (int i = 0; i < K; ++i) if (i < nred && ((red[i] >> h) & 1u)) red[i] ^= x;
#pragma unroll
    for (int i = 0; i < K; ++i) if (i == nred) { red[i] = x; hbit[i] = h; }
    pivots |= 1u << h;
    ++nred;
  };
  // pass 1: independent partner masks take their own slot
  bool own[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    own[j] = false;
    g[j] = 0;
    flip[j] = 0;
    if (j < cnt) {
      op[j] = ops[(size_t)b * A.max_ops + o + j];
      const int k = op[j].kind & 0xff;
      if (k == OP_RX || k == OP_RY) {
        const uint32_t r = reduce(op[j].xm);
        if (r) { push(r); g[j] = op[j].xm; own[j] = true; flip[j] = 1 << j; }
      }
    }
  }
  // pass 2: fillers for the other slots
  {
    int q = 0;
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (!own[j]) {
        uint32_t r = 0;
        while ((r = reduce(1u << q)) == 0) ++q;
        push(r);
        g[j] = 1u << q;
        ++q;
      }
  }
  // pass 3: flip codes of the dependent partner masks (brute force over the 2^K - 1 combinations)
#pragma unroll
  for (int j = 0; j < K; ++j)
    if (j < cnt && !own[j]) {
      const int k = op[j].kind & 0xff;
      if (k == OP_RX || k == OP_RY)
        for (int f = 1; f < E; ++f) {
          uint32_t x = 0;
#pragma unroll
          for (int i = 0; i < K; ++i) if ((f >> i) & 1) x ^= g[i];
          if (x == op[j].xm) flip[j] = f;
        }
    }
  // coset representative: zeros inserted at the pivot bits, lowest first
  uint32_t p0 = t;
  for (int q = 0; q < A.n; ++q) if ((pivots >> q) & 1u) p0 = insert0(p0, q);
  uint32_t idx[E];
  double2 v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    uint32_t x = p0;
#pragma unroll
    for (int i = 0; i < K; ++i) if ((e >> i) & 1) x ^= g[i];
    idx[e] = x;
    v[e] = psi[x];
  }
  const double2* csb = cs + (size_t)b * A.max_params;
#pragma unroll
  for (int j = 0; j < K; ++j) if (j < cnt) s_apply_k<K>(v, idx, op[j], csb, j, flip[j]);
#pragma unroll
  for (int e = 0; e < E; ++e) psi[idx[e]] = v[e];
}

constexpr int kOpsPerSweep = 4;

// Pauli masks of the Hamiltonian expressed in each stream's physical layout.
__global__ void k_s_terms(BatchArgs A, const uint32_t* masks, const int32_t* meta, int n_terms,
                          uint32_t* gxp, uint32_t* tzp, double* tsg) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* xm = masks + (size_t)b * 64;
  const uint32_t* zm = xm + 32;
  const uint32_t c = (uint32_t)meta[(size_t)b * 8 + 1];
  if (i < A.ham.n_groups) {
    const uint32_t x = A.ham.gx[i];
    uint32_t xp = 0;
    for (int q = 0; q < A.n; ++q) if ((x >> q) & 1u) xp ^= xm[q];
    gxp[(size_t)b * A.ham.n_groups + i] = xp;
  }
  if (i < n_terms) {
    const uint32_t z = A.ham.term_z[i];
    uint32_t zp = 0;
    for (int q = 0; q < A.n; ++q) if ((z >> q) & 1u) zp ^= zm[q];
    tzp[(size_t)b * n_terms + i] = zp;
    tsg[(size_t)b * n_terms + i] = parity32(z & c) ? -1.0 : 1.0;
  }
}

constexpr int kEnergyApt = 4;  // amplitudes (= 2 pairs) per thread in the energy sweep

// <psi|H_shard|psi>, pair based: for every X-mask group each amplitude is read exactly once
// (as one member of a pair p0 <-> p0 ^ x'), so the sweep moves 16 * 2^n bytes per group and
// the time of a rank is proportional to the number of groups it owns (the quantity that
// Pauli-term sharding divides).  Signs are evaluated on the fly from the physical Z masks.
__global__ void __launch_bounds__(kThreads) k_s_energy(BatchArgs A, const double2* states, int n_terms,
                                                       const uint32_t* gxp, const uint32_t* tzp,
                                                       const double* tsg, double* partial) {
  __shared__ double red[8];
  const int b = blockIdx.y;
  const size_t dim = (size_t)1 << A.n;
  const double2* psi = states + (size_t)b * dim;
  const uint32_t* gx = gxp + (size_t)b * A.ham.n_groups;
  const uint32_t* tz = tzp + (size_t)b * n_terms;
  const double* ts = tsg + (size_t)b * n_terms;
  const uint32_t blk = blockIdx.x + (uint32_t)A.amp_rank * gridDim.x;   // this rank's slice of the sweep
  const uint32_t qbase = blk * (kThreads * (kEnergyApt / 2)) + threadIdx.x;
  double acc = 0.0;
  for (int g = 0; g < A.ham.n_groups; ++g) {
    const uint32_t x = gx[g];
    const int t0 = A.ham.term_off[g], t1 = A.ham.term_off[g + 1];
    if (x == 0) {   // diagonal group: two neighbouring amplitudes per "pair" slot
#pragma unroll
      for (int k = 0; k < kEnergyApt / 2; ++k) {
        const uint32_t p = 2 * (qbase + k * kThreads);
        const double2 a0 = psi[p], a1 = psi[p + 1];
        double d0 = 0.0, d1 = 0.0;
        for (int t = t0; t < t1; ++t) {
          const double c = ts[t] * A.ham.term_cr[t];
          d0 += parity32(p & tz[t]) ? -c : c;
          d1 += parity32((p + 1) & tz[t]) ? -c : c;
        }
        acc += (a0.x * a0.x + a0.y * a0.y) * d0 + (a1.x * a1.x + a1.y * a1.y) * d1;
      }
    } else {
      const int hb = 31 - __clz((int)x);
#pragma unroll
      for (int k = 0; k < kEnergyApt / 2; ++k) {
        const uint32_t p0 = insert0(qbase + k * kThreads, hb);
        const double2 bb = psi[p0], a = psi[p0 ^ x];
        double dr = 0.0, di = 0.0;
        for (int t = t0; t < t1; ++t) {
          const double s = parity32(p0 & tz[t]) ? -ts[t] : ts[t];
          dr += s * A.ham.term_cr[t];
          di += s * A.ham.term_ci[t];
        }
        acc += 2.0 * ((a.x * bb.x + a.y * bb.y) * dr - (a.x * bb.y - a.y * bb.x) * di);
      }
    }
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kThreads) k_s_reduce(const double* partial, int nblk, double* fout,
                                                       NoiseCfg noise, uint64_t eval_id, int add_shot) {
  __shared__ double red[8];
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += kThreads) acc += partial[(size_t)b * nblk + i];
  double tot = block_sum(acc, red);
  if (add_shot && noise.shot_sigma != 0.0) tot += noise.shot_sigma * noise_gauss(noise.seed, (uint64_t)b, eval_id);
  if (threadIdx.x == 0) fout[b] = tot;
}

// logical-layout copy of stream 0 for vqe_get_state
__global__ void __launch_bounds__(kThreads) k_s_state_out(BatchArgs A, const double2* states,
                                                          const uint32_t* masks, const int32_t* meta) {
  const size_t dim = (size_t)1 << A.n;
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= dim) return;
  const uint32_t c = (uint32_t)meta[1];
  const int ph = meta[2];
  uint32_t v = i ^ c, p = 0;
  for (int q = 0; q < A.n; ++q) if ((v >> q) & 1u) p ^= masks[q];
  double2 a = states[p];
  if (ph == 1) a = make_double2(-a.y, a.x);
  else if (ph == 2) a = make_double2(-a.x, -a.y);
  else if (ph == 3) a = make_double2(a.y, -a.x);
  A.state_out[i] = a;
}

}  // namespace vqe
#include "vqe_tile.h"
namespace vqe {

#define SW_TRY(expr)                                                            \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) {                                                     \
      err = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
      return _e == hipErrorOutOfMemory ? -12 : -5;                              \
    }                                                                           \
  } while (0)

// circuit + (partial) energy of every resident stream, results in A.fout (device).
// Default: the LDS-tiled kernels of vqe_tile.h; VQE_STREAM_TILED=0 (or a Hamiltonian shard with too many
// X-mask groups for the pass table) selects the one-sweep-per-four-ops kernels above.
inline bool stream_tiled(int n_groups, int n_terms) {
  static const bool env_on = [] { const char* e = getenv("VQE_STREAM_TILED"); return !(e && e[0] == '0'); }();
  return env_on && (n_groups + kETileFree - 1) / kETileFree + 1 <= kMaxEnergyPasses &&
         n_terms <= 4096;
}

inline int stream_evaluate(StreamWork& sw, const BatchArgs& A, int n_terms, hipStream_t st,
                           uint64_t eval_id, bool want_energy, std::string& err, bool want_circuit = true,
                           uint64_t generation = 0) {
  const size_t dim = (size_t)1 << A.n;
  const bool noisy = A.noise.p1 > 0.0 || A.noise.p2 > 0.0;
  if (sw.plan_src != (const void*)A.gates || sw.plan_gen != generation || noisy) sw.plan_ops_ok = sw.plan_energy_ok = false;
  sw.plan_src = (const void*)A.gates;
  sw.plan_gen = generation;
  const int B = A.batch;
  SW_TRY(sw_reserve(sw.states, sw.states_cap, (size_t)B * dim));
  SW_TRY(sw_reserve(sw.ops, sw.ops_cap, (size_t)B * A.max_ops));
  SW_TRY(sw_reserve(sw.masks, sw.masks_cap, (size_t)B * 64));
  SW_TRY(sw_reserve(sw.meta, sw.meta_cap, (size_t)B * 8));
  SW_TRY(sw_reserve(sw.cs, sw.cs_cap, (size_t)B * A.max_params));
  const int nt = n_terms > 0 ? n_terms : 1, ng = A.ham.n_groups > 0 ? A.ham.n_groups : 1;
  SW_TRY(sw_reserve(sw.gxp, sw.gxp_cap, (size_t)B * ng));
  SW_TRY(sw_reserve(sw.tzp, sw.tzp_cap, (size_t)B * nt));
  SW_TRY(sw_reserve(sw.tsg, sw.tsg_cap, (size_t)B * nt));
  const int world = A.amp_world > 0 ? A.amp_world : 1;
  const bool tiled = stream_tiled(A.ham.n_groups, n_terms);
  if (tiled) {
    const int tiles = (int)(dim >> kTileBits);
    const int max_pass = A.max_ops / kTileFree + 2;          // a pass is closed by its (kTileFree + 1)-th independent mask
    const int e_pass = std::min(kMaxEnergyPasses, (A.ham.n_groups + kETileFree - 1) / kETileFree + 1);
    {
      TilePass* tp = (TilePass*)sw.passes; size_t cap = sw.passes_cap;
      SW_TRY(sw_reserve(tp, cap, (size_t)B * max_pass)); sw.passes = tp; sw.passes_cap = cap;
      OpCoord* oc = (OpCoord*)sw.opc; cap = sw.opc_cap;
      SW_TRY(sw_reserve(oc, cap, (size_t)B * A.max_ops)); sw.opc = oc; sw.opc_cap = cap;
      ChunkRec* cr = (ChunkRec*)sw.chunks; cap = sw.chunks_cap;
      SW_TRY(sw_reserve(cr, cap, (size_t)B * A.max_ops)); sw.chunks = cr; sw.chunks_cap = cap;
      SW_TRY(sw_reserve(sw.csop, sw.csop_cap, (size_t)B * A.max_ops));
      ETilePass* ep = (ETilePass*)sw.epasses; cap = sw.epasses_cap;
      SW_TRY(sw_reserve(ep, cap, (size_t)B * kMaxEnergyPasses)); sw.epasses = ep; sw.epasses_cap = cap;
    }
    SW_TRY(sw_reserve(sw.npass, sw.npass_cap, (size_t)2 * B));
    SW_TRY(sw_reserve(sw.eorder, sw.eorder_cap, (size_t)2 * B * ng));
    SW_TRY(sw_reserve(sw.gcx, sw.gcx_cap, (size_t)B * ng));
    {
      TermRec* tr = (TermRec*)sw.trec; size_t cap = sw.trec_cap;
      SW_TRY(sw_reserve(tr, cap, (size_t)B * nt)); sw.trec = tr; sw.trec_cap = cap;
    }
    SW_TRY(sw_reserve(sw.grec, sw.grec_cap, (size_t)B * ng));
    {
      EGroupRec* eg = (EGroupRec*)sw.egrp; size_t cap = sw.egrp_cap;
      SW_TRY(sw_reserve(eg, cap, (size_t)B * ng)); sw.egrp = eg; sw.egrp_cap = cap;
      ETermRec* et = (ETermRec*)sw.eterm; cap = sw.eterm_cap;
      SW_TRY(sw_reserve(et, cap, (size_t)B * nt)); sw.eterm = et; sw.eterm_cap = cap;
      SW_TRY(sw_reserve(sw.ewi, sw.ewi_cap, (size_t)B * nt));
    }
    const int tiles_rank = (int)(dim >> kETileBits) / world;     // tiles of the Pauli-term reduction in this rank's slice
    const int e_blocks = (tiles_rank + kTilesPerBlock - 1) / kTilesPerBlock;     // k_t_energy: one partial per workgroup
    const int o_blocks = (tiles + kOpsTilesPerBlock - 1) / kOpsTilesPerBlock;
    // partial sums of a stream: [e_pass][e_blocks] of k_t_energy, then [o_blocks] of the fused pass (the pair groups that
    // close inside the tile of the last circuit pass are evaluated there, before the final state leaves the LDS)
    const int p_stride = e_pass * e_blocks + o_blocks;
    SW_TRY(sw_reserve(sw.partial, sw.partial_cap, (size_t)B * p_stride));
    static const bool fuse_on = [] { const char* e = getenv("VQE_STREAM_FUSE"); return !(e && e[0] == '0'); }();   // A/B knob
    auto plan_energy = [&](bool with_last_pass) {
      hipLaunchKernelGGL(k_t_plan_energy, dim3((B + 63) / 64), dim3(64), 0, st, A, n_terms, sw.gxp, sw.tzp, sw.tsg,
                         (ETilePass*)sw.epasses, sw.npass + B, sw.eorder, sw.gcx, sw.grec, (TermRec*)sw.trec,
                         sw.eorder + (size_t)B * ng, (EGroupRec*)sw.egrp, (ETermRec*)sw.eterm, sw.ewi,
                         with_last_pass ? (const TilePass*)sw.passes : (const TilePass*)nullptr, (const int32_t*)sw.npass, max_pass);
      sw.plan_fused = with_last_pass;
    };
    const int m_terms = std::max(nt, ng);
    bool fused = false;
    bool energy_ready = sw.plan_energy_ok;      // (a noisy run plans anew in every call: the flags stay false)
    if (want_circuit) {
      hipLaunchKernelGGL(k_s_sincos, dim3((A.max_params + 63) / 64, B), dim3(64), 0, st, A, sw.cs);
      if (!sw.plan_ops_ok) {
        hipLaunchKernelGGL(k_s_compile, dim3((B + 63) / 64), dim3(64), 0, st, A, sw.ops, sw.masks, sw.meta, eval_id);
        const bool with_groups = want_energy && fuse_on;      // (the Pauli masks follow the layout the circuit leaves: sw.masks)
        if (with_groups)
          hipLaunchKernelGGL(k_s_terms, dim3((m_terms + 63) / 64, B), dim3(64), 0, st, A, sw.masks, sw.meta, n_terms, sw.gxp,
                             sw.tzp, sw.tsg);
        hipLaunchKernelGGL(k_t_plan_ops, dim3((B + 63) / 64), dim3(64), 0, st, A, sw.ops, sw.meta, (TilePass*)sw.passes,
                           (OpCoord*)sw.opc, (ChunkRec*)sw.chunks, sw.npass, max_pass,
                           with_groups ? (const uint32_t*)sw.gxp : (const uint32_t*)nullptr);
        sw.plan_ops_ok = !noisy;
        sw.plan_energy_ok = false;
        energy_ready = false;
        if (with_groups) {
          plan_energy(true);
          sw.plan_energy_ok = !noisy;
          energy_ready = true;
        }
      }
      if (want_energy && (!energy_ready || sw.plan_fused != fuse_on)) {      // (an op plan cached from a circuit-only call, or an
        hipLaunchKernelGGL(k_s_terms, dim3((m_terms + 63) / 64, B), dim3(64), 0, st, A, sw.masks, sw.meta, n_terms, sw.gxp,   // energy-only plan)
                           sw.tzp, sw.tsg);
        plan_energy(fuse_on);
        sw.plan_energy_ok = !noisy && sw.plan_ops_ok;
        energy_ready = true;
      }
      fused = want_energy && sw.plan_fused;
      hipLaunchKernelGGL(k_t_cs_ops, dim3((A.max_ops + 63) / 64, B), dim3(64), 0, st, A, sw.ops, sw.meta, sw.cs, sw.csop);
      FusedEnergy F{fused ? sw.partial : nullptr, (const ETilePass*)sw.epasses, (const EGroupRec*)sw.egrp, (const ETermRec*)sw.eterm,
                    (const double*)sw.ewi, n_terms, tiles_rank, p_stride, e_pass * e_blocks};
      for (int p = 0; p < max_pass; ++p) {
        if (p + 1 < max_pass || !fused)      // (a stream's last pass is pass max_pass - 1 at the latest)
          hipLaunchKernelGGL(k_t_ops<false>, dim3((unsigned)o_blocks, B), dim3(kThreads), 0, st, A, sw.states,
                             (const ChunkRec*)sw.chunks, (const double2*)sw.csop, (const TilePass*)sw.passes, sw.npass, p,
                             max_pass, tiles, F);
        if (fused)
          hipLaunchKernelGGL(k_t_ops<true>, dim3((unsigned)o_blocks, B), dim3(kThreads), 0, st, A, sw.states,
                             (const ChunkRec*)sw.chunks, (const double2*)sw.csop, (const TilePass*)sw.passes, sw.npass, p,
                             max_pass, tiles, F);
      }
    }
    if (want_energy) {
      if (!energy_ready || (!fused && sw.plan_fused)) {      // (energy of the resident states without their circuits: a plan whose
        hipLaunchKernelGGL(k_s_terms, dim3((m_terms + 63) / 64, B), dim3(64), 0, st, A, sw.masks, sw.meta, n_terms, sw.gxp,   // pass 0 is a tile of k_t_energy's own)
                           sw.tzp, sw.tsg);
        plan_energy(false);
        sw.plan_energy_ok = !noisy && sw.plan_ops_ok;
      }
      if (!fused) SW_TRY(hipMemsetAsync(sw.partial, 0, (size_t)B * p_stride * sizeof(double), st));     // (the fused pass's slots)
      hipLaunchKernelGGL(k_t_energy, dim3((unsigned)e_blocks, B, e_pass), dim3(kThreads), 0, st, A, sw.states, n_terms,
                         (const ETilePass*)sw.epasses, sw.npass + B, (const EGroupRec*)sw.egrp, (const ETermRec*)sw.eterm,
                         sw.ewi, sw.partial, tiles_rank, p_stride, fused ? 1 : 0);
      hipLaunchKernelGGL(k_s_reduce, dim3(B), dim3(kThreads), 0, st, sw.partial, p_stride, A.fout, A.noise,
                         eval_id, A.amp_rank == 0 ? 1 : 0);
    }
    SW_TRY(hipGetLastError());
    return 0;
  }
  const int eblk = (int)(dim / (kThreads * kEnergyApt)) / world;
  SW_TRY(sw_reserve(sw.partial, sw.partial_cap, (size_t)B * eblk));

  if (want_circuit) {
    hipLaunchKernelGGL(k_s_compile, dim3((B + 63) / 64), dim3(64), 0, st, A, sw.ops, sw.masks, sw.meta, eval_id);
    hipLaunchKernelGGL(k_s_sincos, dim3((A.max_params + 63) / 64, B), dim3(64), 0, st, A, sw.cs);
    hipLaunchKernelGGL(k_s_init, dim3((unsigned)(dim / kThreads), B), dim3(kThreads), 0, st, A, sw.states);
    for (int o = 0; o < A.max_ops; o += kOpsPerSweep)
      hipLaunchKernelGGL(k_s_opk<kOpsPerSweep>, dim3((unsigned)((dim >> kOpsPerSweep) / kThreads), B), dim3(kThreads), 0, st,
                         A, sw.states, sw.ops, sw.meta, sw.cs, o);
  }
  if (want_energy) {
    const int m = std::max(nt, ng);
    hipLaunchKernelGGL(k_s_terms, dim3((m + 63) / 64, B), dim3(64), 0, st, A, sw.masks, sw.meta, n_terms,
                       sw.gxp, sw.tzp, sw.tsg);
    hipLaunchKernelGGL(k_s_energy, dim3(eblk, B), dim3(kThreads), 0, st, A, sw.states, n_terms, sw.gxp,
                       sw.tzp, sw.tsg, sw.partial);
    // shot noise belongs to the full energy: only an unsharded handle (or slice 0) adds it
    hipLaunchKernelGGL(k_s_reduce, dim3(B), dim3(kThreads), 0, st, sw.partial, eblk, A.fout, A.noise, eval_id,
                       A.amp_rank == 0 ? 1 : 0);
  }
  SW_TRY(hipGetLastError());
  return 0;
}

// ---- device-resident lock-step COBYLA of the streaming path ------------------------------------------------------
// One THREAD per stream runs cobyla_m0.h in the host context (one thread, sums in index order, no FMA contraction): the
// same bits as the library's host COBYLA, which is pinned against scipy's traces (tests/test_abi.py) - an evaluation of
// a batch of streams takes milliseconds here, the optimiser's O(P^2) update is noise beside it, so nothing is gained
// by spreading it over a wave and the trajectories stay those of the host build.  FIRST: start() from x0; else tell(f)
// of the evaluation just made.  The trial point of every running stream goes to xtrial (the next evaluation's theta);
// a stream that finishes writes its result (x, f, nfev as vqe_cobyla_result returns them) and clears its flag;
// n_active counts the streams that want another evaluation.
// Replaces the per-evaluation H2D / D2H / hipStreamSynchronize of the round-2 host loop (missing 3).
template <bool FIRST>
__global__ void __launch_bounds__(64) k_s_cobyla(int B, const int64_t* __restrict__ pbeg, const int32_t* __restrict__ pcnt,
                                                 const int64_t* __restrict__ sbeg, double* __restrict__ scratch,
                                                 const double* __restrict__ x0, double* __restrict__ xtrial,
                                                 const double* __restrict__ f, double rhobeg, double rhoend, int maxfun,
                                                 int32_t* __restrict__ active, int32_t* __restrict__ n_active,
                                                 double* __restrict__ xres, double* __restrict__ fres,
                                                 int32_t* __restrict__ nfres) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int n = pcnt[b];
  const int64_t p0 = pbeg[b];
  double* mem = scratch + sbeg[b];
  cby::CobylaM0<cby::HostCtx, false> c;
  c.bind(mem, n);
  double* extra = mem + c.words();         // [0] the value told last
  int want;
  if (FIRST) {
    for (int i = 0; i < n; ++i) c.x[i] = x0[p0 + i];
    if (n == 0) { want = 1; c.nfvals = 1; c.status = cby::RUNNING; c.rho = rhobeg; c.rhoend = rhoend; c.maxfun = maxfun; c.ifull = 1;
                  c.prerem = c.parsig = c.pareta = c.fbest_ret = 0.0; c.jdrop = c.ibrnch = c.iflag = 0; c.vcol = c.vrow = -2; }
    else want = c.start(rhobeg, rhoend, maxfun);
    extra[0] = 0.0;
  } else {
    if (!active[b]) return;
    c.load_state();
    const double fv = f[b];
    extra[0] = fv;
    if (n == 0) { want = 0; c.status = cby::DONE_RHOEND; c.ifull = 1; }
    else want = c.tell(fv);
  }
  c.save_state();
  active[b] = want;
  if (want) {
    for (int i = 0; i < n; ++i) xtrial[p0 + i] = c.x[i];
    atomicAdd(n_active, 1);
  } else {
    for (int i = 0; i < n; ++i) xres[p0 + i] = c.x[i];
    fres[b] = (c.status == cby::DONE_RHOEND && c.ifull == 1) ? extra[0] : c.fbest_ret;
    nfres[b] = c.nfvals;
  }
}

}  // namespace vqe
