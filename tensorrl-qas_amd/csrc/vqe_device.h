// vqe_device.h - device-side data layout and the LDS-resident VQE kernels (n <= 13).
//
// Design (gfx950 / MI355X):
//  * one 256-thread workgroup owns one evaluation stream (one parallel environment); the
//    whole 2^n complex128 amplitude vector lives in LDS (64 KiB at n = 12, two workgroups
//    per CU inside the 160 KiB LDS), so HBM sees one read of the shared initial state per
//    evaluation and nothing else;
//  * CNOTs and Pauli-X errors never move data: the kernel tracks the affine GF(2) map
//    logical index = A * physical index ^ c  and turns every rotation on a logical qubit
//    into a pair-exchange with a (multi-bit) XOR mask plus a parity mask; one gather pass at
//    the end of the circuit restores the logical layout;
//  * <psi|H|psi> is evaluated per X-mask group against precomputed sign-sum tables
//    D_x(p) = sum_k c_k (-1)^{popc(p & z_k)} (state independent, built once per
//    Hamiltonian), using the p <-> p^x pair symmetry, with one block reduction per
//    evaluation;
//  * the COBYLA loop of the reference (scipy, sequential callbacks) runs inside the same
//    workgroup (cobyla_m0.h), so an environment step is ONE kernel launch for all
//    environments and there is no host round trip per evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cobyla_m0.h"

namespace vqe {

constexpr int kThreads = 256;

enum : int { G_CNOT = 0, G_RX = 1, G_RY = 2, G_RZ = 3, G_DEPOL1 = 4, G_DEPOL2 = 5 };
enum : int { OP_RX = 1, OP_RY = 2, OP_RZ = 3, OP_PZ = 4 };

struct GateRec { int32_t kind, q0, q1, pidx; };           // as uploaded by the host
struct Op { uint32_t xm, zm; int32_t pidx; int32_t kind; };  // kind | (inv << 8)

// Hamiltonian in device memory.
struct HamDev {
  int n_groups;             // X-mask groups evaluated by this handle (after sharding)
  const uint32_t* gx;       // [n_groups] X mask
  const int64_t* tab_r;     // [n_groups] offset of the real sign-sum table (doubles)
  const int64_t* tab_i;     // [n_groups] offset of the imaginary table or -1
  const double* tables;     // LDS path: pair-compacted tables
  // streaming path (n >= 14): explicit terms
  int n_terms;              // terms of the groups above
  const int32_t* term_off;  // [n_groups + 1]
  const uint32_t* term_z;   // [n_terms]
  const double* term_cr;    // [n_terms] real part of c_k (incl. i^{#Y})
  const double* term_ci;    // [n_terms]
};

struct NoiseCfg { double p1, p2; uint64_t seed; uint64_t eval_base; };

struct BatchArgs {
  int n;                       // qubits
  int batch;
  const GateRec* gates;
  const int64_t* gate_begin;   // [batch]
  const int32_t* gate_count;   // [batch]
  const int64_t* par_begin;    // [batch]
  const int32_t* par_count;    // [batch]
  double* theta;               // [sum P] in: theta / x0   out (minimize): x
  double* fout;                // [batch]
  int32_t* nfev;               // [batch]
  double* scratch;             // COBYLA scratch
  const int64_t* scratch_begin;// [batch]
  const double2* init;         // [2^n]
  HamDev ham;
  NoiseCfg noise;
  int max_ops;                 // LDS capacity (ops) of this launch
  int max_params;              // LDS capacity (cos/sin pairs)
  double rhobeg, rhoend;
  int maxfun;
  double2* state_out;          // get_state: [2^n]
};

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Uniform [0,1) draw for (stream b, evaluation e, gate g): a pure function of the seed.
__device__ __forceinline__ double noise_uniform(uint64_t seed, uint64_t b, uint64_t e, uint64_t g) {
  uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ull * (b + 1));
  k = mix64(k ^ (e * 0xBF58476D1CE4E5B9ull));
  k = mix64(k ^ (g * 0x94D049BB133111EBull));
  return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ uint32_t insert0(uint32_t q, int hb) {
  return ((q >> hb) << (hb + 1)) | (q & ((1u << hb) - 1u));
}
__device__ __forceinline__ int parity32(uint32_t v) { return __popc(v) & 1; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the 256 threads of the block; every thread receives the identical value.
__device__ __forceinline__ double block_sum(double v, double* red /* >= 4 doubles LDS */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct DevCtx {
  int tid;
  static constexpr int nth = kThreads;
  __device__ void sync() const { __syncthreads(); }
  __device__ int all_or(int v) const { return __syncthreads_or(v); }
};

// LDS carve-up of one workgroup.
struct Lds {
  double2* psi;     // [2^n]
  Op* ops;          // [max_ops]
  double2* cs;      // [max_params] (cos, sin)(theta/2)
  double* red;      // [8]
  uint32_t* xm;     // [32] columns of A^-1
  int32_t* meta;    // [8]: n_ops, offset c, phase power, permuted flag
};

__host__ __device__ inline size_t lds_bytes(int n, int max_ops, int max_params) {
  return ((size_t)16 << n) + (size_t)16 * max_ops + (size_t)16 * max_params + 64 + 128 + 32;
}

__device__ __forceinline__ Lds carve(unsigned char* base, int n, int max_ops, int max_params) {
  Lds l;
  l.psi = (double2*)base; base += (size_t)16 << n;
  l.ops = (Op*)base; base += (size_t)16 * max_ops;
  l.cs = (double2*)base; base += (size_t)16 * max_params;
  l.red = (double*)base; base += 64;
  l.xm = (uint32_t*)base; base += 128;
  l.meta = (int32_t*)base;
  return l;
}

// Thread 0: translate the gate list of problem b into pair-exchange ops (CNOT / Pauli-X
// become updates of the affine map).  Noise Paulis are drawn per evaluation.
__device__ inline void compile_ops(const BatchArgs& A, int b, uint64_t eval_id, const Lds& L) {
  const int n = A.n;
  uint32_t xm[32], zm[32];
  for (int q = 0; q < n; ++q) { xm[q] = 1u << q; zm[q] = 1u << q; }
  uint32_t c = 0;
  int phase = 0, nops = 0;
  const GateRec* g = A.gates + A.gate_begin[b];
  const int G = A.gate_count[b];
  auto pauli = [&](int q, int p) {  // 1=X 2=Y 3=Z on logical qubit q
    if (p == 0) return;
    if (p == 1 || p == 2) c ^= 1u << q;
    if (p == 2 || p == 3) {
      if (nops < A.max_ops) L.ops[nops] = Op{0u, zm[q], -1, OP_PZ | (int)(((c >> q) & 1u) << 8)};
      ++nops;
    }
    if (p == 2) phase = (phase + 3) & 3;
  };
  for (int i = 0; i < G; ++i) {
    const GateRec r = g[i];
    switch (r.kind) {
      case G_CNOT:
        zm[r.q1] ^= zm[r.q0];
        xm[r.q0] ^= xm[r.q1];
        if ((c >> r.q0) & 1u) c ^= 1u << r.q1;
        break;
      case G_RX: case G_RY: case G_RZ:
        if (nops < A.max_ops)
          L.ops[nops] = Op{xm[r.q0], zm[r.q0], r.pidx, r.kind | (int)(((c >> r.q0) & 1u) << 8)};
        ++nops;
        break;
      case G_DEPOL1: {
        const double u = noise_uniform(A.noise.seed, (uint64_t)b, eval_id, (uint64_t)i);
        if (u < A.noise.p1) pauli(r.q0, 1 + (int)(u / A.noise.p1 * 3.0));
        break;
      }
      case G_DEPOL2: {
        const double u = noise_uniform(A.noise.seed, (uint64_t)b, eval_id, (uint64_t)i);
        if (u < A.noise.p2) {
          const int idx = 1 + (int)(u / A.noise.p2 * 15.0);
          pauli(r.q0, idx & 3);
          pauli(r.q1, idx >> 2);
        }
        break;
      }
      default: break;
    }
  }
  int permuted = (c != 0);
  for (int q = 0; q < n; ++q) { L.xm[q] = xm[q]; if (xm[q] != (1u << q)) permuted = 1; }
  L.meta[0] = nops < A.max_ops ? nops : A.max_ops;
  L.meta[1] = (int32_t)c;
  L.meta[2] = phase;
  L.meta[3] = permuted;
}

// Apply the compiled ops to the LDS-resident state, then restore the logical layout.
template <int N>
__device__ inline void run_ops(const Lds& L, const double* theta, int P) {
  constexpr uint32_t DIM = 1u << N;
  const int tid = threadIdx.x;
  for (int j = tid; j < P; j += kThreads) {
    double s, c;
    sincos(0.5 * theta[j], &s, &c);
    L.cs[j] = make_double2(c, s);
  }
  __syncthreads();
  const int nops = L.meta[0];
  for (int o = 0; o < nops; ++o) {
    const Op op = L.ops[o];
    const int kind = op.kind & 0xff;
    const int inv = (op.kind >> 8) & 1;
    if (kind == OP_RX || kind == OP_RY) {
      const double2 cs = L.cs[op.pidx];
      const int hb = 31 - __clz((int)op.xm);
      if (kind == OP_RX) {
        for (uint32_t q = tid; q < DIM / 2; q += kThreads) {
          const uint32_t p0 = insert0(q, hb), p1 = p0 ^ op.xm;
          const double2 a0 = L.psi[p0], a1 = L.psi[p1];
          L.psi[p0] = make_double2(cs.x * a0.x - cs.y * a1.y, cs.x * a0.y + cs.y * a1.x);
          L.psi[p1] = make_double2(cs.x * a1.x - cs.y * a0.y, cs.x * a1.y + cs.y * a0.x);
        }
      } else {
        for (uint32_t q = tid; q < DIM / 2; q += kThreads) {
          const uint32_t p0 = insert0(q, hb), p1 = p0 ^ op.xm;
          const double s0 = (parity32(p0 & op.zm) ^ inv) ? -cs.y : cs.y;
          const double2 a0 = L.psi[p0], a1 = L.psi[p1];
          L.psi[p0] = make_double2(cs.x * a0.x + s0 * a1.x, cs.x * a0.y + s0 * a1.y);
          L.psi[p1] = make_double2(cs.x * a1.x - s0 * a0.x, cs.x * a1.y - s0 * a0.y);
        }
      }
    } else if (kind == OP_RZ) {
      const double2 cs = L.cs[op.pidx];
      for (uint32_t p = tid; p < DIM; p += kThreads) {
        const double s = (parity32(p & op.zm) ^ inv) ? -cs.y : cs.y;
        const double2 a = L.psi[p];
        L.psi[p] = make_double2(cs.x * a.x - s * a.y, cs.x * a.y + s * a.x);
      }
    } else {  // OP_PZ
      for (uint32_t p = tid; p < DIM; p += kThreads)
        if (parity32(p & op.zm) ^ inv) {
          const double2 a = L.psi[p];
          L.psi[p] = make_double2(-a.x, -a.y);
        }
    }
    __syncthreads();
  }
  if (L.meta[3]) {  // psi_logical[i] = phi[A^-1 (i ^ c)]
    constexpr int APT = (DIM + kThreads - 1) / kThreads;
    const uint32_t c = (uint32_t)L.meta[1];
    double2 tmp[APT];
#pragma unroll
    for (int k = 0; k < APT; ++k) {
      const uint32_t i = tid + k * kThreads;
      if (i < DIM) {
        uint32_t v = i ^ c, p = 0;
#pragma unroll
        for (int q = 0; q < N; ++q) p ^= ((v >> q) & 1u) ? L.xm[q] : 0u;
        tmp[k] = L.psi[p];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < APT; ++k) {
      const uint32_t i = tid + k * kThreads;
      if (i < DIM) L.psi[i] = tmp[k];
    }
    __syncthreads();
  }
}

template <int N>
__device__ inline void load_init(const Lds& L, const double2* init) {
  constexpr uint32_t DIM = 1u << N;
  for (uint32_t p = threadIdx.x; p < DIM; p += kThreads) L.psi[p] = init[p];
}

// <psi|H|psi> over this handle's X-mask groups; identical result in every thread.
template <int N>
__device__ inline double lds_energy(const Lds& L, const HamDev& H) {
  constexpr uint32_t DIM = 1u << N;
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int g = 0; g < H.n_groups; ++g) {
    const uint32_t x = H.gx[g];
    const double* __restrict__ tr = H.tables + H.tab_r[g];
    const int64_t oi = H.tab_i[g];
    if (x == 0) {
      for (uint32_t p = tid; p < DIM; p += kThreads) {
        const double2 a = L.psi[p];
        acc += (a.x * a.x + a.y * a.y) * tr[p];
      }
    } else {
      const int hb = 31 - __clz((int)x);
      if (oi < 0) {
        double part = 0.0;
        for (uint32_t q = tid; q < DIM / 2; q += kThreads) {
          const uint32_t p0 = insert0(q, hb);
          const double2 b = L.psi[p0], a = L.psi[p0 ^ x];
          part += (a.x * b.x + a.y * b.y) * tr[q];
        }
        acc += 2.0 * part;
      } else {
        const double* __restrict__ ti = H.tables + oi;
        double part = 0.0;
        for (uint32_t q = tid; q < DIM / 2; q += kThreads) {
          const uint32_t p0 = insert0(q, hb);
          const double2 b = L.psi[p0], a = L.psi[p0 ^ x];
          part += (a.x * b.x + a.y * b.y) * tr[q] - (a.x * b.y - a.y * b.x) * ti[q];
        }
        acc += 2.0 * part;
      }
    }
  }
  return block_sum(acc, L.red);
}

template <int N>
__device__ inline double lds_evaluate(const BatchArgs& A, int b, const Lds& L, const double* theta,
                                      int P, bool noisy, uint64_t eval_id) {
  if (noisy) {
    if (threadIdx.x == 0) compile_ops(A, b, eval_id, L);
  }
  load_init<N>(L, A.init);
  __syncthreads();
  run_ops<N>(L, theta, P);
  return lds_energy<N>(L, A.ham);
}

// ---- kernels -----------------------------------------------------------------------------
template <int N>
__global__ void __launch_bounds__(kThreads) k_lds_energy(BatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, N, A.max_ops, A.max_params);
  const int b = blockIdx.x;
  const bool noisy = (A.noise.p1 > 0.0 || A.noise.p2 > 0.0);
  if (!noisy && threadIdx.x == 0) compile_ops(A, b, 0, L);
  __syncthreads();
  const double e = lds_evaluate<N>(A, b, L, A.theta + A.par_begin[b], A.par_count[b], noisy,
                                   A.noise.eval_base);
  if (threadIdx.x == 0) { A.fout[b] = e; if (A.nfev) A.nfev[b] = 1; }
}

template <int N>
__global__ void __launch_bounds__(kThreads) k_lds_state(BatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, N, A.max_ops, A.max_params);
  if (threadIdx.x == 0) compile_ops(A, 0, A.noise.eval_base, L);
  load_init<N>(L, A.init);
  __syncthreads();
  run_ops<N>(L, A.theta + A.par_begin[0], A.par_count[0]);
  const int ph = L.meta[2];
  for (uint32_t p = threadIdx.x; p < (1u << N); p += kThreads) {
    double2 a = L.psi[p];
    if (ph == 1) a = make_double2(-a.y, a.x);
    else if (ph == 2) a = make_double2(-a.x, -a.y);
    else if (ph == 3) a = make_double2(a.y, -a.x);
    A.state_out[p] = a;
  }
}

// Whole inner VQE loop of one environment in one workgroup.
template <int N>
__global__ void __launch_bounds__(kThreads) k_lds_minimize(BatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, N, A.max_ops, A.max_params);
  const int b = blockIdx.x;
  const int P = A.par_count[b];
  double* theta = A.theta + A.par_begin[b];
  const bool noisy = (A.noise.p1 > 0.0 || A.noise.p2 > 0.0);
  if (!noisy && threadIdx.x == 0) compile_ops(A, b, 0, L);
  __syncthreads();
  if (P == 0) {  // scipy returns after a single evaluation for an empty x0
    const double e = lds_evaluate<N>(A, b, L, theta, 0, noisy, A.noise.eval_base);
    if (threadIdx.x == 0) { A.fout[b] = e; A.nfev[b] = 1; }
    return;
  }
  cby::CobylaM0<DevCtx> cob;
  cob.ctx.tid = threadIdx.x;
  cob.bind(A.scratch + A.scratch_begin[b], P);
  for (int i = threadIdx.x; i < P; i += kThreads) cob.x[i] = theta[i];
  __syncthreads();
  int want = cob.start(A.rhobeg, A.rhoend, A.maxfun);
  double flast = 0.0;
  while (want) {
    flast = lds_evaluate<N>(A, b, L, cob.x, P, noisy, A.noise.eval_base + (uint64_t)cob.nfvals);
    want = cob.tell(flast);
  }
  for (int i = threadIdx.x; i < P; i += kThreads) theta[i] = cob.x[i];
  if (threadIdx.x == 0) {
    A.fout[b] = (cob.status == cby::DONE_RHOEND && cob.ifull == 1) ? flast : cob.fbest_ret;
    A.nfev[b] = cob.nfvals;
  }
}

}  // namespace vqe
