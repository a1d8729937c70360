// mps2qc_fit.hip - MPS -> brickwork-PQC fit on MI355X (gfx950): C ABI of include/mps2qc_hip.h.
//
// One workgroup = one fit (one target state, one random start) for the WHOLE optimisation loop
// of the reference (dmrg-to-qc/stiefel_opt.py:91-152): per step
//   forward   psi = U_G ... U_1 |0>                      (state resident in LDS)
//   loss      o = <t|psi>, L = 1 - |o|                   (dmrg-to-qc/mps2qc.py:283-293)
//   backward  for k = G..1: psi <- U_k^H psi (the gates are unitary: nothing is stored),
//             E_k[a,b] = sum_r conj(phi[a,r]) psi[b,r]   (a 4 x 2^(n-2) x 4 complex GEMM, on the
//             matrix cores: v_mfma_f64_16x16x4_f64), phi <- U_k^H phi
//   update    StiefelAdam.update (stiefel_opt.py:297-347) for all gates, 16 lanes per 4x4 matrix
// The reference obtains E_k by jax autodiff through a quimb contraction; o is linear in every
// gate, so dL/dU_k follows from E_k in closed form (see euclid_grads in the oracle).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/mps2qc_hip.h"

namespace {

enum { E_OK = 0, E_INVAL = -22, E_NOMEM = -12, E_NODEV = -19, E_HIP = -5 };
thread_local char g_err[256] = "";

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kMat = 16;          // complex entries of a gate
constexpr int kSlotMats = 10;     // LDS matrices per 16-lane update group
constexpr int kLdsLimit = 160 * 1024;

struct FitArgs {
  int G, max_iter, frozen, use_mfma, target_shared;
  double beta1, beta2, eps, tol, param_tol;
  const int* lo;          // [G] low bit of each gate's qubit pair
  const double* lr_t;     // [max_iter] bias-corrected learning rate of step it+1
  const double2* target;  // [B or 1][2^n]
  const double2* init;    // [B][G][16]
  double2* final_g;       // [B][G][16]
  double2* best_g;        // [B][G][16]
  double2* mom;           // [B][G][16]
  double2* vel;           // [B][G][16]
  double* hist;           // [B][max_iter]
  double* best_val;       // [B]
  int* n_iter;            // [B]
  double2* envs;          // [B][G][16]
  double2* overlap;       // [B]
  // LDS layout (bytes from the start of dynamic LDS), decided by the host
  int off_u, off_e, off_red, off_sc, off_dn, off_lo, off_scratch, off_grp;
  const int* grp;         // [n_groups][2] first gate and size of every run of mutually disjoint gates
  int n_groups;
  int red_slots;          // partial buffers of NW*64 doubles available in LDS (>= 2)
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {  // a * conj(b)
  return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// s + a * b as four fused multiply-adds
__device__ __forceinline__ double2 cmac(double2 s, double2 a, double2 b) {
  double x = __builtin_fma(a.x, b.x, s.x), y = __builtin_fma(a.x, b.y, s.y);
  return make_double2(__builtin_fma(-a.y, b.y, x), __builtin_fma(a.y, b.x, y));
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cscale(double s, double2 a) { return make_double2(s * a.x, s * a.y); }
__device__ __forceinline__ double2 cinv(double2 a) {
  double d = 1.0 / (a.x * a.x + a.y * a.y);
  return make_double2(a.x * d, -a.y * d);
}
__device__ __forceinline__ double2 csqrt_principal(double2 z) {
  double ax = fabs(z.x), ay = fabs(z.y);
  if (ax == 0.0 && ay == 0.0) return make_double2(0.0, z.y);
  double h = hypot(z.x, z.y);
  double t = sqrt(0.5 * (h + ax));
  if (z.x >= 0.0) return make_double2(t, z.y / (2.0 * t));
  return make_double2(ay / (2.0 * t), copysign(t, z.y));
}
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// Physical LDS slot of amplitude idx.  A thread reads the four amplitudes idx | b << lo of its
// 4-vector with ds_read_b128; for lo < 4 sixteen neighbouring lanes would otherwise meet in 4 (lo
// <= 2) or 8 of the 16 16-byte slots of a bank row.  XORing bits 0/2 with bit 4 and bits 1/3 with
// bit 5 spreads them over all 16 (measured: -6 % per step at 12 qubits).  The map is GF(2)-linear,
// so sw(base | field) = sw(base) ^ sw(field): one XOR per address, the wave-uniform part in scalar
// registers.  (A wider map that also folds bits 6..11 in, meant to separate the four rows of an MFMA
// operand read and the lanes of a pair sweep, was measured 15 % SLOWER and dropped.)
__device__ __forceinline__ int sw(int idx) {
  return idx ^ (((idx >> 4) & 3) * 5);
}
__device__ __forceinline__ int ins2(int r, int lo) { return ((r >> lo) << (lo + 2)) | (r & ((1 << lo) - 1)); }
__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }

// 4x4 gate (or its adjoint) from LDS into registers: wave-uniform broadcast reads.  (Moving the
// matrix to SGPRs with v_readfirstlane was measured: 64 extra instructions per sweep and SGPR
// spills, 13 % slower at 12 qubits.)
__device__ __forceinline__ void load_gate(double2 (&m)[16], const double2* M, bool dagger) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double2 v = dagger ? M[b * 4 + a] : M[a * 4 + b];
      if (dagger) v.y = -v.y;
      m[a * 4 + b] = v;
    }
}

// v <- m v for the 4-vectors already in registers
template <int IT>
__device__ __forceinline__ void mat4(const double2 (&m)[16], double2 (&v)[IT][4]) {
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    double2 w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double2 s = cmul(m[a * 4], v[t][0]);
#pragma unroll
      for (int b = 1; b < 4; ++b) s = cmac(s, m[a * 4 + b], v[t][b]);
      w[a] = s;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) v[t][a] = w[a];
  }
}

// st <- m on the qubit pair whose low bit is `lo`; every thread owns whole 4-vectors.  The trip
// count is a compile-time constant and all LDS reads are issued before the arithmetic.
template <int N, int NT>
__device__ __forceinline__ void apply_gate(double2* st, const double2 (&m)[16], int lo_, int tid) {
  constexpr int R = 1 << (N - 2);
  constexpr int IT = R >= NT ? R / NT : 1;
  const int lo = sgpr(lo_);
  int fb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) fb[b] = sw(b << lo);
  double2 v[IT][4];
  int base[IT];
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    const int r = tid + t * NT;
    base[t] = sw(ins2(r, lo));
    if (R >= NT || r < R) {
#pragma unroll
      for (int b = 0; b < 4; ++b) v[t][b] = st[base[t] ^ fb[b]];
    }
  }
  mat4<IT>(m, v);
#pragma unroll
  for (int t = 0; t < IT; ++t)
    if (R >= NT || tid + t * NT < R) {
#pragma unroll
      for (int a = 0; a < 4; ++a) st[base[t] ^ fb[a]] = v[t][a];
    }
}

// Two independent sweeps (phi <- m1 phi, psi <- m2 psi) in one barrier interval, their LDS reads
// in flight together.
template <int N, int NT>
__device__ __forceinline__ void apply_two(double2* s1, const double2* M1, int lo1_, double2* s2, const double2* M2,
                                          int lo2_, int tid) {
  constexpr int R = 1 << (N - 2);
  constexpr int IT = R >= NT ? R / NT : 1;
  const int lo1 = sgpr(lo1_), lo2 = sgpr(lo2_);
  int f1[4], f2[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) f1[b] = sw(b << lo1), f2[b] = sw(b << lo2);
  double2 v1[IT][4], v2[IT][4];
  int b1[IT], b2[IT];
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    const int r = tid + t * NT;
    b1[t] = sw(ins2(r, lo1));
    b2[t] = sw(ins2(r, lo2));
    if (R >= NT || r < R) {
#pragma unroll
      for (int b = 0; b < 4; ++b) v1[t][b] = s1[b1[t] ^ f1[b]];
#pragma unroll
      for (int b = 0; b < 4; ++b) v2[t][b] = s2[b2[t] ^ f2[b]];
    }
  }
  {
    double2 m[16];
    load_gate(m, M1, true);
    mat4<IT>(m, v1);
  }
#pragma unroll
  for (int t = 0; t < IT; ++t)
    if (R >= NT || tid + t * NT < R) {
#pragma unroll
      for (int a = 0; a < 4; ++a) s1[b1[t] ^ f1[a]] = v1[t][a];
    }
  {
    double2 m[16];
    load_gate(m, M2, true);
    mat4<IT>(m, v2);
  }
#pragma unroll
  for (int t = 0; t < IT; ++t)
    if (R >= NT || tid + t * NT < R) {
#pragma unroll
      for (int a = 0; a < 4; ++a) s2[b2[t] ^ f2[a]] = v2[t][a];
    }
}

// ---- sweeps of the half-layer scheme ---------------------------------------------------------
// The gates of a run of mutually disjoint gates (a brickwork half layer) commute.  With
// psi_b = the state before the run and phi_H = the target with the WHOLE run removed,
//   E_k = conj(U_k) F_k,   F_k[a,b] = sum_r conj(phi_H[a,r]) psi_b[b,r]
// for every gate k of the run: no intermediate state is needed, so two gates are applied per
// sweep (a thread owns the 16 amplitudes of both qubit pairs: half the LDS traffic and half
// the barriers) and the environments of a run are MFMA passes over the same two vectors.
// v[.][b] <- m v[.][b] for the four column vectors of a 4x4 block of amplitudes
__device__ __forceinline__ void mat4_cols(const double2 (&m)[16], double2 (&v)[4][4]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    double2 w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double2 s = cmul(m[a * 4], v[0][b]);
#pragma unroll
      for (int c = 1; c < 4; ++c) s = cmac(s, m[a * 4 + c], v[c][b]);
      w[a] = s;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a][b] = w[a];
  }
}

// NS state vectors (contiguous in LDS) <- gate A (x) gate B, loA > loB + 1
template <int N, int NT, int NS>
__device__ __forceinline__ void sweep_pair(double2* st, const double2* MA, int loA_, const double2* MB, int loB_,
                                           bool dagger, int tid) {
  const int loA = sgpr(loA_), loB = sgpr(loB_);
  constexpr int V = 1 << (N - 4);
  constexpr int ITEMS = NS * V;
  constexpr int IT = (ITEMS + NT - 1) / NT;
#pragma unroll 1
  for (int t = 0; t < IT; ++t) {
    const int item = tid + t * NT;
    if (ITEMS % NT == 0 || item < ITEMS) {
      double2* s = st + (item >> (N - 4)) * (1 << N);
      const int base = sw(ins2(ins2(item & (V - 1), loB), loA));
      double2 v[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) v[a][b] = s[base ^ sw((a << loA) | (b << loB))];
      {
        double2 m[16];
        load_gate(m, MB, dagger);
        mat4<4>(m, v);
      }
      __asm__ volatile("" ::: "memory");  // keep the second matrix out of registers until the first is done
      {
        double2 m[16];
        load_gate(m, MA, dagger);
        mat4_cols(m, v);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[base ^ sw((a << loA) | (b << loB))] = v[a][b];
    }
  }
}

// NS state vectors <- one gate (the odd gate of a run)
template <int N, int NT, int NS>
__device__ __forceinline__ void sweep_single(double2* st, const double2* M, int lo_, bool dagger, int tid) {
  const int lo = sgpr(lo_);
  constexpr int R = 1 << (N - 2);
  constexpr int ITEMS = NS * R;
  constexpr int IT = (ITEMS + NT - 1) / NT;
  double2 m[16];
  load_gate(m, M, dagger);
#pragma unroll 1
  for (int t = 0; t < IT; ++t) {
    const int item = tid + t * NT;
    if (ITEMS % NT == 0 || item < ITEMS) {
      double2* s = st + (item >> (N - 2)) * (1 << N);
      const int base = sw(ins2(item & (R - 1), lo));
      double2 v[1][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) v[0][b] = s[base ^ sw(b << lo)];
      mat4<1>(m, v);
#pragma unroll
      for (int b = 0; b < 4; ++b) s[base ^ sw(b << lo)] = v[0][b];
    }
  }
}

// all gates of run [first, first + cnt) on NS state vectors, a barrier after every sweep
template <int N, int NT, int NS>
__device__ __forceinline__ void sweep_run(double2* st, const double2* U, const int* lo_s, int first, int cnt,
                                          bool dagger, int tid) {
  for (int i = 0; i < cnt; i += 2) {
    const int g0 = first + i;
    if (i + 1 < cnt) {
      const int l0 = lo_s[g0], l1 = lo_s[g0 + 1];
      if (l0 > l1) sweep_pair<N, NT, NS>(st, U + g0 * kMat, l0, U + (g0 + 1) * kMat, l1, dagger, tid);
      else sweep_pair<N, NT, NS>(st, U + (g0 + 1) * kMat, l1, U + g0 * kMat, l0, dagger, tid);
    } else {
      sweep_single<N, NT, NS>(st, U + g0 * kMat, lo_s[g0], dagger, tid);
    }
    __syncthreads();
  }
}

// Environment partials on the matrix cores.  One v_mfma_f64_16x16x4_f64 consumes 8 values of
// r: rows (s, c, a) = (r-slice, re/im, row of the gate) of A = phi, columns (s', c', b) of
// B = psi, k = 4 values of r per slice; only the s == s' blocks are meaningful and kept.
// A[i][k]: lane = i + 16 k; C[row][col]: col = lane & 15, row = (lane >> 4) + 4 reg.
template <int N, int NT>
__device__ __forceinline__ void env_mfma(const double2* phi, const double2* psi, int lo_, double* red,
                                         int tid) {
  const int lo = sgpr(lo_);
  constexpr int R = 1 << (N - 2);
  constexpr int NW = NT / 64;
  constexpr int CH = (R + 7) / 8;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, k = lane >> 4;
  const int s = i >> 3, c = (i >> 2) & 1, a = i & 3;
  // compile-time trip count: all operand reads of the wave are issued before the first MFMA
  // and two accumulators break the dependency chain
  constexpr int TR = (CH + NW - 1) / NW;
  double av[TR], bv[TR];
  // address = (lane part: row a, slice s, k-step, re/im) ^ (chunk part, wave-uniform): bit
  // permutations and the swizzle are GF(2)-linear
  const int lane_off = sw(ins2(s * 4 + k, lo) | (a << lo)) * 2 + c;
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int ch = sgpr(wave + t * NW);
    av[t] = 0.0, bv[t] = 0.0;
    if ((R >= 8 && CH % NW == 0) || ch * 8 + s * 4 + k < R) {
      const int off = lane_off ^ (sw(ins2(ch * 8, lo)) * 2);
      av[t] = reinterpret_cast<const double*>(phi)[off];
      bv[t] = reinterpret_cast<const double*>(psi)[off];
    }
  }
  d4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    if (t & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc2, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc, 0, 0, 0);
  }
  if (TR > 1) acc += acc2;
  // fold the two r-slices (columns 8..15 hold slice 1, in registers 2 and 3) and keep the 64
  // meaningful sums of the wave: red[wave][c][a][c'][b]
  const double x0 = acc[0] + shfl_xor_d(acc[2], 8);
  const double x1 = acc[1] + shfl_xor_d(acc[3], 8);
  if ((lane & 8) == 0) {
    const int cl = (lane >> 4) * 8 + (lane & 7);
    red[wave * 64 + cl] = x0;
    red[wave * 64 + 32 + cl] = x1;
  }
}

// E[a][b] = (M_00 + M_11) + i (M_01 - M_10) with M_cc'[a][b] = sum_r phi_c[a,r] psi_c'[b,r].
template <int NT>
__device__ __forceinline__ double2 env_mfma_combine(const double* red, int t) {
  constexpr int NW = NT / 64;
  const int a = t >> 2, b = t & 3;
  double m00 = 0, m01 = 0, m10 = 0, m11 = 0;
  for (int w = 0; w < NW; ++w) {
    const double* p = red + w * 64 + a * 8 + b;
    m00 += p[0];
    m01 += p[4];
    m10 += p[32];
    m11 += p[36];
  }
  return make_double2(m00 + m11, m01 - m10);
}

// The same partials with vector FMAs and a cross-lane reduction (A/B variant).
template <int N, int NT>
__device__ __forceinline__ void env_valu(const double2* phi, const double2* psi, int lo, double* red,
                                         int tid) {
  constexpr int R = 1 << (N - 2);
  const int lane = tid & 63, wave = tid >> 6;
  const int lmask = (1 << lo) - 1;
  double2 acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = make_double2(0.0, 0.0);
  for (int r = tid; r < R; r += NT) {
    const int base = ((r >> lo) << (lo + 2)) | (r & lmask);
    double2 f[4], p[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f[b] = phi[sw(base | (b << lo))];
      p[b] = psi[sw(base | (b << lo))];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a * 4 + b] = cadd(acc[a * 4 + b], cmulc(p[b], f[a]));
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    double x = acc[e].x, y = acc[e].y;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      x += shfl_xor_d(x, m);
      y += shfl_xor_d(y, m);
    }
    if (lane == 0) {
      red[wave * 32 + 2 * e] = x;
      red[wave * 32 + 2 * e + 1] = y;
    }
  }
}
template <int NT>
__device__ __forceinline__ double2 env_valu_combine(const double* red, int t) {
  constexpr int NW = NT / 64;
  double x = 0, y = 0;
  for (int w = 0; w < NW; ++w) {
    x += red[w * 32 + 2 * t];
    y += red[w * 32 + 2 * t + 1];
  }
  return make_double2(x, y);
}

template <int N, int NT, bool MFMA, bool GRP>
__global__ __launch_bounds__(NT) void k_fit(FitArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int DIM = 1 << N;
  constexpr int NW = NT / 64;
  constexpr int SLOTS = NT / 16 < 16 ? NT / 16 : 16;  // 16-lane groups that run the update
  double2* psi = reinterpret_cast<double2*>(smem);
  double2* phi = psi + DIM;
  double2* U = reinterpret_cast<double2*>(smem + A.off_u);
  double2* E = reinterpret_cast<double2*>(smem + A.off_e);
  double* red = reinterpret_cast<double*>(smem + A.off_red);
  double* sc = reinterpret_cast<double*>(smem + A.off_sc);
  double* dnorm = reinterpret_cast<double*>(smem + A.off_dn);
  int* lo_s = reinterpret_cast<int*>(smem + A.off_lo);
  int* grp_s = reinterpret_cast<int*>(smem + A.off_grp);
  double2* scratch = reinterpret_cast<double2*>(smem + A.off_scratch);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = A.G;
  const long inst = blockIdx.x;
  const double2* tgt = A.target + (A.target_shared ? 0 : inst * (long)DIM);
  const long gbase = inst * (long)G * kMat;

  for (int e = tid; e < G * kMat; e += NT) {
    U[e] = A.init[gbase + e];
    if (!A.frozen) {
      A.mom[gbase + e] = make_double2(0.0, 0.0);
      A.vel[gbase + e] = make_double2(0.0, 0.0);
    }
  }
  for (int k = tid; k < G; k += NT) lo_s[k] = A.lo[k];
  if (GRP)
    for (int k = tid; k < 2 * A.n_groups; k += NT) grp_s[k] = A.grp[k];
  __syncthreads();

  double best_val = 10000.0;  // stiefel_opt.py:122
#ifdef MPS2QC_STAMPS
  long long st_c[5] = {0, 0, 0, 0, 0}, st_t = clock64(), st_n;
#define STAMP(i) st_n = clock64(), st_c[i] += st_n - st_t, st_t = st_n
#else
#define STAMP(i)
#endif
  int it = 0;
  for (; it < A.max_iter;) {
    // ---- forward
    for (int i = tid; i < DIM; i += NT) psi[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);  // sw(0) = 0
    __syncthreads();
    if constexpr (GRP) {
      for (int gi = 0; gi < A.n_groups; ++gi)
        sweep_run<N, NT, 1>(psi, U, lo_s, grp_s[2 * gi], grp_s[2 * gi + 1], false, tid);
    } else {
      for (int k = 0; k < G; ++k) {
        double2 m[16];
        load_gate(m, U + k * kMat, false);
        apply_gate<N, NT>(psi, m, lo_s[k], tid);
        __syncthreads();
      }
    }
    STAMP(0);
    // ---- overlap o = <t|psi>, phi <- t
    {
      double2 part = make_double2(0.0, 0.0);
      for (int i = tid; i < DIM; i += NT) {
        const double2 t = tgt[i];
        phi[sw(i)] = t;
        part = cadd(part, cmulc(psi[sw(i)], t));
      }
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        part.x += shfl_xor_d(part.x, m);
        part.y += shfl_xor_d(part.y, m);
      }
      if (lane == 0) {
        sc[2 * wave] = part.x;
        sc[2 * wave + 1] = part.y;
      }
    }
    __syncthreads();
    if constexpr (!GRP) {
      double2 m[16];
      load_gate(m, U + (G - 1) * kMat, true);
      apply_gate<N, NT>(psi, m, lo_s[G - 1], tid);
      __syncthreads();
    }
    double2 o = make_double2(0.0, 0.0);
    for (int w = 0; w < NW; ++w) o = cadd(o, make_double2(sc[2 * w], sc[2 * w + 1]));
    const double abso = hypot(o.x, o.y);
    const double val = 1.0 - abso;
    const double2 ph = make_double2(o.x / abso, o.y / abso);

    STAMP(1);
    // ---- backward: environments
    if constexpr (GRP) {
      // per run, last to first: remove the whole run from psi and phi (pair sweeps on both
      // vectors at once), then one MFMA pass per gate over the same two vectors; E holds F_k
      int nenv = 0;
      for (int gi = A.n_groups - 1; gi >= 0; --gi) {
        const int first = grp_s[2 * gi], cnt = grp_s[2 * gi + 1];
        sweep_run<N, NT, 2>(psi, U, lo_s, first, cnt, true, tid);
        STAMP(3);
        if (cnt <= A.red_slots) {
          // one partial buffer per gate of the run: no barrier between the MFMA passes, the waves
          // drift apart and hide each other's operand reads
          for (int k = first; k < first + cnt; ++k) env_mfma<N, NT>(phi, psi, lo_s[k], red + (k - first) * (NW * 64), tid);
          __syncthreads();
          if (tid < 16 * cnt) E[(first + (tid >> 4)) * kMat + (tid & 15)] = env_mfma_combine<NT>(red + (tid >> 4) * (NW * 64), tid & 15);
          __syncthreads();
        } else {
          for (int k = first + cnt - 1; k >= first; --k, ++nenv) {
            double* rb = red + (nenv & 1) * (NW * 64);  // double buffered: one barrier per gate
            env_mfma<N, NT>(phi, psi, lo_s[k], rb, tid);
            __syncthreads();
            if (tid < 16) E[k * kMat + tid] = env_mfma_combine<NT>(rb, tid);
          }
        }
        STAMP(2);
      }
      __syncthreads();
    } else {
    for (int k = G - 1; k >= 0; --k) {
      if constexpr (MFMA) env_mfma<N, NT>(phi, psi, lo_s[k], red, tid);
      else env_valu<N, NT>(phi, psi, lo_s[k], red, tid);
      __syncthreads();
      STAMP(2);
      if (tid < 16)
        E[k * kMat + tid] = MFMA ? env_mfma_combine<NT>(red, tid) : env_valu_combine<NT>(red, tid);
      if (k > 0) {  // nothing reads phi_0 / psi_{-1}
        if constexpr ((1 << (N - 2)) <= NT) {
          apply_two<N, NT>(phi, U + k * kMat, lo_s[k], psi, U + (k - 1) * kMat, lo_s[k - 1], tid);
        } else {  // two 4-vectors per thread and sweep: one sweep at a time keeps it in registers
          double2 m[16];
          load_gate(m, U + k * kMat, true);
          apply_gate<N, NT>(phi, m, lo_s[k], tid);
          load_gate(m, U + (k - 1) * kMat, true);
          apply_gate<N, NT>(psi, m, lo_s[k - 1], tid);
        }
      }
      __syncthreads();
      STAMP(3);
    }
    }
    if (A.envs)
      for (int e = tid; e < G * kMat; e += NT) {
        double2 ev = E[e];
        if constexpr (GRP) {  // E_k = conj(U_k) F_k
          const int k = e >> 4, i = (e >> 2) & 3, j = e & 3;
          ev = make_double2(0.0, 0.0);
          for (int l = 0; l < 4; ++l) ev = cadd(ev, cmulc(E[k * kMat + l * 4 + j], U[k * kMat + i * 4 + l]));
        }
        A.envs[gbase + e] = ev;
      }
    if (A.overlap && tid == 0) A.overlap[inst] = o;

    // ---- StiefelAdam.update, 16 lanes per gate; scratch overlays the idle state vectors
    const double lr = A.lr_t[it];
    const int sub = tid & 15, slot = (tid >> 4) & (SLOTS - 1);
    const bool upd = tid < SLOTS * 16;  // the other waves only keep the barriers company
    const int i = sub >> 2, j = sub & 3;
    double2* S = scratch + slot * (kSlotMats * kMat);
    __syncthreads();
    for (int chunk = 0; chunk * SLOTS < G; ++chunk) {
      const int k = chunk * SLOTS + slot;
      const bool act = upd && k < G;
      const int kk = act ? k : 0;
      const double2* Uk = U + kk * kMat;
      const double2 u = Uk[sub];
      double2 Ee = E[kk * kMat + sub];
      if constexpr (GRP) {  // E_k = conj(U_k) F_k
        Ee = make_double2(0.0, 0.0);
#pragma unroll
        for (int l = 0; l < 4; ++l) Ee = cadd(Ee, cmulc(E[kk * kMat + l * 4 + j], Uk[i * 4 + l]));
      }
      // Euclidean gradient handed to update(): -(o/|o|) conj(E)
      const double2 g = cmul(make_double2(-ph.x, -ph.y), make_double2(Ee.x, -Ee.y));
      if (upd) S[0 * kMat + sub] = g;
      __syncthreads();
      double2 t1 = make_double2(0.0, 0.0);  // U G^H
#pragma unroll
      for (int l = 0; l < 4; ++l) t1 = cadd(t1, cmulc(Uk[i * 4 + l], S[0 * kMat + j * 4 + l]));
      if (upd) S[1 * kMat + sub] = t1;
      __syncthreads();
      double2 rg = g;  // riemannian gradient G - U G^H U (:36-42)
#pragma unroll
      for (int l = 0; l < 4; ++l) rg = csub(rg, cmul(S[1 * kMat + i * 4 + l], Uk[l * 4 + j]));
      double met = rg.x * rg.x + rg.y * rg.y;  // Re tr(rg^H rg) (:84-90)
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) met += shfl_xor_d(met, m);
      double2 m0 = make_double2(0.0, 0.0), v0 = make_double2(0.0, 0.0);
      if (!A.frozen) {
        m0 = A.mom[gbase + kk * kMat + sub];
        v0 = A.vel[gbase + kk * kMat + sub];
      }
      const double2 mom = cadd(cscale(A.beta1, m0), cscale(1.0 - A.beta1, rg));
      double2 vel = cscale(A.beta2, v0);
      vel.x += (1.0 - A.beta2) * met;  // scalar broadcast into the matrix (:325-328)
      double2 den = csqrt_principal(vel);
      den.x += A.eps;
      const double2 dir = cmul(mom, cinv(den));
      const double2 X = cscale(-lr, dir);
      if (upd) S[2 * kMat + sub] = X;
      if (upd) S[3 * kMat + sub] = mom;
      if (upd) S[4 * kMat + sub] = vel;
      __syncthreads();
      double2 a = make_double2(0.0, 0.0);  // X U^H - U X^H (:54)
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        a = cadd(a, cmulc(S[2 * kMat + i * 4 + l], Uk[j * 4 + l]));
        a = csub(a, cmulc(Uk[i * 4 + l], S[2 * kMat + j * 4 + l]));
      }
      const double dg = (i == j) ? 1.0 : 0.0;
      double2 mm = make_double2(dg - 0.5 * a.x, -0.5 * a.y);
      if (upd) S[5 * kMat + sub] = make_double2(dg + 0.5 * a.x, 0.5 * a.y);
      __syncthreads();
      double2 y = make_double2(0.0, 0.0);  // (I + a/2) U
#pragma unroll
      for (int l = 0; l < 4; ++l) y = cadd(y, cmul(S[5 * kMat + i * 4 + l], Uk[l * 4 + j]));
      // (I - a/2) Y = (I + a/2) U by Gauss-Jordan; a is skew-Hermitian, so the Hermitian part of
      // the matrix is the identity and elimination without pivoting is stable.
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (upd) S[6 * kMat + sub] = mm;
        if (upd) S[7 * kMat + sub] = y;
        __syncthreads();
        const double2 inv = cinv(S[6 * kMat + p * 4 + p]);
        const double2 mpj = S[6 * kMat + p * 4 + j], ypj = S[7 * kMat + p * 4 + j];
        if (i == p) {
          mm = cmul(mpj, inv);
          y = cmul(ypj, inv);
        } else {
          const double2 f = cmul(S[6 * kMat + i * 4 + p], inv);
          mm = csub(mm, cmul(f, mpj));
          y = csub(y, cmul(f, ypj));
        }
        __syncthreads();
      }
      if (upd) S[8 * kMat + sub] = y;  // the new gate
      double df = (y.x - u.x) * (y.x - u.x) + (y.y - u.y) * (y.y - u.y);
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) df += shfl_xor_d(df, m);
      if (act && sub == 0) dnorm[k] = sqrt(df);
      __syncthreads();
      if (!A.frozen) {  // vector transport 0.5 (M - U M^H U) of both moments (:344-345)
        double2 tm = make_double2(0.0, 0.0), tv = make_double2(0.0, 0.0);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          tm = cadd(tm, cmulc(S[8 * kMat + i * 4 + l], S[3 * kMat + j * 4 + l]));
          tv = cadd(tv, cmulc(S[8 * kMat + i * 4 + l], S[4 * kMat + j * 4 + l]));
        }
        if (upd) S[0 * kMat + sub] = tm;
        if (upd) S[1 * kMat + sub] = tv;
      }
      __syncthreads();
      if (!A.frozen) {
        double2 mn = mom, vn = vel;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          mn = csub(mn, cmul(S[0 * kMat + i * 4 + l], S[8 * kMat + l * 4 + j]));
          vn = csub(vn, cmul(S[1 * kMat + i * 4 + l], S[8 * kMat + l * 4 + j]));
        }
        if (act) {
          A.mom[gbase + k * kMat + sub] = cscale(0.5, mn);
          A.vel[gbase + k * kMat + sub] = cscale(0.5, vn);
        }
      }
      if (act) U[k * kMat + sub] = y;
      __syncthreads();
    }

    STAMP(4);
    // ---- bookkeeping of minimize() (:124-147), identical in every thread
    if (tid == 0) A.hist[inst * (long)A.max_iter + it] = val;
    ++it;
    if (val < best_val) {
      best_val = val;
      for (int e = tid; e < G * kMat; e += NT) A.best_g[gbase + e] = U[e];
    }
    if (val < A.tol) break;
    double dsum = 0.0;
    for (int k = 0; k < G; ++k) dsum += dnorm[k];
    if (dsum / G < A.param_tol) break;
  }
  for (int e = tid; e < G * kMat; e += NT) A.final_g[gbase + e] = U[e];
#ifdef MPS2QC_STAMPS
  if (tid == 0)  // diagnostic build only: cycles of forward / overlap / env / backward applies / update
    for (int q = 0; q < 5; ++q) A.envs[gbase + q] = make_double2((double)st_c[q], 0.0);
#endif
  if (tid == 0) {
    A.best_val[inst] = best_val;
    A.n_iter[inst] = it;
  }
}

#define HIP_TRY(x)                                                                        \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      snprintf(g_err, sizeof g_err, "%s: %s", #x, hipGetErrorString(e_));                 \
      rc = E_HIP;                                                                         \
      goto done;                                                                          \
    }                                                                                     \
  } while (0)

template <int N, int NT, bool MFMA, bool GRP>
hipError_t launch2(const FitArgs& A, int batch, size_t lds, hipStream_t st) {
  auto fn = reinterpret_cast<const void*>(&k_fit<N, NT, MFMA, GRP>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_fit<N, NT, MFMA, GRP>), dim3((unsigned)batch), dim3(NT), lds, st, A);
  return hipGetLastError();
}
// the vector-FMA environments are a kernel of their own: their 32 accumulators per thread would
// otherwise set the register allocation of the MFMA kernel too
template <int N, int NT>
hipError_t launch(const FitArgs& A, int batch, size_t lds, hipStream_t st, bool grouped) {
  if constexpr (N >= 10) {
    if (grouped) return launch2<N, NT, true, true>(A, batch, lds, st);
  }
  return A.use_mfma ? launch2<N, NT, true, false>(A, batch, lds, st) : launch2<N, NT, false, false>(A, batch, lds, st);
}

}  // namespace

extern "C" {

const char* mps2qc_last_error(void) { return g_err; }

int mps2qc_brickwork_sites(int n_qubits, int n_layers, int32_t* sites, int cap) {
  if (n_qubits < 2 || n_layers < 0 || (!sites && cap > 0)) {
    snprintf(g_err, sizeof g_err, "mps2qc_brickwork_sites: bad argument");
    return E_INVAL;
  }
  int cnt = 0;
  for (int l = 0; l < n_layers; ++l)
    for (int par = 0; par < 2; ++par)
      for (int i = par; i < n_qubits - 1; i += 2) {
        if (cnt < cap) sites[cnt] = i;
        ++cnt;
      }
  return cnt;
}

int mps2qc_fit_brickwork(int device_id, int n, int G, const int32_t* sites, int batch,
                         const double* target, int target_shared, const double* init_gates, double lr,
                         double beta1, double beta2, double eps, int jit_frozen, int max_iter,
                         double tol, double param_tol, int use_mfma, double* opt_gates,
                         double* final_gates, double* loss_history, double* best_val, int32_t* n_iter,
                         double* last_envs, double* last_overlap, float* kernel_ms) {
  int rc = E_OK;
  g_err[0] = 0;
  if (n < 2 || n > MPS2QC_MAX_QUBITS || G < 1 || batch < 1 || max_iter < 1 || !sites || !target ||
      !init_gates) {
    snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork: bad argument (2 <= n <= %d, G, batch, max_iter >= 1)",
             MPS2QC_MAX_QUBITS);
    return E_INVAL;
  }
  std::vector<int> lo(G);
  for (int k = 0; k < G; ++k) {
    if (sites[k] < 0 || sites[k] > n - 2) {
      snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork: gate %d on sites (%d,%d) outside the register", k,
               sites[k], sites[k] + 1);
      return E_INVAL;
    }
    lo[k] = n - 2 - sites[k];
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device_id < 0 || device_id >= ndev) {
    snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork: no usable HIP device (there is no CPU fallback)");
    return E_NODEV;
  }
  // bias-corrected learning rate of every step (stiefel_opt.py:333-335); t = 1 when frozen
  std::vector<double> lr_t(max_iter);
  for (int it = 0; it < max_iter; ++it) {
    const double t = jit_frozen ? 1.0 : (double)(it + 1);
    lr_t[it] = lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t));
  }

  // threads per fit: enough waves per SIMD to hide the LDS latency of the gate sweeps
  // threads per fit: enough waves per SIMD to hide the LDS latency of the gate sweeps
  const int NT = n <= 8 ? 64 : n <= 10 ? 256 : 512;
  // runs of mutually disjoint gates (brickwork half layers) for the half-layer scheme
  std::vector<int> grp;
  {
    unsigned used = 0;
    for (int k = 0; k < G; ++k) {
      const unsigned bits = 3u << lo[k];
      if (grp.empty() || (used & bits)) {
        grp.push_back(k), grp.push_back(0);
        used = 0;
      }
      used |= bits;
      ++grp.back();
    }
  }
  // the half-layer scheme pays at 12 qubits (-9 % per step); at 10 / 11 qubits it leaves too many
  // threads idle in the pair sweeps and is 5-20 % slower
  bool grouped = use_mfma && n >= (getenv("MPS2QC_GROUPED") ? 10 : 12) && !(getenv("MPS2QC_GROUPED") && atoi(getenv("MPS2QC_GROUPED")) == 0);
  const size_t dim = (size_t)1 << n;
  FitArgs A;
  memset(&A, 0, sizeof A);
  size_t off = 2 * dim * 16;
  A.off_u = (int)off, off += (size_t)G * kMat * 16;
  A.off_e = (int)off, off += (size_t)G * kMat * 16;
  A.off_red = (int)off;
  const size_t red_off = off;
  off += (size_t)2 * (NT / 64) * 64 * 8;  // double buffered (half-layer scheme)
  A.off_sc = (int)off, off += 2 * 16 * 8;  // one complex partial per wave (<= 16 waves)
  A.off_dn = (int)off, off += (size_t)((G + 1) & ~1) * 8;
  A.off_lo = (int)off, off += (size_t)((G + 3) & ~3) * 4;
  A.off_grp = (int)off, off += (size_t)((grp.size() + 3) & ~(size_t)3) * 4;
  const size_t scratch = (size_t)(NT / 16 < 16 ? NT / 16 : 16) * kSlotMats * kMat * 16;
  if (2 * dim * 16 >= scratch) A.off_scratch = 0;  // overlay on psi / phi, idle during the update
  else A.off_scratch = (int)off, off += scratch;
  int red_slots = 2;
  if (grouped && !getenv("MPS2QC_RED2")) {  // one partial buffer per gate of the largest run when LDS has room: moves everything behind red
    int big = 2;
    for (size_t g = 1; g < grp.size(); g += 2) big = grp[g] > big ? grp[g] : big;
    const size_t extra = (size_t)(big - 2) * (NT / 64) * 64 * 8;
    if (off + extra <= (size_t)kLdsLimit) {
      red_slots = big;
      off += extra;
      A.off_sc += (int)extra, A.off_dn += (int)extra, A.off_lo += (int)extra, A.off_grp += (int)extra;
      if (A.off_scratch) A.off_scratch += (int)extra;
    }
  }
  (void)red_off;
  if (off > (size_t)kLdsLimit) {
    snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork: %d gates at %d qubits need %zu B of LDS (limit %d)", G, n,
             off, kLdsLimit);
    return E_INVAL;
  }

  const size_t gsz = (size_t)batch * G * kMat * 16;
  const size_t tsz = (target_shared ? 1 : (size_t)batch) * dim * 16;
  int *d_lo = nullptr, *d_grp = nullptr;
  double *d_lr = nullptr, *d_hist = nullptr, *d_bv = nullptr;
  double2 *d_t = nullptr, *d_init = nullptr, *d_fin = nullptr, *d_best = nullptr, *d_m = nullptr, *d_v = nullptr,
          *d_env = nullptr, *d_ov = nullptr;
  int* d_ni = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;

  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipStreamCreate(&st));
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipMalloc(&d_lo, G * sizeof(int)));
  HIP_TRY(hipMalloc(&d_grp, grp.size() * sizeof(int)));
  HIP_TRY(hipMalloc(&d_lr, max_iter * sizeof(double)));
  HIP_TRY(hipMalloc(&d_hist, (size_t)batch * max_iter * sizeof(double)));
  HIP_TRY(hipMalloc(&d_bv, batch * sizeof(double)));
  HIP_TRY(hipMalloc(&d_ni, batch * sizeof(int)));
  HIP_TRY(hipMalloc(&d_t, tsz));
  HIP_TRY(hipMalloc(&d_init, gsz));
  HIP_TRY(hipMalloc(&d_fin, gsz));
  HIP_TRY(hipMalloc(&d_best, gsz));
  HIP_TRY(hipMalloc(&d_m, gsz));
  HIP_TRY(hipMalloc(&d_v, gsz));
  HIP_TRY(hipMalloc(&d_env, gsz));
  HIP_TRY(hipMalloc(&d_ov, batch * 16));
  HIP_TRY(hipMemcpyAsync(d_lo, lo.data(), G * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_grp, grp.data(), grp.size() * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_lr, lr_t.data(), max_iter * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_t, target, tsz, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_init, init_gates, gsz, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(d_hist, 0, (size_t)batch * max_iter * sizeof(double), st));
  HIP_TRY(hipMemcpyAsync(d_best, init_gates, gsz, hipMemcpyHostToDevice, st));

  A.G = G, A.max_iter = max_iter, A.frozen = jit_frozen ? 1 : 0, A.use_mfma = use_mfma ? 1 : 0;
  A.target_shared = target_shared ? 1 : 0;
  A.beta1 = beta1, A.beta2 = beta2, A.eps = eps, A.tol = tol, A.param_tol = param_tol;
  A.grp = d_grp, A.n_groups = (int)grp.size() / 2, A.red_slots = red_slots;
  A.lo = d_lo, A.lr_t = d_lr, A.target = d_t, A.init = d_init, A.final_g = d_fin, A.best_g = d_best;
  A.mom = d_m, A.vel = d_v, A.hist = d_hist, A.best_val = d_bv, A.n_iter = d_ni, A.envs = d_env, A.overlap = d_ov;

  HIP_TRY(hipEventRecord(e0, st));
  {
    hipError_t le = hipErrorInvalidValue;
    switch (n) {
#define CASE(NN, TT) case NN: le = launch<NN, TT>(A, batch, off, st, grouped); break;
      CASE(2, 64) CASE(3, 64) CASE(4, 64) CASE(5, 64) CASE(6, 64) CASE(7, 64) CASE(8, 64)
      CASE(9, 256) CASE(10, 256) CASE(11, 512)
      CASE(12, 512)
#undef CASE
    }
    HIP_TRY(le);
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (kernel_ms) *kernel_ms = ms;
  if (opt_gates) HIP_TRY(hipMemcpy(opt_gates, d_best, gsz, hipMemcpyDeviceToHost));
  if (final_gates) HIP_TRY(hipMemcpy(final_gates, d_fin, gsz, hipMemcpyDeviceToHost));
  if (loss_history)
    HIP_TRY(hipMemcpy(loss_history, d_hist, (size_t)batch * max_iter * sizeof(double), hipMemcpyDeviceToHost));
  if (best_val) HIP_TRY(hipMemcpy(best_val, d_bv, batch * sizeof(double), hipMemcpyDeviceToHost));
  if (n_iter) HIP_TRY(hipMemcpy(n_iter, d_ni, batch * sizeof(int), hipMemcpyDeviceToHost));
  if (last_envs) HIP_TRY(hipMemcpy(last_envs, d_env, gsz, hipMemcpyDeviceToHost));
  if (last_overlap) HIP_TRY(hipMemcpy(last_overlap, d_ov, (size_t)batch * 16, hipMemcpyDeviceToHost));

done:
  for (void* p : {(void*)d_grp, (void*)d_lo, (void*)d_lr, (void*)d_hist, (void*)d_bv, (void*)d_ni, (void*)d_t, (void*)d_init,
                  (void*)d_fin, (void*)d_best, (void*)d_m, (void*)d_v, (void*)d_env, (void*)d_ov})
    (void)hipFree(p);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

}  // extern "C"

// =====================================================================================================
// HBM-streaming variant for registers beyond MPS2QC_MAX_QUBITS (13 .. 26 qubits): the states no longer
// fit LDS, but a dense 2^n target is small by MI355X standards (16 MiB at 20 qubits; 288 GB of HBM hold
// the two work vectors of a fit up to ~33 qubits) - this is the dense-target answer to the reference's
// "MPS form" contraction (dmrg-to-qc/mps2qc.py:242-339) at sizes where quimb contracts a tensor network.
// Same optimisation loop as k_fit, cut at the kernel boundary:
//   forward   G coalesced sweeps psi <- U_k psi             (k_sf_apply: a thread owns one 4-vector)
//   overlap   o = <t|psi>                                    (k_sf_dot + fixed-order second stage)
//   backward  per gate ONE fused sweep: psi <- U_k^H psi, E_k += conj(phi_a) psi_b, phi <- U_k^H phi
//             (k_sf_back: per-block partials of the 16 complex entries, fixed-order second stage)
//   update    StiefelAdam.update + the bookkeeping of minimize() on the HOST (16 x G complex numbers
//             per fit and step; one device round trip per step against ~1 ms of sweeps at 20 qubits)
namespace {

constexpr int kSfThreads = 256;

struct SfGate { double2 m[16]; };      // one gate (or its adjoint), passed by value: scalar registers

__global__ void __launch_bounds__(kSfThreads) k_sf_zero_state(double2* psi, size_t dim) {
  const size_t i = (size_t)blockIdx.x * kSfThreads + threadIdx.x;
  double2* p = psi + (size_t)blockIdx.y * dim;
  if (i < dim) p[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);
}

__global__ void __launch_bounds__(kSfThreads) k_sf_copy(double2* dst, const double2* src, size_t dim, int src_shared) {
  const size_t i = (size_t)blockIdx.x * kSfThreads + threadIdx.x;
  if (i < dim) dst[(size_t)blockIdx.y * dim + i] = src[(src_shared ? 0 : (size_t)blockIdx.y * dim) + i];
}

// st <- M st on the qubit pair with low bit `lo`; gates: [B][G][16], gate k of fit blockIdx.y (adjoint if dagger)
__global__ void __launch_bounds__(kSfThreads) k_sf_apply(double2* states, size_t dim, const double2* gates, int G, int k,
                                                         int lo, int dagger) {
  const size_t r = (size_t)blockIdx.x * kSfThreads + threadIdx.x;
  if (r >= dim / 4) return;
  double2* st = states + (size_t)blockIdx.y * dim;
  const double2* M = gates + ((size_t)blockIdx.y * G + k) * kMat;
  const size_t base = ((r >> lo) << (lo + 2)) | (r & (((size_t)1 << lo) - 1));
  double2 v[4], w[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) v[b] = st[base | ((size_t)b << lo)];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double2 s = make_double2(0.0, 0.0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double2 m = dagger ? M[b * 4 + a] : M[a * 4 + b];
      if (dagger) m.y = -m.y;
      s = b == 0 ? cmul(m, v[0]) : cmac(s, m, v[b]);
    }
    w[a] = s;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) st[base | ((size_t)a << lo)] = w[a];
}

__device__ __forceinline__ double block_sum_sf(double v, double* red) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += shfl_xor_d(v, m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[b][block] = sum over the block's amplitudes of conj(t) psi
__global__ void __launch_bounds__(kSfThreads) k_sf_dot(const double2* target, int target_shared, const double2* psi, size_t dim,
                                                       double2* partial) {
  __shared__ double red[4];
  const size_t i = (size_t)blockIdx.x * kSfThreads + threadIdx.x;
  const double2* t = target + (target_shared ? 0 : (size_t)blockIdx.y * dim);
  const double2* p = psi + (size_t)blockIdx.y * dim;
  double2 c = make_double2(0.0, 0.0);
  if (i < dim) c = cmulc(p[i], t[i]);          // psi * conj(t)
  const double x = block_sum_sf(c.x, red), y = block_sum_sf(c.y, red);
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = make_double2(x, y);
}

// fused backward step of gate k: psi <- U^H psi, E[a][b] += conj(phi[a]) psi[b], phi <- U^H phi
__global__ void __launch_bounds__(kSfThreads) k_sf_back(double2* psi_all, double2* phi_all, size_t dim, const double2* gates, int G,
                                                        int k, int lo, double2* partial /* [B][blocks][16] */) {
  __shared__ double red[4];
  const size_t r = (size_t)blockIdx.x * kSfThreads + threadIdx.x;
  double2* psi = psi_all + (size_t)blockIdx.y * dim;
  double2* phi = phi_all + (size_t)blockIdx.y * dim;
  const double2* M = gates + ((size_t)blockIdx.y * G + k) * kMat;
  double2 e[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) e[q] = make_double2(0.0, 0.0);
  if (r < dim / 4) {
    const size_t base = ((r >> lo) << (lo + 2)) | (r & (((size_t)1 << lo) - 1));
    double2 p[4], f[4], pw[4], fw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { p[b] = psi[base | ((size_t)b << lo)]; f[b] = phi[base | ((size_t)b << lo)]; }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double2 sp = make_double2(0.0, 0.0), sf = make_double2(0.0, 0.0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double2 m = M[b * 4 + a];
        m.y = -m.y;                                   // (U^H)[a][b] = conj(U[b][a])
        sp = b == 0 ? cmul(m, p[0]) : cmac(sp, m, p[b]);
        sf = b == 0 ? cmul(m, f[0]) : cmac(sf, m, f[b]);
      }
      pw[a] = sp; fw[a] = sf;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) e[a * 4 + b] = cmulc(pw[b], f[a]);     // conj(phi_k[a]) * psi_{k-1}[b]
#pragma unroll
    for (int a = 0; a < 4; ++a) { psi[base | ((size_t)a << lo)] = pw[a]; phi[base | ((size_t)a << lo)] = fw[a]; }
  }
  double2* out = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const double x = block_sum_sf(e[q].x, red), y = block_sum_sf(e[q].y, red);
    if (threadIdx.x == 0) out[q] = make_double2(x, y);
  }
}

// out[b][slot * width + q] = sum over blocks (fixed order) of partial[b][block][q]
__global__ void __launch_bounds__(kSfThreads) k_sf_reduce(const double2* partial, int nblk, int width, double2* out, int out_stride,
                                                          int slot) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  for (int q = 0; q < width; ++q) {
    double x = 0.0, y = 0.0;
    for (int i = threadIdx.x; i < nblk; i += kSfThreads) {
      const double2 v = partial[((size_t)b * nblk + i) * width + q];
      x += v.x; y += v.y;
    }
    x = block_sum_sf(x, red); y = block_sum_sf(y, red);
    if (threadIdx.x == 0) out[(size_t)b * out_stride + (size_t)slot * width + q] = make_double2(x, y);
  }
}

// ---- host side of StiefelAdam.update (stiefel_opt.py:297-347) on 4x4 complex matrices -------------------
struct C4 { double2 a[16]; };
inline double2 h_mul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
inline double2 h_mulc(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a conj(b)
inline double2 h_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
inline double2 h_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
inline double2 h_inv(double2 a) { const double d = 1.0 / (a.x * a.x + a.y * a.y); return make_double2(a.x * d, -a.y * d); }
inline double2 h_sqrt(double2 z) {         // principal branch, as numpy
  const double ax = fabs(z.x), ay = fabs(z.y);
  if (ax == 0.0 && ay == 0.0) return make_double2(0.0, z.y);
  const double h = hypot(z.x, z.y), t = sqrt(0.5 * (h + ax));
  if (z.x >= 0.0) return make_double2(t, z.y / (2.0 * t));
  return make_double2(ay / (2.0 * t), copysign(t, z.y));
}
inline C4 h_matmul(const C4& A, const C4& B, bool conjT_B) {      // A * B or A * B^H
  C4 R;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double2 s = make_double2(0.0, 0.0);
      for (int l = 0; l < 4; ++l) s = h_add(s, conjT_B ? h_mulc(A.a[i * 4 + l], B.a[j * 4 + l]) : h_mul(A.a[i * 4 + l], B.a[l * 4 + j]));
      R.a[i * 4 + j] = s;
    }
  return R;
}
inline C4 h_riem(const C4& g, const C4& p) {      // g - p g^H p
  const C4 t = h_matmul(h_matmul(p, g, true), p, false);
  C4 R;
  for (int e = 0; e < 16; ++e) R.a[e] = h_sub(g.a[e], t.a[e]);
  return R;
}
// one gate: returns the new gate, updates the moments (unless frozen) and the Frobenius norm of the change
inline C4 h_update(const C4& p, const C4& g, C4& mom_io, C4& vel_io, bool frozen, double lr, double b1, double b2, double eps,
                   double* dnorm) {
  const C4 rg = h_riem(g, p);
  double tr = 0.0;
  for (int e = 0; e < 16; ++e) tr += rg.a[e].x * rg.a[e].x + rg.a[e].y * rg.a[e].y;      // Re tr(rg^H rg)
  C4 mom, vel, dir;
  for (int e = 0; e < 16; ++e) {
    const double2 m0 = frozen ? make_double2(0.0, 0.0) : mom_io.a[e], v0 = frozen ? make_double2(0.0, 0.0) : vel_io.a[e];
    mom.a[e] = make_double2(b1 * m0.x + (1 - b1) * rg.a[e].x, b1 * m0.y + (1 - b1) * rg.a[e].y);
    vel.a[e] = make_double2(b2 * v0.x + (1 - b2) * tr, b2 * v0.y);                          // the scalar metric broadcast into the matrix (:325-328)
    double2 den = h_sqrt(vel.a[e]);
    den.x += eps;
    dir.a[e] = h_mul(mom.a[e], h_inv(den));
    dir.a[e] = make_double2(-lr * dir.a[e].x, -lr * dir.a[e].y);
  }
  // Cayley retraction (:48-57): a = g' p^H - p g'^H, new = (I - a/2)^-1 (I + a/2) p
  const C4 gp = h_matmul(dir, p, true), pg = h_matmul(p, dir, true);
  C4 mm, y;
  C4 cplus;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const double2 a = h_sub(gp.a[i * 4 + j], pg.a[i * 4 + j]);
      const double id = i == j ? 1.0 : 0.0;
      mm.a[i * 4 + j] = make_double2(id - 0.5 * a.x, -0.5 * a.y);
      cplus.a[i * 4 + j] = make_double2(id + 0.5 * a.x, 0.5 * a.y);
    }
  y = h_matmul(cplus, p, false);
  for (int pv = 0; pv < 4; ++pv) {      // Gauss-Jordan without pivoting: the Hermitian part of (I - a/2) is the identity
    const double2 inv = h_inv(mm.a[pv * 4 + pv]);
    double2 mrow[4], yrow[4];
    for (int j = 0; j < 4; ++j) { mrow[j] = mm.a[pv * 4 + j]; yrow[j] = y.a[pv * 4 + j]; }
    for (int i = 0; i < 4; ++i) {
      if (i == pv) {
        for (int j = 0; j < 4; ++j) { mm.a[i * 4 + j] = h_mul(mrow[j], inv); y.a[i * 4 + j] = h_mul(yrow[j], inv); }
      } else {
        const double2 f = h_mul(mm.a[i * 4 + pv], inv);
        for (int j = 0; j < 4; ++j) {
          mm.a[i * 4 + j] = h_sub(mm.a[i * 4 + j], h_mul(f, mrow[j]));
          y.a[i * 4 + j] = h_sub(y.a[i * 4 + j], h_mul(f, yrow[j]));
        }
      }
    }
  }
  double df = 0.0;
  for (int e = 0; e < 16; ++e) df += (y.a[e].x - p.a[e].x) * (y.a[e].x - p.a[e].x) + (y.a[e].y - p.a[e].y) * (y.a[e].y - p.a[e].y);
  *dnorm = sqrt(df);
  if (!frozen) {      // vector transport 0.5 (M - U M^H U) of both moments (:344-345)
    const C4 tm = h_matmul(h_matmul(y, mom, true), y, false), tv = h_matmul(h_matmul(y, vel, true), y, false);
    for (int e = 0; e < 16; ++e) {
      mom_io.a[e] = make_double2(0.5 * (mom.a[e].x - tm.a[e].x), 0.5 * (mom.a[e].y - tm.a[e].y));
      vel_io.a[e] = make_double2(0.5 * (vel.a[e].x - tv.a[e].x), 0.5 * (vel.a[e].y - tv.a[e].y));
    }
  }
  return y;
}

}  // namespace

extern "C" int mps2qc_fit_brickwork_stream(int device_id, int n, int G, const int32_t* sites, int batch, const double* target,
                                           int target_shared, const double* init_gates, double lr, double beta1, double beta2,
                                           double eps, int jit_frozen, int max_iter, double tol, double param_tol,
                                           double* opt_gates, double* final_gates, double* loss_history, double* best_val,
                                           int32_t* n_iter, double* last_envs, double* last_overlap, float* total_ms) {
  int rc = E_OK;
  g_err[0] = 0;
  if (n < 2 || n > MPS2QC_STREAM_MAX_QUBITS || G < 1 || batch < 1 || max_iter < 1 || !sites || !target || !init_gates) {
    snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork_stream: bad argument (2 <= n <= %d, G, batch, max_iter >= 1)",
             MPS2QC_STREAM_MAX_QUBITS);
    return E_INVAL;
  }
  std::vector<int> lo(G);
  for (int k = 0; k < G; ++k) {
    if (sites[k] < 0 || sites[k] > n - 2) {
      snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork_stream: gate %d on sites (%d,%d) outside the register", k, sites[k], sites[k] + 1);
      return E_INVAL;
    }
    lo[k] = n - 2 - sites[k];
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device_id < 0 || device_id >= ndev) {
    snprintf(g_err, sizeof g_err, "mps2qc_fit_brickwork_stream: no usable HIP device (there is no CPU fallback)");
    return E_NODEV;
  }
  const size_t dim = (size_t)1 << n;
  const int B = batch;
  const size_t gcount = (size_t)B * G * kMat;
  const unsigned blk_amp = (unsigned)((dim + kSfThreads - 1) / kSfThreads), blk_vec = (unsigned)((dim / 4 + kSfThreads - 1) / kSfThreads);
  std::vector<C4> U((size_t)B * G), Ubest((size_t)B * G), mom((size_t)B * G), vel((size_t)B * G);
  memcpy(U.data(), init_gates, gcount * 16);
  Ubest = U;
  memset(mom.data(), 0, gcount * 16);
  memset(vel.data(), 0, gcount * 16);
  std::vector<double> bv(B, 10000.0);
  std::vector<int> nit(B, 0), active(B, 1);
  if (loss_history) memset(loss_history, 0, (size_t)B * max_iter * sizeof(double));
  // One optimiser step is the same 3 G + 5 launches and three copies every time - only the gate values change: the
  // sequence is captured ONCE into a hipGraph and replayed (a launch of ~10 us on a 4 MiB vector otherwise waits for
  // the host to enqueue it; MPS2QC_STREAM_GRAPH=0: plain launches, for comparison).  The graph's copy nodes need
  // page-locked host buffers: h_pin = [gates in | overlaps out | environments out].
  double2 *h_pin = nullptr, *h_gin = nullptr, *h_ov = nullptr, *h_env = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  static const bool graph_on = [] { const char* e = getenv("MPS2QC_STREAM_GRAPH"); return !(e && e[0] == '0'); }();

  double2 *d_psi = nullptr, *d_phi = nullptr, *d_t = nullptr, *d_g = nullptr, *d_part = nullptr, *d_ov = nullptr, *d_env = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  const size_t tsz = (target_shared ? 1 : (size_t)B) * dim * 16;
  HIP_TRY(hipSetDevice(device_id));
  HIP_TRY(hipStreamCreate(&st));
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipMalloc(&d_psi, (size_t)B * dim * 16));
  HIP_TRY(hipMalloc(&d_phi, (size_t)B * dim * 16));
  HIP_TRY(hipMalloc(&d_t, tsz));
  HIP_TRY(hipMalloc(&d_g, gcount * 16));
  HIP_TRY(hipMalloc(&d_part, (size_t)B * (blk_vec > blk_amp ? blk_vec : blk_amp) * 16 * 16));
  HIP_TRY(hipMalloc(&d_ov, (size_t)B * 16));
  HIP_TRY(hipMalloc(&d_env, gcount * 16));
  HIP_TRY(hipHostMalloc((void**)&h_pin, (2 * gcount + (size_t)B) * 16, hipHostMallocDefault));
  h_gin = h_pin; h_ov = h_pin + gcount; h_env = h_ov + B;
  HIP_TRY(hipMemcpyAsync(d_t, target, tsz, hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(e0, st));
  {
    auto enqueue_step = [&]() -> hipError_t {
      hipError_t e = hipMemcpyAsync(d_g, h_gin, gcount * 16, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_sf_zero_state, dim3(blk_amp, B), dim3(kSfThreads), 0, st, d_psi, dim);
      for (int k = 0; k < G; ++k)
        hipLaunchKernelGGL(k_sf_apply, dim3(blk_vec, B), dim3(kSfThreads), 0, st, d_psi, dim, d_g, G, k, lo[k], 0);
      hipLaunchKernelGGL(k_sf_dot, dim3(blk_amp, B), dim3(kSfThreads), 0, st, d_t, target_shared ? 1 : 0, d_psi, dim, d_part);
      hipLaunchKernelGGL(k_sf_reduce, dim3(B), dim3(kSfThreads), 0, st, d_part, (int)blk_amp, 1, d_ov, 1, 0);
      hipLaunchKernelGGL(k_sf_copy, dim3(blk_amp, B), dim3(kSfThreads), 0, st, d_phi, d_t, dim, target_shared ? 1 : 0);
      for (int k = G - 1; k >= 0; --k) {
        hipLaunchKernelGGL(k_sf_back, dim3(blk_vec, B), dim3(kSfThreads), 0, st, d_psi, d_phi, dim, d_g, G, k, lo[k], d_part);
        hipLaunchKernelGGL(k_sf_reduce, dim3(B), dim3(kSfThreads), 0, st, d_part, (int)blk_vec, 16, d_env, G * kMat, k);
      }
      if ((e = hipGetLastError()) != hipSuccess) return e;
      if ((e = hipMemcpyAsync(h_ov, d_ov, (size_t)B * 16, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
      return hipMemcpyAsync(h_env, d_env, gcount * 16, hipMemcpyDeviceToHost, st);
    };
    if (graph_on) {
      // (the target copy above must have been issued before the capture begins; a failed capture falls back to plain launches)
      HIP_TRY(hipStreamSynchronize(st));
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const hipError_t e = enqueue_step();
        hipGraph_t g = nullptr;
        const hipError_t e2 = hipStreamEndCapture(st, &g);
        if (e == hipSuccess && e2 == hipSuccess && g && hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0) == hipSuccess) graph = g;
        else {
          if (g) (void)hipGraphDestroy(g);
          gexec = nullptr;
          (void)hipGetLastError();
        }
      } else (void)hipGetLastError();
    }
  for (int it = 0; it < max_iter; ++it) {
    int any = 0;
    for (int b = 0; b < B; ++b) any |= active[b];
    if (!any) break;
    memcpy(h_gin, U.data(), gcount * 16);
    if (gexec) HIP_TRY(hipGraphLaunch(gexec, st));
    else HIP_TRY(enqueue_step());
    HIP_TRY(hipStreamSynchronize(st));
    const double t = jit_frozen ? 1.0 : (double)(it + 1);
    const double lr_t = lr * sqrt(1.0 - pow(beta2, t)) / (1.0 - pow(beta1, t));      // (stiefel_opt.py:333-335)
    for (int b = 0; b < B; ++b) {
      if (!active[b]) continue;
      const double2 o = h_ov[b];
      const double ao = hypot(o.x, o.y), val = 1.0 - ao;
      const double2 ph = make_double2(o.x / ao, o.y / ao);
      double dsum = 0.0;
      for (int k = 0; k < G; ++k) {
        C4 g;      // -(o/|o|) conj(E_k): what step() hands to update() (:107-109)
        for (int e = 0; e < 16; ++e) {
          const double2 E = h_env[((size_t)b * G + k) * kMat + e];
          const double2 v = h_mul(ph, make_double2(E.x, -E.y));
          g.a[e] = make_double2(-v.x, -v.y);
        }
        double dn;
        U[(size_t)b * G + k] = h_update(U[(size_t)b * G + k], g, mom[(size_t)b * G + k], vel[(size_t)b * G + k], jit_frozen != 0, lr_t,
                                        beta1, beta2, eps, &dn);
        dsum += dn;
      }
      // bookkeeping of minimize() (:124-147)
      if (loss_history) loss_history[(size_t)b * max_iter + it] = val;
      nit[b] = it + 1;
      if (val < bv[b]) {
        bv[b] = val;
        for (int k = 0; k < G; ++k) Ubest[(size_t)b * G + k] = U[(size_t)b * G + k];
      }
      if (val < tol || dsum / G < param_tol) active[b] = 0;
    }
  }
  }
  HIP_TRY(hipEventRecord(e1, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (total_ms) *total_ms = ms;
  if (opt_gates) memcpy(opt_gates, Ubest.data(), gcount * 16);
  if (final_gates) memcpy(final_gates, U.data(), gcount * 16);
  if (best_val) memcpy(best_val, bv.data(), B * sizeof(double));
  if (n_iter) for (int b = 0; b < B; ++b) n_iter[b] = nit[b];
  if (last_envs) memcpy(last_envs, h_env, gcount * 16);
  if (last_overlap) memcpy(last_overlap, h_ov, (size_t)B * 16);

done:
  for (void* p : {(void*)d_psi, (void*)d_phi, (void*)d_t, (void*)d_g, (void*)d_part, (void*)d_ov, (void*)d_env}) (void)hipFree(p);
  if (gexec) (void)hipGraphExecDestroy(gexec);
  if (graph) (void)hipGraphDestroy(graph);
  if (h_pin) (void)hipHostFree(h_pin);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
