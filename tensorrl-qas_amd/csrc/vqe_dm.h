// vqe_dm.h - exact channel mode of the noisy path: density-matrix evolution on the matrix cores (n <= 13).
//
// The reference's noisy circuits (environments/VQAs/VQE_qulacs_TN_notin_RL_noise.py:13-54) put a qulacs
// DepolarizingNoise(q, 0.01) behind every rotation and a TwoQubitDepolarizingNoise(c, t, 0.05) behind every CNOT and
// draw ONE Pauli trajectory per update_quantum_state call (:94-101); the trajectory kernels of vqe_device.h do the
// same.  This file evaluates the CHANNEL those draws sample from:
//     rho <- (1 - p) rho + p/3 (X rho X + Y rho Y + Z rho Z)                 (one qubit)
//     rho <- (1 - p) rho + p/15 sum_{P != II} P rho P                        (two qubits)
// on the density matrix rho (4^n complex128 in HBM, flat index = ket | bra << n, both little-endian), and
// E = tr(rho H) over the Pauli terms.  It is the first check of the noisy path that is not "same draws on both
// sides": the mean of the trajectory sampler has to converge to it (tests/test_dm_gpu.py).
//
// Design for the hardware (BASELINE config 5: "batched unitary MFMA kernel"): the host fuses every run of gates and
// channels that stays inside a two-qubit window (a, b) into ONE 16 x 16 complex superoperator S acting on the four
// index bits (ket a, ket b, bra a, bra b); a sweep applies S to all 4^n / 16 groups of 16 entries:
//     [Out_r; Out_i] = [S_r -S_i; S_i S_r] [V_r; V_i]      16 v_mfma_f64_16x16x4_f64 per 16 groups and wavefront,
// 4 KiB in and 4 KiB out per 1024 matrix-pipe cycles = 19 TB/s over the chip: the sweep is HBM bound (n = 12: 268 MB
// read + 268 MB written per block) with the matrix pipe 40 % busy, where the vector pipe would need 8 flop per byte.
// A lane loads the four entries (e = q + 4 c, q = lane >> 4) of its group that it also stores (the D rows of the
// f64 MFMA are q + 4 reg): in place, no LDS, no exchange.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vqe {

typedef double dm_d4 __attribute__((ext_vector_type(4)));

struct DmBlockArgs {
  double2* rho;
  const double* S;      // [2][16][16]: real part, imaginary part, row-major S[m][k]
  int n;                // qubits
  int a, b;             // window qubits: entry bit 0 = ket a, 1 = ket b, 2 = bra a, 3 = bra b
  int hole[4];          // the four index bits of the window, ascending
  uint32_t n_groups;    // 4^n / 16
};

__global__ void __launch_bounds__(256) k_dm_init(double2* __restrict__ rho, const double2* __restrict__ psi0, int n) {
  const size_t total = (size_t)1 << (2 * n);
  const uint32_t mask = (1u << n) - 1u;
  for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (size_t)gridDim.x * 256) {
    const double2 k = psi0[(uint32_t)f & mask], bq = psi0[(uint32_t)(f >> n)];
    rho[f] = make_double2(k.x * bq.x + k.y * bq.y, k.y * bq.x - k.x * bq.y);      // psi_i conj(psi_j)
  }
}

__global__ void __launch_bounds__(256) k_dm_block(DmBlockArgs A) {
  const int lane = threadIdx.x & 63, q = lane >> 4, col = lane & 15;
  // A operands: S[m = lane & 15][k = 4 c + q] (A[i][k]: lane = i + 16 k)
  double sr[4], si[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    sr[c] = A.S[col * 16 + 4 * c + q];
    si[c] = A.S[256 + col * 16 + 4 * c + q];
  }
  // the four entries of a group this lane loads AND stores: e = q + 4 c
  uint32_t eoff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t e = (uint32_t)(q + 4 * c);
    eoff[c] = ((e & 1u) << A.a) | (((e >> 1) & 1u) << A.b) | (((e >> 2) & 1u) << (A.a + A.n)) | (((e >> 3) & 1u) << (A.b + A.n));
  }
  const uint32_t n_tiles = (A.n_groups + 15u) >> 4;
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = gridDim.x * 4u;
  for (uint32_t t = wave; t < n_tiles; t += n_waves) {
    const uint32_t g = t * 16u + (uint32_t)col;
    const bool live = g < A.n_groups;
    uint32_t idx = live ? g : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // a zero at every window bit, lowest first
      const int hb = A.hole[i];
      idx = ((idx >> hb) << (hb + 1)) | (idx & ((1u << hb) - 1u));
    }
    double2 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = live ? A.rho[idx | eoff[c]] : make_double2(0.0, 0.0);
    dm_d4 accr = {0.0, 0.0, 0.0, 0.0}, acci = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {      // B[k][n]: lane = n + 16 k holds entry k = 4 c + q of group n
      accr = __builtin_amdgcn_mfma_f64_16x16x4f64(sr[c], v[c].x, accr, 0, 0, 0);
      acci = __builtin_amdgcn_mfma_f64_16x16x4f64(si[c], v[c].x, acci, 0, 0, 0);
      accr = __builtin_amdgcn_mfma_f64_16x16x4f64(-si[c], v[c].y, accr, 0, 0, 0);
      acci = __builtin_amdgcn_mfma_f64_16x16x4f64(sr[c], v[c].y, acci, 0, 0, 0);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) A.rho[idx | eoff[r]] = make_double2(accr[r], acci[r]);   // D row = q + 4 reg
    }
  }
}

// E = Re sum_x sum_i rho[i, i ^ x] D_x(i),  D_x(i) = sum_{k in x} c_k (-1)^{popc(i & z_k)}  (c_k incl. i^{#Y}):
// one thread per ket index, per-block partials, fixed-order second pass (bitwise reproducible).
__global__ void __launch_bounds__(256) k_dm_energy(const double2* __restrict__ rho, int n, int n_groups,
                                                  const uint32_t* __restrict__ gx, const int32_t* __restrict__ term_off,
                                                  const uint32_t* __restrict__ term_z, const double* __restrict__ cr,
                                                  const double* __restrict__ ci, double* __restrict__ partial) {
  __shared__ double red[4];
  const uint32_t dim = 1u << n, i = blockIdx.x * 256u + threadIdx.x;
  double acc = 0.0;
  if (i < dim) {
    for (int g = 0; g < n_groups; ++g) {
      const uint32_t x = gx[g];
      const double2 r = rho[(size_t)i | ((size_t)(i ^ x) << n)];
      double dr = 0.0, di = 0.0;
      for (int k = term_off[g]; k < term_off[g + 1]; ++k) {
        const bool neg = __popc(i & term_z[k]) & 1;
        dr += neg ? -cr[k] : cr[k];
        di += neg ? -ci[k] : ci[k];
      }
      acc += r.x * dr - r.y * di;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void k_dm_sum(const double* __restrict__ partial, int n_blocks, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += partial[b];
    *out = s;
  }
}

}  // namespace vqe
