/* vqe_oracle.c - plain C restatement of the reference's CPU hot path.  TEST INFRASTRUCTURE
 * ONLY (parity checker and the timed CPU baseline of bench.py); the product never links or
 * loads it.  "Parity unpinned" by the reference's own tests (it has none); pinned instead
 * by the known answers of SURVEY.md section 8c (tests/test_oracle.py).
 *
 * Follows the reference literally, one evaluation =
 *   state = QuantumState(n); state.load(TN_state)        VQE_qulacs_TN_notin_RL.py:82-83
 *   circuit.update_quantum_state(state)                   :84   (G in-place gate sweeps)
 *   psi = state.get_vector()                              :85
 *   (conj(psi).T @ op @ psi).real                         :86   (dense 2^n x 2^n operator)
 * qulacs (third-party, unpinned, requirements.txt:1) conventions: little-endian qubits,
 * R{X,Y,Z}(theta) = exp(+i theta/2 P), CNOT(control, target).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

enum { G_CNOT = 0, G_RX = 1, G_RY = 2, G_RZ = 3, G_DEPOL1 = 4, G_DEPOL2 = 5 };

static void apply_1q(cplx* psi, size_t dim, int q, cplx m00, cplx m01, cplx m10, cplx m11) {
  const size_t bit = (size_t)1 << q;
  for (size_t i = 0; i < dim; ++i) {
    if (i & bit) continue;
    const cplx a = psi[i], b = psi[i | bit];
    psi[i] = m00 * a + m01 * b;
    psi[i | bit] = m10 * a + m11 * b;
  }
}

static void apply_pauli(cplx* psi, size_t dim, int q, int p) {
  if (p == 1) apply_1q(psi, dim, q, 0, 1, 1, 0);
  else if (p == 2) apply_1q(psi, dim, q, 0, -I, I, 0);
  else if (p == 3) apply_1q(psi, dim, q, 1, 0, 0, -1);
}

/* psi_out <- U_G ... U_1 psi0 ; noise_draw may be NULL (no Pauli errors applied) */
void orc_run_circuit(int n, const double* psi0, int n_gates, const int32_t* kind, const int32_t* q0,
                     const int32_t* q1, const int32_t* pidx, const double* theta,
                     const int32_t* noise_draw, double* psi_out) {
  const size_t dim = (size_t)1 << n;
  cplx* psi = (cplx*)psi_out;
  memcpy(psi, psi0, dim * sizeof(cplx)); /* state.load(TN_state) */
  for (int g = 0; g < n_gates; ++g) {
    const int k = kind[g];
    if (k == G_CNOT) {
      const size_t cb = (size_t)1 << q0[g], tb = (size_t)1 << q1[g];
      for (size_t i = 0; i < dim; ++i)
        if ((i & cb) && !(i & tb)) { const cplx t = psi[i]; psi[i] = psi[i | tb]; psi[i | tb] = t; }
    } else if (k >= G_RX && k <= G_RZ) {
      const double c = cos(0.5 * theta[pidx[g]]), s = sin(0.5 * theta[pidx[g]]);
      if (k == G_RX) apply_1q(psi, dim, q0[g], c, I * s, I * s, c);
      else if (k == G_RY) apply_1q(psi, dim, q0[g], c, s, -s, c);
      else apply_1q(psi, dim, q0[g], c + I * s, 0, 0, c - I * s);
    } else if (k == G_DEPOL1) {
      if (noise_draw) apply_pauli(psi, dim, q0[g], noise_draw[g]);
    } else if (k == G_DEPOL2) {
      if (noise_draw) { apply_pauli(psi, dim, q0[g], noise_draw[g] & 3); apply_pauli(psi, dim, q1[g], noise_draw[g] >> 2); }
    }
  }
}

/* (conj(psi).T @ op @ psi).real with op row-major dense: t = conj(psi) @ op, then t @ psi */
double orc_energy_dense(int n, const double* psi_, const double* op_) {
  const size_t dim = (size_t)1 << n;
  const cplx* psi = (const cplx*)psi_;
  const cplx* op = (const cplx*)op_;
  cplx* t = (cplx*)calloc(dim, sizeof(cplx));
  for (size_t i = 0; i < dim; ++i) {
    const cplx ci = conj(psi[i]);
    const cplx* row = op + i * dim;
    for (size_t j = 0; j < dim; ++j) t[j] += ci * row[j];
  }
  cplx e = 0;
  for (size_t j = 0; j < dim; ++j) e += t[j] * psi[j];
  free(t);
  return creal(e);
}

/* sum_k w_k <psi|P_k|psi>,  P|i> = i^{#Y} (-1)^{popc(i & z)} |i ^ x> */
double orc_energy_pauli(int n, const double* psi_, int n_terms, const uint64_t* xmask,
                        const uint64_t* zmask, const double* coeff) {
  const size_t dim = (size_t)1 << n;
  const cplx* psi = (const cplx*)psi_;
  static const cplx ipow[4] = {1, I, -1, -I};
  double e = 0.0;
  for (int k = 0; k < n_terms; ++k) {
    const uint64_t x = xmask[k], z = zmask[k];
    cplx acc = 0;
    for (size_t i = 0; i < dim; ++i) {
      const cplx v = conj(psi[i ^ x]) * psi[i];
      acc += (__builtin_popcountll(i & z) & 1) ? -v : v;
    }
    e += coeff[k] * creal(acc * ipow[__builtin_popcountll(x & z) & 3]);
  }
  return e;
}

/* one reference evaluation: circuit from psi0, then dense (op != NULL) or Pauli-sum energy */
double orc_evaluate(int n, const double* psi0, int n_gates, const int32_t* kind, const int32_t* q0,
                    const int32_t* q1, const int32_t* pidx, const double* theta, const double* op_dense,
                    int n_terms, const uint64_t* xmask, const uint64_t* zmask, const double* coeff) {
  const size_t dim = (size_t)1 << n;
  double* psi = (double*)malloc(dim * 2 * sizeof(double)); /* QuantumState(n) */
  orc_run_circuit(n, psi0, n_gates, kind, q0, q1, pidx, theta, NULL, psi);
  const double e = op_dense ? orc_energy_dense(n, psi, op_dense)
                            : orc_energy_pauli(n, psi, n_terms, xmask, zmask, coeff);
  free(psi);
  return e;
}

/* splitmix-style draw shared BY SPECIFICATION with the device (a pure function of
 * seed, stream, evaluation, gate): restated here, not shared code. */
static uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
double orc_noise_uniform(uint64_t seed, uint64_t b, uint64_t e, uint64_t g) {
  uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ull * (b + 1));
  k = mix64(k ^ (e * 0xBF58476D1CE4E5B9ull));
  k = mix64(k ^ (g * 0x94D049BB133111EBull));
  return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}
/* standard normal for (stream, evaluation): Box-Muller on gate slots 2^40 and 2^40 + 1 */
double orc_noise_gauss(uint64_t seed, uint64_t b, uint64_t e) {
  const double u1 = orc_noise_uniform(seed, b, e, (uint64_t)1 << 40);
  const double u2 = orc_noise_uniform(seed, b, e, ((uint64_t)1 << 40) + 1);
  return sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
}
/* qulacs DepolarizingNoise / TwoQubitDepolarizingNoise draw for every gate of a circuit */
void orc_noise_draws(uint64_t seed, uint64_t stream, uint64_t eval, int n_gates, const int32_t* kind,
                     double p1, double p2, int32_t* draw) {
  for (int g = 0; g < n_gates; ++g) {
    draw[g] = 0;
    if (kind[g] == G_DEPOL1) {
      const double u = orc_noise_uniform(seed, stream, eval, (uint64_t)g);
      if (u < p1) draw[g] = 1 + (int)(u / p1 * 3.0);
    } else if (kind[g] == G_DEPOL2) {
      const double u = orc_noise_uniform(seed, stream, eval, (uint64_t)g);
      if (u < p2) draw[g] = 1 + (int)(u / p2 * 15.0);
    }
  }
}

/* n_traj stochastic evaluations of one circuit: trajectory t uses the draws of
 * (seed, stream0 + t, eval) and returns sum_k w_k <psi_t|P_k|psi_t>.  Plain loop over the
 * functions above (OpenMP over trajectories when built with -fopenmp): the checker for the
 * distributional test of the noisy 12-qubit configuration. */
void orc_noisy_energies(int n, const double* psi0, int n_gates, const int32_t* kind, const int32_t* q0,
                        const int32_t* q1, const int32_t* pidx, const double* theta, int n_terms,
                        const uint64_t* xmask, const uint64_t* zmask, const double* coeff, uint64_t seed,
                        uint64_t stream0, int n_traj, uint64_t eval, double p1, double p2, double* out) {
  const size_t dim = (size_t)1 << n;
#pragma omp parallel
  {
    double* psi = (double*)malloc(dim * 2 * sizeof(double));
    int32_t* draw = (int32_t*)malloc((size_t)(n_gates > 0 ? n_gates : 1) * sizeof(int32_t));
#pragma omp for schedule(static)
    for (int t = 0; t < n_traj; ++t) {
      orc_noise_draws(seed, stream0 + (uint64_t)t, eval, n_gates, kind, p1, p2, draw);
      orc_run_circuit(n, psi0, n_gates, kind, q0, q1, pidx, theta, draw, psi);
      out[t] = orc_energy_pauli(n, psi, n_terms, xmask, zmask, coeff);
    }
    free(psi);
    free(draw);
  }
}
