/* vqe_hip.h - C ABI of the MI355X VQE environment-step engine (libvqe_hip.so).
 *
 * The reference (Aqasch/TensorRL-QAS) has no FFI: its hot path is Python calling the
 * third-party qulacs / numpy / scipy packages.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference checkout).  Plain C types only;
 * every function returns 0 on success or a negative VQE_E* code and never throws or
 * aborts across the ABI; vqe_last_error() returns the message of the last failure.
 *
 * Conventions (identical to the reference's simulator, qulacs):
 *   - qubit k is bit k of the basis-state index (little-endian);
 *   - amplitudes are complex128, interleaved (re, im);
 *   - rotations are exp(+i*theta/2*P)  (qulacs add_parametric_R{X,Y,Z}_gate);
 *   - Pauli terms are (xmask, zmask, coeff): bit k of xmask set iff the factor on qubit k
 *     is X or Y, bit k of zmask set iff it is Z or Y; the i^{#Y} phase is applied by the
 *     callee.
 * A handle is not thread-safe; distinct handles are independent.  All work of a handle is
 * issued on one HIP stream (vqe_set_stream); calls that return results in host memory
 * block until those results are ready.
 */
#ifndef VQE_HIP_H
#define VQE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vqe_handle vqe_t;
typedef struct vqe_cobyla vqe_cobyla_t;

enum {
  VQE_OK = 0,
  VQE_EINVAL = -22,   /* bad argument */
  VQE_ENOMEM = -12,   /* host or device allocation failed */
  VQE_ENODEV = -19,   /* no usable HIP device */
  VQE_EHIP = -5,      /* a HIP runtime call failed */
  VQE_ESTATE = -1     /* call sequence error (e.g. no circuit / Hamiltonian set) */
};

/* gate kinds of a circuit description (order = reference construct_ansatz order,
 * environments/VQAs/VQE_qulacs_TN_notin_RL.py:13-45; noise kinds:
 * VQE_qulacs_TN_notin_RL_noise.py:26-28,40-50) */
enum {
  VQE_GATE_CNOT = 0,   /* q0 = control, q1 = target                       */
  VQE_GATE_RX = 1,     /* q0 = qubit, param_idx = index into theta        */
  VQE_GATE_RY = 2,
  VQE_GATE_RZ = 3,
  VQE_GATE_DEPOL1 = 4, /* DepolarizingNoise(q0, p1)                       */
  VQE_GATE_DEPOL2 = 5  /* TwoQubitDepolarizingNoise(q0, q1, p2)           */
};

/* ---- lifetime ------------------------------------------------------------------------
 * replaces: qulacs.QuantumState(n) / ParametricQuantumCircuit(n) construction,
 * environments/VQAs/VQE_qulacs_TN_notin_RL.py:10,82 */
int vqe_create(int n_qubits, int device_id, vqe_t** out);
void vqe_destroy(vqe_t* h);
const char* vqe_last_error(const vqe_t* h); /* h may be NULL: last error of vqe_create */
int vqe_set_stream(vqe_t* h, void* hip_stream); /* NULL: the handle's own stream */
/* the stream all work of the handle is issued on (its own one unless vqe_set_stream replaced it):
 * lets the caller order its own streams against it (e.g. torch.cuda.ExternalStream + wait_stream
 * before an RCCL all-reduce of vqe_batch_copy_energy's destination) */
int vqe_get_stream(vqe_t* h, void** hip_stream);
int vqe_sync(vqe_t* h);
/* device facts for the caller's roofline arithmetic: [0]=CU count, [1]=LDS bytes/CU,
 * [2]=workgroups resident per CU for the last LDS-path launch, [3]=1 if the LDS-resident
 * path serves this n_qubits else 0 */
int vqe_device_info(vqe_t* h, int64_t info[4]);

/* ---- problem definition --------------------------------------------------------------
 * replaces: state.load(TN_state)            VQE_qulacs_TN_notin_RL.py:83
 *           (NULL = |0...0>, the VQE_qulacs.py:81 path) */
int vqe_set_init_state(vqe_t* h, const double* amps_re_im /* 2 * 2^n, or NULL */);
/* The same with the amplitudes in DEVICE memory (complex128, 2^n, e.g. a torch tensor): an asynchronous
 * device-to-device copy on the handle's stream.  vqe_get_state_dev is vqe_get_state with a device destination.
 * Used by the amplitude-sharded states (tensorrl-qas_amd/parallel.py), where a rank's shard of a larger register is
 * the "state" of an (n - log2 world)-qubit handle and never leaves the GPU between two exchanges. */
int vqe_set_init_state_dev(vqe_t* h, const void* dev_amps_re_im);
int vqe_get_state_dev(vqe_t* h, const double* theta, void* dev_amps_re_im);
/* replaces: the dense operator handed to get_exp_val (VQE_qulacs_TN_notin_RL.py:80,86;
 * built at environment_qulacs_TN_notin_agent.py:126-131,162) by its Pauli-sum form */
int vqe_set_hamiltonian_pauli(vqe_t* h, int n_terms, const uint64_t* xmask,
                              const uint64_t* zmask, const double* coeff);
/* The same operator as the reference hands it to get_exp_val: a dense 2^n x 2^n complex matrix in the
 * simulator's little-endian basis, row-major, (re, im) interleaved - i.e. the product of
 * Operator(H).reverse_qargs().to_matrix() (environment_qulacs_TN_notin_agent.py:162) on the fixed path, the raw
 * file matrix on the trainable path.  Decomposed into Pauli terms on the host (one Walsh-Hadamard transform
 * per X mask; coefficients below tol * max|H_ij| are dropped; n <= 13).  vqe_hamiltonian_terms reports what
 * was found. */
int vqe_set_hamiltonian_dense(vqe_t* h, const double* op_re_im /* 2 * 4^n */, double tol);
int vqe_hamiltonian_terms(vqe_t* h, int32_t* n_terms, int32_t* n_xgroups);
/* How the LDS-resident kernels hold this handle's share of the Hamiltonian (diagnostic; n <= 13):
 * out[0] = X-mask groups evaluated through full sign-sum tables, out[1] = units - sub-cubes of
 * one pair per thread on which the table of a mostly-zero group (a fermionic excitation operator
 * connects one occupation pattern in 2^w) does not vanish; groups stored as units are not in
 * out[0] -, out[2] = groups of out[0] whose partner index sits in the register bits, out[3] = 1
 * if there is a diagonal group.  The reference has no counterpart: its get_exp_val multiplies
 * by the dense matrix (VQE_qulacs_TN_notin_RL.py:86). */
int vqe_hamiltonian_layout(vqe_t* h, int32_t out[4]);
/* Evaluate only the X-mask groups owned by `rank` of `world` (Pauli-term sharding; the
 * caller sums the partial energies of all ranks, e.g. one RCCL all-reduce). */
int vqe_set_term_shard(vqe_t* h, int rank, int world);
/* Streaming path (n >= 14) only: sweep slice `rank` of `world` equal slices of the amplitude
 * index range for ALL terms (the other partition of the double sum over terms and basis
 * states).  Every rank still applies the whole circuit to its own copy of the state; the
 * partial energies are summed by the caller exactly as for term sharding.  Work and memory
 * traffic of the reduction are 1/world per rank. */
int vqe_set_amplitude_shard(vqe_t* h, int rank, int world);
/* The one collective of the term-sharded sum as a library call (RCCL over xGMI; librccl is opened lazily).  Rank 0
 * makes the 128-byte id (vqe_comm_unique_id) and hands it to the other ranks by any means; every rank calls
 * vqe_comm_init on its handle; vqe_comm_allreduce_energy sums the batch's energy array (float64[batch], what
 * vqe_batch_run_energy left on the device and vqe_batch_fetch returns) over the ranks, in place, asynchronously on
 * the handle's stream.  The reference has no counterpart (single process). */
int vqe_comm_unique_id(void* id128);
int vqe_comm_init(vqe_t* h, int rank, int world, const void* id128);
int vqe_comm_allreduce_energy(vqe_t* h);
int vqe_comm_destroy(vqe_t* h);
/* host only: the rank that vqe_set_term_shard(.., world) makes responsible for each term
 * (terms sharing an X mask stay together); needs no device */
int vqe_term_owner(int n_qubits, int n_terms, const uint64_t* xmask, int world, int32_t* owner);
/* Stochastic Pauli noise (qulacs DepolarizingNoise / TwoQubitDepolarizingNoise
 * semantics: each non-identity Pauli with probability p/3 resp. p/15, one trajectory per
 * evaluation).  The draw for (stream, evaluation, gate) is a pure function of `seed`. */
int vqe_set_noise(vqe_t* h, double p1, double p2, uint64_t seed);
/* How the noise gates of a circuit are evaluated.  0 (default): one Pauli trajectory per evaluation, as qulacs does
 * inside update_quantum_state (VQE_qulacs_TN_notin_RL_noise.py:94-101).  1: the exact channel those draws sample
 * from - density-matrix evolution rho -> (1-p) rho + p/3 sum_P P rho P (two qubits: p/15 over the 15 non-identity
 * Paulis), E = tr(rho H); 2 <= n <= 13 (4^n complex128 in HBM), runs of gates inside a two-qubit window fused into
 * 16 x 16 superoperators applied with FP64 MFMA; vqe_energy*, vqe_batch_run_energy, vqe_minimize_cobyla,
 * vqe_batch_run_minimize / _env_step (COBYLA then driven by the host on the exact energies).  The seed of
 * vqe_set_noise and vqe_set_shot_noise play no part in mode 1. */
int vqe_set_noise_mode(vqe_t* h, int mode);
/* host only (needs no device): the superoperator blocks the exact channel mode would sweep for this circuit - windows[2k],
 * windows[2k+1] = the two qubits (a, b) of block k, S[k][2][16][16] = real and imaginary part of its 16 x 16 matrix, entry
 * index = ket_a + 2 ket_b + 4 bra_a + 8 bra_b.  windows == NULL or S == NULL: only the count.  For tests of the fusion. */
int vqe_dm_plan(int n_qubits, int n_gates, const int32_t* kind, const int32_t* q0, const int32_t* q1, const int32_t* param_idx,
                const double* theta, double p1, double p2, int cap_blocks, int32_t* n_blocks, int32_t* windows, double* S);
/* out[0] = the mode, out[1] = superoperator blocks (sweeps over rho) of the last exact-mode evaluation; for the
 * caller's roofline arithmetic (a sweep reads and writes 4^n complex128) */
int vqe_noise_mode_info(vqe_t* h, int32_t out[2]);
/* Finite-shot model of the reference's restricted variant
 * (environments/VQAs/VQE_qulacs_TN_notin_RL_noise_restricted.py:47-48,84-96): every evaluation
 * returns E + weights . N(0, sigma^2 I), sigma = n_shots^-1/2, i.e. E + sigma_total * N(0,1)
 * with sigma_total = sigma * |weights|_2 (the same distribution, one draw per evaluation from
 * the seeded counter-based generator).  0 switches it off. */
int vqe_set_shot_noise(vqe_t* h, double sigma_total, uint64_t seed);

/* ---- one circuit ---------------------------------------------------------------------
 * replaces: Parametric_Circuit.construct_ansatz product (the qulacs circuit handle),
 * VQE_qulacs_TN_notin_RL.py:13-45 */
int vqe_set_circuit(vqe_t* h, int n_gates, const int32_t* kind, const int32_t* q0,
                    const int32_t* q1, const int32_t* param_idx, int n_params);
/* replaces: get_energy_qulacs / get_exp_val, VQE_qulacs_TN_notin_RL.py:48-87 */
int vqe_energy(vqe_t* h, const double* theta, double* energy);
int vqe_energy_batch(vqe_t* h, int batch, const double* theta /* batch x n_params */,
                     double* energy /* batch */);
/* replaces: circuit.update_quantum_state(state); state.get_vector()  (:84-85) */
int vqe_get_state(vqe_t* h, const double* theta, double* amps_re_im /* 2 * 2^n */);
/* replaces: scipy.optimize.minimize(cost, x0, method='COBYLA', options={'maxiter': m})
 * at environment_qulacs_TN_notin_agent.py:478 (scipy 1.15: rhobeg=1.0, rhoend=1e-4,
 * maxfun=m).  The whole loop (simplex algebra + evaluations) runs on the device. */
int vqe_minimize_cobyla(vqe_t* h, const double* x0, double rhobeg, double rhoend,
                        int maxfun, double* x, double* f, int32_t* nfev);

/* ---- batches of independent circuits (one per parallel environment / RL seed) ---------
 * Circuits are concatenated; circuit b owns gates [gate_off[b], gate_off[b+1]) and
 * parameters [par_off[b], par_off[b+1]); param_idx is local to the circuit.
 * vqe_batch_load copies the description (and x0 / theta) into device memory; the
 * *_run calls only launch device work on resident data and never modify it (a run can be
 * repeated); vqe_batch_fetch copies results back.  replaces: B independent CircuitEnv.scipy_optim / get_energy calls
 * (environment_qulacs_TN_notin_agent.py:442-482). */
int vqe_batch_load(vqe_t* h, int batch, const int64_t* gate_off, const int32_t* kind,
                   const int32_t* q0, const int32_t* q1, const int32_t* param_idx,
                   const int64_t* par_off, const double* theta0);
int vqe_batch_run_energy(vqe_t* h);
int vqe_batch_run_minimize(vqe_t* h, double rhobeg, double rhoend, int maxfun);
/* streaming path (n >= 14): redo only the Pauli-term reduction <psi|H_shard|psi> on the
 * states left by the previous vqe_batch_run_energy (for timing the sharded reduction) */
int vqe_batch_run_reduction(vqe_t* h);
/* One CircuitEnv.step() worth of arithmetic per circuit, in ONE launch
 * (environment_qulacs_TN_notin_agent.py:283-291): new_gate[b] is the index, inside circuit
 * b, of the gate the RL action just added (-1: none).  COBYLA runs on the circuit WITHOUT
 * that gate from x0 = theta0 (the reference optimises the pre-action state, :453); the
 * optimum is rounded to float32 (state-tensor dtype, :480) and the energy of the FULL
 * circuit at those angles is returned in f (the extra get_energy() of :291).  x receives
 * all n_params angles (the new rotation keeps its theta0 value), nfev COBYLA's count.
 * A noise gate that directly follows gate new_gate[b] and acts on the same qubit(s) is the
 * channel construct_ansatz attached to it (VQE_qulacs_TN_notin_RL_noise.py:26-28,40-50) and is
 * left out of the optimised circuit with it; the noise draws of the COBYLA phase are numbered by
 * gate position in that pre-action circuit, those of the final evaluation by position in the
 * full one.  n <= 13: one fused launch; n >= 14 (streaming path): the same steps with the COBYLA loop
 * driven by the host over batched evaluations. */
int vqe_batch_set_new_gate(vqe_t* h, const int32_t* new_gate /* batch, or NULL */);
int vqe_batch_run_env_step(vqe_t* h, double rhobeg, double rhoend, int maxfun);
int vqe_batch_fetch(vqe_t* h, double* x /* sum of n_params, may be NULL */,
                    double* f /* batch */, int32_t* nfev /* batch, may be NULL */);
/* the optimiser's result before the float32 rounding of vqe_batch_run_env_step
 * (scipy's result.x, stored by the reference as env.opt_ang_save, :288); same layout as x */
int vqe_batch_fetch_xopt(vqe_t* h, double* x /* sum of n_params */);
/* device pointer to the batch's f / energy array (float64[batch]) for on-device
 * reductions by the caller (e.g. torch.distributed all_reduce over RCCL) */
int vqe_batch_energy_devptr(vqe_t* h, void** dev_ptr);
/* asynchronous device-to-device copy of that array into caller-owned device memory
 * (e.g. a torch tensor) on the handle's stream */
int vqe_batch_copy_energy(vqe_t* h, void* dst_dev /* float64[batch] */);
/* Diagnostic: record (f, x[0..P)) of every COBYLA evaluation of the next vqe_batch_run_minimize /
 * vqe_batch_run_env_step launches (LDS-resident path), for trajectory-level comparison with
 * scipy's callback sequence (environment_qulacs_TN_notin_agent.py:464-468,478).
 * vqe_batch_fetch_trace copies circuit b's records: out[k * stride] = f of evaluation k + 1,
 * out[k * stride + 1 + j] = its trial point (the optimised parameters only: the new gate's angle is
 * not among them), zero beyond nfev; out == NULL only queries maxfun / stride. */
int vqe_batch_set_trace(vqe_t* h, int enable);
int vqe_batch_fetch_trace(vqe_t* h, int circuit, double* out /* maxfun * stride, or NULL */,
                          int32_t* maxfun, int32_t* stride);
/* diagnostic builds only (-DVQE_STAMPS): [0] evaluations, [1] cycles in the circuit phase,
 * [2] in the energy phase, [3] in the optimiser update, summed over workgroups; read and
 * cleared.  All zero in the shipped library (no stamp executes there). */
int vqe_debug_counters(vqe_t* h, uint64_t out[8]);
/* kernel time of the last *_run call measured with HIP events on the handle's stream */
int vqe_last_kernel_ms(vqe_t* h, float* ms);

/* ---- host-side COBYLA (ask/tell) -------------------------------------------------------
 * Same algorithm as the device loop; used when every evaluation needs a collective
 * (Pauli-term sharding across GPUs) or by a caller with its own cost function.
 * replaces: scipy.optimize.minimize(..., method='COBYLA') as above. */
int vqe_cobyla_create(int n, const double* x0, double rhobeg, double rhoend, int maxfun,
                      vqe_cobyla_t** out);
/* returns 1 and fills x[n] when f(x) is wanted, 0 when finished, <0 on error */
int vqe_cobyla_ask(vqe_cobyla_t* c, double* x);
int vqe_cobyla_tell(vqe_cobyla_t* c, double f);
int vqe_cobyla_result(vqe_cobyla_t* c, double* x, double* f, int32_t* nfev, int32_t* status);
void vqe_cobyla_destroy(vqe_cobyla_t* c);

#ifdef __cplusplus
}
#endif
#endif /* VQE_HIP_H */
