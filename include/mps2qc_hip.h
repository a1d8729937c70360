/* mps2qc_hip.h - C ABI of the MI355X MPS -> PQC fit (libmps2qc_hip.so).
 *
 * Replaces the offline block of the reference (Aqasch/TensorRL-QAS) that turns a target
 * matrix-product state into the brickwork circuit the RL episodes start from:
 *   dmrg-to-qc/mps2qc.py:242-339      mps_to_qc(): loss 1 - |<mps|qc>|, random SU(4) start,
 *                                     optimizer.minimize(...)
 *   dmrg-to-qc/stiefel_opt.py:91-152  StiefelOptimizer.minimize(): loop, best tracking,
 *                                     tol / param_tol termination
 *   dmrg-to-qc/stiefel_opt.py:257-347 StiefelAdam.update(): Riemannian gradient, Adam
 *                                     moments, Cayley retraction, vector transport
 *   dmrg-to-qc/tnqc_ansatze.py:46-98  brickwork_ansatz(): gate order
 * The reference does this with jax autodiff over a quimb tensor network, one fit at a time; here
 * a whole batch of fits (random restarts and / or different targets) runs in ONE launch, one
 * workgroup per fit, the complete optimisation loop on the device.
 *
 * Conventions (quimb): MPS site 0 is the most significant bit of the dense index; a gate on
 * sites (i, i+1) is a row-major 4x4 complex matrix indexed by 2*s_i + s_{i+1}.  Complex
 * numbers are interleaved (re, im) doubles.  Plain C types only; functions return 0 or a
 * negative errno-style code (same values as vqe_hip.h) and never throw across the ABI.
 */
#ifndef MPS2QC_HIP_H
#define MPS2QC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Largest register that is fitted with the state resident in LDS. */
#define MPS2QC_MAX_QUBITS 12

/* Gate order of brickwork_ansatz (tnqc_ansatze.py:85-95): per layer the even bonds, then the odd
 * bonds.  Writes the first site of every gate to sites[] (capacity cap) and returns the number
 * of gates, or a negative code. */
int mps2qc_brickwork_sites(int n_qubits, int n_layers, int32_t* sites, int cap);

/* Fit `batch` brickwork circuits on device `device_id`; blocks until the results are in host
 * memory.
 *   sites[n_gates]               first site of each gate, in application order
 *   target                       dense target states, [batch][2^n] complex (or one shared
 *                                state when target_shared != 0)
 *   init_gates                   [batch][n_gates][16] complex, unitary (mps2qc.py:296)
 *   lr, beta1, beta2, eps        StiefelAdam.__init__ (stiefel_opt.py:267-277)
 *   jit_frozen                   0: the optimiser as written (moments carried, t = step count);
 *                                1: the optimiser as the reference executes it under jax.jit
 *                                   (moments and t are trace-time constants: m = v = 0, t = 1)
 *   max_iter, tol, param_tol     StiefelOptimizer.minimize (stiefel_opt.py:91-96)
 *   use_mfma                     1: environments E_k = sum_r conj(phi)[a,r] psi[b,r] on the
 *                                matrix cores (v_mfma_f64_16x16x4_f64); 0: vector FMA +
 *                                cross-lane reduction (kept for A/B measurements)
 * Outputs (any may be NULL):
 *   opt_gates    [batch][n_gates][16] complex   optimizer.opt_params (gates AFTER the update of
 *                                               the step whose loss was the best, :133-135)
 *   final_gates  [batch][n_gates][16] complex   gates after the last step
 *   loss_history [batch][max_iter]              optimizer.loss_history (entries >= n_iter unset)
 *   best_val     [batch]                        optimizer.best_val
 *   n_iter       [batch]                        steps executed
 *   last_envs    [batch][n_gates][16] complex   environments of the last executed step
 *   last_overlap [batch] complex                <target|qc> of the last executed step
 *   kernel_ms                                   duration of the launch (HIP events)
 */
int mps2qc_fit_brickwork(int device_id, int n_qubits, int n_gates, const int32_t* sites,
                         int batch, const double* target, int target_shared,
                         const double* init_gates,
                         double lr, double beta1, double beta2, double eps, int jit_frozen,
                         int max_iter, double tol, double param_tol, int use_mfma,
                         double* opt_gates, double* final_gates, double* loss_history,
                         double* best_val, int32_t* n_iter,
                         double* last_envs, double* last_overlap, float* kernel_ms);

/* The same fit with the state vectors in HBM instead of LDS: registers of 2 .. MPS2QC_STREAM_MAX_QUBITS qubits
 * (what the reference reaches beyond a dozen qubits only as a tensor-network contraction, mps2qc.py:242-339; a
 * dense 2^n target is 16 MiB at 20 qubits).  Same arguments, same optimiser and bookkeeping semantics and the
 * same outputs as mps2qc_fit_brickwork (no use_mfma: the 4x4 environments are reduced by a fused backward
 * sweep per gate); the optimiser update runs on the host, so an active fit that has stopped does not hold the
 * others back.  total_ms: duration of the whole loop. */
#define MPS2QC_STREAM_MAX_QUBITS 26
int mps2qc_fit_brickwork_stream(int device_id, int n_qubits, int n_gates, const int32_t* sites,
                                int batch, const double* target, int target_shared,
                                const double* init_gates,
                                double lr, double beta1, double beta2, double eps, int jit_frozen,
                                int max_iter, double tol, double param_tol,
                                double* opt_gates, double* final_gates, double* loss_history,
                                double* best_val, int32_t* n_iter,
                                double* last_envs, double* last_overlap, float* total_ms);

/* Message of the last failure on the calling thread ("" if none). */
const char* mps2qc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
