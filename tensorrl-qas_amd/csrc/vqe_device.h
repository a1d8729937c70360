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
//    Hamiltonian, L2 resident), using the p <-> p^x pair symmetry; table values stream
//    through a 4-group-deep register ring so that L2 latency is hidden; one block reduction
//    per evaluation;
//  * the COBYLA loop of the reference (scipy, sequential callbacks) runs inside the same
//    workgroup (cobyla_m0.h), so an environment step is ONE kernel launch for all
//    environments and there is no host round trip per evaluation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifdef VQE_STAMPS   // diagnostic build: cycles between marked points of the device-side COBYLA
__device__ unsigned long long g_cby_dbg[8];
__device__ unsigned long long g_cby_in_out[2];
__device__ long long g_cby_t0_unused;
#if defined(__HIP_DEVICE_COMPILE__)
#define CBY_STAMP_RESET() long long cby_t0_ = (long long)__builtin_readcyclecounter(); (void)cby_t0_
#define CBY_STAMP(k)                                                                               \
  {                                                                                                \
    const long long t_ = (long long)__builtin_readcyclecounter();                                  \
    if (ctx.tid == 0 && Ctx::nth > 1) __hip_atomic_fetch_add(&g_cby_dbg[k], (unsigned long long)(t_ - cby_t0_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    cby_t0_ = (long long)__builtin_readcyclecounter();                                             \
  }
#else
#define CBY_STAMP_RESET()
#define CBY_STAMP(k)
#endif
#endif
#include "cobyla_m0.h"

namespace vqe {

constexpr int kThreads = 256;   // default workgroup size (n <= 11 and the streaming path)

// Workgroup geometry of the LDS-resident kernels.
//  * n <= 11: 256 threads, registers capped for 4 waves per SIMD (4 workgroups per CU fit the
//    LDS): measured at n = 11 +31 % over 2 waves per SIMD although the cap costs spills - the
//    vector-memory, LDS and VALU pipes of the energy step overlap better across more waves;
//  * n = 12: 256 threads x 16 amplitudes, 2 workgroups per CU (LDS bound), 256 VGPRs.  The
//    512-thread variant (8 amplitudes per thread, 4 waves per SIMD in 128 VGPRs) was measured
//    15 % slower: 124 spilled registers and one more re-layout per ~3 rotations;
//  * n = 13: 512 threads x 16 amplitudes, one workgroup per CU.
#ifndef VQE_WIDE_MIN
#define VQE_WIDE_MIN 13
#endif
constexpr int kWideMinQubits = VQE_WIDE_MIN;   // 512-thread workgroups from this size on

#ifndef VQE_ONE_WAVE_MAX
#define VQE_ONE_WAVE_MAX 9
#endif
// Up to this size an environment is ONE wavefront (64 threads, 4 amplitudes per thread at 8 qubits).  With four waves
// per environment three of them sit at a barrier while wave 0 runs the optimiser update, and at this size that update
// is most of an evaluation: one-wave workgroups keep 16 environments per CU busy instead of 4 (8 qubits, 20 gates:
// 46.9 -> 92.4 M evaluations/s; 150 gates: 12.9 -> 13.2 M; 129 variables: 6.3 -> 5.8 M at 4096 environments, 6.4 M
// at 16384 - the price of having no second wave for the workgroup-wide update).
constexpr int kOneWaveMaxQubits = VQE_ONE_WAVE_MAX;

#ifndef VQE_ONE_WAVE_REG
#define VQE_ONE_WAVE_REG 0      // 1: 10 qubits on the register path with one wave (16 amplitudes per thread) - parity green, 0..9 % slower than four waves x 4 amplitudes
#endif
__host__ __device__ constexpr bool geo_one_wave(int n) { return n <= kOneWaveMaxQubits || (VQE_ONE_WAVE_REG && n == 10); }
__host__ __device__ constexpr int geo_lt(int n) { return n >= kWideMinQubits ? 9 : (geo_one_wave(n) ? 6 : 8); }

#ifndef VQE_WPS_SMALL
#define VQE_WPS_SMALL 2     // n <= 9 (one wave per environment): all 256 registers - at 128 these kernels spilled 470..690 B per lane; eight environments per CU without spills beat sixteen with them by 18..32 %
#endif
#ifndef VQE_WPS10
#define VQE_WPS10 4
#endif
#ifndef VQE_WPS11
#define VQE_WPS11 3     // n = 11: 170 registers per wave instead of 128 (464 B of spills), three workgroups per CU (the LDS rarely admits a fourth): +2..3 %
#endif
template <int N>
struct Geo {
  static constexpr int NT = 1 << geo_lt(N);        // threads per workgroup
  static constexpr int LT = geo_lt(N);             // log2(NT)
  static constexpr int NW = NT / 64;               // waves per workgroup
  static constexpr int WPS = N == 11 ? VQE_WPS11 : (N == 10 ? (VQE_ONE_WAVE_REG ? 2 : VQE_WPS10) : (N <= 9 ? VQE_WPS_SMALL : 2));      // waves per SIMD asked of the register allocator
};

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef double d2v_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const unsigned char lds_cbyte;
typedef __attribute__((address_space(3))) double lds_double;

enum : int { G_CNOT = 0, G_RX = 1, G_RY = 2, G_RZ = 3, G_DEPOL1 = 4, G_DEPOL2 = 5 };
enum : int { OP_RX = 1, OP_RY = 2, OP_RZ = 3, OP_PZ = 4, OP_NOP = 6 };   // (5 = OP_RELAYOUT, vqe_reg.h)

struct GateRec { int32_t kind, q0, q1, pidx; };           // as uploaded by the host
struct Op { uint32_t xm, zm; int32_t pidx; int32_t kind; };  // kind | (inv << 8)

// Hamiltonian in device memory.
struct HamDev {
  int n_groups;             // X-mask groups evaluated by this handle (after sharding)
  const uint32_t* gx;       // [n_groups] X mask
  // LDS path (n <= 13): pair-compacted sign-sum tables (diagonal group, if any, first)
  const int32_t* tab_r;     // [n_groups] offset of the real table (doubles)
  const int32_t* tab_i;     // [n_groups] offset of the imaginary table or -1
  const double* tables;
  int has_diag;             // 1: group 0 is the diagonal (x == 0) group
  int n_real;               // real-table pair groups incl. zero padding (multiple of energy_pd(n))
  // register path (10 <= n <= 13): the state is handed to the energy step in the CANONICAL index
  // p' = M p (GF(2)-linear, chosen by the host so that the X mask of every real group touches one
  // of the R register bits LT..n-1 of p'); masks and tables below are expressed in p'.
  int n_cls;                // leading real groups whose x' has a register bit (multiple of energy_pd(n))
  uint32_t mrow[16];        // row i of M: bit i of p' = parity(mrow[i] & p)
  // unit path (LDS-resident kernels): X-mask groups whose sign-sum table is mostly EXACT zeros (fermionic excitation
  // operators connect one occupation pattern in 2^w) are stored as *units* - sub-cubes of NT pairs on which the table
  // does not vanish - and never enter the group lists above.  A unit fixes F = n-1-LT bit positions (plus the
  // selector bit that tells the two members of a pair apart); thread t owns the pair whose remaining LT index bits
  // are the bits of t.  urec: 8 words per unit {m0..m4: byte-address masks of the bit deposit, s16: the fixed bits,
  // x16: the X mask, toff: byte offset of its NT table doubles in utab}; n_units is a multiple of kUnitUnroll
  // (zero-table padding).
  int n_units;
  const uint32_t* urec;
  const double* utab;
  // streaming path (n >= 14): explicit terms
  int n_terms;              // terms of the groups above
  const int32_t* term_off;  // [n_groups + 1]
  const uint32_t* term_z;   // [n_terms]
  const double* term_cr;    // [n_terms] real part of c_k (incl. i^{#Y})
  const double* term_ci;    // [n_terms]
};

// LDS bank shear of the unit path.  ds_read_b128 serves a wavefront in four groups of 16 lanes - (lane bit 5, parity of
// lane bits 2..4) - and a group is conflict free when its 16 lanes hit 16 different 16-byte slots modulo 256 B
// (MI355X_MICROARCH.md, LDS).  The lanes of a unit differ in the lowest FREE index bits; wherever a low index bit is a
// hole of the unit (a fixed or the selector bit) the plain layout stacks a group 2, 4 or 8 deep on the same slots:
// 2.95 x the conflict-free LDS cycles averaged over all four-hole patterns of 12 index bits, 2.14 x for the bench
// Hamiltonian's 174 units (tools/conflict_model.py).  So the canonical index map gets a shear on top of its qubit
// permutation: the four low index bits are XORed with a 4-bit code of bits 4..7 (kSwzCode, the best of all 15^4
// assignments: 1.25 x on average).  The shear is part of the index map M - free in the final scatter of the
// circuit, invisible to the table paths, which work in canonical space for any M; only the unit loop, whose cubes are
// axis aligned BEFORE the shear, applies it to the deposited thread id: a 16-entry table of 4-bit codes in one 64-bit
// constant (entry u at bits [4u, 4u+4)).
constexpr uint32_t kSwzCode[4] = {1u, 15u, 2u, 12u};
constexpr unsigned long long kSwzTable = 0x1fe23dccd32ef10ull;
__host__ __device__ constexpr uint32_t swz_index(uint32_t p) {      // permuted index -> canonical (sheared) index
  return p ^ (((p >> 4) & 1u) * kSwzCode[0]) ^ (((p >> 5) & 1u) * kSwzCode[1]) ^ (((p >> 6) & 1u) * kSwzCode[2]) ^
         (((p >> 7) & 1u) * kSwzCode[3]);
}
constexpr int kUnitMinQubits = 8;             // below: a group has no more pairs than a workgroup has threads
constexpr int kUnitTrip = 4;                  // units per trip of the unit loop
constexpr int kUnitUnroll = 3 * kUnitTrip;    // HamDev::n_units is padded to a multiple of this (three trips per turn of the loop)

struct NoiseCfg { double p1, p2; uint64_t seed; uint64_t eval_base; double shot_sigma; };

struct BatchArgs {
  int n;                       // qubits
  int batch;
  const GateRec* gates;
  const int64_t* gate_begin;   // [batch]
  const int32_t* gate_count;   // [batch]
  const int64_t* par_begin;    // [batch]
  const int32_t* par_count;    // [batch]
  const int32_t* order;        // [batch] circuit run by workgroup i: longest expected first (LPT), shortens the launch tail
  const double* theta;         // [sum P] in: theta / x0 (never written: runs are repeatable)
  double* xout;                // [sum P] out: optimised parameters (minimize / env_step)
  double* xraw;                // [sum P] out: the same before the float32 rounding of env_step
  const int32_t* new_gate;     // [batch] or NULL: index of the gate the RL action just added
  int env_step;                // 1: after COBYLA round x to float32 and evaluate the full circuit
  double* fout;                // [batch]
  int32_t* nfev;               // [batch]
  double* scratch;             // COBYLA scratch
  const int64_t* scratch_begin;// [batch]
  const double2* init;         // [2^n]
  HamDev ham;
  NoiseCfg noise;
  int max_ops;                 // LDS capacity (ops) of this launch
  int max_pair;                // ... of which pair ops (RX / RY gates: the only ops that can open a new register layout)
  int max_params;              // LDS capacity (cos/sin pairs)
  double rhobeg, rhoend;
  int maxfun;
  double2* state_out;          // get_state: [2^n]
  unsigned long long* dbg;     // [8] phase cycle counters, written only by -DVQE_STAMPS builds
  double* trace;               // NULL, or [batch][maxfun][1 + max_params]: (f, x[]) of every COBYLA evaluation (vqe_batch_set_trace)
  int amp_rank, amp_world;     // streaming path: this handle sweeps slice amp_rank of amp_world of the amplitudes
};

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Uniform [0,1) draw for (stream b, evaluation e, gate g): a pure function of the seed.
// noise_key() is the part that does not depend on the gate (hoisted out of per-gate loops).
__device__ __forceinline__ uint64_t noise_key(uint64_t seed, uint64_t b, uint64_t e) {
  const uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ull * (b + 1));
  return mix64(k ^ (e * 0xBF58476D1CE4E5B9ull));
}
__device__ __forceinline__ double noise_uniform_k(uint64_t key, uint64_t g) {
  const uint64_t k = mix64(key ^ (g * 0x94D049BB133111EBull));
  return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double noise_uniform(uint64_t seed, uint64_t b, uint64_t e, uint64_t g) {
  return noise_uniform_k(noise_key(seed, b, e), g);
}

// Standard normal draw for (stream b, evaluation e): Box-Muller on two draws of the same
// counter-based generator (gate slots 2^40, 2^40+1 are never used by circuits).
__device__ __forceinline__ double noise_gauss(uint64_t seed, uint64_t b, uint64_t e) {
  const double u1 = noise_uniform(seed, b, e, (uint64_t)1 << 40);
  const double u2 = noise_uniform(seed, b, e, ((uint64_t)1 << 40) + 1);
  return sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
}

__device__ __forceinline__ uint32_t insert0(uint32_t q, int hb) {
  return ((q >> hb) << (hb + 1)) | (q & ((1u << hb) - 1u));
}
__device__ __forceinline__ int parity32(uint32_t v) { return __popc(v) & 1; }

// Cross-lane moves by DPP (no LDS traffic, unlike ds_bpermute behind __shfl_xor): within a row
// of 16 lanes the butterfly is quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror; the four row results are then combined through v_readlane.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Sum over the NW*64 threads of the block; every thread receives the identical value.
template <int NW = 4>
__device__ __forceinline__ double block_sum(double v, double* red /* >= NW doubles LDS */) {
  v = wave_sum(v);
  if constexpr (NW == 1) return v;     // one-wave workgroups: every lane has it already
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NW; w += 2) t += red[w] + red[w + 1];
  return t;
}

// Execution context of cobyla_m0.h on the device: ONE wavefront runs the optimiser's
// bookkeeping (n <= a few dozen: 64 lanes cover every loop), so its reductions are lane
// shuffles and its synchronisation points need no s_barrier; the other waves of the workgroup
// wait at the barrier that follows tell().  Sums are taken in a different order than on the
// host (same algorithm, results differ in the last bits).
struct WaveCtx {
  int tid;   // lane
  static constexpr int nth = 64;
  static constexpr int kPad = 8;   // inner loops in batches of 8, matrices zero-padded (cobyla_m0.h)
  static constexpr bool kSplit = true;   // <= 32 rows: lanes l and l + 32 share a row
  static constexpr bool kColumns = false;
  static constexpr bool kTile = false;
  static constexpr bool kTileWalks = false;
  // v + (the value of v in lane ^ 32)
  __device__ __forceinline__ double pair_sum(double v) const {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool up = tid >= 32;
    return v + __hiloint2double(up ? r1[0] : r1[1], up ? r0[0] : r0[1]);
  }
  __device__ __forceinline__ void lockstep() const {}   // a wavefront IS in lock step
  // LDS / global accesses of one wave execute in order; the fence keeps the compiler (and the
  // memory counters) from moving accesses across the point where lanes exchange data.
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ int all_or(int v) const {
    sync();
    return __ballot(v != 0) != 0ull;
  }
  template <class F>
  __device__ __forceinline__ double sum(int n, F f) const {
    double a = 0.0;
    for (int i = tid; i < n; i += 64) a += f(i);
    return wave_sum(a);
  }
  template <class F>
  __device__ __forceinline__ int arg_first(int n, F f, double thresh, bool want_max, double* val) const {
    double best = thresh;
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += 64) {
      const double v = f(i);
      if (want_max ? (v > best) : (v < best)) { best = v; idx = i; }
    }
    auto merge = [&](double ob, int oi) {
      const bool better = want_max ? (ob > best) : (ob < best);
      if (better || (ob == best && oi < idx)) { best = ob; idx = oi; }
    };
    merge(dpp_f64<0xB1>(best), dpp_i32<0xB1>(idx));
    merge(dpp_f64<0x4E>(best), dpp_i32<0x4E>(idx));
    merge(dpp_f64<0x141>(best), dpp_i32<0x141>(idx));
    merge(dpp_f64<0x140>(best), dpp_i32<0x140>(idx));
    const double b0 = best;
    const int i0 = idx;
    best = readlane_f64(b0, 0); idx = __builtin_amdgcn_readlane(i0, 0);
    merge(readlane_f64(b0, 16), __builtin_amdgcn_readlane(i0, 16));
    merge(readlane_f64(b0, 32), __builtin_amdgcn_readlane(i0, 32));
    merge(readlane_f64(b0, 48), __builtin_amdgcn_readlane(i0, 48));
    *val = best;
    return idx == 0x7fffffff ? -1 : idx;
  }
};

// Block transfers of cobyla_m0.h's walk_tiled for the device contexts whose matrices live in global memory: buffer
// loads / stores through a descriptor of the whole array - rows beyond it read as zero and are not written (no
// predicates), the part of the address that is the same for all lanes sits in a scalar register (no 64-bit vector
// address arithmetic: the plain-C version of this walk spent 130 vector instructions per block on addresses).
struct TileOps {
  lds_double* tile;      // this wavefront's transposition tile (kTileDoubles)
  lds_double* shared;    // the vector every row meets (nv doubles; one per workgroup)
  __amdgpu_buffer_rsrc_t t_rs;
  uint32_t t_voff, t_row4;      // byte offset of this lane's (row lane >> 4, entry lane & 15); bytes of four rows
  uint32_t t_ld8;
  __device__ __forceinline__ void tile_bind(const double* m, int m_rows, int ld) {
    const uint32_t lane = threadIdx.x & 63u;
    t_rs = __builtin_amdgcn_make_buffer_rsrc((void*)m, 0, m_rows * ld * 8, 0x00020000);
    t_ld8 = (uint32_t)ld * 8u;
    t_voff = (lane >> 4) * t_ld8 + (lane & 15u) * 8u;
    t_row4 = 4u * t_ld8;
  }
  __device__ __forceinline__ void tile_fetch(int jb, int i0, double (&g)[16]) const {
    const uint32_t base = (uint32_t)jb * t_ld8 + (uint32_t)i0 * 8u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(t_rs, (int)t_voff, (int)(base + (uint32_t)k * t_row4), 0);
      g[k] = __hiloint2double(v[1], v[0]);
    }
  }
  __device__ __forceinline__ void tile_store(int jb, int i0, const double (&o)[16]) const {
    const uint32_t base = (uint32_t)jb * t_ld8 + (uint32_t)i0 * 8u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v2i_t v;
      v[0] = __double2loint(o[k]); v[1] = __double2hiint(o[k]);
      __builtin_amdgcn_raw_buffer_store_b64(v, t_rs, (int)t_voff, (int)(base + (uint32_t)k * t_row4), 0);
    }
  }
  // LDS instructions of one wavefront execute in order: only the compiler has to be kept from moving them (and
  // nothing waits for the global loads of the next block, which a fence would)
  __device__ __forceinline__ void tile_sync() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ unsigned long long ballot(bool v) const { return __ballot(v); }
  __device__ __forceinline__ int popc(unsigned long long m) const { return __popcll(m); }
  __device__ __forceinline__ int ctz(unsigned long long m) const { return __builtin_ctzll(m); }
};
constexpr size_t kCobTileBytes = (size_t)(64 * 17) * 8;     // cby::CobylaM0::kTileDoubles doubles per wavefront, + the shared vector

// The one-wave context of the TRAINABLE regime on the one-wave kernels (n <= 9; 129 parameters for H2O-8q): arrays in
// global memory, row walks through an LDS transposition tile (cobyla_m0.h: walk_tiled) instead of one lane per row -
// a lane per row touches 64 cache lines per load instruction and the CU's vector L1 looks them up one per cycle, which
// is what bounded this regime.  Same arithmetic in the same order as WaveCtx (bit-identical trial points); padding to
// 16 so that rows are whole 128-byte lines.  Only the kernel variant launched for such batches carries it (registers,
// LDS for the tile).
struct WaveRowsCtx : WaveCtx, TileOps {
  static constexpr int kPad = 16;
  static constexpr bool kSplit = false;      // (only problems with more than 32 variables come here)
#ifndef VQE_ROWS_TILE
#define VQE_ROWS_TILE 1
#endif
  static constexpr bool kTile = VQE_ROWS_TILE != 0;
  static constexpr bool kTileWalks = true;      // every row walk of tell() goes through the tile
  __device__ __forceinline__ void tile_sync_all() const { tile_sync(); }      // the workgroup is this wavefront
};
__host__ __device__ inline size_t cobyla_tile_bytes(int n, int max_params) {
  return geo_one_wave(n) && max_params > 64 ? kCobTileBytes + (size_t)cby::padded(max_params, 16) * 8 : 0;
}

// Workgroup-wide context of cobyla_m0.h for LARGE problems (more rows than a wave has lanes,
// matrices in the global scratch - the trainable-path regime, ~129 parameters): every thread
// takes a row, reductions go through LDS and s_barrier.  With one wave the row loops would run
// several passes of a latency-bound inner loop over L2/HBM-resident matrices.
// TILE: the row walks go through per-wavefront LDS transposition tiles in the (dead) state region (walk_tiled: every
// wavefront takes 64-row blocks) - n >= 12, where that region holds NW tiles.
template <int NT, bool TILE = false>
struct BlockCtx : TileOps {
  int tid;
  double* red;   // >= 12 doubles of LDS (NW sums + NW indices)
  static constexpr int nth = NT;
  static constexpr int NW = NT / 64;
  static constexpr int kPad = TILE ? 16 : 8;    // (16 - twice the loads in flight per batch - measured +5 % at 12 qubits / 202 variables)
  static constexpr bool kSplit = false;
  static constexpr bool kColumns = true;   // element-wise matrix passes with the lanes along a row (cobyla_m0.h: update_simi)
  static constexpr bool kTile = TILE;
#ifndef VQE_BLOCK_TILE_WALKS
#define VQE_BLOCK_TILE_WALKS 0
#endif
  // a thread per row in ONE pass walks its row as fast as the tile hands it over (measured, 12 qubits / 202 variables:
  // simi . dx 7.8 -> 11.1, acceptability 21 -> 29 G cycles with tiles): only the rank-one update - whose element-wise
  // form leaves the row norms to a later full pass - takes the tile here
  static constexpr bool kTileWalks = VQE_BLOCK_TILE_WALKS != 0;
  __device__ __forceinline__ void tile_sync_all() const { __syncthreads(); }
  __device__ __forceinline__ double pair_sum(double v) const { return v; }
  __device__ __forceinline__ void lockstep() const {}   // kSplit = false: a row has one owner
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ int all_or(int v) const {   // (HIP's __syncthreads_or allocates static LDS)
    const int any = __ballot(v != 0) != 0ull;
    int* ired = (int*)(red + NW);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ired[threadIdx.x >> 6] = any;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r |= ired[w];
    return r != 0;
  }
  template <class F>
  __device__ __forceinline__ double sum(int n, F f) const {
    double a = 0.0;
    for (int i = tid; i < n; i += NT) a += f(i);
    return block_sum<NW>(a, red);
  }
  template <class F>
  __device__ __forceinline__ int arg_first(int n, F f, double thresh, bool want_max, double* val) const {
    double best = thresh;
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += NT) {
      const double v = f(i);
      if (want_max ? (v > best) : (v < best)) { best = v; idx = i; }
    }
    auto merge = [&](double ob, int oi) {
      const bool better = want_max ? (ob > best) : (ob < best);
      if (better || (ob == best && oi < idx)) { best = ob; idx = oi; }
    };
    merge(dpp_f64<0xB1>(best), dpp_i32<0xB1>(idx));
    merge(dpp_f64<0x4E>(best), dpp_i32<0x4E>(idx));
    merge(dpp_f64<0x141>(best), dpp_i32<0x141>(idx));
    merge(dpp_f64<0x140>(best), dpp_i32<0x140>(idx));
    const double b0 = best;
    const int i0 = idx;
    best = readlane_f64(b0, 0); idx = __builtin_amdgcn_readlane(i0, 0);
    merge(readlane_f64(b0, 16), __builtin_amdgcn_readlane(i0, 16));
    merge(readlane_f64(b0, 32), __builtin_amdgcn_readlane(i0, 32));
    merge(readlane_f64(b0, 48), __builtin_amdgcn_readlane(i0, 48));
    int* ired = (int*)(red + NW);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = best; ired[threadIdx.x >> 6] = idx; }
    __syncthreads();
    best = red[0]; idx = ired[0];
#pragma unroll
    for (int wv = 1; wv < NW; ++wv) merge(red[wv], ired[wv]);
    __syncthreads();   // red is reused by the next reduction
    *val = best;
    return idx == 0x7fffffff ? -1 : idx;
  }
};

// LDS carve-up of one workgroup.

struct GroupMeta { uint32_t x; int32_t hb; int32_t off_r; int32_t off_i; };
// Per-group addressing record of the register energy path, staged into LDS once per launch
// (stage_groups): the wave issues ~1 instruction per 4 cycles, so anything recomputed per
// group - scalar or vector - costs as much as the floating-point work it serves.
struct ClsMeta {
  uint32_t xt16;     // (x' & (NT-1)) << 4 : XORed into tid*16
  int32_t cls;       // highest register bit of x'
  uint32_t pad0, pad1;
  uint32_t off[8];   // byte offset of the partner of own amplitude r = insert0(j, cls): (r ^ xr) << (LT+4)
};

struct LayoutRec;
struct Lds {
  double2* psi;     // [2^n]  (also: gate staging during compile, COBYLA matrices during tell)
  Op* ops;          // [max_ops] raw ops
  Op* sched;        // [max_ops+max_pair+2] scheduled ops (register path, n >= 10): every op + a re-layout in front of a pair op at most
  LayoutRec* lay;   // [max_pair+2] layouts (register path)
  double2* cs;      // [max_params] (cos, sin)(theta/2)
  GroupMeta* gm;    // [n_groups]
  ClsMeta* cm;    // [n_groups] addressing records of the register energy path (n >= 10)
  double* red;      // [16]
  uint32_t* xm;     // [32] columns of A^-1
  uint32_t* zm;     // [32] rows of A
  int32_t* meta;    // [8]: n_ops, offset c (as the final scatter wants it), phase power, permuted flag, n_sched, n_layouts, permuted by CNOTs alone, raw offset c
  uint32_t* sb;     // [16] scheduler scratch
  uint16_t* sidx;   // [max_ops] position of raw op k in the executed list (sched for n >= 10, ops below)
  double* cob;      // the optimiser's arrays (cobyla_resident_bytes), or nullptr
  double* tile;     // transposition tile of the one-wave optimiser on more than 64 variables (cobyla_tile_bytes), or nullptr
};

// n >= 10: the raw ops only live while the schedule is built, in the (idle) state region: in its
// upper half next to the staged gate records when they fit there, else in all of it (launch_lds
// refuses circuits with more than 2^n ops).
__host__ __device__ inline bool ops_fit_upper_half(int n, int max_ops) { return (size_t)max_ops <= ((size_t)1 << n) / 2; }
// max_pair: RX / RY gates of the longest circuit (<= max_ops).  Only a pair op whose partner mask lies outside the
// current layout opens a new one (vqe_reg.h: schedule_ops), so the schedule has at most max_ops + max_pair records and
// max_pair + 1 layouts - for the trainable regime at 12 qubits (203 rotations, two thirds of them RX / RY) the
// difference to the 2 max_ops bound is what lets a second workgroup into the CU (82.8 KB -> 79.6 KB).
__host__ __device__ inline size_t lds_bytes_base(int n, int max_ops, int max_params, int n_groups, int max_pair) {
  const int ng = n_groups > 0 ? n_groups : 1;
  size_t b = (size_t)16 << n;
  b += n >= 10 ? (size_t)16 * (max_ops + max_pair + 2) + (size_t)32 * (max_pair + 2) : (size_t)16 * max_ops;
  b += (size_t)16 * max_params + (size_t)16 * ng + (n >= 10 ? (size_t)48 * ng : 0);
  return b + 128 + 128 + 128 + 32 + 64 + (((size_t)2 * max_ops + 15) & ~(size_t)15);
}
// One-wave workgroups (n <= kOneWaveMaxQubits): the optimiser's arrays get an LDS region of their own when the
// workgroup then still fits eight times into a CU - no staging copies, no global-memory round trips in the update,
// which is all of the critical path when an environment is one wave.  0: none (the arrays are staged into the idle
// state region when they fit there, else they stay in the global scratch).
#ifndef VQE_RESIDENT_BUDGET
#define VQE_RESIDENT_BUDGET (160 * 1024 / 8)
#endif
constexpr size_t kResidentLdsBudget = VQE_RESIDENT_BUDGET;
__host__ __device__ inline size_t cobyla_resident_bytes(int n, int max_ops, int max_params, int n_groups, int max_pair) {
  if (n > kOneWaveMaxQubits || max_params <= 0) return 0;
  const size_t need = (cby::scratch_doubles(max_params, 8) * 8 + 15) & ~(size_t)15;
  return lds_bytes_base(n, max_ops, max_params, n_groups, max_pair) + need + 16 <= kResidentLdsBudget ? need : 0;
}
__host__ __device__ inline size_t lds_bytes(int n, int max_ops, int max_params, int n_groups, int max_pair) {
  const size_t r = cobyla_resident_bytes(n, max_ops, max_params, n_groups, max_pair);
  return ((lds_bytes_base(n, max_ops, max_params, n_groups, max_pair) + 15) & ~(size_t)15) + r + cobyla_tile_bytes(n, max_params);
}

__device__ __forceinline__ Lds carve(unsigned char* base, int n, int max_ops, int max_params, int n_groups, int max_pair) {
  const int ng = n_groups > 0 ? n_groups : 1;
  Lds l;
  l.psi = (double2*)base; base += (size_t)16 << n;
  if (n >= 10) {
    // raw ops: upper half of the state region (the lower half stages the gate records), or - for
    // circuits with more than 2^(n-1) ops - the whole region (gates are then read from global memory)
    l.ops = (Op*)(l.psi) + (ops_fit_upper_half(n, max_ops) ? ((size_t)1 << n) / 2 : 0);
    l.sched = (Op*)base; base += (size_t)16 * (max_ops + max_pair + 2);
    l.lay = (LayoutRec*)base; base += (size_t)32 * (max_pair + 2);
  } else {
    l.ops = (Op*)base; base += (size_t)16 * max_ops;
    l.sched = l.ops; l.lay = (LayoutRec*)l.ops;
  }
  l.cs = (double2*)base; base += (size_t)16 * max_params;
  l.gm = (GroupMeta*)base; base += (size_t)16 * ng;
  l.cm = (ClsMeta*)base; if (n >= 10) base += (size_t)48 * ng;
  l.red = (double*)base; base += 128;
  l.xm = (uint32_t*)base; base += 128;
  l.zm = (uint32_t*)base; base += 128;
  l.meta = (int32_t*)base; base += 32;
  l.sb = (uint32_t*)base; base += 64;
  l.sidx = (uint16_t*)base;
  l.cob = cobyla_resident_bytes(n, max_ops, max_params, n_groups, max_pair)
              ? (double*)((unsigned char*)l.psi + ((lds_bytes_base(n, max_ops, max_params, n_groups, max_pair) + 15) & ~(size_t)15))
              : nullptr;
  l.tile = cobyla_tile_bytes(n, max_params)
               ? (double*)((unsigned char*)l.psi + ((lds_bytes_base(n, max_ops, max_params, n_groups, max_pair) + 15) & ~(size_t)15) +
                           cobyla_resident_bytes(n, max_ops, max_params, n_groups, max_pair))
               : nullptr;
  return l;
}

// Once per kernel: X-mask group descriptors into LDS (no dependent global loads later).
__device__ __forceinline__ void stage_groups(const HamDev& H, const Lds& L) {
  for (int g = threadIdx.x; g < H.n_groups; g += (int)blockDim.x) {
    const uint32_t x = H.gx[g];
    L.gm[g] = GroupMeta{x, x ? 31 - __clz((int)x) : 0, H.tab_r[g], H.tab_i[g]};
  }
}

// Addressing records of the class groups (register energy path); group index relative to the
// first class group.
template <int N>
__device__ __forceinline__ void stage_cls(const HamDev& H, const Lds& L) {
  constexpr int LT = Geo<N>::LT;
  constexpr int NP = 1 << (N - LT - 1);
  for (int i = threadIdx.x; i < H.n_cls; i += (int)blockDim.x) {
    const uint32_t x = H.gx[i + H.has_diag];
    const int cls = (31 - __clz((int)x)) - LT;
    const uint32_t xr = x >> LT;
    uint32_t* w = (uint32_t*)(L.cm + i);
    w[0] = (x & ((1u << LT) - 1u)) << 4;
    w[1] = (uint32_t)cls;
    w[2] = w[3] = 0u;
    for (int j = 0; j < 8; ++j) w[4 + j] = j < NP ? ((insert0((uint32_t)j, cls) ^ xr) << (LT + 4)) : 0u;
  }
}

// Translate the gate list of problem b into pair-exchange ops (CNOT / Pauli-X become
// updates of the affine map).  All threads stage the gate records into the (idle) state
// region, thread 0 walks them.  Noise Paulis are drawn per evaluation.  Ends with a barrier.
// `slots`: structure-only compile for stochastic runs.  Every noise gate leaves one inactive
// Pauli-Z slot per qubit it acts on (kind OP_NOP) and draws nothing; patch_noise() then sets,
// for each evaluation, which slots are active and the sign bits that X errors flip - the
// masks, the layouts and the schedule do not depend on the errors drawn.
// Gates [skip, skip_end) are left out (the action just taken and the noise gate that belongs to it).
__device__ __forceinline__ void compile_ops(const BatchArgs& A, int b, uint64_t /*eval_id*/, const Lds& L, int skip = -1,
                                            bool slots = false, int skip_end = 0) {
  const int n = A.n;
  const int G = A.gate_count[b];
  const int cap = n >= 10 ? (ops_fit_upper_half(n, A.max_ops) ? (int)(((size_t)1 << n) / 2) : 0)
                          : (int)(((size_t)16 << n) / sizeof(GateRec));
  const GateRec* gsrc = A.gates + A.gate_begin[b];
  GateRec* gl = (GateRec*)L.psi;
  const bool staged = G <= cap;
  if (staged) {
    const int4* s4 = (const int4*)gsrc;
    int4* d4 = (int4*)gl;
    for (int i = threadIdx.x; i < G; i += (int)blockDim.x) d4[i] = s4[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const GateRec* g = staged ? gl : gsrc;
    uint32_t* xm = L.xm;
    uint32_t* zm = L.zm;
    for (int q = 0; q < n; ++q) { xm[q] = 1u << q; zm[q] = 1u << q; }
    uint32_t c = 0;
    int phase = 0, nops = 0;
    auto slot = [&](int q) {
      if (nops < A.max_ops) L.ops[nops] = Op{0u, zm[q], -1, OP_NOP};
      ++nops;
    };
    for (int i = 0; i < G; ++i) {
      if (i >= skip && i < skip_end) continue;
      const GateRec r = g[i];
      if (slots && (r.kind == G_DEPOL1 || r.kind == G_DEPOL2)) {
        slot(r.q0);
        if (r.kind == G_DEPOL2) slot(r.q1);
        continue;
      }
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
        // (noise gates: slots above, or ignored - the errors are applied by patch_noise)
        default: break;
      }
    }
    int permuted = (c != 0);
    for (int q = 0; q < n; ++q) if (xm[q] != (1u << q)) permuted = 1;
    L.meta[0] = nops < A.max_ops ? nops : A.max_ops;
    L.meta[1] = (int32_t)c;
    L.meta[2] = phase;
    L.meta[3] = permuted;
    L.meta[6] = permuted;
    L.meta[7] = (int32_t)c;
  }
  __syncthreads();
}

// Per-evaluation part of a stochastic run (after a `slots` compile): draw the Pauli errors of
// this (stream, evaluation) in parallel, then one thread walks the gate list tracking only the
// offset c and patches the executed records in place: sign bit of every rotation, activity and
// sign bit of every Pauli-Z slot.  `canonical`: the final scatter wants M c (see compile_all).
// Gates [skip, skip_end) are not part of the circuit (as in compile_ops); the draw of a noise gate
// is numbered by its position in the circuit that is actually executed, so a stochastic COBYLA
// phase is exactly a run on the pre-action gate list (reference scipy_optim builds its circuit
// from the pre-action state: environment_qulacs_TN_notin_agent_noise.py:372-380).
template <int N>
__device__ __forceinline__ void patch_noise_wave(const BatchArgs& A, int b, uint64_t eval_id, const Lds& L, int skip,
                                                 bool canonical, int lane, int skip_end) {
  // One wave.  Lane l holds gate base+l and draws its Pauli error; everything that does not depend
  // on the running offset c (record numbers by a wave prefix sum, Z activity, the masks each gate
  // XORs into c) is prepared per lane, so the serial walk is a short branch-free scalar loop that
  // reads two packed words per gate with v_readlane and collects the sign bits in 64-bit masks.
  const int G = A.gate_count[b];
  const int4* gsrc = (const int4*)(A.gates + A.gate_begin[b]);
  {
    Op* exe = N >= 10 ? L.sched : L.ops;
    const int kmax = L.meta[0];
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint64_t nkey = noise_key(A.noise.seed, (uint64_t)b, eval_id);
    uint32_t c = 0;
    int ny = 0, kbase = 0;
    int4 rnext = lane < G ? gsrc[lane] : make_int4(-1, 0, 0, 0);
    for (int base = 0; base < G; base += 64) {
      const int i = base + lane;
      int kind = -1, q0 = 0, q1 = 0, code = 0;
      const int4 r = rnext;                      // (the records of the next chunk are requested before this one is walked)
      if (i + 64 < G) rnext = gsrc[i + 64];
      if (i < G && !(i >= skip && i < skip_end)) {
        kind = r.x; q0 = r.y; q1 = r.z < 0 ? 0 : r.z;
        if (kind == G_DEPOL1 || kind == G_DEPOL2) {
          const double u = noise_uniform_k(nkey, (uint64_t)(i - ((skip >= 0 && i >= skip_end) ? skip_end - skip : 0)));
          if (kind == G_DEPOL1) { if (u < A.noise.p1) code = 1 + (int)(u / A.noise.p1 * 3.0); }
          else if (u < A.noise.p2) code = 1 + (int)(u / A.noise.p2 * 15.0);
        }
      }
      // Pauli on q0 / q1: 1=X 2=Y 3=Z (compile_ops' rule: X part first, then the Z slot reads c)
      const int pa = kind == G_DEPOL1 ? code : (kind == G_DEPOL2 ? (code & 3) : 0);
      const int pb = kind == G_DEPOL2 ? (code >> 2) : 0;
      const uint32_t flip_a = (pa == 1 || pa == 2) ? 1u << q0 : 0u;
      const uint32_t flip_b = (pb == 1 || pb == 2) ? 1u << q1 : 0u;
      const uint32_t cnot_m = kind == G_CNOT ? 1u << q1 : 0u;
      const int nrec = (kind >= G_RX && kind <= G_DEPOL1) ? 1 : (kind == G_DEPOL2 ? 2 : 0);
      const uint32_t w1 = flip_a | ((uint32_t)q0 << 16) | ((uint32_t)q1 << 24);
      const uint32_t w2 = cnot_m | (flip_b << 16);
      const uint64_t has1 = __ballot(nrec >= 1), has2 = __ballot(nrec == 2);
      const int k0 = kbase + __popcll(has1 & lt_mask) + __popcll(has2 & lt_mask);
      kbase += __popcll(has1) + __popcll(has2);
      ny += __popcll(__ballot(pa == 2)) + __popcll(__ballot(pb == 2));
      uint64_t inv0 = 0, inv1 = 0;
      const int cnt = G - base < 64 ? G - base : 64;
      // while no X / Y error has happened the offset c is zero, CNOTs keep it zero and every sign bit is clear: the
      // serial walk starts at the first gate of the chunk that flips a bit (with ~2 errors per evaluation most
      // chunks are skipped in part or entirely)
      int j0 = 0;
      if (c == 0) {
        const uint64_t flips = __ballot((flip_a | flip_b) != 0u);
        j0 = flips ? __ffsll((long long)flips) - 1 : cnt;
      }
      for (int j = j0; j < cnt; ++j) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)w1, j);
        const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)w2, j);
        c ^= a & 0xffffu;                                   // X error on q0
        const uint32_t b0 = (c >> ((a >> 16) & 31u)) & 1u;  // sign bit of record 0 / CNOT control
        c ^= (0u - b0) & (bw & 0xffffu);                    // CNOT: target follows the control
        inv0 |= (uint64_t)b0 << j;
        c ^= bw >> 16;                                      // X error on q1
        inv1 |= (uint64_t)((c >> (a >> 24)) & 1u) << j;     // sign bit of record 1
      }
      auto apply = [&](int pk, int sign, int pauli) {   // pauli < 0: a rotation, kind stays
        if (pk < kmax) {
          Op* o = exe + L.sidx[pk];
          int kd = o->kind;
          if (pauli >= 0) kd = (kd & ~0xff) | ((pauli == 2 || pauli == 3) ? OP_PZ : OP_NOP);
          o->kind = (kd & ~0x100) | (sign << 8);
        }
      };
      if (nrec >= 1) apply(k0, (int)((inv0 >> lane) & 1ull), kind <= G_RZ ? -1 : pa);
      if (nrec == 2) apply(k0 + 1, (int)((inv1 >> lane) & 1ull), pb);
    }
    {
      // jump counts for the run loop: every record learns how many inactive slots follow it (walked from the
      // end in chunks of 64 records, `lead` = inactive records at the head of the chunk behind)
      const int ns = N >= 10 ? L.meta[4] : kmax;
      int lead = 0;
      for (int base = ((ns - 1) / 64) * 64; base >= 0; base -= 64) {
        const int k = base + lane;
        const int kd = k < ns ? exe[k].kind : 0;
        const bool nop = k < ns && (kd & 0xff) == OP_NOP;
        const uint64_t m = __ballot(nop);
        const uint64_t behind = lane < 63 ? (m >> (lane + 1)) : 0ull;          // inactivity of the records behind this lane's
        const int in_chunk = 63 - lane;                                          // records behind it inside the chunk
        int run = behind == ~0ull >> (lane + 1) && lane < 63 ? in_chunk : (lane < 63 ? __ffsll((long long)~behind) - 1 : 0);
        if (run >= in_chunk) run = in_chunk + lead;                              // the run reaches the next chunk
        if (k < ns) exe[k].kind = (kd & 0x00ffffff) | ((run < 255 ? run : 255) << 24);
        lead = (m == ~0ull) ? 64 + lead : __ffsll((long long)~m) - 1;            // (a short last chunk has zeros beyond ns)
      }
    }
    if (lane == 0) {
      uint32_t cs = c;
      if (N >= 10 && canonical) {
        cs = 0;
        for (int q = 0; q < N; ++q) cs |= (uint32_t)parity32(A.ham.mrow[q] & c) << q;
      }
      L.meta[1] = (int32_t)cs;
      L.meta[2] = (3 * ny) & 3;
      L.meta[3] = L.meta[6] | (c != 0);
      L.meta[7] = (int32_t)c;
    }
  }
}

// ... by wave 0 of the workgroup, followed by a barrier.
template <int N>
__device__ __forceinline__ void patch_noise(const BatchArgs& A, int b, uint64_t eval_id, const Lds& L, int skip,
                                            bool canonical, int skip_end = 0) {
  if (threadIdx.x < 64) patch_noise_wave<N>(A, b, eval_id, L, skip, canonical, (int)threadIdx.x, skip_end);
  __syncthreads();
}

}  // namespace vqe
#include "vqe_reg.h"
namespace vqe {

// compile + (n >= 10) schedule for the register-resident path; ends with a barrier
template <int N>
__device__ __forceinline__ void compile_all(const BatchArgs& A, int b, uint64_t eval_id, const Lds& L, int skip = -1,
                                            bool canonical = true, bool slots = false, int skip_end = 0) {
  compile_ops(A, b, eval_id, L, skip, slots, skip_end);
  if constexpr (N < 10) {   // (kRegMinQubits)
    if (slots) {
      for (int k = threadIdx.x; k < L.meta[0]; k += (int)blockDim.x) L.sidx[k] = (uint16_t)k;
      __syncthreads();
    }
  }
  if constexpr (N >= kRegMinQubits) {
    if (threadIdx.x == 0) schedule_ops<N>(L);
    __syncthreads();
    if (canonical) {
      // the energy step wants the state in canonical order p' = M (A p ^ c): compose M into the
      // rows of A and into c that the final scatter of run_ops_reg uses
      uint32_t zrow = 0, cbit = 0;
      if (threadIdx.x < N) {
        const uint32_t m = A.ham.mrow[threadIdx.x];
        for (int q = 0; q < N; ++q) if ((m >> q) & 1u) zrow ^= L.zm[q];
        cbit = (uint32_t)parity32(m & (uint32_t)L.meta[1]) << threadIdx.x;
      }
      // N <= 13 < 64: the contributing threads sit in wave 0
      uint32_t cnew = cbit;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) cnew |= __shfl_xor(cnew, o, 16);
      __syncthreads();
      if (threadIdx.x < N) L.zm[threadIdx.x] = zrow;
      if (threadIdx.x == 0) L.meta[1] = (int32_t)cnew;
      __syncthreads();
    }
  }
}

// Apply the compiled ops to the LDS-resident state, then restore the logical layout.
// `theta` holds the P parameters of the circuit, or - when p_hole >= 0 - the P-1 parameters
// of the circuit without the rotation whose parameter index is p_hole.
template <int N, bool SLOTS = true>
__device__ __forceinline__ void run_ops(const Lds& L, const double* theta, int P, int p_hole = -1) {
  constexpr int kThreads = Geo<N>::NT;   // shadows the default: this kernel family's block size
  constexpr uint32_t DIM = 1u << N;
  constexpr int NP = (DIM / 2 + kThreads - 1) / kThreads;   // pairs per thread
  constexpr int NA = (DIM + kThreads - 1) / kThreads;       // amplitudes per thread
  const int tid = threadIdx.x;
  for (int j = tid; j < P; j += kThreads) {
    if (j == p_hole) continue;
    double s, c;
    sincos(0.5 * theta[j - (p_hole >= 0 && j > p_hole)], &s, &c);
    L.cs[j] = make_double2(c, s);
  }
  __syncthreads();
  const int nops = L.meta[0];
  for (int o = 0; o < nops; ++o) {
    const Op op = L.ops[o];
    const int kind = op.kind & 0xff;
    if constexpr (SLOTS) o += (int)((uint32_t)op.kind >> 24);   // inactive noise slots behind this record (jump counts of patch_noise_wave)
    if (kind == OP_NOP) continue;   // inactive noise slot: nothing to do, no barrier needed
    const int inv = (op.kind >> 8) & 1;
    if (kind == OP_RX || kind == OP_RY) {
      const double2 cs = L.cs[op.pidx];
      const int hb = 31 - __clz((int)op.xm);
      if (kind == OP_RX) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const uint32_t q = tid + k * kThreads;
          if (DIM / 2 >= kThreads || q < DIM / 2) {
            const uint32_t p0 = insert0(q, hb), p1 = p0 ^ op.xm;
            const double2 a0 = L.psi[p0], a1 = L.psi[p1];
            L.psi[p0] = make_double2(cs.x * a0.x - cs.y * a1.y, cs.x * a0.y + cs.y * a1.x);
            L.psi[p1] = make_double2(cs.x * a1.x - cs.y * a0.y, cs.x * a1.y + cs.y * a0.x);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const uint32_t q = tid + k * kThreads;
          if (DIM / 2 >= kThreads || q < DIM / 2) {
            const uint32_t p0 = insert0(q, hb), p1 = p0 ^ op.xm;
            const double s0 = (parity32(p0 & op.zm) ^ inv) ? -cs.y : cs.y;
            const double2 a0 = L.psi[p0], a1 = L.psi[p1];
            L.psi[p0] = make_double2(cs.x * a0.x + s0 * a1.x, cs.x * a0.y + s0 * a1.y);
            L.psi[p1] = make_double2(cs.x * a1.x - s0 * a0.x, cs.x * a1.y - s0 * a0.y);
          }
        }
      }
    } else if (kind == OP_RZ) {
      const double2 cs = L.cs[op.pidx];
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const uint32_t p = tid + k * kThreads;
        if (DIM >= kThreads || p < DIM) {
          const double s = (parity32(p & op.zm) ^ inv) ? -cs.y : cs.y;
          const double2 a = L.psi[p];
          L.psi[p] = make_double2(cs.x * a.x - s * a.y, cs.x * a.y + s * a.x);
        }
      }
    } else if (kind == OP_PZ) {   // (OP_NOP: an inactive noise slot)
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const uint32_t p = tid + k * kThreads;
        if ((DIM >= kThreads || p < DIM) && (parity32(p & op.zm) ^ inv)) {
          const double2 a = L.psi[p];
          L.psi[p] = make_double2(-a.x, -a.y);
        }
      }
    }
    __syncthreads();
  }
  if (L.meta[3]) {  // psi_logical[i] = phi[A^-1 (i ^ c)]
    const uint32_t c = (uint32_t)L.meta[1];
    double2 tmp[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const uint32_t i = tid + k * kThreads;
      if (DIM >= kThreads || i < DIM) {
        uint32_t v = i ^ c, p = 0;
#pragma unroll
        for (int q = 0; q < N; ++q) p ^= ((v >> q) & 1u) ? L.xm[q] : 0u;
        tmp[k] = L.psi[p];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const uint32_t i = tid + k * kThreads;
      if (DIM >= kThreads || i < DIM) L.psi[i] = tmp[k];
    }
    __syncthreads();
  }
}

template <int N>
__device__ __forceinline__ void load_init(const Lds& L, const double2* init) {
  constexpr int kThreads = Geo<N>::NT;   // shadows the default: this kernel family's block size
  constexpr uint32_t DIM = 1u << N;
  constexpr int NA = (DIM + kThreads - 1) / kThreads;
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const uint32_t p = threadIdx.x + k * kThreads;
    if (DIM >= kThreads || p < DIM) L.psi[p] = init[p];
  }
}

// <psi|H|psi> over this handle's X-mask groups; identical result in every thread.
// Group order (host): [diagonal] [real-table groups, padded with zero-table dummies to a
// multiple of PD] [groups with an imaginary table].  Table values stream from L2 through a
// PD-deep register ring (group g+PD is requested when group g is consumed); the PD-unrolled
// body is branch free so LDS reads of one group overlap the FMAs of the previous one.
// (depth of the table ring by size: the shallower ring frees 32 registers where the kernel sits at its register cap -
// n = 11 ... 13: +1..2 % - and costs 2..3 % where it does not)
__host__ __device__ constexpr int energy_pd(int n) { return n >= 11 ? 2 : 4; }

// All pairs of one group, both members read from LDS in batches of kEnergyBatch pairs (all
// reads of a batch are in flight together, then the arithmetic).  DEXPR yields D for pair k
// (register ring or F-table lookup).  Expanded in place (see VQE_PAIR_SWITCH in vqe_reg.h).
constexpr int kEnergyBatch = 4;

#define VQE_ENERGY_PAIRS(X, HB, DEXPR)                                                             \
  {                                                                                                \
    const uint32_t base = insert0(tid, (HB)) << 4, basex = base ^ ((X) << 4);                     \
    uint32_t kbit[KB > 0 ? KB : 1];                                                                \
    _Pragma("unroll") for (int i = 0; i < KB; ++i)                                                \
      kbit[i] = 16u << (LT + i + ((LT + i) >= (HB) ? 1 : 0));                                     \
    double p0s = 0.0, p1s = 0.0;                                                                   \
    _Pragma("unroll") for (int k0 = 0; k0 < NP; k0 += kEnergyBatch) {                             \
      constexpr int NB = NP < kEnergyBatch ? NP : kEnergyBatch;                                   \
      double2 pa[NB], pb[NB];                                                                      \
      _Pragma("unroll") for (int kk = 0; kk < NB; ++kk) {                                         \
        const int k = k0 + kk;                                                                     \
        uint32_t kc = 0;                                                                           \
        _Pragma("unroll") for (int i = 0; i < KB; ++i) if ((k >> i) & 1) kc ^= kbit[i];           \
        const bool live = FULL || tid + (uint32_t)k * kThreads < DIM / 2;                         \
        pb[kk] = live ? *(const double2*)(psi_b + (base ^ kc)) : make_double2(0.0, 0.0);         \
        pa[kk] = live ? *(const double2*)(psi_b + (basex ^ kc)) : make_double2(0.0, 0.0);        \
      }                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      _Pragma("unroll") for (int kk = 0; kk < NB; ++kk) {                                         \
        const int k = k0 + kk;                                                                     \
        const double v = (pa[kk].x * pb[kk].x + pa[kk].y * pb[kk].y) * (DEXPR);                   \
        if (k & 1) p1s += v; else p0s += v;                                                       \
      }                                                                                            \
    }                                                                                              \
    acc0 += p0s;   /* pair tables hold 2*D */                                                      \
    acc1 += p1s;                                                                                   \
  }

#define VQE_ENERGY_CONSUME(G, D)                                                                   \
  {                                                                                                \
    const GroupMeta m = L.gm[G];                                                                   \
    const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.x);                        \
    const int hb = __builtin_amdgcn_readfirstlane(m.hb);                                          \
    VQE_ENERGY_PAIRS(x, hb, D[k])                                                                  \
  }

// Real-table groups [g0, g1) (g1 - g0 a multiple of energy_pd(n), tables contiguous) with BOTH
// members of every pair read from LDS.
template <int N>
__device__ __forceinline__ void energy_real_lds(const Lds& L, const double* __restrict__ tables, int g0, int g1,
                                                double& acc0, double& acc1) {
  constexpr int kThreads = Geo<N>::NT;   // shadows the default: this kernel family's block size
  constexpr uint32_t DIM = 1u << N;
  constexpr int NP = (DIM / 2 + kThreads - 1) / kThreads;   // pairs per thread
  constexpr int LT = Geo<N>::LT;
  constexpr int KB = N > LT + 1 ? N - LT - 1 : 0;           // log2(NP)
  constexpr int PD = energy_pd(N);
  constexpr bool FULL = DIM / 2 >= kThreads;                // every thread owns NP pairs
  const uint32_t tid = threadIdx.x;
  const unsigned char* psi_b = (const unsigned char*)L.psi;
  double buf[PD][NP];
  if (g0 < g1) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const double* t = tables + __builtin_amdgcn_readfirstlane(L.gm[g0 + j].off_r);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const uint32_t q = tid + (uint32_t)k * kThreads;
        buf[j][k] = (FULL || q < DIM / 2) ? t[q] : 0.0;
      }
    }
    if constexpr (NP >= 4) {
      // Software pipeline over half groups: while the arithmetic of one half runs, the LDS
      // reads of the next half (possibly of the next group) are already in flight.
      constexpr int HP = NP / 2;
      double2 aA[HP], bA[HP], aB[HP], bB[HP];
      // addressing context of the group being loaded.  The X mask of a group is fetched
      // from LDS one step ahead (xq), so no LDS latency sits in front of the address math;
      // everything is branch free (conditionals inside this unrolled body make hipcc spill).
      uint32_t cb, cbx, ckb[KB];
      uint32_t xq;
#define VQE_E_CTX(XV)                                                                              \
      {                                                                                            \
        const uint32_t x_ = (XV);                                                                  \
        const int hb_ = 31 - __builtin_clz((int)x_);                                              \
        cb = insert0(tid, hb_) << 4;                                                               \
        _Pragma("unroll") for (int i = 0; i < KB; ++i) ckb[i] = 16u << (LT + i + ((LT + i) >= hb_ ? 1 : 0)); \
        cbx = cb ^ (x_ << 4);                                                                      \
      }
#define VQE_E_XFETCH(G) (uint32_t)__builtin_amdgcn_readfirstlane((int)L.gm[(G) < g1 ? (G) : g1 - 1].x)
#define VQE_E_LOAD(H, PA, PB)                                                                      \
      _Pragma("unroll") for (int kk = 0; kk < HP; ++kk) {                                         \
        const int k = (H) * HP + kk;                                                               \
        uint32_t kc = 0;                                                                           \
        _Pragma("unroll") for (int i = 0; i < KB; ++i) if ((k >> i) & 1) kc ^= ckb[i];            \
        PB[kk] = *(const double2*)(psi_b + (cb ^ kc));                                            \
        PA[kk] = *(const double2*)(psi_b + (cbx ^ kc));                                           \
      }
#define VQE_E_COMP(H, PA, PB, D)                                                                   \
      _Pragma("unroll") for (int kk = 0; kk < HP; ++kk) {                                         \
        const double v = (PA[kk].x * PB[kk].x + PA[kk].y * PB[kk].y) * D[(H) * HP + kk];          \
        if (kk & 1) acc1 += v; else acc0 += v;       /* pair tables hold 2*D */                   \
      }
      const double* treal = tables + __builtin_amdgcn_readfirstlane(L.gm[g0].off_r);
      VQE_E_CTX(VQE_E_XFETCH(g0))
      xq = VQE_E_XFETCH(g0 + 1);
      VQE_E_LOAD(0, aA, bA)
      for (int g = g0; g < g1; g += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
          // Refill the ring slot consumed in the PREVIOUS step first, so that these loads
          // have a whole step to land before anything waits on the vector-memory counter.
          {
            const int sl = (j + PD - 1) % PD;
            int gr = (j == 0) ? (g == g0 ? g0 + PD - 1 : g - 1 + PD) : g + j - 1 + PD;
            gr = gr < g1 ? gr : g1 - 1;            // slots past the end are never consumed
            const double* t = treal + (size_t)(gr - g0) * (DIM / 2);   // contiguous, fixed stride
#pragma unroll
            for (int k = 0; k < NP; ++k) buf[sl][k] = t[tid + (uint32_t)k * kThreads];
          }
          VQE_E_LOAD(1, aB, bB)                    // second half of group g+j
          __builtin_amdgcn_sched_barrier(0);
          VQE_E_COMP(0, aA, bA, buf[j])
          __builtin_amdgcn_sched_barrier(0);
          VQE_E_CTX(xq)                            // next group (g+j+1); its x came in a step ago
          xq = VQE_E_XFETCH(g + j + 2);
          VQE_E_LOAD(0, aA, bA)                    // first half of the next group
          __builtin_amdgcn_sched_barrier(0);
          VQE_E_COMP(1, aB, bB, buf[j])
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef VQE_E_CTX
#undef VQE_E_XFETCH
#undef VQE_E_LOAD
#undef VQE_E_COMP
    } else {
      for (int g = g0; g < g1; g += PD) {
        const int gn = g + PD < g1 ? g + PD : g;
#pragma unroll
        for (int j = 0; j < PD; ++j) {
          VQE_ENERGY_CONSUME(g + j, buf[j])
          const double* t = tables + __builtin_amdgcn_readfirstlane(L.gm[gn + j].off_r);
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const uint32_t q = tid + (uint32_t)k * kThreads;
            buf[j][k] = (FULL || q < DIM / 2) ? t[q] : 0.0;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

// Groups with imaginary parts (odd number of Y factors): rare, plain loop.
template <int N>
__device__ __forceinline__ void energy_imag_lds(const Lds& L, const double* __restrict__ tables, int g1, int g2,
                                                double& acc0) {
  constexpr int kThreads = Geo<N>::NT;
  constexpr uint32_t DIM = 1u << N;
  constexpr int NP = (DIM / 2 + kThreads - 1) / kThreads;
  constexpr bool FULL = DIM / 2 >= kThreads;
  const uint32_t tid = threadIdx.x;
  for (int g = g1; g < g2; ++g) {
    const GroupMeta m = L.gm[g];
    const double* tr = tables + m.off_r;
    const double* ti = tables + m.off_i;
    double part = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const uint32_t q = tid + (uint32_t)k * kThreads;
      if (FULL || q < DIM / 2) {
        const uint32_t p0 = insert0(q, m.hb);
        const double2 b = L.psi[p0], a = L.psi[p0 ^ m.x];
        part += (a.x * b.x + a.y * b.y) * tr[q] - (a.x * b.y - a.y * b.x) * ti[q];
      }
    }
    acc0 += part;   // pair tables hold 2*D
  }
}

template <int N>
__device__ __forceinline__ void unit_energy(const Lds& L, const HamDev& H, double& acc0, double& acc1);   // (below)

template <int N>
__device__ __forceinline__ double lds_energy(const Lds& L, const HamDev& H) {
  constexpr int kThreads = Geo<N>::NT;
  constexpr uint32_t DIM = 1u << N;
  constexpr int NA = (DIM + kThreads - 1) / kThreads;       // own amplitudes per thread
  const uint32_t tid = threadIdx.x;
  const double* __restrict__ tables = H.tables;
  double acc0 = 0.0, acc1 = 0.0;
  int g0 = 0;
  if (H.has_diag) {   // diagonal group: full-length table
    const double* t = tables + L.gm[0].off_r;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const uint32_t p = tid + (uint32_t)k * kThreads;
      if (DIM >= kThreads || p < DIM) {
        const double2 a = L.psi[p];
        const double v = (a.x * a.x + a.y * a.y) * t[p];
        if (k & 1) acc1 += v; else acc0 += v;
      }
    }
    g0 = 1;
  }
  const int g1 = g0 + H.n_real;        // multiple of PD groups with real tables
  energy_real_lds<N>(L, tables, g0, g1, acc0, acc1);
  energy_imag_lds<N>(L, tables, g1, H.n_groups, acc0);
  if constexpr (N >= kUnitMinQubits && N - 1 - Geo<N>::LT >= 1) unit_energy<N>(L, H, acc0, acc1);      // the mostly-zero groups (state in logical order here)
  return block_sum<Geo<N>::NW>(acc0 + acc1, L.red);
}

// Register path (n >= 10).  The state arrives in LDS in canonical order p' = tid | r << LT; each
// thread keeps its 2^R amplitudes own[r] in registers.  For a group whose x' has register bit
// `cls` (the highest one) the pairs {p', p' ^ x'} are covered exactly once by taking as one
// member the own amplitudes with bit cls of r clear: only the partner is read from LDS - half
// the LDS traffic of energy_real_lds, and none for the diagonal group.  Table entry [j][tid]
// belongs to r = insert0(j, cls).  Same software pipeline as energy_real_lds (half groups,
// PD-deep table ring, X mask fetched one step ahead); the class dispatch is a wave-uniform
// switch around the arithmetic only.

// Table reads through a buffer descriptor: one instruction per load (SGPR descriptor + SGPR
// group offset + loop-invariant VGPR lane offset), no 64-bit address arithmetic per group.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000);
}
__device__ __forceinline__ double2 buf_load_d2(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
  return make_double2(__hiloint2double(v.y, v.x), __hiloint2double(v.w, v.z));
}
__device__ __forceinline__ double buf_load_d(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
  return __hiloint2double(v.y, v.x);
}
__device__ __forceinline__ double2 lds_load_d2(lds_cbyte* base, uint32_t off) {
  const d2v_t v = *(const __attribute__((address_space(3))) d2v_t*)(base + off);
  return make_double2(v.x, v.y);
}

// NP accumulations  acc += (ax*bx + ay*by) * d  as ONE asm block: keeps the per-class bodies of
// the switch apart (identical C++ bodies are merged behind a register-select PHI = 2^R moves
// per half group) and leaves no hazard padding between them.
__device__ __forceinline__ void pair_fma4(double& acc0, double& acc1, const double2& a0, const double2& a1,
                                          const double2& a2, const double2& a3, const double2 (&b)[4], double d0,
                                          double d1, double d2, double d3) {
  double t0, t1, t2, t3;
  asm("v_mul_f64 %2, %6, %14\n\tv_mul_f64 %3, %8, %16\n\tv_mul_f64 %4, %10, %18\n\tv_mul_f64 %5, %12, %20\n\t"
      "v_fmac_f64 %2, %7, %15\n\tv_fmac_f64 %3, %9, %17\n\tv_fmac_f64 %4, %11, %19\n\tv_fmac_f64 %5, %13, %21\n\t"
      "v_fmac_f64 %0, %2, %22\n\tv_fmac_f64 %1, %3, %23\n\tv_fmac_f64 %0, %4, %24\n\tv_fmac_f64 %1, %5, %25"
      : "+v"(acc0), "+v"(acc1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(a2.x), "v"(a2.y), "v"(a3.x), "v"(a3.y),
        "v"(b[0].x), "v"(b[0].y), "v"(b[1].x), "v"(b[1].y), "v"(b[2].x), "v"(b[2].y), "v"(b[3].x), "v"(b[3].y),
        "v"(d0), "v"(d1), "v"(d2), "v"(d3));
}
__device__ __forceinline__ void pair_fma2(double& acc0, double& acc1, const double2& a0, const double2& a1,
                                          const double2 (&b)[2], double d0, double d1) {
  double t0, t1;
  asm("v_mul_f64 %2, %4, %8\n\tv_mul_f64 %3, %6, %10\n\t"
      "v_fmac_f64 %2, %5, %9\n\tv_fmac_f64 %3, %7, %11\n\t"
      "v_fmac_f64 %0, %2, %12\n\tv_fmac_f64 %1, %3, %13"
      : "+v"(acc0), "+v"(acc1), "=&v"(t0), "=&v"(t1)
      : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(b[0].x), "v"(b[0].y), "v"(b[1].x), "v"(b[1].y),
        "v"(d0), "v"(d1));
}
__device__ __forceinline__ void pair_fma1(double& acc0, const double2& a0, const double2 (&b)[1], double d0) {
  double t0;
  asm("v_mul_f64 %1, %2, %4\n\tv_fmac_f64 %1, %3, %5\n\tv_fmac_f64 %0, %1, %6"
      : "+v"(acc0), "=&v"(t0) : "v"(a0.x), "v"(a0.y), "v"(b[0].x), "v"(b[0].y), "v"(d0));
}

// Unit path: the X-mask groups whose sign-sum tables are mostly exact zeros (HamDev::urec, built by the host).  One
// pair per thread and unit, both members from LDS: the record (8 scalars, read with scalar loads - the records sit
// in the constant address space) gives the byte-address masks that deposit the thread id into the free index bits,
// the fixed bits, the X mask and the offset of the unit's NT table values.  No dispatch of any kind: 7 integer
// instructions, 2 ds_read_b128, one 8-byte table load and 3 FP64 instructions per unit and thread, against ~60
// instructions, 8 ds_read_b128 and 64 table bytes for a group of the class path below.
// (Measured and dropped: trips whose four units share ONE deposit - a hopping pair's four units differ in two filler
// bits only -: 13 instead of 28 integer instructions for 28 of the bench Hamiltonian's 45 trips, but a second loop, its
// own prologue and padding of both phases to whole turns: 326.4 ms against 327.8, not worth a second code path.)
// LDS read at an ABSOLUTE byte address.  The state region starts the dynamic LDS of these kernels and they have no
// static LDS (launch_lds checks the function's static size), so its address is 0; going through the `smem` symbol
// costs a v_add_u32 of that late-bound zero per read.
__device__ __forceinline__ double2 lds_load_abs(uint32_t addr) {
  const d2v_t v = *(const __attribute__((address_space(3))) d2v_t*)(uintptr_t)addr;
  return make_double2(v.x, v.y);
}
#ifndef VQE_UNIT_SHEAR
#define VQE_UNIT_SHEAR 0
#endif
constexpr bool kUnitShear = VQE_UNIT_SHEAR != 0;
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(4))) const u32x8_t const_u32x8;

template <int N>
__device__ __forceinline__ void unit_energy(const Lds& L, const HamDev& H, double& acc0, double& acc1) {
  constexpr int LT = Geo<N>::LT;
  constexpr int SEG = (N - 1 - LT) + 2;      // holes (fixed bits + selector) + 1 runs of free positions
  static_assert(SEG >= 2 && SEG <= 5, "unit records hold five deposit masks");
  constexpr int U = kUnitTrip;
  static_assert(kUnitUnroll == 3 * kUnitTrip, "three trips per turn of the loop (ring of three table buffers)");
  constexpr uint32_t TSTRIDE = (uint32_t)8 << LT;   // table bytes of one unit
  static_assert(U == 4, "the tables of a trip are interleaved [thread][unit of the trip]: two 16-byte loads per thread and trip");
  constexpr uint32_t TRIP_BYTES = U * TSTRIDE;
  const int nu = __builtin_amdgcn_readfirstlane(H.n_units);
  if (nu <= 0) return;
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const __amdgpu_buffer_rsrc_t ru = table_rsrc(H.utab);
  const const_u32x8* rec = (const const_u32x8*)H.urec;
  uint32_t tsh[SEG];
#pragma unroll
  for (int i = 0; i < SEG; ++i) tsh[i] = tid << (4 + i);
  // A trip = U units.  Table values come from L2 two trips ahead (ring of three register buffers, the loop body is
  // three trips so that the buffer indices are static).  The records of the next trip are requested into the SAME
  // scalar registers once the addresses of this trip are formed, and land while its LDS reads are in flight
  // (scalar and LDS loads share one counter: requested earlier, a wait for a record would also wait for LDS reads).
  double d[3][U];
  u32x8_t R[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    R[j] = rec[j];
  }
  const uint32_t tid32 = tid << 5;      // the thread's U table values of a trip are 32 contiguous bytes
  auto load_trip = [&](double (&dst)[U], uint32_t first_unit) {
    const uint32_t soff = (first_unit / U) * TRIP_BYTES;
    const double2 lo = buf_load_d2(ru, tid32, soff), hi = buf_load_d2(ru, tid32 + 16u, soff);
    dst[0] = lo.x; dst[1] = lo.y; dst[2] = hi.x; dst[3] = hi.y;
  };
  load_trip(d[0], 0u);
  load_trip(d[1], (uint32_t)(U < nu ? U : 0));
  const int last = nu - U;       // first unit of the last trip
  for (int u = 0; u < nu; u += 3 * U) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ub = u + k * U;
      const int u1 = ub + U < last ? ub + U : last;             // (past the end: re-read the last trip)
      const int u2 = ub + 2 * U < last ? ub + 2 * U : last;
      double2 pa[U], pb[U];
      uint32_t a[U], ax[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        // deposit of the thread id: one v_and_or_b32 per run of free positions (left to itself the compiler builds
        // an AND / OR3 tree of 7 instructions for the 5 runs)
        uint32_t v;
        asm("v_and_b32 %0, %1, %2" : "=v"(v) : "s"(R[j][0]), "v"(tsh[0]));
#pragma unroll
        for (int i = 1; i < SEG; ++i) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(v) : "v"(tsh[i]), "s"(R[j][i]));
        if constexpr (kUnitShear) {
          // the bank shear of the canonical index (kSwzTable) on the deposited thread id; the record's fixed bits
          // and X mask arrive sheared
          const uint32_t u4 = (v >> 6) & 0x3Cu;                  // 4 x (index bits 4..7)
          v ^= ((uint32_t)(kSwzTable >> u4) << 4) & 0xF0u;
        }
        a[j] = v ^ R[j][5];
        ax[j] = a[j] ^ R[j][6];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        pb[j] = lds_load_abs(a[j]);
        pa[j] = lds_load_abs(ax[j]);
      }
      load_trip(d[(k + 2) % 3], (uint32_t)u2);
      __builtin_amdgcn_sched_barrier(0);
      const const_u32x8* rn = rec + u1;
#pragma unroll
      for (int j = 0; j < U; ++j) R[j] = rn[j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const double v = (pa[j].x * pb[j].x + pa[j].y * pb[j].y) * d[k][j];
        if (j & 1) acc1 += v; else acc0 += v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Register path (n >= 10).  The state arrives in LDS in canonical order p' = tid | r << LT; each
// thread keeps its 2^R amplitudes own[r] in registers.  For a group whose x' has register bit
// `cls` (the highest one) the pairs {p', p' ^ x'} are covered exactly once by taking as one
// member the own amplitudes with bit cls of r clear: only the partner is read from LDS - half
// the LDS traffic of energy_real_lds, and none for the diagonal group.  Table entry
// [j/2][tid][j&1] belongs to r = insert0(j, cls).  Software pipeline over half groups with a
// PD-deep table ring as in energy_real_lds; the addressing records (ClsMeta) are fetched from
// LDS two groups ahead into two alternating register sets; the class dispatch is a
// wave-uniform switch around the arithmetic only.
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_pairs` runs once the class groups are done (the own amplitudes and the table ring are
// dead from there on): the env-step kernel uses it to start fetching the optimiser's matrices.
template <int N, class Hook = NoHook>
__device__ __forceinline__ double reg_energy(const Lds& L, const HamDev& H, Hook after_pairs = Hook()) {
  constexpr int kThreads = Geo<N>::NT;
  constexpr uint32_t DIM = 1u << N;
  constexpr int LT = Geo<N>::LT;
  constexpr int R = N - LT;
  constexpr int NA = 1 << R;
  constexpr int NP = NA / 2;
  constexpr int HP = NP / 2;                 // pairs per half group
  constexpr int PD = energy_pd(N);
  static_assert(NP >= 2 && NP <= 8 && PD % 2 == 0, "register energy path: 2..8 pairs per thread");
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque: address terms derived from it are rebuilt per evaluation, not kept live across the kernel
  const double* __restrict__ tables = H.tables;
  lds_cbyte* psi_l = (lds_cbyte*)L.psi;
  double2 own[NA];
#pragma unroll
  for (int r = 0; r < NA; ++r) own[r] = lds_load_d2(psi_l, (tid + (uint32_t)r * kThreads) << 4);
  double acc0 = 0.0, acc1 = 0.0;
  int g0 = 0;
  if (H.has_diag) {
    const __amdgpu_buffer_rsrc_t rd = table_rsrc(tables + __builtin_amdgcn_readfirstlane(L.gm[0].off_r));
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      const double v = (own[r].x * own[r].x + own[r].y * own[r].y) * buf_load_d(rd, (tid + (uint32_t)r * kThreads) << 3, 0);
      if (r & 1) acc1 += v; else acc0 += v;
    }
    g0 = 1;
  }
  const int gc = g0 + H.n_cls;
  if (g0 < gc) {
    const int ncls = H.n_cls;
    constexpr uint32_t GSTRIDE = DIM * 4;   // bytes of one pair table
    double2 buf[PD][NP / 2];
    const __amdgpu_buffer_rsrc_t rt = table_rsrc(tables + __builtin_amdgcn_readfirstlane(L.gm[g0].off_r));
    uint32_t voff[NP / 2];
#pragma unroll
    for (int k = 0; k < NP / 2; ++k) voff[k] = (tid + (uint32_t)k * kThreads) << 4;
    // slots 0 .. PD-2 only: slot PD-1 is requested by the first step like every later refill, so the
    // outstanding loads look the same on loop entry and on the back edge (the compiler's
    // s_waitcnt bookkeeping merges both and would otherwise wait for nearly everything there)
#pragma unroll
    for (int j = 0; j < PD - 1; ++j) {
#pragma unroll
      for (int k = 0; k < NP / 2; ++k) buf[j][k] = buf_load_d2(rt, voff[k], (uint32_t)j * GSTRIDE);
      __builtin_amdgcn_sched_barrier(0);   // keep the slots in issue order
    }
#pragma unroll
    for (int k = 0; k < NP / 2; ++k) buf[PD - 1][k] = make_double2(0.0, 0.0);
    double2 bA[HP], bB[HP];
    // addressing records of the group being loaded / the next one: two alternating sets
    uint2 mh[2];
    uint4 mo[2][2];
    const uint32_t tid16 = tid << 4;
    uint32_t tb;
    int ccls;
#define VQE_R_FETCH(S, GI)                                                                         \
    {                                                                                              \
      const ClsMeta* cm_ = L.cm + ((GI) < ncls ? (GI) : ncls - 1);                                 \
      mh[S] = *(const uint2*)cm_;                                                                  \
      mo[S][0] = *(const uint4*)cm_->off;                                                          \
      if constexpr (NP > 4) mo[S][1] = *(const uint4*)(cm_->off + 4);                             \
    }
#define VQE_R_OFF(S, J) ((J) == 0 ? mo[S][0].x : (J) == 1 ? mo[S][0].y : (J) == 2 ? mo[S][0].z : (J) == 3 ? mo[S][0].w : \
                         (J) == 4 ? mo[S][1].x : (J) == 5 ? mo[S][1].y : (J) == 6 ? mo[S][1].z : mo[S][1].w)
#define VQE_R_LOAD(S, HH, PB)                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < HP; ++kk)                                             \
      PB[kk] = lds_load_d2(psi_l, tb + VQE_R_OFF(S, (HH) * HP + kk));
#define VQE_R_OWN(C, J) own[(((J) >> (C)) << ((C) + 1)) | ((J) & ((1 << (C)) - 1))]
#define VQE_R_DV(D, J) (((J) & 1) ? D[(J) / 2].y : D[(J) / 2].x)
#define VQE_R_COMP_C(C, HH, PB, D)                                                                 \
    if constexpr (HP == 4)                                                                         \
      pair_fma4(acc0, acc1, VQE_R_OWN(C, (HH) * 4), VQE_R_OWN(C, (HH) * 4 + 1), VQE_R_OWN(C, (HH) * 4 + 2),     \
                VQE_R_OWN(C, (HH) * 4 + 3), PB, VQE_R_DV(D, (HH) * 4), VQE_R_DV(D, (HH) * 4 + 1),            \
                VQE_R_DV(D, (HH) * 4 + 2), VQE_R_DV(D, (HH) * 4 + 3));                                       \
    else if constexpr (HP == 2)                                                                    \
      pair_fma2(acc0, acc1, VQE_R_OWN(C, (HH) * 2), VQE_R_OWN(C, (HH) * 2 + 1), PB, VQE_R_DV(D, (HH) * 2),     \
                VQE_R_DV(D, (HH) * 2 + 1));                                                                  \
    else                                                                                           \
      pair_fma1(acc0, VQE_R_OWN(C, (HH)), PB, VQE_R_DV(D, (HH)));
    // one class dispatch per group: first half, loads of the next group's first half, second half
#define VQE_R_BODY(C, JJ)                                                                          \
    {                                                                                              \
      VQE_R_COMP_C(C, 0, bA, buf[JJ])                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      tb = tid16 ^ mh[((JJ) + 1) & 1].x;                                                           \
      VQE_R_LOAD(((JJ) + 1) & 1, 0, bA)                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      VQE_R_COMP_C(C, 1, bB, buf[JJ])                                                              \
    }
    VQE_R_FETCH(0, 0)
    VQE_R_FETCH(1, 1)
    tb = tid16 ^ mh[0].x;
    VQE_R_LOAD(0, 0, bA)
    for (int g = 0; g < ncls; g += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        {   // refill the ring slot consumed in the previous step (see energy_real_lds)
          const int sl = (j + PD - 1) % PD;
          int gr = (j == 0) ? (g == 0 ? PD - 1 : g - 1 + PD) : g + j - 1 + PD;
          gr = gr < ncls ? gr : ncls - 1;
          const uint32_t soff = (uint32_t)gr * GSTRIDE;
#pragma unroll
          for (int k = 0; k < NP / 2; ++k) buf[sl][k] = buf_load_d2(rt, voff[k], soff);
        }
        ccls = __builtin_amdgcn_readfirstlane((int)mh[j & 1].y);
        VQE_R_LOAD(j & 1, 1, bB)             // second half of group g+j
        VQE_R_FETCH(j & 1, g + j + 2)        // its record is free now: fetch that of group g+j+2
        __builtin_amdgcn_sched_barrier(0);
        switch (ccls) {
          case 0: VQE_R_BODY(0, j) break;
          case 1: VQE_R_BODY(1, j) break;
          case 2: if constexpr (R > 2) VQE_R_BODY(2, j) break;
          default: if constexpr (R > 3) VQE_R_BODY(3, j) break;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef VQE_R_FETCH
#undef VQE_R_OFF
#undef VQE_R_LOAD
#undef VQE_R_OWN
#undef VQE_R_DV
#undef VQE_R_COMP_C
#undef VQE_R_BODY
  }
  after_pairs();
  unit_energy<N>(L, H, acc0, acc1);
  const int g1 = g0 + H.n_real;
  energy_real_lds<N>(L, tables, gc, g1, acc0, acc1);
  energy_imag_lds<N>(L, tables, g1, H.n_groups, acc0);
  return block_sum<Geo<N>::NW>(acc0 + acc1, L.red);
}

// One evaluation with the ops already compiled: circuit, then <psi|H|psi>.
// cs_ready: the (cos, sin) of the angles are in L.cs already (StagedCobyla::out formed them from the trial point)
template <int N, bool SLOTS = false, class Hook = NoHook>
__device__ __forceinline__ double lds_evaluate(const BatchArgs& A, const Lds& L, const double* theta, int P,
                                               int p_hole = -1, Hook after_pairs = Hook(), bool cs_ready = false) {
#ifdef VQE_STAMPS
  const long long t0 = clock64();
#endif
  if constexpr (N >= kRegMinQubits) {
    run_ops_reg<N, SLOTS>(L, A.init, theta, P, p_hole, A.dbg, cs_ready);
  } else {
    load_init<N>(L, A.init);
    __syncthreads();
    run_ops<N, SLOTS>(L, theta, P, p_hole);
  }
#ifdef VQE_STAMPS
  const long long t1 = clock64();
#endif
  double e;
  if constexpr (N >= kRegMinQubits) e = reg_energy<N>(L, A.ham, after_pairs); else e = lds_energy<N>(L, A.ham);
#ifdef VQE_STAMPS   // diagnostic build only: cycles per phase, summed over workgroups (thread 0)
  if (threadIdx.x == 0) {
    atomicAdd(A.dbg + 0, 1ull);
    atomicAdd(A.dbg + 1, (unsigned long long)(t1 - t0));
    atomicAdd(A.dbg + 2, (unsigned long long)(clock64() - t1));
  }
#endif
  return e;
}

// COBYLA with its matrices staged into the state region of LDS while the state is dead
// (between two evaluations); falls back to the global scratch when they do not fit.
struct NoSide { __device__ __forceinline__ void operator()() const {} };

// K double2 values that stay in registers: members of a recursive struct are scalars from the start (a plain member
// array that is indexed inside loops and lambdas is left in scratch memory by the compiler)
template <int K>
struct KeepRegs {
  double2 v;
  KeepRegs<K - 1> rest;
  template <class F> __device__ __forceinline__ void load(F f, int k = 0) { v = f(k); rest.load(f, k + 1); }
  template <class F> __device__ __forceinline__ void store(F f, int k = 0) const { f(k, v); rest.store(f, k + 1); }
};
template <>
struct KeepRegs<0> {
  template <class F> __device__ __forceinline__ void load(F, int = 0) {}
  template <class F> __device__ __forceinline__ void store(F, int = 0) const {}
};

// WIDE: compile the workgroup-wide update in (always for n >= 10; below that only in the kernel variant that is
// launched for batches with more than 64 parameters - the trainable regime - because its registers cost the
// 128-VGPR kernels 10-20 % at small parameter counts)
template <int N, bool WIDE = (N >= 10 && Geo<N>::NT >= 256)>
struct StagedCobyla {
  static constexpr int kThreads = Geo<N>::NT;
  typedef cby::CobylaM0<WaveCtx, false, lds_double> CobL;   // arrays in LDS (ds_ instructions)
  // WIDE on a one-wave workgroup: no second wave to spread the update over - the lanes walk their rows side by side
  static constexpr bool kBlock = WIDE && Geo<N>::NT >= 256;
  static constexpr bool kRowsG = WIDE && Geo<N>::NT == 64;
  typedef cby::CobylaM0<WaveCtx, false, double> CobG;       // arrays in the global scratch, one wave
  // ... and with the row / column walks of the trainable regime (WaveRowsCtx): problems with more than 32 variables,
  // i.e. those whose rows are not split over lane pairs - the padding to 16 moves the split point of a pair, nothing
  // else: for every other problem the two contexts produce the same bits, so a circuit's result does not depend on
  // which kernel variant its batch selected
  typedef cby::CobylaM0<WaveRowsCtx, false, double> CobR;
  __device__ __forceinline__ bool rows_ctx() const { return kRowsG && n > 32; }
#ifndef VQE_BLOCK_TILE
#define VQE_BLOCK_TILE 1
#endif
  // (tiles of the workgroup context: NW x 8.7 KB + the shared vector in the dead state region - 16 << N bytes)
  static constexpr bool kBlockTile = VQE_BLOCK_TILE && ((size_t)16 << N) >= (Geo<N>::NT / 64) * kCobTileBytes + 4096;
  typedef cby::CobylaM0<BlockCtx<Geo<N>::NT, kBlockTile>, false, double> CobB;   // the same, whole workgroup (more rows than a wave has lanes)
  double* red;     // LDS words of the block reductions
  bool block;      // more rows than a wave has lanes: workgroup-wide context on the global scratch
  double* gmem;    // per-problem scratch; x[] is its first array
  double* lmem;    // the (dead) state region of LDS
  double* tile;    // LDS transposition tile of the one-wave context on more than 64 variables (kRowsG)
  int* pub;        // LDS: what wave 0 publishes to the workgroup after a call
  int n, words;
  bool staged;
  bool resident;   // the arrays live in an LDS region of their own (Lds::cob): nothing is copied
  int want, nfvals;   // published after start()/tell()
  // PRE: the first kStage double2 per thread of the staging copy are requested from L2 while the energy step still
  // runs (prefetch(), called from reg_energy once its class loop has released its registers) and written to LDS by
  // in(): the 2.7 k cycles of L2 latency that in() used to wait for are gone (round 3; n >= 10 only)
#ifndef VQE_STAGE_PREFETCH
#define VQE_STAGE_PREFETCH 1
#endif
#ifndef VQE_LIGHT_OUT
#define VQE_LIGHT_OUT 0      // measured (12 qubits, bench workload): prefetch alone 322.9 ms, light barrier alone 329.6, both 329.2, neither 324.5
#endif
  static constexpr bool PRE = VQE_STAGE_PREFETCH && N >= 10 && Geo<N>::NT >= 256;
  KeepRegs<PRE ? 6 : 0> pre;
  bool pre_valid;
  __device__ __forceinline__ void init(double* global_scratch, const Lds& L, int n_) {
    pre_valid = false;
    gmem = global_scratch;
    resident = L.cob != nullptr;
    lmem = resident ? L.cob : (double*)L.psi;
    tile = L.tile;
    pub = (int*)(L.red + 8);
    n = n_;
    words = (int)cby::scratch_doubles(n, WaveCtx::kPad);
    red = L.red;
    block = kBlock && n > 64;
    staged = resident || (!block && (size_t)words * 8 <= ((size_t)16 << N));
    if (block) words = (int)cby::scratch_doubles(n, CobB::P);
    // arrays that stay in the global scratch use the 64-byte row layout (cobyla_m0.h: lead_dim_global)
    // (one-wave context only: with a thread per row - BlockCtx - the aligned stride measured 3.5 % slower at 12 qubits /
    // 202 variables, 4.6 % faster for the one-wave context at 8 qubits / 129 variables)
    if (!staged && !block) {
      const int pad = rows_ctx() ? CobR::P : CobG::P;      // (16 with the tile context: rows are whole 128-byte lines)
      words = (int)cby::scratch_doubles_ld(n, pad, cby::lead_dim_global(cby::padded(n, pad)));
    }
  }
  __device__ __forceinline__ double* x() const { return resident ? lmem : gmem; }
  // the optimiser's scalars as parked in the scratch (valid after start()/tell())
  __device__ __forceinline__ const double* state() const { return (resident ? lmem : gmem) + words - cby::kStateDoubles; }
  // Staging copies: all loads of a thread are issued before its first store (a plain strided loop
  // waits for every load in turn: ~5 dependent L2 round trips per call), in chunks of kStage
  // double2 per thread.
  static constexpr int kStage = 6;
  __device__ __forceinline__ void prefetch() {
    if constexpr (PRE) {
      if (!staged || resident) return;
      const double2* s = (const double2*)gmem;
      const int nw = (words + 1) / 2;
      pre.load([&](int k) {
        const int i = (int)threadIdx.x + k * kThreads;
        return s[i < nw ? i : nw - 1];
      });
      pre_valid = true;
    }
  }
  __device__ __forceinline__ void in() {
    if (!staged || resident) return;
    const double2* s = (const double2*)gmem;
    double2* d = (double2*)lmem;
    const int nw = (words + 1) / 2;
    int first = threadIdx.x;
    if constexpr (PRE) {
      if (pre_valid) {      // chunk 0 is in registers already
        pre.store([&](int k, const double2& v) {
          const int i = (int)threadIdx.x + k * kThreads;
          if (i < nw) d[i] = v;
        });
        pre_valid = false;
        first += kStage * kThreads;
      }
    }
    for (int i0 = first; i0 < nw; i0 += kStage * kThreads) {
      double2 v[kStage];
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        const int i = i0 + k * kThreads;
        v[k] = s[i < nw ? i : nw - 1];
      }
      // (opaque uses: left alone, the compiler sinks every load into the guarded store below and
      // waits for it there - kStage dependent round trips)
#pragma unroll
      for (int k = 0; k < kStage; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        const int i = i0 + k * kThreads;
        if (i < nw) d[i] = v[k];
      }
    }
    __syncthreads();
  }
  // cs_out != nullptr (and the optimiser wants another evaluation): the (cos, sin)(x/2) of the trial point are formed
  // HERE from the LDS copy of x (parameter j of the circuit at cs_out[j]; p_hole: the parameter that is not a
  // variable), so the coming evaluation reads nothing that out() writes to global memory and the closing barrier need
  // not wait for the write-back to be acknowledged (LDS-only barrier: 2.5 k cycles of every tell()).  `full`: the caller
  // reads x / the scalars from the scratch next (finish, trace): the ordinary barrier.
  __device__ __forceinline__ bool out(bool full = true, double2* cs_out = nullptr, int P = 0, int p_hole = -1) {
    if (threadIdx.x == 0) { pub[0] = want; pub[1] = nfvals; }
    __syncthreads();
    want = pub[0]; nfvals = pub[1];
    if (!staged || resident) return false;
    const bool light = VQE_LIGHT_OUT && !full && want && cs_out != nullptr;
    if (light) {
      for (int j = threadIdx.x; j < P; j += kThreads) {
        if (j == p_hole) continue;
        double sn, cn;
        sincos(0.5 * lmem[j - (p_hole >= 0 && j > p_hole)], &sn, &cn);
        cs_out[j] = make_double2(cn, sn);
      }
    }
    const double2* s = (const double2*)lmem;
    double2* d = (double2*)gmem;
    const int nw = (words + 1) / 2;
    for (int i0 = threadIdx.x; i0 < nw; i0 += kStage * kThreads) {
      double2 v[kStage];
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        const int i = i0 + k * kThreads;
        v[k] = s[i < nw ? i : nw - 1];
      }
      // (opaque uses: left alone, the compiler sinks every load into the guarded store below and
      // waits for it there - kStage dependent round trips)
#pragma unroll
      for (int k = 0; k < kStage; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
#pragma unroll
      for (int k = 0; k < kStage; ++k) {
        const int i = i0 + k * kThreads;
        if (i < nw) d[i] = v[k];
      }
    }
    // every thread has its part of the staging area in registers by now (the stores above waited for
    // the LDS reads), and whatever overwrites the state region next - the first re-layout or the
    // final scatter of the coming evaluation, the x[] copy of the result - sits behind a barrier
    if (light) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS reads done; the global stores drain on their own
      return true;
    }
    __syncthreads();
    return false;
  }
  // FIRST: start() instead of tell().  The optimiser object lives only inside this call.
  template <bool FIRST, class Cob, class Ptr>
  __device__ __forceinline__ void call(Ptr mem, double f, double rhobeg, double rhoend, int maxfun) {
    Cob cob;
    cob.ctx.tid = Cob::P == 0 ? 0 : (int)(threadIdx.x % (unsigned)decltype(cob.ctx)::nth);
    // the workgroup contexts reduce through LDS words.  (Round 2 tested `nth != 64` here to tell them from WaveCtx:
    // BlockCtx<64> has 64 threads too, so a one-wave instantiation of the workgroup-wide update ran arg_first / all_or
    // through an UNINITIALISED `red` pointer - the "hang" recorded in round 2)
    if constexpr (!std::is_base_of<WaveCtx, decltype(cob.ctx)>::value) cob.ctx.red = red;
    if constexpr (decltype(cob.ctx)::kTile) {
      if constexpr (std::is_base_of<WaveCtx, decltype(cob.ctx)>::value) {
        cob.ctx.tile = (lds_double*)tile;
        cob.ctx.shared = (lds_double*)tile + Cob::kTileDoubles;
      } else {      // workgroup context: a tile per wavefront and the shared vector (<= 512 variables) in the dead state region
        cob.ctx.tile = (lds_double*)lmem + (threadIdx.x >> 6) * Cob::kTileDoubles;
        cob.ctx.shared = (lds_double*)lmem + (kThreads / 64) * Cob::kTileDoubles;
      }
    }
    // opaque copies: otherwise the array addresses bind() derives are loop invariants of the
    // evaluation loop, get hoisted out of it, spilled across the energy step (where registers
    // are scarcest) and reloaded from scratch inside every tell()
    int nn = n;
    if constexpr (N >= 10) asm volatile("" : "+s"(mem), "+s"(nn));   // (uniform: scalar registers; n <= 9: measured -15 %)
    cob.bind(mem, nn);
    if (FIRST) {
      want = cob.start(rhobeg, rhoend, maxfun);
    } else {
      cob.load_state();
      want = cob.tell(f);
    }
    nfvals = cob.nfvals;
    cob.save_state();
  }
  // `side()` runs on the second wave while the first one does the optimiser's bookkeeping (the
  // env-step kernel prepares the noise patches of the next evaluation there)
  bool cs_ready;      // out() formed the (cos, sin) of the coming evaluation
  template <bool FIRST, class Side>
  __device__ __forceinline__ int run(double f, double rhobeg, double rhoend, int maxfun, Side side, bool full = true,
                                     double2* cs_out = nullptr, int P = 0, int p_hole = -1) {
#ifdef VQE_STAMPS
    const long long t0 = clock64();
#endif
    in();
#ifdef VQE_STAMPS
    const long long t1 = clock64();
#endif
    if (kBlock && block) {
      if constexpr (kBlock) {
        if (kThreads == 64 || (threadIdx.x >= 64 && threadIdx.x < 128)) side();
        call<FIRST, CobB>(gmem, f, rhobeg, rhoend, maxfun);     // all threads: contains barriers
      }
    } else if (threadIdx.x < 64) {
      if (staged) call<FIRST, CobL>((lds_double*)lmem, f, rhobeg, rhoend, maxfun);
      else if (rows_ctx()) {
        if constexpr (kRowsG) call<FIRST, CobR>(gmem, f, rhobeg, rhoend, maxfun);
      } else call<FIRST, CobG>(gmem, f, rhobeg, rhoend, maxfun);
      if constexpr (kThreads == 64) side();     // one-wave workgroups: no second wave to give the side job to
    } else if (threadIdx.x < 128) {
#ifdef VQE_STAMPS
      const long long ts0 = clock64();
#endif
      side();
#ifdef VQE_STAMPS
      if (threadIdx.x == 64 && !FIRST) atomicAdd(&g_cby_dbg[6], (unsigned long long)(clock64() - ts0));   // the side job of wave 1
#endif
    }
#ifdef VQE_STAMPS
    const long long t2 = clock64();
#endif
    cs_ready = out(full, cs_out, P, p_hole);
#ifdef VQE_STAMPS
    if (threadIdx.x == 0 && !FIRST) {
      atomicAdd(&g_cby_dbg[7], (unsigned long long)(t2 - t1));          // wave 0: load_state + tell + save_state
      atomicAdd(&g_cby_in_out[0], (unsigned long long)(t1 - t0));       // staging in
      atomicAdd(&g_cby_in_out[1], (unsigned long long)(clock64() - t2)); // publish + staging out
    }
#endif
    return want;
  }
  __device__ __forceinline__ int start(double rhobeg, double rhoend, int maxfun, bool full = true, double2* cs_out = nullptr,
                                       int P = 0, int p_hole = -1) {
    return run<true>(0.0, rhobeg, rhoend, maxfun, NoSide(), full, cs_out, P, p_hole);
  }
  template <class Side>
  __device__ __forceinline__ int tell(double f, unsigned long long* __restrict__ dbg, Side side, bool full = true,
                                      double2* cs_out = nullptr, int P = 0, int p_hole = -1) {
#ifdef VQE_STAMPS
    const long long t0 = clock64();
#endif
    const int w = run<false>(f, 0.0, 0.0, 0, side, full, cs_out, P, p_hole);
#ifdef VQE_STAMPS
    if (threadIdx.x == 0) atomicAdd(dbg + 4, (unsigned long long)(clock64() - t0));
#endif
    return w;
  }
};

// ---- kernels -----------------------------------------------------------------------------
template <int N>
__global__ void __launch_bounds__(Geo<N>::NT) k_lds_energy(BatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, N, A.max_ops, A.max_params, A.ham.n_groups, A.max_pair);
  const int b = blockIdx.x;
  const bool noisy = (A.noise.p1 > 0.0 || A.noise.p2 > 0.0);
  stage_groups(A.ham, L);
  if constexpr (N >= 10) stage_cls<N>(A.ham, L);
  compile_all<N>(A, b, 0, L, -1, true, noisy);
  if (noisy) patch_noise<N>(A, b, A.noise.eval_base, L, -1, true);
  double e = lds_evaluate<N, true>(A, L, A.theta + A.par_begin[b], A.par_count[b]);
  if (A.noise.shot_sigma != 0.0) e += A.noise.shot_sigma * noise_gauss(A.noise.seed, (uint64_t)b, A.noise.eval_base);
  if (threadIdx.x == 0) { A.fout[b] = e; if (A.nfev) A.nfev[b] = 1; }
}

template <int N>
__global__ void __launch_bounds__(Geo<N>::NT) k_lds_state(BatchArgs A) {
  constexpr int kThreads = Geo<N>::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, N, A.max_ops, A.max_params, A.ham.n_groups, A.max_pair);
  const bool noisy_state = (A.noise.p1 > 0.0 || A.noise.p2 > 0.0);
  compile_all<N>(A, 0, 0, L, -1, false, noisy_state);   // logical order for the read-out
  if (noisy_state) patch_noise<N>(A, 0, A.noise.eval_base, L, -1, false);
  if constexpr (N >= kRegMinQubits) {
    run_ops_reg<N, true>(L, A.init, A.theta + A.par_begin[0], A.par_count[0]);
  } else {
    load_init<N>(L, A.init);
    __syncthreads();
    run_ops<N>(L, A.theta + A.par_begin[0], A.par_count[0]);
  }
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
//
// Plain mode: minimise E(theta) of circuit b from x0 = theta.
// With new_gate[b] = g >= 0 the loop follows CircuitEnv.step of the reference
// (environments/environment_qulacs_TN_notin_agent.py:283-291,452-482): COBYLA optimises the
// circuit WITHOUT gate g (the action just taken; its angle, if it is a rotation, is not a
// variable), and with env_step = 1 the optimum is rounded to float32 (the state tensor's
// dtype, :480) and the energy of the FULL circuit is reported (:291).
// NOISY: the stochastic instantiation (depolarising channels): the noiseless one carries none of its code
template <int N, bool WIDE = (N >= 10 && Geo<N>::NT >= 256), bool NOISY = false>
__global__ void __launch_bounds__(Geo<N>::NT, Geo<N>::WPS) k_lds_minimize(BatchArgs A) {
  constexpr int kThreads = Geo<N>::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, N, A.max_ops, A.max_params, A.ham.n_groups, A.max_pair);
  const int b = A.order[blockIdx.x];
  const int P = A.par_count[b];
  const double* theta = A.theta + A.par_begin[b];
  double* xout = A.xout + A.par_begin[b];
  constexpr bool noisy = NOISY;
  const int skip = A.new_gate ? A.new_gate[b] : -1;
  int p_hole = -1;
  int skip_end = skip + 1;   // (no new gate: the empty range [-1, 0))
  if (skip >= 0) {
    const GateRec r = A.gates[A.gate_begin[b] + skip];
    if (r.kind >= G_RX && r.kind <= G_RZ) p_hole = r.pidx;
    // the noise channel construct_ansatz puts behind every gate belongs to that gate: the
    // pre-action circuit contains neither (VQE_qulacs_TN_notin_RL_noise.py:26-28,40-50)
    if (skip + 1 < A.gate_count[b]) {
      const GateRec f = A.gates[A.gate_begin[b] + skip + 1];
      if ((f.kind == G_DEPOL1 && r.kind >= G_RX && r.kind <= G_RZ && f.q0 == r.q0) ||
          (f.kind == G_DEPOL2 && r.kind == G_CNOT && f.q0 == r.q0 && f.q1 == r.q1))
        skip_end = skip + 2;
    }
  }
  const int Popt = P - (p_hole >= 0);
  stage_groups(A.ham, L);
  if constexpr (N >= 10) stage_cls<N>(A.ham, L);
  StagedCobyla<N, WIDE> sc;
  sc.cs_ready = false;
  // the write-back of the optimiser's arrays must be visible to the rest of the workgroup only when somebody reads it
  // from the scratch: the trace does (the result path asks for it itself)
  const bool full_out = N < 10 || A.trace != nullptr;
  // phases: 0 = single evaluation (empty x0: scipy returns after one call), 1 = COBYLA loop,
  // 2 = post-action evaluation of env_step.  ONE evaluation call site keeps everything inlined.
  int phase = 0, nfev = 1;
  bool need_compile = true;
  bool patched = false;   // the noise patches of the coming evaluation are already in (made during tell())
  double fret = 0.0, flast = 0.0;
  if (Popt > 0) {
    sc.init(A.scratch + A.scratch_begin[b], L, Popt);
    for (int j = threadIdx.x; j < P; j += kThreads)
      if (j != p_hole) sc.x()[j - (p_hole >= 0 && j > p_hole)] = theta[j];
    __syncthreads();
    sc.start(A.rhobeg, A.rhoend, A.maxfun, full_out, L.cs, P, p_hole);   // always asks for f(x0)
    phase = 1;
  }
  for (;;) {
    const double* th = phase == 0 ? theta : (phase == 1 ? sc.x() : xout);
    const int sk = phase == 2 ? -1 : skip;
    const int ske = phase == 2 ? 0 : skip_end;
    const int ph = phase == 2 ? -1 : p_hole;
    const uint64_t eid = A.noise.eval_base +
                         (phase == 1 ? (uint64_t)sc.nfvals : (phase == 2 ? (uint64_t)A.maxfun + 1 : 0));
    if (need_compile) { compile_all<N>(A, b, 0, L, sk, true, noisy, ske); need_compile = false; patched = false; }
    if (noisy && !patched) patch_noise<N>(A, b, eid, L, sk, true, ske);
    patched = false;
    const bool csr = phase == 1 && sc.cs_ready;
    double e = lds_evaluate<N, NOISY>(A, L, th, P, ph, [&]() { if (phase == 1) sc.prefetch(); }, csr);
    // finite-shot estimate of <H>: Gaussian with the total standard deviation the caller set
    if (A.noise.shot_sigma != 0.0) e += A.noise.shot_sigma * noise_gauss(A.noise.seed, (uint64_t)b, eid);
    bool finished_opt = false;
    if (phase == 0) {
      fret = e;
      for (int j = threadIdx.x; j < P; j += kThreads) {
        xout[j] = A.env_step ? (double)(float)theta[j] : theta[j];
        A.xraw[A.par_begin[b] + j] = theta[j];
      }
      finished_opt = true;
    } else if (phase == 1) {
      flast = e;
      if (A.trace) {   // diagnostic: the trial point and its value, for trajectory-level parity tests
        double* tr = A.trace + ((size_t)b * A.maxfun + (size_t)(sc.nfvals - 1)) * (size_t)(1 + A.max_params);
        if (threadIdx.x == 0) tr[0] = e;
        for (int j = threadIdx.x; j < Popt; j += kThreads) tr[1 + j] = sc.x()[j];
      }
#ifdef VQE_STAMPS
      const long long tt0 = clock64();
#endif
      // while wave 0 updates the simplex, wave 1 draws and applies the errors of the next evaluation
      const uint64_t eid_next = A.noise.eval_base + (uint64_t)sc.nfvals + 1;
      const int want = sc.tell(e, A.dbg, [&]() {
        if (noisy) patch_noise_wave<N>(A, b, eid_next, L, sk, true, (int)(threadIdx.x & 63), ske);
      }, full_out, L.cs, P, p_hole);
      patched = noisy && want;
#ifdef VQE_STAMPS
      if (threadIdx.x == 0) atomicAdd(A.dbg + 3, (unsigned long long)(clock64() - tt0));
#endif
      if (!want) {
        for (int j = threadIdx.x; j < P; j += kThreads) {
          const double v = (j == p_hole) ? theta[j] : sc.x()[j - (p_hole >= 0 && j > p_hole)];
          xout[j] = A.env_step ? (double)(float)v : v;
          A.xraw[A.par_begin[b] + j] = v;
        }
        const double* st = sc.state();
        fret = ((int)st[11] == cby::DONE_RHOEND && (int)st[10] == 1) ? flast : st[4];
        nfev = sc.nfvals;
        finished_opt = true;
      }
    } else {
      fret = e;
      break;
    }
    if (finished_opt) {
      if (!A.env_step) break;
      __syncthreads();          // xout visible to the whole workgroup
      phase = 2;
      need_compile = true;      // the full circuit, including the new gate
    }
  }
  if (threadIdx.x == 0) { A.fout[b] = fret; A.nfev[b] = nfev; }
}

}  // namespace vqe
