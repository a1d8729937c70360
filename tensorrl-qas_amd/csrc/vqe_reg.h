// vqe_reg.h - register-resident circuit application for 10 <= n <= 13 (included by
// vqe_device.h).
//
// Every thread keeps NA = 2^(n-8) amplitudes in VGPRs.  A *layout* is a basis of GF(2)^n
// split into R = n-8 "register directions" e_0..e_{R-1} (fully reduced echelon form, pivot =
// highest set bit) and 8 unit vectors on the non-pivot bit positions that are addressed by
// the thread id:   physical index (tid, r) = deposit(tid -> non-pivot bits) ^ XOR_{i in r} e_i.
//
// A rotation whose pair mask xm lies in span(e) pairs register r with register r ^ j
// (j = coordinates of xm), entirely inside the thread: no LDS traffic, no barrier.  RZ and
// Pauli-Z are diagonal and always local.  Thread 0 schedules the circuit greedily: when a
// mask falls outside the current span it opens a new layout whose register directions are
// the next R independent masks of the circuit, and the workgroup re-distributes the state
// through LDS once (write 16 B * 2^n, barrier, read).  LDS bandwidth - the limiter of the
// plain LDS-state version (ds_write_b128 ~79 B/clk/CU) - is then paid once per layout
// (~4 two-level rotations + all diagonal ones) instead of once per rotation.  After the last
// op every thread scatters its amplitudes to LDS in LOGICAL order for the energy phase.
#pragma once

namespace vqe {

constexpr int kRegMinQubits = 10;
enum : int { OP_RELAYOUT = 5 };

struct LayoutRec {  // 32 bytes
  uint32_t e[5];     // register directions (reduced echelon form)
  uint32_t pos;      // 8 x 4 bits: non-pivot bit positions 0..7, ascending
  uint32_t pos8;     // the ninth position (512-thread workgroups)
  uint32_t pad;
};

// ---- scheduler (thread 0) ---------------------------------------------------------------------
struct Basis {           // storage lives in LDS (runtime-indexed arrays would land in scratch)
  uint32_t* e;           // [5]
  int32_t* piv;          // [5]
  int k;
  __device__ void clear() { k = 0; }
  __device__ uint32_t reduce(uint32_t v) const {
    for (int i = 0; i < k; ++i) if ((v >> piv[i]) & 1u) v ^= e[i];
    return v;
  }
  __device__ bool add(uint32_t v) {      // keeps the basis fully reduced
    v = reduce(v);
    if (!v) return false;
    const int p = 31 - __clz((int)v);
    for (int i = 0; i < k; ++i) if ((e[i] >> p) & 1u) e[i] ^= v;
    e[k] = v; piv[k] = p; ++k;
    return true;
  }
  __device__ uint32_t coords(uint32_t v) const {   // v must be in the span
    uint32_t j = 0;
    for (int i = 0; i < k; ++i) j |= ((v >> piv[i]) & 1u) << i;
    return j;
  }
};

template <int N>
__device__ __forceinline__ void emit_layout(const Basis& B, LayoutRec* out) {
  uint32_t pivmask = 0;
  for (int i = 0; i < 5; ++i) out->e[i] = i < B.k ? B.e[i] : 0u;
  for (int i = 0; i < B.k; ++i) pivmask |= 1u << B.piv[i];
  uint32_t pos = 0, pos8 = 0;
  int cnt = 0;
  for (int b = 0; b < N; ++b)
    if (!((pivmask >> b) & 1u)) {
      if (cnt < 8) pos |= (uint32_t)b << (4 * cnt); else pos8 = (uint32_t)b;
      ++cnt;
    }
  out->pos = pos;
  out->pos8 = pos8;
}

// bit r of the result = parity(r & svec)
template <int R>
__device__ __forceinline__ uint32_t sign_word(uint32_t svec) {
  constexpr uint32_t pat[5] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};
  uint32_t w = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) if ((svec >> i) & 1u) w ^= pat[i];
  return w;
}

// raw ops (L.ops, n = meta[0]) -> scheduled ops (L.sched, n = meta[4]) + layouts (L.lay).
// Scheduled record: xm = sign word of the op in its layout (bit r: parity of the op's Z mask
// with register combination r), or the layout index of a RELAYOUT; kind = op | inv << 8 |
// partner mask j << 16; pidx = BYTE offset of the op's (cos, sin) in L.cs (0 when it has none):
// everything a thread would otherwise recompute for every evaluation.  Bits 24..31 of kind: number of records
// behind this one that the run loop may jump over (inactive noise slots; set per evaluation by patch_noise_wave).
template <int N>
__device__ __forceinline__ void schedule_ops(const Lds& L) {
  constexpr int R = N - Geo<N>::LT;
  const int nraw = L.meta[0];
  int ns = 0, nl = 0;
  Basis B;
  B.e = (uint32_t*)L.sb;
  B.piv = (int32_t*)L.sb + 8;
  auto open_layout = [&](int from) {
    B.clear();
    for (int k = from; k < nraw && B.k < R; ++k) {
      const int kd = L.ops[k].kind & 0xff;
      if (kd == OP_RX || kd == OP_RY) B.add(L.ops[k].xm);
    }
    for (int b = N - 1; b >= 0 && B.k < R; --b) B.add(1u << b);   // fill with unit directions
    emit_layout<N>(B, &L.lay[nl]);
    ++nl;
  };
  open_layout(0);
  for (int o = 0; o < nraw; ++o) {
    const Op op = L.ops[o];
    const int kd = op.kind & 0xff;
    uint32_t j = 0;
    if (kd == OP_RX || kd == OP_RY) {
      if (B.reduce(op.xm) != 0) {
        open_layout(o);
        L.sched[ns++] = Op{(uint32_t)(nl - 1), 0u, -1, OP_RELAYOUT};
      }
      j = B.coords(op.xm);
    }
    uint32_t svec = 0;
    for (int i = 0; i < R; ++i) svec |= (uint32_t)parity32(op.zm & B.e[i]) << i;
    L.sidx[o] = (uint16_t)ns;
    L.sched[ns++] = Op{sign_word<R>(svec), op.zm, op.pidx >= 0 ? op.pidx * 16 : 0, op.kind | (int32_t)(j << 16)};
  }
  L.sched[ns] = Op{0u, 0u, 0, 0};   // the run loop prefetches one record past the end
  L.meta[4] = ns;
  L.meta[5] = nl;
}

// ---- per-thread helpers -----------------------------------------------------------------------
template <int LT>
__device__ __forceinline__ uint32_t deposit(uint32_t tid, uint32_t pos, uint32_t pos8) {
  uint32_t b = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) b |= ((tid >> j) & 1u) << ((pos >> (4 * j)) & 15u);
  if (LT > 8) b |= ((tid >> 8) & 1u) << pos8;
  return b;
}

template <int R>
__device__ __forceinline__ uint32_t combo(const uint32_t (&e)[5], int r) {
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) if ((r >> i) & 1) v ^= e[i];
  return v;
}

__device__ __forceinline__ double flip_if(double v, uint32_t w, int r) {
  // v with its sign flipped iff bit r of w is set
  const uint32_t hi = (uint32_t)__double2hiint(v) ^ (((w >> r) & 1u) << 31);
  return __hiloint2double((int)hi, __double2loint(v));
}

// Two plane rotations  (x, y) <- (c x - s y, c y + s x)  updated IN PLACE.  Written as one asm
// block with tied operands: left to itself hipcc keeps the old and the new amplitudes in two
// register sets and copies the whole state (2^R v_mov_b64 per gate) where the switch cases
// merge, which also doubles the register pressure of the loop.
__device__ __forceinline__ void rot2x2(double& x0, double& y0, double& x1, double& y1, double c, double s0,
                                       double s1) {
  double t0, t1;
  asm("v_mul_f64 %4, %7, %1\n\t"
      "v_mul_f64 %5, %8, %3\n\t"
      "v_mul_f64 %1, %6, %1\n\t"
      "v_mul_f64 %3, %6, %3\n\t"
      "v_fmac_f64 %1, %7, %0\n\t"
      "v_fmac_f64 %3, %8, %2\n\t"
      "v_fma_f64 %0, %6, %0, -%4\n\t"
      "v_fma_f64 %2, %6, %2, -%5"
      : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "=&v"(t0), "=&v"(t1)
      : "v"(c), "v"(s0), "v"(s1));
}

template <int NA, int J>
__device__ __forceinline__ void rx_pairs(double2 (&amp)[NA], double c, double s) {
  static_assert(J > 0 && J < NA, "partner mask out of range");
  constexpr int HB = 31 - __builtin_clz(J);
#pragma unroll
  for (int r = 0; r < NA; ++r) {
    if ((r >> HB) & 1) continue;
    const int r2 = r ^ J;
    // amp[r] = (c a0.x - s a1.y, c a0.y + s a1.x), amp[r2] = (c a1.x - s a0.y, c a1.y + s a0.x)
    rot2x2(amp[r].x, amp[r2].y, amp[r2].x, amp[r].y, c, s, s);
  }
}

template <int NA, int J>
__device__ __forceinline__ void ry_pairs(double2 (&amp)[NA], double c, double s, uint32_t w) {
  constexpr int HB = 31 - __builtin_clz(J);
#pragma unroll
  for (int r = 0; r < NA; ++r) {
    if ((r >> HB) & 1) continue;
    const int r2 = r ^ J;
    const double s0 = flip_if(s, w, r);
    // amp[r] = c a0 + s0 a1, amp[r2] = c a1 - s0 a0 (componentwise)
    rot2x2(amp[r2].x, amp[r].x, amp[r2].y, amp[r].y, c, s0, s0);
  }
}

// The partner-mask dispatch is a switch written INSIDE run_ops_reg (not a helper taking the
// array by reference): a helper is optimised on its own first, where branch-merging turns the
// per-case constant register indices into pointer PHIs and the amplitudes end up in scratch.
#define VQE_PAIR_CASE(J, CALL) case J: if constexpr (J < NA) { CALL; } break;
#define VQE_PAIR_SWITCH(j, F, ...)                                                                 \
  switch (j) {                                                                                     \
    VQE_PAIR_CASE(1, (F<NA, 1>(__VA_ARGS__))) VQE_PAIR_CASE(2, (F<NA, 2>(__VA_ARGS__)))            \
    VQE_PAIR_CASE(3, (F<NA, 3>(__VA_ARGS__))) VQE_PAIR_CASE(4, (F<NA, 4>(__VA_ARGS__)))            \
    VQE_PAIR_CASE(5, (F<NA, 5>(__VA_ARGS__))) VQE_PAIR_CASE(6, (F<NA, 6>(__VA_ARGS__)))            \
    VQE_PAIR_CASE(7, (F<NA, 7>(__VA_ARGS__))) VQE_PAIR_CASE(8, (F<NA, 8>(__VA_ARGS__)))            \
    VQE_PAIR_CASE(9, (F<NA, 9>(__VA_ARGS__))) VQE_PAIR_CASE(10, (F<NA, 10>(__VA_ARGS__)))          \
    VQE_PAIR_CASE(11, (F<NA, 11>(__VA_ARGS__))) VQE_PAIR_CASE(12, (F<NA, 12>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(13, (F<NA, 13>(__VA_ARGS__))) VQE_PAIR_CASE(14, (F<NA, 14>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(15, (F<NA, 15>(__VA_ARGS__))) VQE_PAIR_CASE(16, (F<NA, 16>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(17, (F<NA, 17>(__VA_ARGS__))) VQE_PAIR_CASE(18, (F<NA, 18>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(19, (F<NA, 19>(__VA_ARGS__))) VQE_PAIR_CASE(20, (F<NA, 20>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(21, (F<NA, 21>(__VA_ARGS__))) VQE_PAIR_CASE(22, (F<NA, 22>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(23, (F<NA, 23>(__VA_ARGS__))) VQE_PAIR_CASE(24, (F<NA, 24>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(25, (F<NA, 25>(__VA_ARGS__))) VQE_PAIR_CASE(26, (F<NA, 26>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(27, (F<NA, 27>(__VA_ARGS__))) VQE_PAIR_CASE(28, (F<NA, 28>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(29, (F<NA, 29>(__VA_ARGS__))) VQE_PAIR_CASE(30, (F<NA, 30>(__VA_ARGS__)))        \
    VQE_PAIR_CASE(31, (F<NA, 31>(__VA_ARGS__)))                                                    \
    default: break;                                                                                \
  }

// Apply the scheduled ops with the amplitudes in registers; leaves the state in L.psi in
// LOGICAL order (same contract as run_ops).
template <int N, bool SLOTS = false>
__device__ __forceinline__ void run_ops_reg(const Lds& L, const double2* __restrict__ init, const double* theta, int P,
                                   int p_hole = -1, unsigned long long* __restrict__ dbg = nullptr, bool cs_ready = false) {
#ifdef VQE_STAMPS
  const long long ts0 = clock64();
  long long trel = 0;
#endif
  constexpr int kThreads = Geo<N>::NT;
  constexpr int LT = Geo<N>::LT;
  constexpr int R = N - LT;
  constexpr int NA = 1 << R;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // opaque (see reg_energy)
  if (!cs_ready) {      // (else: formed by StagedCobyla::out from the trial point while it was in LDS)
    for (int j = tid; j < P; j += kThreads) {
      if (j == p_hole) continue;
      double s, c;
      sincos(0.5 * theta[j - (p_hole >= 0 && j > p_hole)], &s, &c);
      L.cs[j] = make_double2(c, s);
    }
  }
  double2 amp[NA];
  // layout vectors and the thread's base index are kept as BYTE offsets (<< 4): one XOR per
  // LDS address instead of XOR + shift
  uint32_t e[5];
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  lds_byte* psi_l = (lds_byte*)L.psi;
  auto lds_get = [&](uint32_t off) {
    const d2v_t v = *(const __attribute__((address_space(3))) d2v_t*)(psi_l + off);
    return make_double2(v.x, v.y);
  };
  auto lds_put = [&](uint32_t off, const double2& a) {
    d2v_t v; v.x = a.x; v.y = a.y;
    *(__attribute__((address_space(3))) d2v_t*)(psi_l + off) = v;
  };
  uint32_t base = deposit<LT>(tid, L.lay[0].pos, L.lay[0].pos8);
  {
    const LayoutRec lr = L.lay[0];
#pragma unroll
    for (int i = 0; i < 5; ++i) e[i] = lr.e[i] << 4;
    const unsigned char* init_b = (const unsigned char*)init;
#pragma unroll
    for (int r = 0; r < NA; ++r) amp[r] = *(const double2*)(init_b + ((base << 4) ^ combo<R>(e, r)));
  }
  // all initial amplitudes in: otherwise every case of the gate switch carries its own waits
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();   // cs[] visible
  const int nops = L.meta[4];
#ifdef VQE_STAMPS
  const long long ts1 = clock64();
#endif
  // descriptor and (cos, sin) of op o+1 are fetched while op o runs (the sched array has a
  // spare slot; an index outside [0, P) reads entry 0: always valid LDS addresses)
  const unsigned char* cs_b = (const unsigned char*)L.cs;
  Op nxt = L.sched[0];
  double2 ncs = *(const double2*)(cs_b + nxt.pidx);
  for (int o = 0; o < nops; ++o) {
    const Op op = nxt;
    const double2 cs = ncs;
    // stochastic runs (SLOTS): jump over the inactive noise slots behind this record - an iteration costs ~250 cycles
    // even when it does nothing.  Compiled into the noisy instantiation only: the address of the next record then waits
    // for this one, which costs the noiseless kernel 1 % (2 % behind a run-time branch).
    if constexpr (SLOTS) o += (int)((uint32_t)__builtin_amdgcn_readfirstlane(op.kind) >> 24);
    nxt = L.sched[o + 1];
    ncs = *(const double2*)(cs_b + nxt.pidx);
    const int kind = op.kind & 0xff;
    if (kind == OP_RELAYOUT) {
#ifdef VQE_STAMPS
      const long long tr0 = clock64();
#endif
      __syncthreads();                                   // earlier reads of psi are done
      {
        const uint32_t b16 = base << 4;
#pragma unroll
        for (int r = 0; r < NA; ++r) lds_put(b16 ^ combo<R>(e, r), amp[r]);
      }
      const LayoutRec lr = L.lay[op.xm];
#pragma unroll
      for (int i = 0; i < 5; ++i) e[i] = lr.e[i] << 4;
      base = deposit<LT>(tid, lr.pos, lr.pos8);
      __syncthreads();
      {
        const uint32_t b16 = base << 4;
#pragma unroll
        for (int r = 0; r < NA; ++r) amp[r] = lds_get(b16 ^ combo<R>(e, r));
      }
#ifdef VQE_STAMPS
      trel += clock64() - tr0;
#endif
      continue;
    }
    if (kind == OP_NOP) continue;   // inactive noise slot
    const int inv = (op.kind >> 8) & 1;
    const uint32_t flip = (uint32_t)(parity32(op.zm & base) ^ inv);
    const uint32_t w = op.xm ^ (0u - flip);
    const int jm = (op.kind >> 16) & 0xff;
    if (kind == OP_RX) {
      VQE_PAIR_SWITCH(jm, rx_pairs, amp, cs.x, cs.y)
    } else if (kind == OP_RY) {
      VQE_PAIR_SWITCH(jm, ry_pairs, amp, cs.x, cs.y, w)
    } else if (kind == OP_RZ) {
#pragma unroll
      for (int r = 0; r < NA; r += 2)
        rot2x2(amp[r].x, amp[r].y, amp[r + 1].x, amp[r + 1].y, cs.x, flip_if(cs.y, w, r), flip_if(cs.y, w, r + 1));
    } else if (kind == OP_PZ) {   // (OP_NOP: an inactive noise slot)
#pragma unroll
      for (int r = 0; r < NA; ++r) amp[r] = make_double2(flip_if(amp[r].x, w, r), flip_if(amp[r].y, w, r));
    }
  }
#ifdef VQE_STAMPS
  const long long ts2 = clock64();
#endif
  // scatter to logical order: i = A * p ^ c, bit q of A*p = parity(zm[q] & p)
  uint32_t ib = (uint32_t)L.meta[1] << 4;            // byte offsets again
  uint32_t ae[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const uint32_t z = L.zm[q];
    ib ^= (uint32_t)parity32(z & base) << (q + 4);
#pragma unroll
    for (int i = 0; i < R; ++i) ae[i] |= (uint32_t)parity32(z & (e[i] >> 4)) << (q + 4);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NA; ++r) lds_put(ib ^ combo<R>(ae, r), amp[r]);
  __syncthreads();
#ifdef VQE_STAMPS
  if (threadIdx.x == 0 && dbg) {
    atomicAdd(dbg + 5, (unsigned long long)(ts1 - ts0));          // sincos + initial load
    atomicAdd(dbg + 6, (unsigned long long)trel);                 // re-layouts
    atomicAdd(dbg + 7, (unsigned long long)(clock64() - ts2));    // final scatter
  }
#endif
}

}  // namespace vqe
