// cobyla_m0.h - Powell's COBYLA for the UNCONSTRAINED case (m = 0), written as an
// ask/tell state machine in SPMD style so that the same source runs
//   * on the host (one "thread"), and
//   * inside a HIP workgroup (all threads of the block execute it together; loops are
//     strided over the block, scalars are computed redundantly and stay bit-identical in
//     every thread, shared vectors live in per-problem scratch memory).
//
// It replaces the reference's inner VQE loop
//   scipy.optimize.minimize(cost, x0, method='COBYLA', options={'maxiter': 1000})
// (reference environments/environment_qulacs_TN_notin_agent.py:478; scipy 1.15 defaults
// rhobeg=1.0, rhoend=tol=1e-4, maxfun=maxiter), i.e. M.J.D. Powell's algorithm as
// published ("A direct search optimization method that models the objective and constraint
// functions by linear interpolation", 1994) and shipped by scipy <= 1.15 as cobyla2.f.  The
// sequence of simplex operations, acceptance tests and the order of floating-point
// accumulations follow that publication so that iterates agree with scipy's to rounding
// (tests/test_abi.py::test_host_cobyla_reproduces_scipy_traces pins this against recorded scipy 1.15.3 traces).
//
// With m = 0 the trust-region LP has the closed form dx = rho * a / |a| (a = minus the
// model gradient); trstlp_m0() evaluates it with the same Givens accumulation the general
// routine would perform on an identity Z matrix.
//
// Known quirk kept on purpose (it is observable through result.x in the reference): on
// normal termination the returned point is the LAST trust-region trial point, not the best
// vertex; on maxfun termination it is the best vertex.
#pragma once
#include <math.h>
#include <type_traits>

#if defined(__HIPCC__)
#define CBY_HD __host__ __device__ __forceinline__
#define CBY_L __attribute__((always_inline))     // lambdas handed to walk_tiled: their operands are register arrays
#else
#define CBY_HD
#define CBY_L
#endif

// The optimiser's arithmetic is kept un-contracted (no FMA fusion): that is what makes the
// host build reproduce scipy's iterates bit for bit.  The pragma is scoped to this header.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#ifndef CBY_STAMP
#define CBY_STAMP(k)
#define CBY_STAMP_RESET()
#endif
#ifndef CBY_LOG   // decision log, compiled in only by tests/cpp/cobyla_wave_emulation.cpp
#define CBY_LOG(...)
#endif

namespace cby {

#if defined(__HIPCC__)
#define CBY_UNROLL _Pragma("unroll 8")
#define CBY_FULL_UNROLL _Pragma("unroll")
#else
#define CBY_UNROLL
#define CBY_FULL_UNROLL
#endif

// Execution context for the host: one thread, no synchronisation, sums in index order (the
// order of the published algorithm, so iterates match scipy's Fortran build bit for bit).
// A device context provides the same members with block-wide reductions (vqe_device.h).
struct HostCtx {
  static constexpr int tid = 0;
  static constexpr int nth = 1;
  static constexpr int kPad = 1;   // inner loops run to exactly n
  static constexpr bool kSplit = false;   // no second lane to share a row with
  static constexpr bool kColumns = false; // (workgroup contexts: element-wise matrix passes run with the lanes along a row)
  static constexpr bool kTile = false;    // (one-wave context on global arrays: row walks through an LDS transposition tile, walk_tiled)
  static constexpr bool kTileWalks = false;
  CBY_HD double pair_sum(double v) const { return v; }
  CBY_HD void lockstep() const {}   // see the one call site

  CBY_HD void sync() const {}
  CBY_HD int all_or(int v) const { return v; }
  // sum_{i<n} f(i), identical in every thread
  template <class F>
  CBY_HD double sum(int n, F f) const {
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += f(i);
    return a;
  }
  // first index whose f(i) is the strict extreme beyond `thresh` (max if want_max else min);
  // -1 when no f(i) beats thresh.  *val receives the extreme (or thresh).
  template <class F>
  CBY_HD int arg_first(int n, F f, double thresh, bool want_max, double* val) const {
    int idx = -1;
    double best = thresh;
    for (int i = 0; i < n; ++i) {
      const double v = f(i);
      if (want_max ? (v > best) : (v < best)) { best = v; idx = i; }
    }
    *val = best;
    return idx;
  }
};

enum Status { RUNNING = 0, DONE_RHOEND = 1, DONE_MAXFUN = 2, DONE_ROUNDING = 3 };

// Scalars of the optimiser, parked behind the arrays between two calls (save_state /
// load_state) so that a device kernel holds none of them in registers while it evaluates f.
constexpr int kStateDoubles = 15;

// Leading dimension of sim / simi: odd, so that walking a matrix along either index touches
// distinct LDS banks (a stride of n doubles with n = 32 puts a whole column in one bank).
CBY_HD int lead_dim(int nv) { return nv | 1; }
// ... and for device arrays in GLOBAL memory a multiple of 8 doubles: every row then starts on a 64-byte boundary and
// a batch of 8 consecutive elements of a row is ONE 64-byte sector (with the odd stride of the LDS layout every batch
// straddled two: twice the L2 / HBM traffic of the row loops, which is what bounds the trainable regime).
CBY_HD int lead_dim_global(int nv) { return (nv + 7) & ~7; }      // (nv is a multiple of the context's kPad: 16 -> rows are whole 128-byte lines)

// Parallel contexts run the inner (serial) loops of a row to nv = n rounded up to kPad, in
// batches of kPad with every load of a batch in flight together; the padding entries are
// zero and stay zero, dummy vertices n..nv-1 sit between the real ones and the pole (index nv).
CBY_HD int padded(int n, int pad) { return (n + pad - 1) / pad * pad; }

CBY_HD size_t scratch_doubles_ld(int n, int pad, int ld_) {
  // x, sim, simi, datmat, a, vsig, veta, sigbar, dx, w, tdot, state
  const size_t nv = (size_t)padded(n, pad), ld = (size_t)ld_;
  return nv + (nv + 1) * ld + nv * ld + (nv + 1) + 7 * nv + 2 + kStateDoubles;
}
CBY_HD size_t scratch_doubles(int n, int pad = 8) {      // (odd leading dimension: an upper bound for the global layout too)
  return scratch_doubles_ld(n, pad, lead_dim(padded(n, pad)));
}

// Real is `double`, or an address-space qualified double (LDS) on the device so that the
// arrays are reached with ds_ instructions instead of flat ones.
template <class Ctx, bool CHECK_INVERSE = false, class Real = double>
struct CobylaM0 {
  typedef Ctx Context;
  Ctx ctx;
  static constexpr int P = Ctx::kPad;   // batch of the inner loops (all loads of a batch, then the arithmetic)
  int n, nv, ld, maxfun;   // nv: inner-loop bound (n padded to Ctx::kPad) and index of the pole
  // Row-parallel loops: this lane works on rows rlane, rlane + rstep, ... and on the part
  // [ilo, ihi) of each row's inner loop.  Contexts with kSplit put two lanes on a row when all
  // rows fit in half of the lanes (partial sums are then combined with ctx.pair_sum).
  int rlane, rstep, ilo, ihi;
  bool split;
  double rhoend;
  // shared (per problem) arrays
  Real *x, *sim, *simi, *datmat, *a, *vsig, *veta, *sigbar, *dx, *w, *tdot, *st;
  // scalars, identical in every thread
  double rho, prerem, parsig, pareta, fbest_ret;
  int nfvals, jdrop, ibrnch, iflag, ifull, status;
  // veta[] (vertex lengths) depends on sim alone and a step replaces ONE vertex: vcol = -2: all of it is stale, -1: all
  // valid, j >= 0: only entry j is stale (same values as a full recomputation, one sweep over sim saved per iteration)
  int vcol;
  int vrow;   // the same for vsig[] (row norms of simi): update_simi leaves them all valid, a pole move spoils one row
  // (only where the arrays live in global memory - plain `double` - : with LDS-resident arrays the sweep is cheap and
  // the extra branch costs the fused 12-qubit kernel 0.8 %)
  static constexpr bool kIncrementalEta = std::is_same<Real, double>::value;
  // device contexts on arrays in global memory: rows on 64-byte boundaries (lead_dim_global)
  static constexpr bool kGlobalRows = std::is_same<Real, double>::value && (Ctx::nth == 64);

  CBY_HD Real &SIM(int i, int j) { return sim[(size_t)j * ld + i]; }    // coordinate i of vertex j
  CBY_HD Real &SIMI(int j, int i) { return simi[(size_t)j * ld + i]; }  // row j of the inverse

  CBY_HD void bind(Real *mem, int n_) {
    n = n_;
    nv = padded(n, Ctx::kPad);
    ld = kGlobalRows ? lead_dim_global(nv) : lead_dim(nv);
    x = mem; mem += nv;
    sim = mem; mem += (size_t)(nv + 1) * ld;
    simi = mem; mem += (size_t)nv * ld;
    datmat = mem; mem += nv + 1;
    a = mem; mem += nv;
    vsig = mem; mem += nv;
    veta = mem; mem += nv;
    sigbar = mem; mem += nv;
    dx = mem; mem += nv;
    w = mem; mem += nv + 2;
    tdot = mem; mem += nv;     // signed simi_j . dx of the trust-region branch, reused by update_simi
    st = mem;
    split = Ctx::kSplit && 2 * nv <= ctx.nth;
    if (split) {
      const int half = ctx.nth / 2, part = ctx.tid / half;
      const int mid = ((nv / P + 1) / 2) * P;
      rlane = ctx.tid % half; rstep = half;
      ilo = part ? mid : 0; ihi = part ? nv : mid;
    } else {
      rlane = ctx.tid; rstep = ctx.nth; ilo = 0; ihi = nv;
    }
  }
  CBY_HD size_t words() const { return (size_t)(st - x) + kStateDoubles; }

  // ---- row walks through an LDS tile (contexts with kTile: ONE wavefront, arrays in global memory) -------------------
  // The O(n^2) passes below whose lane owns a ROW j and walks its entries i (row norms, simi_j . dx, edge lengths, the
  // rank-one update) read m[j * ld + i] with a stride of a whole row between neighbouring lanes: every load
  // instruction touches 64 cache lines, and the CU's vector L1 looks lines up one per cycle - measured on the
  // trainable H2O-8q workload (129 variables): 22 L1 accesses per load instruction, L1 busy 97 % of the kernel, 55 % of
  // it waiting on misses, the wavefronts idle 70 % of theirs.  walk_tiled() moves a block of 64 rows x 16
  // entries per step with the lanes ALONG the rows (16 loads of four 128-byte lines each, the next block already
  // in flight), turns it round in LDS (row stride 17 doubles: conflict-free both ways) and hands lane l the 16 entries
  // of row jb + l in index order - the arithmetic of a row, its order and therefore its bits are those of the plain
  // loop; tests/cpp/cobyla_wave_emulation.cpp (plain loops) reproduces the device's trial points bit for bit.
  //   m      : row-major array with leading dimension ld (simi, or sim whose "rows" are the vertices)
  //   shared : vector every row meets (dx, or the rescaled row jdrop), staged in LDS behind the tile; or nullptr
  //   init(j, acc) -> bool : per-row set-up (accumulators start at 0); false = the row is not walked
  //   fn(j, i, v, u, acc)  : entry i of row j, v = m[j * ld + i], u = shared[i] (WRITE: may change v, the block is stored back)
  //   post(j, walked, acc) : finish row j
  // Needs nv % 16 == 0 (Ctx::kPad == 16) and rows 0..nv-1 present in m.
  static constexpr int kAcc = 3;
  static constexpr int kTileStride = 17;
  static constexpr int kTileDoubles = 64 * kTileStride;      // followed by the shared vector (nv doubles)
#ifndef CBY_TILE_FEW
#define CBY_TILE_FEW 6
#endif
  static constexpr int kTileFewRows = CBY_TILE_FEW;
  // ... and the passes whose lane owns a COLUMN i and walks down the rows (the linear model, the pole move) are
  // coalesced as they stand but latency bound: a batch of loads, a round trip, its arithmetic, the next batch - and a
  // lane on 129 variables owns columns l, l + 64, l + 128, three walks in a row, the last one for lane 0 alone.
  // walk_cols() takes a lane's (up to kColsSide) columns side by side: a third of the round trips, three times the
  // loads in flight in each.  Per column nothing changes.
  //   load(k, i, v, u)    : the two operands of row k, column i
  //   fn(k, i, v, u, acc) : arithmetic (and stores) of row k
  //   post(i, acc)        : finish column i
  static constexpr int kColsSide = Ctx::nth == 64 ? 3 : 1;      // (a workgroup context has a thread per column)
  template <int PB, class Init, class Load, class Fn, class Post>      // PB: rows per batch (divides nv)
  CBY_HD void walk_cols(Init init, Load load, Fn fn, Post post) {
    for (int ib = ctx.tid; ib < n; ib += Ctx::nth * kColsSide) {
      int ic[kColsSide];
      bool on[kColsSide];
      double acc[kColsSide][kAcc];
      CBY_FULL_UNROLL
      for (int c = 0; c < kColsSide; ++c) {
        ic[c] = ib + Ctx::nth * c;
        on[c] = ic[c] < n;
        acc[c][0] = 0.0; acc[c][1] = 0.0; acc[c][2] = 0.0;
        if (on[c]) init(ic[c], acc[c]);
      }
      for (int k0 = 0; k0 < nv; k0 += PB) {
        double v[kColsSide][PB], u[kColsSide][PB];
        CBY_FULL_UNROLL
        for (int c = 0; c < kColsSide; ++c)
          if (on[c]) {
            CBY_FULL_UNROLL
            for (int q = 0; q < PB; ++q) load(k0 + q, ic[c], v[c][q], u[c][q]);
          }
        CBY_FULL_UNROLL
        for (int c = 0; c < kColsSide; ++c)
          if (on[c]) {
            CBY_FULL_UNROLL
            for (int q = 0; q < PB; ++q) fn(k0 + q, ic[c], v[c][q], u[c][q], acc[c]);
          }
      }
      CBY_FULL_UNROLL
      for (int c = 0; c < kColsSide; ++c)
        if (on[c]) post(ic[c], acc[c]);
    }
  }

  template <bool WRITE, bool SHARED, class Init, class Fn, class Post>
  CBY_HD void walk_tiled(Real* m, int m_rows, const Real* shared, Init init, Fn fn, Post post) {
    const int lane = ctx.tid & 63;      // (a workgroup context: every wavefront walks 64-row blocks of its own, through its own tile)
    auto* Tw = ctx.tile + (lane >> 4) * kTileStride + (lane & 15);     // lanes along the rows: row (lane >> 4) + 4 k, entry lane & 15
    auto* Tr = ctx.tile + lane * kTileStride;                          // a lane per row
    auto* S = ctx.shared;
    ctx.tile_bind(m, m_rows, ld);
    if (SHARED) {
      ctx.tile_sync_all();      // (the previous walk's readers of S are done)
      for (int i = ctx.tid; i < nv; i += Ctx::nth) S[i] = shared[i];
      ctx.tile_sync_all();
    }
    for (int jb = (ctx.tid >> 6) * 64; jb < n; jb += Ctx::nth) {
      const int j = jb + lane;
      double acc[kAcc] = {0.0, 0.0, 0.0};
      const bool on = j < n && init(j, acc);
      // one block: g (fetched a block ago) -> LDS -> the lane's row, entries in index order -> (WRITE) back
      auto block = [&](int i0, double (&g)[16]) CBY_L {
        CBY_FULL_UNROLL
        for (int k = 0; k < 16; ++k) Tw[4 * k * kTileStride] = g[k];
        ctx.tile_sync();
        if (on) {
          CBY_FULL_UNROLL
          for (int h = 0; h < 16; h += 8) {
            double v[8], u[8];
            CBY_FULL_UNROLL
            for (int q = 0; q < 8; ++q) { v[q] = Tr[h + q]; u[q] = SHARED ? (double)S[i0 + h + q] : 0.0; }
            CBY_FULL_UNROLL
            for (int q = 0; q < 8; ++q) fn(j, i0 + h + q, v[q], u[q], acc);
            if (WRITE) {
              CBY_FULL_UNROLL
              for (int q = 0; q < 8; ++q) Tr[h + q] = v[q];
            }
          }
        }
        if (WRITE) {
          ctx.tile_sync();
          double o[16];
          CBY_FULL_UNROLL
          for (int k = 0; k < 16; ++k) o[k] = Tw[4 * k * kTileStride];
          ctx.tile_store(jb, i0, o);
        }
        ctx.tile_sync();
      };
      // few rows of the block want the walk (the 129th row of 129, one changed row in the acceptability test, the far
      // vertices of the edge test): the wavefront fetches just those rows, lanes along each, into row buffers in the
      // tile area - one round trip for all of them - and their owners walk them from LDS side by side
      const unsigned long long mask = ctx.ballot(on);
      const int cnt = ctx.popc(mask);
      if (cnt == 0) {
        if (j < n) post(j, on, acc);
        continue;
      }
      if (cnt <= kTileFewRows && nv <= 192 && cnt * nv <= kTileDoubles) {
        auto* R = ctx.tile;
        int rowof[kTileFewRows];
        {
          unsigned long long mm = mask;
          CBY_FULL_UNROLL
          for (int t = 0; t < kTileFewRows; ++t) { rowof[t] = mm ? ctx.ctz(mm) : 0; mm &= mm - 1; }
        }
        double r[kTileFewRows][3];
        CBY_FULL_UNROLL
        for (int t = 0; t < kTileFewRows; ++t)
          if (t < cnt) {
            CBY_FULL_UNROLL
            for (int c = 0; c < 3; ++c) r[t][c] = lane + 64 * c < nv ? (double)m[(size_t)(jb + rowof[t]) * ld + lane + 64 * c] : 0.0;
          }
        CBY_FULL_UNROLL
        for (int t = 0; t < kTileFewRows; ++t)
          if (t < cnt) {
            CBY_FULL_UNROLL
            for (int c = 0; c < 3; ++c)
              if (lane + 64 * c < nv) R[t * nv + lane + 64 * c] = r[t][c];
          }
        ctx.tile_sync();
        if (on) {
          auto* row = R + ctx.popc(mask & ((1ull << lane) - 1ull)) * nv;
          for (int i0 = 0; i0 < nv; i0 += 8) {
            double v[8], u[8];
            CBY_FULL_UNROLL
            for (int q = 0; q < 8; ++q) { v[q] = row[i0 + q]; u[q] = SHARED ? (double)S[i0 + q] : 0.0; }
            CBY_FULL_UNROLL
            for (int q = 0; q < 8; ++q) fn(j, i0 + q, v[q], u[q], acc);
            if (WRITE) {
              CBY_FULL_UNROLL
              for (int q = 0; q < 8; ++q) row[i0 + q] = v[q];
            }
          }
        }
        if (WRITE) {
          ctx.tile_sync();
          CBY_FULL_UNROLL
          for (int t = 0; t < kTileFewRows; ++t)
            if (t < cnt) {
              CBY_FULL_UNROLL
              for (int c = 0; c < 3; ++c)
                if (lane + 64 * c < nv) m[(size_t)(jb + rowof[t]) * ld + lane + 64 * c] = R[t * nv + lane + 64 * c];
            }
        }
        ctx.tile_sync();
        if (j < n) post(j, on, acc);
        continue;
      }
      // two register buffers in turn (no copies: a copy would wait for the block just requested)
      double ga[16], gb[16];
      // (no branch around a fetch: the wait-count pass would then assume the older buffer's loads are the newest
      // and wait for everything; a fetch past the end of the rows brings values nobody uses)
      ctx.tile_fetch(jb, 0, ga);
      int i0 = 0;
      for (; i0 + 32 <= nv; i0 += 32) {
        ctx.tile_fetch(jb, i0 + 16, gb);
        block(i0, ga);
        ctx.tile_fetch(jb, i0 + 32, ga);
        block(i0 + 16, gb);
      }
      if (i0 < nv) block(i0, ga);
      if (j < n) post(j, on, acc);
    }
  }

  CBY_HD void save_state() {
    if (ctx.tid == 0) {
      st[0] = rho; st[1] = prerem; st[2] = parsig; st[3] = pareta; st[4] = fbest_ret; st[5] = rhoend;
      st[6] = (double)nfvals; st[7] = (double)jdrop; st[8] = (double)ibrnch; st[9] = (double)iflag;
      st[10] = (double)ifull; st[11] = (double)status; st[12] = (double)maxfun; st[13] = (double)vcol; st[14] = (double)vrow;
    }
    ctx.sync();
  }
  CBY_HD void load_state() {
    rho = st[0]; prerem = st[1]; parsig = st[2]; pareta = st[3]; fbest_ret = st[4]; rhoend = st[5];
    nfvals = (int)st[6]; jdrop = (int)st[7]; ibrnch = (int)st[8]; iflag = (int)st[9];
    ifull = (int)st[10]; status = (int)st[11]; maxfun = (int)st[12]; vcol = (int)st[13]; vrow = (int)st[14];
  }

  // Begin a minimisation; x[] must already hold x0.  Returns 1 when f(x) is wanted.
  CBY_HD int start(double rhobeg, double rhoend_, int maxfun_) {
    rho = rhobeg; rhoend = rhoend_; maxfun = maxfun_;
    nfvals = 0; ibrnch = 0; iflag = 0; ifull = 1; status = RUNNING; prerem = 0.0; vcol = -2; vrow = -2;
    const double temp = 1.0 / rho;
    if (Ctx::kPad > 1) {   // padding entries must read as zero from now on
      const int total = (int)(st - sim);
      for (int k = ctx.tid; k < total; k += ctx.nth) sim[k] = 0.0;
      ctx.sync();
    }
    for (int i = ctx.tid; i < n; i += ctx.nth) {
      SIM(i, nv) = x[i];
      for (int j = 0; j < n; ++j) { SIM(i, j) = 0.0; SIMI(i, j) = 0.0; }
      SIM(i, i) = rho;
      SIMI(i, i) = temp;
    }
    jdrop = nv;   // "the pole"
    ctx.sync();
    return request_eval();
  }

  CBY_HD void vertex_changed(int j) { vcol = (vcol == -1 || vcol == j) ? j : -2; }

  // Label 40: decide whether another evaluation may be made.
  CBY_HD int request_eval() {
    if (nfvals >= maxfun && nfvals > 0) { status = DONE_MAXFUN; return finish(false); }
    ++nfvals;
    return 1;
  }

  CBY_HD int finish(bool keep_x) {
    if (!keep_x) {
      for (int i = ctx.tid; i < n; i += ctx.nth) x[i] = SIM(i, nv);
      fbest_ret = datmat[nv];
    }
    ctx.sync();
    return 0;
  }

  // m = 0 trust-region step: dx = rho * a/|a| accumulated as the general routine does.
  // Uses w[0..n) for the running direction and w[n..2n) is not needed.  Sets ifull.
  CBY_HD void trstlp_m0() {
    if (Ctx::nth > 1) {
      // Parallel contexts: the closed form dx = rho * a / |a| directly (one reduction, no
      // serial chain of n square roots and 2n divisions).  Same step as the Givens form
      // below up to rounding.
      const double nrm2 = ctx.sum(n, [&](int i) { return a[i] * a[i]; });
      if (nrm2 == 0.0) {
        for (int i = ctx.tid; i < n; i += ctx.nth) dx[i] = 0.0;
        ifull = 0;
        ctx.sync();
        return;
      }
      const double scale = rho / sqrt(nrm2);
      for (int i = ctx.tid; i < n; i += ctx.nth) dx[i] = scale * a[i];
      ifull = 1;
      ctx.sync();
      return;
    }
    // Givens accumulation from k = n-1 down to 0 on Z = I (all threads, redundantly, for
    // the scalar chain; the direction vector is built in parallel afterwards).
    double tot = 0.0;
    // alpha_k, beta_k are stored in vsig-independent scratch: reuse sigbar (alpha) and w (beta)
    for (int k = n - 1; k >= 0; --k) {
      double sp = a[k];
      const double spabs = fabs(sp);
      const double acca = spabs + 0.1 * fabs(sp);
      const double accb = spabs + 0.2 * fabs(sp);
      if (spabs >= acca || acca >= accb) sp = 0.0;
      double al, be;
      if (tot == 0.0) { tot = sp; al = 1.0; be = 0.0; }  // column k keeps e_k
      else {
        const double temp = sqrt(sp * sp + tot * tot);
        al = sp / temp; be = tot / temp; tot = temp;
      }
      if (ctx.tid == 0) { sigbar[k] = al; w[k] = be; }
    }
    ctx.sync();
    if (tot == 0.0) {  // no descent direction: short (zero) step
      for (int i = ctx.tid; i < n; i += ctx.nth) dx[i] = 0.0;
      ifull = 0;
      ctx.sync();
      return;
    }
    // z(:,0) after all rotations: z_i = alpha_i * beta_{i-1} * ... * beta_0 (rotation k
    // multiplies the tail by beta_k, applied for k = i-1 down to 0).  Entries before the
    // first non-zero sp (counting from the end) stay those of e_k, handled by al=1, be=0.
    const double zdota = tot;
    const double tinv = 1.0 / zdota;
    for (int i = ctx.tid; i < n; i += ctx.nth) {
      double z = sigbar[i];
      CBY_UNROLL
      for (int k = i - 1; k >= 0; --k) z = w[k] * z;
      dx[i] = tinv * z;  // sdirn
    }
    ctx.sync();
    const double ss = ctx.sum(n, [&](int i) { return dx[i] * dx[i]; });
    const double dd = rho * rho;
    const double temp = sqrt(ss * dd);
    const double step = dd / (temp + 0.0);
    ctx.sync();
    for (int i = ctx.tid; i < n; i += ctx.nth) dx[i] = 0.0 + step * dx[i];
    ifull = 1;
    ctx.sync();
  }

  // Report f(x) of the point handed out by the previous start()/tell().  Returns 1 when
  // another evaluation (of the new x[]) is wanted, 0 when finished.
  CBY_HD int tell(double f) {
    CBY_STAMP_RESET();
    int lbl;
    if (ibrnch == 1) {
      lbl = 440;
    } else {
      if (nfvals <= n + 1) {
        const double fpole = (jdrop < n) ? datmat[nv] : 0.0;
        ctx.sync();
        if (jdrop < n) {
          if (fpole <= f) {
            if (ctx.tid == 0) { x[jdrop] = SIM(jdrop, nv); datmat[jdrop] = f; }
          } else {
            if (ctx.tid == 0) {
              SIM(jdrop, nv) = x[jdrop];
              datmat[jdrop] = fpole;
              datmat[nv] = f;
              for (int k = 0; k <= jdrop; ++k) {
                SIM(jdrop, k) = -rho;
                double temp = 0.0;
                for (int i = k; i <= jdrop; ++i) temp -= SIMI(i, k);
                SIMI(jdrop, k) = temp;
              }
            }
          }
        } else if (ctx.tid == 0) {
          datmat[jdrop] = f;
        }
        ctx.sync();
        if (nfvals <= n) {
          jdrop = nfvals - 1;
          if (ctx.tid == 0) x[jdrop] += rho;
          ctx.sync();
          return request_eval();
        }
      } else {
        if (ctx.tid == 0) datmat[jdrop] = f;
        ctx.sync();
      }
      ibrnch = 1;
      lbl = 140;
    }

    double trured = 0.0;
    for (;;) {
      if (lbl == 140) {
        // ---- identify the optimal vertex and move it to the pole position
        // (the tie rule compares the all-zero constraint violations: never switches)
        double phimin;
        int nbest = ctx.arg_first(n, [&](int j) { return datmat[j]; }, datmat[nv], false, &phimin);
        if (nbest < 0) nbest = n;
        CBY_LOG("140: nfvals %d nbest %d phimin %.17g rho %g ibrnch %d", nfvals, nbest, phimin, rho, ibrnch);
        ctx.sync();
        if (nbest < n) {
          vcol = -2;   // every vertex moves relative to the new pole
          vrow = (vrow == -1 || vrow == nbest) ? nbest : -2;   // row nbest of simi is replaced below
          if (ctx.tid == 0) { const double t = datmat[nv]; datmat[nv] = datmat[nbest]; datmat[nbest] = t; }
          bool walked_cols = false;
          if constexpr (Ctx::kTile) {
            if (!split) {
              walk_cols<8>(
                  [&](int i, double* acc) CBY_L {
                    const double temp = SIM(i, nbest);
                    SIM(i, nv) += temp;
                    acc[1] = temp;
                  },
                  [&](int k, int i, double& v, double& u) CBY_L { v = SIM(i, k); u = SIMI(k, i); },
                  [&](int k, int i, double v, double u, double* acc) CBY_L { SIM(i, k) = v - acc[1]; acc[0] -= u; },
                  [&](int i, double* acc) CBY_L {
                    SIM(i, nbest) = -acc[1];   // (old vertex nbest = new pole: 0 - temp)
                    w[i] = acc[0];
                  });
              walked_cols = true;
            }
          }
          if (!walked_cols)
          for (int i = rlane; i < n; i += rstep) {
            const double temp = SIM(i, nbest);
            // two lanes may share row i (kSplit): both must have read SIM(i, nbest) before its owner
            // overwrites it below.  One wavefront executes in lock step, so this is a no-op on the
            // GPU; an emulation with free-running threads synchronises the pair here.
            if (split) ctx.lockstep();
            if (ilo == 0) SIM(i, nv) += temp;
            double tempa = 0.0;
            for (int k0 = ilo; k0 < ihi; k0 += P) {   // (dummy vertices n..nv-1: harmless)
              double v[P], u[P];
              CBY_FULL_UNROLL
              for (int q = 0; q < P; ++q) { v[q] = SIM(i, k0 + q); u[q] = SIMI(k0 + q, i); }
              CBY_FULL_UNROLL
              for (int q = 0; q < P; ++q) { SIM(i, k0 + q) = v[q] - temp; tempa -= u[q]; }
            }
            if (nbest >= ilo && nbest < ihi) SIM(i, nbest) = -temp;   // (old vertex nbest = new pole: 0 - temp)
            if (split) tempa = ctx.pair_sum(tempa);
            w[i] = tempa;  // becomes SIMI(nbest, i); deferred so column sums read old values
          }
          ctx.sync();
          for (int i = ctx.tid; i < n; i += ctx.nth) SIMI(nbest, i) = w[i];
          ctx.sync();
        }
        if (CHECK_INVERSE) {
          int bad = 0;
          for (int i = ctx.tid; i < n; i += ctx.nth)
            for (int j = 0; j < n; ++j) {
              double temp = (i == j) ? -1.0 : 0.0;
              for (int k = 0; k < n; ++k) temp += SIMI(i, k) * SIM(k, j);
              if (fabs(temp) > 0.1) bad = 1;
            }
          if (ctx.all_or(bad)) { status = DONE_ROUNDING; return finish(false); }
        }
        CBY_STAMP(3);
        // ---- linear model: a = -grad
        const double fp = datmat[nv];
        bool walked_cols = false;
        if constexpr (Ctx::kTile && Ctx::kTileWalks) {
          if (!split) {
            auto* S = ctx.shared;      // the function values where every lane finds them
            ctx.tile_sync_all();
            for (int j = ctx.tid; j < nv; j += Ctx::nth) S[j] = datmat[j];
            ctx.tile_sync_all();
            walk_cols<8>([&](int, double*) CBY_L {},
                      [&](int j, int i, double& v, double& u) CBY_L { v = S[j]; u = SIMI(j, i); },   // (rows n..nv-1 of simi are zero)
                      [&](int, int, double v, double u, double* acc) CBY_L { acc[0] += (v - fp) * u; },
                      [&](int i, double* acc) CBY_L { a[i] = -acc[0]; });
            walked_cols = true;
          }
        }
        if (!walked_cols)
        for (int i = rlane; i < n; i += rstep) {
          double temp = 0.0;
          for (int j0 = ilo; j0 < ihi; j0 += P) {   // (rows n..nv-1 of simi are zero)
            double v[P], u[P];
            CBY_FULL_UNROLL
            for (int q = 0; q < P; ++q) { v[q] = datmat[j0 + q]; u[q] = SIMI(j0 + q, i); }
            CBY_FULL_UNROLL
            for (int q = 0; q < P; ++q) temp += (v[q] - fp) * u[q];
          }
          if (split) temp = ctx.pair_sum(temp);
          a[i] = -temp;
        }
        CBY_STAMP(4);
        // ---- simplex acceptability
        parsig = 0.25 * rho;
        pareta = 2.1 * rho;
        int flag_bad = 0;
        bool walked_tiles = false;
        if constexpr (Ctx::kTile && Ctx::kTileWalks) {
          if (!split) {
            // (two walks: the row norms of simi, then the vertex norms of sim for the rows that want them)
            walk_tiled<false, false>(
                simi, nv, nullptr,
                [&](int j, double*) CBY_L { return !kIncrementalEta || vrow == -2 || j == vrow || vcol == -2 || j == vcol; },
                [&](int, int, double& v, double, double* acc) CBY_L { acc[0] += v * v; },
                [&](int j, bool walked, double* acc) CBY_L {
                  if (walked) vsig[j] = 1.0 / sqrt(acc[0]);
                });
            walk_tiled<false, false>(
                sim, nv + 1, nullptr,
                [&](int j, double*) CBY_L { return !kIncrementalEta || vcol == -2 || j == vcol; },
                [&](int, int, double& v, double, double* acc) CBY_L { acc[0] += v * v; },
                [&](int j, bool walked, double* acc) CBY_L {
                  if (walked) veta[j] = sqrt(acc[0]);
                  if (vsig[j] < parsig || veta[j] > pareta) flag_bad = 1;
                });
            walked_tiles = true;
          }
        }
        if (!walked_tiles)
        for (int j = rlane; j < n; j += rstep) {
          double wsig = 0.0, weta = 0.0;
          const bool need_eta = !kIncrementalEta || vcol == -2 || j == vcol;
          const bool need_sig = !kIncrementalEta || vrow == -2 || j == vrow;
          if (!need_sig && !need_eta) {
            if (vsig[j] < parsig || veta[j] > pareta) flag_bad = 1;
            continue;
          }
          if (need_eta) {
            for (int i0 = ilo; i0 < ihi; i0 += P) {
              double v[P], u[P];
              CBY_FULL_UNROLL
              for (int q = 0; q < P; ++q) { v[q] = SIMI(j, i0 + q); u[q] = SIM(i0 + q, j); }
              CBY_FULL_UNROLL
              for (int q = 0; q < P; ++q) { wsig += v[q] * v[q]; weta += u[q] * u[q]; }
            }
          } else {
            for (int i0 = ilo; i0 < ihi; i0 += P) {
              double v[P];
              CBY_FULL_UNROLL
              for (int q = 0; q < P; ++q) v[q] = SIMI(j, i0 + q);
              CBY_FULL_UNROLL
              for (int q = 0; q < P; ++q) wsig += v[q] * v[q];
            }
          }
          if (split) { wsig = ctx.pair_sum(wsig); weta = ctx.pair_sum(weta); }
          const double vs = 1.0 / sqrt(wsig), ve = need_eta ? sqrt(weta) : (double)veta[j];
          vsig[j] = vs;
          if (need_eta) veta[j] = ve;
          if (vs < parsig || ve > pareta) flag_bad = 1;
        }
        vcol = -1; vrow = -1;
        iflag = ctx.all_or(flag_bad) ? 0 : 1;  // all_or synchronises
        CBY_LOG("    acceptable %d (parsig %.17g pareta %.17g)", iflag, parsig, pareta);
        CBY_STAMP(5);
        if (ibrnch == 1 || iflag == 1) { lbl = 370; continue; }
        // ---- geometry step: replace the worst-placed vertex
        double temp;
        int jd = ctx.arg_first(n, [&](int j) { return veta[j]; }, pareta, true, &temp);
        if (jd < 0) jd = ctx.arg_first(n, [&](int j) { return vsig[j]; }, pareta, false, &temp);
        jdrop = jd;
        CBY_LOG("    geometry step: jdrop %d (veta/vsig extreme %.17g)", jd, temp);
        temp = 0.5 * rho * vsig[jdrop];
        const double sum = ctx.sum(n, [&](int i) { return a[i] * (temp * SIMI(jdrop, i)); });
        // dxsign = -1 iff parmu*(cvmaxp-cvmaxm) > 2*sum with parmu = 0
        const double dxsign = (0.0 > sum + sum) ? -1.0 : 1.0;
        CBY_LOG("    geometry sum %.17g dxsign %g", sum, dxsign);
        ctx.sync();
        for (int i = ctx.tid; i < n; i += ctx.nth) {
          const double d = dxsign * (temp * SIMI(jdrop, i));
          dx[i] = d;
          SIM(i, jdrop) = d;
        }
        vertex_changed(jdrop);
        ctx.sync();
        update_simi(false);
        for (int j = ctx.tid; j < n; j += ctx.nth) x[j] = SIM(j, nv) + dx[j];
        ctx.sync();
        return request_eval();  // ibrnch == 0: the value lands in datmat[jdrop]
      }
      if (lbl == 370) {
        trstlp_m0();
        if (ifull == 0) {
          const double t = ctx.sum(n, [&](int i) { return dx[i] * dx[i]; });
          if (t < 0.25 * rho * rho) { ibrnch = 1; lbl = 550; continue; }
        }
        // sum = 0 - a.dx accumulated term by term; prerem = parmu*prerec - sum with parmu = 0
        prerem = ctx.sum(n, [&](int i) { return a[i] * dx[i]; });
        CBY_LOG("370: trust-region step, prerem %.17g ifull %d", prerem, ifull);
        ctx.sync();
        for (int i = ctx.tid; i < n; i += ctx.nth) x[i] = SIM(i, nv) + dx[i];
        ibrnch = 1;
        ctx.sync();
        CBY_STAMP(6);
        return request_eval();
      }
      if (lbl == 440) {
        const double vmold = datmat[nv];
        trured = vmold - f;
        if (f == vmold) { prerem = 0.0; trured = 0.0; }
        // ---- which vertex (if any) does x(*) replace
        double ratio = (trured <= 0.0) ? 1.0 : 0.0;
        bool walked_tiles = false;
        if constexpr (Ctx::kTile && Ctx::kTileWalks) {
          if (!split) {
            walk_tiled<false, true>(
                simi, nv, dx, [&](int, double*) CBY_L { return true; },
                [&](int, int, double& v, double u, double* acc) CBY_L { acc[0] += v * u; },
                [&](int j, bool, double* acc) CBY_L {
                  double t = acc[0];
                  tdot[j] = t;
                  t = fabs(t);
                  w[j] = t;
                  sigbar[j] = t * vsig[j];
                });
            walked_tiles = true;
          }
        }
        if (!walked_tiles)
        for (int j = rlane; j < n; j += rstep) {
          double t = 0.0;
          for (int i0 = ilo; i0 < ihi; i0 += P) {
            double v[P], u[P];
            CBY_FULL_UNROLL
            for (int q = 0; q < P; ++q) { v[q] = SIMI(j, i0 + q); u[q] = dx[i0 + q]; }
            CBY_FULL_UNROLL
            for (int q = 0; q < P; ++q) t += v[q] * u[q];
          }
          if (split) t = ctx.pair_sum(t);
          tdot[j] = t;
          t = fabs(t);
          w[j] = t;
          sigbar[j] = t * vsig[j];
        }
        ctx.sync();
        int jd = ctx.arg_first(n, [&](int j) { return w[j]; }, ratio, true, &ratio);
        ctx.sync();
        CBY_STAMP(0);
        walked_tiles = false;
        if constexpr (Ctx::kTile && Ctx::kTileWalks) {
          if (!split) {
            walk_tiled<false, true>(
                sim, nv + 1, dx,
                [&](int j, double* acc) CBY_L {      // acc[1]: the value of a row that is not walked
                  acc[1] = -1.0;
                  if (sigbar[j] >= parsig || sigbar[j] >= vsig[j]) {
                    acc[1] = veta[j];
                    return trured > 0.0;
                  }
                  return false;
                },
                [&](int, int, double& v, double u, double* acc) CBY_L { const double d = u - v; acc[0] += d * d; },
                [&](int j, bool walked, double* acc) CBY_L { w[j] = walked ? sqrt(acc[0]) : acc[1]; });
            walked_tiles = true;
          }
        }
        if (!walked_tiles)
        for (int j = rlane; j < n; j += rstep) {
          double t = -1.0;
          if (sigbar[j] >= parsig || sigbar[j] >= vsig[j]) {
            t = veta[j];
            if (trured > 0.0) {
              t = 0.0;
              for (int i0 = ilo; i0 < ihi; i0 += P) {
                double v[P], u[P];
                CBY_FULL_UNROLL
                for (int q = 0; q < P; ++q) { v[q] = dx[i0 + q]; u[q] = SIM(i0 + q, j); }
                CBY_FULL_UNROLL
                for (int q = 0; q < P; ++q) { const double d = v[q] - u[q]; t += d * d; }
              }
              if (split) t = ctx.pair_sum(t);
              t = sqrt(t);
            }
          }
          w[j] = t;
        }
        ctx.sync();
        double edgmax;
        const int l = ctx.arg_first(n, [&](int j) { return w[j]; }, 1.1 * rho, true, &edgmax);
        CBY_LOG("440: trured %.17g prerem %.17g jd(ratio) %d ratio %.17g l(edge) %d edgmax %.17g", trured, prerem, jd, ratio, l, edgmax);
        if (l >= 0) jd = l;
        if (jd < 0) { lbl = 550; continue; }
        jdrop = jd;
        ctx.sync();
        for (int i = ctx.tid; i < n; i += ctx.nth) SIM(i, jdrop) = dx[i];
        vertex_changed(jdrop);
        if (ctx.tid == 0) datmat[jdrop] = f;
        ctx.sync();
        CBY_STAMP(1);
        update_simi(true);     // tdot[] still holds simi_j . dx for this dx
        CBY_STAMP(2);
        if (trured > 0.0 && trured >= 0.1 * prerem) { lbl = 140; continue; }
        lbl = 550;
        continue;
      }
      // lbl == 550
      CBY_LOG("550: iflag %d rho %g", iflag, rho);
      if (iflag == 0) { ibrnch = 0; lbl = 140; continue; }
      if (rho > rhoend) {
        rho = 0.5 * rho;
        if (rho <= 1.5 * rhoend) rho = rhoend;
        lbl = 140;
        continue;
      }
      status = DONE_RHOEND;
      return finish(ifull == 1);
    }
  }

  // Rank-one update of SIMI after vertex jdrop was replaced by pole + dx.
  // have_tdot: tdot[j] = sum_i SIMI(j, i) * dx[i] was left by the caller (same accumulation
  // order as the loops below would use: identical bits, one pass over simi saved).
  CBY_HD void update_simi(bool have_tdot) {
    const double temp = have_tdot ? (double)tdot[jdrop] : ctx.sum(n, [&](int i) { return SIMI(jdrop, i) * dx[i]; });
    ctx.sync();
    for (int i = ctx.tid; i < n; i += ctx.nth) SIMI(jdrop, i) /= temp;
    ctx.sync();
    if constexpr (Ctx::kColumns && !Ctx::kTile) {
      // Workgroup context: the update is element-wise, so it does not matter who does it - wave w takes rows
      // w, w + NW, ... with its lanes along the row (coalesced, every thread busy, nothing to reduce; a thread
      // per row walks 2 nv doubles through L2 with a stride of a whole row between neighbouring lanes).
      if (!have_tdot) {
        for (int j = rlane; j < n; j += rstep) {
          double t = 0.0;
          for (int i0 = ilo; i0 < ihi; i0 += P) {
            double v[P], u[P];
            CBY_FULL_UNROLL
            for (int q = 0; q < P; ++q) { v[q] = SIMI(j, i0 + q); u[q] = dx[i0 + q]; }
            CBY_FULL_UNROLL
            for (int q = 0; q < P; ++q) t += v[q] * u[q];
          }
          tdot[j] = t;
        }
        ctx.sync();
      }
      const int wv = ctx.tid >> 6, ln = ctx.tid & 63;
      for (int j = wv; j < n; j += Ctx::nth / 64) {
        if (j == jdrop) continue;
        const double t = tdot[j];
        for (int i = ln; i < nv; i += 64) {
          const double v = SIMI(j, i), u = SIMI(jdrop, i);
          SIMI(j, i) = v - t * u;
        }
      }
      vrow = -2;   // (the row norms are not formed on this path)
      ctx.sync();
      return;
    }
    if constexpr (Ctx::kTile) {
      if (!split) {
        if (!have_tdot)
          walk_tiled<false, true>(
              simi, nv, dx, [&](int j, double*) CBY_L { return j != jdrop; },
              [&](int, int, double& v, double u, double* acc) CBY_L { acc[0] += v * u; },
              [&](int j, bool walked, double* acc) CBY_L {
                if (walked) tdot[j] = acc[0];      // (read back by the same lane below)
              });
        // row jdrop (rescaled above) rides along with a factor of zero: its entries stay as they are, its norm - the
        // same sum in the same order as the owner's loop of the plain path - comes out of the same walk
        walk_tiled<true, true>(
            simi, nv, simi + (size_t)jdrop * ld,
            [&](int j, double* acc) CBY_L {
              acc[1] = j == jdrop ? 0.0 : (double)tdot[j];
              return true;
            },
            [&](int j, int, double& v, double u, double* acc) CBY_L {
              const double nw = v - acc[1] * u;
              acc[0] += nw * nw;     // the row norm the next acceptability test wants: same order as there
              v = j == jdrop ? v : nw;
            },
            [&](int j, bool, double* acc) CBY_L { vsig[j] = 1.0 / sqrt(acc[0]); });
        vrow = -1;
        ctx.sync();
        return;
      }
    }
    for (int j = rlane; j < n; j += rstep) {
      if (j == jdrop) continue;
      double t = 0.0;
      if (have_tdot) {
        t = tdot[j];
      } else {
        for (int i0 = ilo; i0 < ihi; i0 += P) {
          double v[P], u[P];
          CBY_FULL_UNROLL
          for (int q = 0; q < P; ++q) { v[q] = SIMI(j, i0 + q); u[q] = dx[i0 + q]; }
          CBY_FULL_UNROLL
          for (int q = 0; q < P; ++q) t += v[q] * u[q];
        }
        if (split) t = ctx.pair_sum(t);
      }
      double ws = 0.0;
      for (int i0 = ilo; i0 < ihi; i0 += P) {
        double v[P], u[P];
        CBY_FULL_UNROLL
        for (int q = 0; q < P; ++q) { v[q] = SIMI(j, i0 + q); u[q] = SIMI(jdrop, i0 + q); }
        CBY_FULL_UNROLL
        for (int q = 0; q < P; ++q) {
          const double nw = v[q] - t * u[q];
          SIMI(j, i0 + q) = nw;
          if (kIncrementalEta) ws += nw * nw;     // the row norm the next acceptability test wants: same order as there
        }
      }
      if (kIncrementalEta) {
        if (split) ws = ctx.pair_sum(ws);
        vsig[j] = 1.0 / sqrt(ws);
      }
    }
    if (kIncrementalEta) {   // ... and the norm of the rescaled row jdrop, by its owner
      for (int j = rlane; j < n; j += rstep) {
        if (j != jdrop) continue;
        double ws = 0.0;
        for (int i0 = ilo; i0 < ihi; i0 += P) {
          double v[P];
          CBY_FULL_UNROLL
          for (int q = 0; q < P; ++q) v[q] = SIMI(j, i0 + q);
          CBY_FULL_UNROLL
          for (int q = 0; q < P; ++q) ws += v[q] * v[q];
        }
        if (split) ws = ctx.pair_sum(ws);
        vsig[j] = 1.0 / sqrt(ws);
      }
      vrow = -1;
    }
    ctx.sync();
  }
};

}  // namespace cby

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
