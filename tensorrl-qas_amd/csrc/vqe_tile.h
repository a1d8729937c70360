// vqe_tile.h - LDS-tiled kernels of the streaming path (n >= 14; included by vqe_stream.h).
//
// One gate sweep over a 20-qubit state moves 32 MiB through HBM; a circuit of ~16 rotations and a
// Hamiltonian of ~20 X-mask groups would move it ~25 times if every op / group streamed the vector.
// Instead a workgroup stages a TILE of 2^11 amplitudes (32 KiB of LDS, four workgroups per CU: measured
// 31.5 k evaluations/s on the 20-qubit bench workload against 25.9 k with 64 KiB tiles and 28.4 k with
// 16 KiB ones - occupancy against passes) and does everything that closes inside it before the tile goes back:
//   * a tile is a coset p0 ^ V of an 11-dimensional subspace V of GF(2)^n that contains the unit vectors
//     e_0..e_2 (circuit passes: aligned 128-byte runs, coalesced loads and stores) resp. e_0..e_3 (the reduction:
//     256-byte runs) plus up to eight resp. seven independent pair masks - the physical partner masks A^-1 e_q of
//     the rotations (vqe_device.h: CNOTs never move data), or the physical X masks of Pauli groups; masks that
//     DEPEND on the basis ride along for free;
//   * a planner thread per stream cuts the op list into passes (maximal runs of ops whose masks fit one
//     V) and packs the X-mask groups into passes first-fit; V is kept as a fully reduced basis, so tile
//     coordinates of a mask are just its bits at the pivot positions, and the sign selector parity(p & z)
//     of a rotation / Pauli term splits into parity(p0 & z) (per tile, scalar) ^ parity(t & cz) with
//     cz_i = parity(basis_i & z);
//   * inside a tile a chunk of up to three pair ops and the diagonal ops between them is applied from registers
//     (the coset trick of k_s_opk, in tile coordinates: 256 threads x 8 amplitudes), so LDS sees one read + one
//     write of the tile per chunk;
//   * the first pass of a stream reads the shared initial state instead of its own buffer (no separate
//     initialisation sweep); the LAST pass takes Pauli-group masks into its basis where it has room and evaluates
//     the pair groups that close inside its tile before the tile is stored (the fused pass: one reduction sweep
//     less for most streams); pair groups of two equal-magnitude real terms (XX + YY of a bond) are evaluated on the
//     half space on which their sum does not vanish (half groups);
//   * a workgroup walks several tiles with the next one already in flight, and everything a kernel needs per
//     chunk / group / term comes precomputed from the planner by scalar loads: a SIMD issues one instruction per
//     4-5 cycles whatever its kind, so the instruction count per amplitude is what these kernels are written for
//     (DESIGN.md 4.3).
// Every floating-point update of the circuit is the per-element form of s_apply_k (vqe_stream.h).
#pragma once

namespace vqe {

#ifndef VQE_TILE_BITS
#define VQE_TILE_BITS 11
#endif
constexpr int kTileBits = VQE_TILE_BITS;
#ifndef VQE_TILE_LOW
#define VQE_TILE_LOW 3
#endif
#ifndef VQE_ETILE_LOW
#define VQE_ETILE_LOW 4
#endif
// e_0 .. e_{kTileLow-1} are in every tile: aligned runs of 16 << kTileLow bytes.  Circuit passes: 128-byte runs (one
// cache line) and EIGHT independent masks per pass - fewer passes: 49.0 -> 50.6 k evaluations/s on the 20-qubit bench
// workload against 256-byte runs and seven masks (512-byte runs / six masks: 47.6 k; 64-byte runs / nine: 50.3 k).  The
// reduction keeps 256-byte runs (its three passes do not become two with eight masks, and it only reads).
constexpr int kTileLow = VQE_TILE_LOW;
constexpr int kETileLow = VQE_ETILE_LOW;
constexpr int kTileAmps = 1 << kTileBits;
constexpr int kTileFree = kTileBits - kTileLow;   // independent masks a pass can take
constexpr int kMaxEnergyPasses = 32;
// The Pauli-term reduction uses its own, larger tiles: it only reads the state, so a pass costs it one sweep
// where a circuit pass costs two, and it is bound by the instructions per amplitude and group, which 16
// amplitudes per thread halve; 2^12 amplitudes = 64 KiB, two workgroups per CU, 8 independent masks per pass
// (20-qubit Heisenberg chain: 2 passes instead of 3).
#ifndef VQE_ETILE_BITS
#define VQE_ETILE_BITS 11
#endif
constexpr int kETileBits = VQE_ETILE_BITS;
constexpr int kETileAmps = 1 << kETileBits;
constexpr int kETileFree = kETileBits - kETileLow;

struct TilePass {
  uint32_t basis[kTileBits];  // fully reduced (every vector has a pivot = highest bit that no other vector has), ascending pivots:
                              // basis[i] = 1 << i below kTileLow, and tile coordinate i <-> basis[i]
  uint32_t pivmask;
  int32_t begin, end;         // ops [begin, end) of the stream / entries [begin, end) of its group order
  int32_t rec_begin, rec_count;   // energy passes: the term records of the pass (TermRec, pass order)
  uint32_t pad;
};
struct ETilePass {            // a pass of the Pauli-term reduction (same fields, kETileBits basis vectors)
  uint32_t basis[kETileBits];
  uint32_t pivmask;
  int32_t begin, end;
  int32_t rec_begin, rec_count;
  uint32_t pad;
};
struct OpCoord { uint32_t cx, cz; };   // tile coordinates of an op's pair mask / sign mask
constexpr int kTileK = kTileBits - 8;   // ops applied per LDS round trip: 256 threads x 2^K amplitudes = one tile
// One chunk of kTileK ops as k_t_ops wants it (made once by the planner, read with scalar loads): slot j of the
// thread's coset basis is the partner mask of op j when that is independent of the earlier slots, else a filler
// unit vector (the bookkeeping of k_s_opk, in tile coordinates).
// A chunk holds up to kTileK pair ops (RX / RY: each needs a slot) and the diagonal ops (RZ, Pauli-Z) between
// them, which act on whatever elements a thread holds: kChunkOps ops at most.
constexpr int kChunkOps = 6;
struct ChunkRec {             // 24 dwords, indexed by the chunk's first op
  uint32_t g16[3];            // slot masks as byte offsets (<< 4)
  uint32_t pvc;               // pivot positions of the reduced slot basis, ascending, 5 bits each | number of ops << 16
  struct {
    uint32_t kfe;             // kind | flip code << 4 (partner of element e: e ^ flip) | inversion << 8 | ebits << 16
    uint32_t cz, zm;          // sign selector in tile coordinates / the op's physical Z mask (sign of the tile origin)
  } op[kChunkOps];            // ebits: bit e = parity(cz & combination e of the slot masks)
  uint32_t pad[2];
};
struct TermRec { double wr, wi; uint32_t tz, cz; };   // coefficient (incl. (-1)^{z.c}), physical Z mask, its tile coordinates

// ---- basis bookkeeping of the planners (one thread per stream; arrays live in scratch: irrelevant here) ----
struct TileBasis {
  uint32_t v[kTileBits];
  int piv[kTileBits];
  int dim;
  __device__ void reset() {
    for (int i = 0; i < kTileLow; ++i) { v[i] = 1u << i; piv[i] = i; }
    dim = kTileLow;
  }
  __device__ uint32_t reduce(uint32_t x) const {
    for (int i = 0; i < dim; ++i) if ((x >> piv[i]) & 1u) x ^= v[i];
    return x;
  }
  __device__ void add(uint32_t r) {   // r != 0, already reduced
    const int p = 31 - __clz((int)r);
    for (int i = 0; i < dim; ++i) if ((v[i] >> p) & 1u) v[i] ^= r;
    v[dim] = r; piv[dim] = p; ++dim;
  }
  __device__ void fill(int n) {       // complete to kTileBits dimensions with the lowest free unit directions
    for (int q = kTileLow; q < n && dim < kTileBits; ++q) {
      const uint32_t r = reduce(1u << q);
      if (r) add(r);
    }
  }
  __device__ void sort() {            // ascending pivots: the low tile coordinates (= lanes of a wave) get the low address bits
    for (int i = 1; i < dim; ++i)
      for (int j = i; j > 0 && piv[j - 1] > piv[j]; --j) {
        const uint32_t tv = v[j]; v[j] = v[j - 1]; v[j - 1] = tv;
        const int tp = piv[j]; piv[j] = piv[j - 1]; piv[j - 1] = tp;
      }
  }
  __device__ void emit(TilePass* out, int begin, int end) const {
    uint32_t pm = 0;
    for (int i = 0; i < kTileBits; ++i) { out->basis[i] = v[i]; pm |= 1u << piv[i]; }
    out->pivmask = pm; out->begin = begin; out->end = end; out->rec_begin = 0; out->rec_count = 0; out->pad = 0;
  }
  __device__ uint32_t coords(uint32_t x) const {     // x in the span
    uint32_t c = 0;
    for (int i = 0; i < kTileBits; ++i) c |= ((x >> piv[i]) & 1u) << i;
    return c;
  }
  __device__ uint32_t zcoords(uint32_t z) const {
    uint32_t c = 0;
    for (int i = 0; i < kTileBits; ++i) c |= (uint32_t)parity32(v[i] & z) << i;
    return c;
  }
};

// ops of every stream -> passes + tile coordinates.  passes: [batch][max_pass]; opc: [batch][max_ops]
static_assert(sizeof(ChunkRec) == 96 && kTileK <= 3, "ChunkRec: three slots, 24 dwords");
// gxp (or nullptr): the physical X masks of the Hamiltonian's groups, [batch][n_groups] - the LAST pass of a stream
// takes as many of them into its basis as it has room for (instead of filler unit vectors): those groups are then
// evaluated by k_t_ops while the final state is still in LDS (k_t_plan_energy makes that basis its pass 0).
__global__ void k_t_plan_ops(BatchArgs A, const Op* ops, const int32_t* meta, TilePass* passes, OpCoord* opc,
                             ChunkRec* chunks, int32_t* npass, int max_pass, const uint32_t* gxp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.batch) return;
  const int nops = meta[(size_t)b * 8];
  const Op* op = ops + (size_t)b * A.max_ops;
  OpCoord* oc = opc + (size_t)b * A.max_ops;
  ChunkRec* ck = chunks + (size_t)b * A.max_ops;
  constexpr int K = kTileK, E = 1 << K;
  // The coset bookkeeping of k_s_opk in tile coordinates for the chunk that starts at op o of a pass ending at
  // `end`: pair ops take the slots in order while their masks are independent of the slots so far (a dependent
  // mask gets its flip code), diagonal ops ride along; returns the number of ops taken.
  auto plan_chunk = [&](int o, int end) {
    uint32_t g[4] = {0u, 0u, 0u, 0u}, red[4] = {0u, 0u, 0u, 0u};
    int hbit[4] = {0, 0, 0, 0};
    int slot_of[kChunkOps];
    uint32_t pivots = 0;
    int nred = 0, nslots = 0, cnt = 0;
    auto reduce = [&](uint32_t x) {
      for (int i = 0; i < nred; ++i) if ((x >> hbit[i]) & 1u) x ^= red[i];
      return x;
    };
    auto push = [&](uint32_t x) {
      const int h = 31 - __clz((int)x);
      for (int i = 0; i < nred; ++i) if ((red[i] >> h) & 1u) red[i] ^= x;
      red[nred] = x; hbit[nred] = h;
      pivots |= 1u << h;
      ++nred;
    };
    while (o + cnt < end && cnt < kChunkOps) {
      const int kd = op[o + cnt].kind & 0xff;
      slot_of[cnt] = -1;
      if (kd == OP_RX || kd == OP_RY) {
        const uint32_t r = reduce(oc[o + cnt].cx);
        if (r) {
          if (nslots == K) break;              // a fourth independent mask: the next chunk
          push(r); g[nslots] = oc[o + cnt].cx; slot_of[cnt] = nslots; ++nslots;
        }
      }
      ++cnt;
    }
    int q = 0;
    for (int j = nslots; j < K; ++j) {         // fillers for the unused slots
      uint32_t r = 0;
      while ((r = reduce(1u << q)) == 0) ++q;
      push(r);
      g[j] = 1u << q;
      ++q;
    }
    ChunkRec c;
    for (int j = 0; j < 3; ++j) c.g16[j] = g[j] << 4;
    uint32_t pvc = (uint32_t)cnt << 16;
    {
      int k = 0;
      for (int qb = 0; qb < kTileBits; ++qb) if ((pivots >> qb) & 1u) { pvc |= (uint32_t)qb << (5 * k); ++k; }
    }
    c.pvc = pvc;
    c.pad[0] = c.pad[1] = 0u;
    for (int j = 0; j < kChunkOps; ++j) {
      c.op[j].kfe = OP_NOP; c.op[j].cz = 0u; c.op[j].zm = 0u;
      if (j < cnt) {
        const int kd = op[o + j].kind & 0xff;
        int flip = 0;
        if (kd == OP_RX || kd == OP_RY) {
          if (slot_of[j] >= 0) flip = 1 << slot_of[j];
          else
            for (int f = 1; f < E; ++f) {
              uint32_t x = 0;
              for (int i = 0; i < K; ++i) if ((f >> i) & 1) x ^= g[i];
              if (x == oc[o + j].cx) flip = f;
            }
        }
        const uint32_t cz = oc[o + j].cz;
        uint32_t ebits = 0;
        for (int e = 0; e < E; ++e) {
          uint32_t x = 0;
          for (int i = 0; i < K; ++i) if ((e >> i) & 1) x ^= g[i];
          ebits |= (uint32_t)parity32(x & cz) << e;
        }
        c.op[j].kfe = (uint32_t)(kd & 0xf) | ((uint32_t)flip << 4) | ((uint32_t)((op[o + j].kind >> 8) & 1) << 8) | (ebits << 16);
        c.op[j].cz = cz;
        c.op[j].zm = op[o + j].zm;
      }
    }
    ck[o] = c;
    return cnt;
  };
  TilePass* P = passes + (size_t)b * max_pass;
  TileBasis B;
  B.reset();
  int begin = 0, np = 0;
  auto close = [&](int end, bool last = false) {
    if (last && gxp) {
      const uint32_t* gx = gxp + (size_t)b * A.ham.n_groups;
      for (int g = 0; g < A.ham.n_groups && B.dim < kTileBits; ++g) {
        const uint32_t r = B.reduce(gx[g]);
        if (r) B.add(r);
      }
    }
    B.fill(A.n);
    B.sort();
    if (np < max_pass) {
      B.emit(P + np, begin, end);
      for (int o = begin; o < end; ++o) {
        const int kd = op[o].kind & 0xff;
        oc[o].cx = (kd == OP_RX || kd == OP_RY) ? B.coords(op[o].xm) : 0u;
        oc[o].cz = B.zcoords(op[o].zm);
      }
      for (int o = begin; o < end;) o += plan_chunk(o, end);
    }
    ++np;
  };
  for (int o = 0; o < nops; ++o) {
    const int kd = op[o].kind & 0xff;
    if (kd != OP_RX && kd != OP_RY) continue;          // diagonal ops fit every tile
    const uint32_t r = B.reduce(op[o].xm);
    if (!r) continue;
    if (B.dim < kTileBits) { B.add(r); continue; }
    close(o);
    begin = o;
    B.reset();
    B.add(B.reduce(op[o].xm));
  }
  close(nops, true);                                     // (a stream without ops still gets its copy pass)
  npass[b] = np < max_pass ? np : max_pass;
}

// X-mask groups of every stream -> passes (first fit) + tile coordinates of the groups + the term records of
// every pass in the order the tile kernel consumes them.
// order: [batch][n_groups] group ids pass by pass; gcx, grec: [batch][n_groups]; trec: [batch][n_terms]
// What k_t_energy reads per group / per term, made once by the planner and fetched with SCALAR loads
// (everything in them is the same for all threads and all tiles of the pass): every instruction a wave issues -
// scalar, LDS or vector - costs the SIMD one issue slot, so whatever can be precomputed is.
constexpr int kEPairs = kETileAmps / 2 / kThreads;   // pairs per thread of k_t_energy
struct EGroupRec {            // 4 + kEPairs dwords
  uint32_t cx16;              // tile coordinates of the X mask, as a byte offset (<< 4); 0: the diagonal group
  uint32_t hb;                // highest set bit of the coordinates
  int32_t nt;                 // its terms: the next nt records of eterm (groups and terms are stored in the order they run)
  uint32_t im;                // 1: some weight has an imaginary part (then wi[] is read as well); 2: a half group (hb, u16: see the planner)
  uint32_t u16[kEPairs];      // byte offset of pair k of a thread: insert0(k * kThreads, hb) << 4
};
struct ETermRec {             // 4 + 2 kEPairs dwords
  double wr;                  // real part of the weight (pair groups: times 2, the p <-> p ^ x symmetry)
  uint32_t cz, tz;            // sign selector in tile coordinates / physical Z mask (sign of the tile origin)
  double sg[kEPairs];         // +-1: parity of cz with the k-th pair of a thread (pair groups)
};
__global__ void k_t_plan_energy(BatchArgs A, int n_terms, const uint32_t* gxp, const uint32_t* tzp, const double* tsg,
                                ETilePass* passes, int32_t* npass, int32_t* order, uint32_t* gcx, int32_t* grec,
                                TermRec* trec, int32_t* gpass, EGroupRec* egrp, ETermRec* eterm, double* ewi,
                                const TilePass* opasses, const int32_t* onpass, int max_pass) {
  static_assert(kETileBits == kTileBits, "the fused pass shares its tile with the last circuit pass");
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.batch) return;
  const int ng = A.ham.n_groups;
  const uint32_t* gx = gxp + (size_t)b * ng;
  int32_t* gp = gpass + (size_t)b * ng;
  ETilePass* P = passes + (size_t)b * kMaxEnergyPasses;
  // pass bases live in the output records while they are built (fully reduced at every moment)
  int np = 0;
  int dims[kMaxEnergyPasses];
  int piv[kMaxEnergyPasses][kETileBits];
  auto reduce = [&](int k, uint32_t x) {
    for (int i = 0; i < dims[k]; ++i) if ((x >> piv[k][i]) & 1u) x ^= P[k].basis[i];
    return x;
  };
  auto add = [&](int k, uint32_t r) {
    const int p = 31 - __clz((int)r);
    for (int i = 0; i < dims[k]; ++i) if ((P[k].basis[i] >> p) & 1u) P[k].basis[i] ^= r;
    P[k].basis[dims[k]] = r; piv[k][dims[k]] = p; ++dims[k];
  };
  auto open = [&]() {
    for (int i = 0; i < kETileLow; ++i) { P[np].basis[i] = 1u << i; piv[np][i] = i; }
    dims[np] = kETileLow;
    return np++;
  };
  const bool fused = opasses != nullptr;
  if (fused) {      // pass 0 = the tile of the stream's last circuit pass (complete: takes the PAIR groups that lie in it, no more;
                    // the diagonal group - the heaviest - stays with k_t_energy, where the sweeps overlap its arithmetic)
    const TilePass& L = opasses[(size_t)b * max_pass + (onpass[b] > 0 ? onpass[b] - 1 : 0)];
    int i = 0;
    for (int q = 0; q < 32 && i < kETileBits; ++q)
      if ((L.pivmask >> q) & 1u) { P[0].basis[i] = L.basis[i]; piv[0][i] = q; ++i; }
    dims[0] = kETileBits;
    np = 1;
  }
  for (int g = 0; g < ng; ++g) {
    const uint32_t x = gx[g];
    int dst = -1;
    for (int k = (fused && x == 0u) ? 1 : 0; k < np && dst < 0; ++k) {
      const uint32_t r = reduce(k, x);
      if (!r) dst = k;
      else if (dims[k] < kETileBits) { add(k, r); dst = k; }
    }
    if (dst < 0) {
      if (np < kMaxEnergyPasses) { dst = open(); const uint32_t r = reduce(dst, x); if (r) add(dst, r); }
      else dst = kMaxEnergyPasses - 1;      // (host bounds n_groups so that this cannot happen)
    }
    gp[g] = dst;
  }
  if (np == 0) open();                       // a Hamiltonian shard without groups: one empty pass
  // close the passes: fill, sort by pivot, group lists pass by pass, coordinates, term records
  int32_t* ord = order + (size_t)b * ng;
  TermRec* rec = trec + (size_t)b * n_terms;
  int pos = 0, rpos = 0;
  for (int k = 0; k < np; ++k) {
    for (int q = kETileLow; q < A.n && dims[k] < kETileBits; ++q) { const uint32_t r = reduce(k, 1u << q); if (r) add(k, r); }
    for (int i = 1; i < kETileBits; ++i)
      for (int j = i; j > 0 && piv[k][j - 1] > piv[k][j]; --j) {
        const uint32_t tv = P[k].basis[j]; P[k].basis[j] = P[k].basis[j - 1]; P[k].basis[j - 1] = tv;
        const int tp = piv[k][j]; piv[k][j] = piv[k][j - 1]; piv[k][j - 1] = tp;
      }
    uint32_t pm = 0;
    for (int i = 0; i < kETileBits; ++i) pm |= 1u << piv[k][i];
    P[k].pivmask = pm; P[k].begin = pos; P[k].rec_begin = rpos; P[k].pad = 0;
    for (int g = 0; g < ng; ++g)
      if (gp[g] == k) {
        uint32_t c = 0;
        for (int i = 0; i < kETileBits; ++i) c |= ((gx[g] >> piv[k][i]) & 1u) << i;
        gcx[(size_t)b * ng + g] = c;
        grec[(size_t)b * ng + g] = rpos - P[k].rec_begin;
        const int hb = c ? 31 - __clz((int)c) : 0;
        EGroupRec G;
        G.cx16 = c << 4; G.hb = (uint32_t)hb; G.nt = A.ham.term_off[g + 1] - A.ham.term_off[g]; G.im = 0u;
        for (int j = 0; j < kEPairs; ++j) G.u16[j] = c ? insert0((uint32_t)j * kThreads, hb) << 4 : 0u;
        // the diagonal group's terms go out class by class (class = the bits of cz that the thread's elements
        // differ in): the kernel sums every class once and combines the classes by a Walsh-Hadamard butterfly
        constexpr int kClsBits = kETileBits - 8, kCls = 1 << kClsBits;
        for (int cls = 0; cls < (c ? 1 : kCls); ++cls) {
          int in_class = 0;
          for (int t = A.ham.term_off[g]; t < A.ham.term_off[g + 1]; ++t) {
            const uint32_t z = tzp[(size_t)b * n_terms + t];
            uint32_t cz = 0;
            for (int i = 0; i < kETileBits; ++i) cz |= (uint32_t)parity32(P[k].basis[i] & z) << i;
            if (!c && (int)(cz >> 8) != cls) continue;
            const double sg = tsg[(size_t)b * n_terms + t];
            const double wr = sg * A.ham.term_cr[t], wi = sg * A.ham.term_ci[t];
            ETermRec E;
            E.wr = c ? 2.0 * wr : wr; E.cz = cz; E.tz = z;
            for (int j = 0; j < kEPairs; ++j) E.sg[j] = parity32(insert0((uint32_t)j * kThreads, hb) & cz) ? -1.0 : 1.0;
            if (wi != 0.0) G.im = 1u;
            eterm[(size_t)b * n_terms + rpos] = E;
            ewi[(size_t)b * n_terms + rpos] = c ? 2.0 * wi : wi;
            rec[rpos++] = TermRec{wr, wi, z, cz};
            ++in_class;
          }
          if (!c) G.u16[cls >> 1] |= (uint32_t)in_class << (16 * (cls & 1));     // (u16 is zero for the diagonal group)
        }
        // A pair group of TWO real terms of equal magnitude (XX + YY of a bond: the hopping part of a number-conserving
        // operator) has D(p) = w1 s1(p) + w2 s2(p) = 2 w1 s1(p) on the half space parity(p & (z1 ^ z2)) = const and
        // EXACTLY zero on the other half: the kernel then visits only the pairs of that half - two per thread instead
        // of four, one signed weight instead of two (the streaming path's form of the unit path of the LDS kernels).
        // Needs a tile coordinate besides the pair bit in z1 ^ z2 to enumerate the half space with.
        if (c && G.nt == 2 && !G.im) {
          ETermRec& E1 = eterm[(size_t)b * n_terms + rpos - 2];
          const ETermRec& E2 = eterm[(size_t)b * n_terms + rpos - 1];
          const uint32_t czd = (E1.cz ^ E2.cz) & ~(1u << hb);
          if ((E1.wr == E2.wr || E1.wr == -E2.wr) && E1.wr != 0.0 && czd != 0u) {
            const int q = 31 - __clz((int)czd);
            const int lo = q < hb ? q : hb, hi = q < hb ? hb : q;
            G.im = 2u;
            G.hb = (uint32_t)hb | ((uint32_t)q << 8) | ((E1.wr == E2.wr ? 0u : 1u) << 16);
            static_assert(kEPairs >= kEPairs / 2 + 2, "half groups: kEPairs / 2 pair offsets and two masks in u16");
            for (int j = 0; j < kEPairs / 2; ++j)      // pair j of a thread: index bits 8.., zeros inserted at the two holes
              G.u16[j] = insert0(insert0((uint32_t)j * kThreads, lo), hi) << 4;
            G.u16[kEPairs / 2] = E1.cz ^ E2.cz;
            G.u16[kEPairs / 2 + 1] = E1.tz ^ E2.tz;
            E1.wr = 2.0 * E1.wr;      // D on the half space (the pair factor 2 is in wr already)
          }
        }
        egrp[(size_t)b * ng + pos] = G;
        ord[pos++] = g;
      }
    P[k].end = pos;
    P[k].rec_count = rpos - P[k].rec_begin;
  }
  npass[b] = np;
}

// ---- tile addressing (uniform per workgroup unless it depends on tid) ---------------------------------
__device__ __forceinline__ uint32_t tile_origin(uint32_t tile, uint32_t pivmask, int n) {
  uint32_t p = 0;
  int k = 0;
  for (int q = 0; q < n; ++q)
    if (!((pivmask >> q) & 1u)) { p |= ((tile >> k) & 1u) << q; ++k; }
  return p;
}
// offset of tile element t = tid + 256 k: low kTileLow bits are address bits, the other 8 coordinates select basis vectors
template <int LOW, int BITS>
__device__ __forceinline__ uint32_t tile_lane_offset(const uint32_t (&basis)[BITS], uint32_t tid) {
  uint32_t x = tid & ((1u << LOW) - 1u);
#pragma unroll
  for (int i = 0; i < 8 - LOW; ++i) if ((tid >> (LOW + i)) & 1u) x ^= basis[LOW + i];
  return x;
}
template <int BITS>
__device__ __forceinline__ uint32_t tile_k_offset(const uint32_t (&basis)[BITS], int k) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < BITS - 8; ++i) if ((k >> i) & 1) x ^= basis[8 + i];
  return x;
}

// All X-mask groups of one energy pass on the tile that sits in LDS (tile coordinate t at tile[t]): used by k_t_energy
// for every pass and by k_t_ops for the groups that close inside the LAST circuit pass of a stream (the fused pass:
// the final state is evaluated before it leaves the LDS, one sweep over the state less).
struct ETileArgs {
  const EGroupRec* __restrict__ G;      // group records of the pass, in the order they run
  const ETermRec* __restrict__ T;       // its term records
  const double* __restrict__ WI;        // imaginary parts of the weights
  int n_in_pass, r_count;
};
constexpr int kEBlobTerms = kEPairs <= 4 ? 2 : 1;   // term records requested one group ahead (scalar registers: 102 in all)
__device__ __forceinline__ void e_tile_groups(const double2* tile, uint32_t tid, uint32_t pt, const ETileArgs& C, double& acc) {
  constexpr int NE = kETileAmps / kThreads;           // elements per thread
  constexpr int NPR = kEPairs;      // pairs per thread
  constexpr int kBlobTerms = kEBlobTerms;
  const EGroupRec* __restrict__ G = C.G;
  const ETermRec* __restrict__ T = C.T;
  const double* __restrict__ WI = C.WI;
  const int n_in_pass = C.n_in_pass, r_count = C.r_count;
  lds_cbyte* tile_b = (lds_cbyte*)tile;
  // One group: its record and its first two term records are in scalar registers already (requested one group
  // ahead: a scalar load that misses its cache takes ~600 cycles, as long as the whole group)
  struct Blob { EGroupRec g; ETermRec t[kBlobTerms]; };
  auto fetch = [&](Blob& B, int gi, int cur) {
    B.g = G[gi < n_in_pass ? gi : n_in_pass - 1];
#pragma unroll
    for (int i = 0; i < kBlobTerms; ++i) B.t[i] = T[cur + i < r_count ? cur + i : 0];
  };
  {
    // Sign sums D_k = sum_terms w (-1)^{parity(t_k & cz)}: the elements of a thread are t_k = t_0 ^ U_k with
    // U_k the same for every thread, so parity(t_k & cz) = parity(t_0 & cz) ^ parity(U_k & cz): one vector
    // parity per term, the k-dependence is the record's +-1 (a scalar operand of the FMA); the sign of the
    // tile origin, parity(pt & tz), is a scalar too
    auto signed_w = [&](double w, uint32_t t0, uint32_t cz, uint32_t tz) {
      const uint32_t flip = ((uint32_t)__builtin_popcount(t0 & cz) ^ (uint32_t)__builtin_popcount(pt & tz)) << 31;
      return __hiloint2double(__double2hiint(w) ^ (int)flip, __double2loint(w));
    };
    auto body = [&](const Blob& B, int cur) {
      const uint32_t cx16 = B.g.cx16;
      const int nt = B.g.nt;
      if (cx16 == 0) {          // diagonal group (one per Hamiltonian shard: records fetched as they are needed)
        double2 a[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) a[k] = tile[tid + (uint32_t)k * kThreads];
        // D_k = sum_t v_t (-1)^{parity(k & class_t)}: one sum per class (terms stored class by class, counts in
        // u16), then the butterfly over the classes - per term one signed add instead of NE
        double d[NE];
        int tc = cur;
#pragma unroll
        for (int cls = 0; cls < NE; ++cls) {
          const int n_cls = (int)((B.g.u16[cls >> 1] >> (16 * (cls & 1))) & 0xffffu);
          double sum = 0.0;
          for (int t = 0; t < n_cls; ++t, ++tc) sum += signed_w(T[tc].wr, tid, T[tc].cz, T[tc].tz);
          d[cls] = sum;
        }
#pragma unroll
        for (int h = 1; h < NE; h <<= 1)
#pragma unroll
          for (int i = 0; i < NE; ++i)
            if (!(i & h)) { const double x = d[i], y = d[i | h]; d[i] = x + y; d[i | h] = x - y; }
#pragma unroll
        for (int k = 0; k < NE; ++k) acc = fma(a[k].x * a[k].x + a[k].y * a[k].y, d[k], acc);
        return;
      }
      if (B.g.im == 2u) {       // two equal-magnitude real terms: only the half space on which their sum does not vanish
        const int hb = (int)(B.g.hb & 0xffu), q = (int)((B.g.hb >> 8) & 0xffu);
        constexpr int NPH = NPR / 2;
        const uint32_t czd = B.g.u16[NPH];
        // pairs with parity(e & czd) == v carry D = 2 w1 s1; the others exactly zero
        const uint32_t v = ((B.g.hb >> 16) ^ (uint32_t)__builtin_popcount(pt & B.g.u16[NPH + 1])) & 1u;
        const int lo = q < hb ? q : hb, hi = q < hb ? hb : q;
        const uint32_t e0t = insert0(insert0(tid, lo), hi);
        double2 ha[NPH], hbv[NPH];
        uint32_t he[NPH];
#pragma unroll
        for (int k = 0; k < NPH; ++k) {
          const uint32_t e0 = k ? e0t | (B.g.u16[k] >> 4) : e0t;
          he[k] = e0 | ((((uint32_t)__builtin_popcount(e0 & czd) ^ v) & 1u) << q);
          hbv[k] = lds_load_d2(tile_b, he[k] << 4);
          ha[k] = lds_load_d2(tile_b, (he[k] << 4) ^ cx16);
        }
        const ETermRec& r = B.t[0];
#pragma unroll
        for (int k = 0; k < NPH; ++k)
          acc = fma(ha[k].x * hbv[k].x + ha[k].y * hbv[k].y, signed_w(r.wr, he[k], r.cz, r.tz), acc);
        return;
      }
      const uint32_t a0 = insert0(tid, (int)B.g.hb) << 4;
      // both members of the thread's pairs: requested before the sign sums, consumed after them
      double2 pa[NPR], pb[NPR];
#pragma unroll
      for (int k = 0; k < NPR; ++k) {
        const uint32_t off = k ? a0 ^ B.g.u16[k] : a0;
        pb[k] = lds_load_d2(tile_b, off);
        pa[k] = lds_load_d2(tile_b, off ^ cx16);
      }
      const uint32_t t0 = a0 >> 4;
      double dr[NPR];
#pragma unroll
      for (int k = 0; k < NPR; ++k) dr[k] = 0.0;
      auto term = [&](const ETermRec& r) {
        const double v = signed_w(r.wr, t0, r.cz, r.tz);
#pragma unroll
        for (int k = 0; k < NPR; ++k) dr[k] = fma(r.sg[k], v, dr[k]);
      };
#pragma unroll
      for (int i = 0; i < kBlobTerms; ++i) if (nt > i) term(B.t[i]);
      for (int t = kBlobTerms; t < nt; ++t) term(T[cur + t]);
#pragma unroll
      for (int k = 0; k < NPR; ++k) acc = fma(pa[k].x * pb[k].x + pa[k].y * pb[k].y, dr[k], acc);
      if (B.g.im & 1u) {       // imaginary parts of the weights (odd number of Y factors): rare
        double di[NPR];
#pragma unroll
        for (int k = 0; k < NPR; ++k) di[k] = 0.0;
        for (int t = 0; t < nt; ++t) {
          const double v = signed_w(WI[cur + t], t0, T[cur + t].cz, T[cur + t].tz);
#pragma unroll
          for (int k = 0; k < NPR; ++k) di[k] = fma(T[cur + t].sg[k], v, di[k]);
        }
#pragma unroll
        for (int k = 0; k < NPR; ++k) acc -= (pa[k].x * pb[k].y - pa[k].y * pb[k].x) * di[k];
      }
    };
    if (n_in_pass > 0) {
      Blob B0, B1;
      int c0 = 0, c1 = 0;
      fetch(B0, 0, 0);
      for (int gi = 0; gi < n_in_pass; gi += 2) {
        c1 = c0 + B0.g.nt;
        fetch(B1, gi + 1, c1);
        body(B0, c0);
        if (gi + 1 >= n_in_pass) break;
        c0 = c1 + B1.g.nt;
        fetch(B0, gi + 2, c0);
        body(B1, c1);
      }
    }
    }
}

// One pass of the circuit: stage the tile, apply the ops of the pass four at a time from registers, store.
// Registers of a tile on its way between HBM and LDS.  A plain array that stays live around the tile loop is
// left in scratch memory by the compiler (one scratch store per load, with a wait); members of a recursive
// struct are scalars from the start.
template <int N>
struct TileRegs {
  double2 v;
  TileRegs<N - 1> rest;
  template <class F> __device__ __forceinline__ void load(F f, int k = 0) { v = f(k); rest.load(f, k + 1); }
  template <class F> __device__ __forceinline__ void store(F f, int k = 0) const { f(k, v); rest.store(f, k + 1); }
};
template <>
struct TileRegs<0> {
  template <class F> __device__ __forceinline__ void load(F, int = 0) {}
  template <class F> __device__ __forceinline__ void store(F, int = 0) const {}
};

// (cos, sin) of every op's parameter in op order: k_t_ops reads them next to the chunk records, with no
// dependent load through the parameter index
__global__ void k_t_cs_ops(BatchArgs A, const Op* ops, const int32_t* meta, const double2* cs, double2* csop) {
  const int b = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= meta[(size_t)b * 8]) return;
  const Op op = ops[(size_t)b * A.max_ops + o];
  const int kd = op.kind & 0xff;
  csop[(size_t)b * A.max_ops + o] =
      (kd == OP_RX || kd == OP_RY || kd == OP_RZ) ? cs[(size_t)b * A.max_params + op.pidx] : make_double2(1.0, 0.0);
}

#ifndef VQE_OPS_TILES_PER_BLOCK
#define VQE_OPS_TILES_PER_BLOCK 4
#endif
constexpr int kOpsTilesPerBlock = VQE_OPS_TILES_PER_BLOCK;
// sign-flipped copy of s: bit 31 of `flipword` decides
__device__ __forceinline__ double t_flip(double s, uint32_t flipword) {
  return __hiloint2double(__double2hiint(s) ^ (int)(flipword & 0x80000000u), __double2loint(s));
}
// w[e] = c v[e] + s(e) v[e ^ F]: the per-element updates of s_apply_k (vqe_stream.h), same expressions
template <int E, int F, bool RX>
__device__ __forceinline__ void t_rot_pairs(double2 (&v)[E], double c, double s, uint32_t fw, uint32_t ebits) {
  double2 w[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const double2 a = v[e], bq = v[e ^ F];
    if (RX) w[e] = make_double2(c * a.x - s * bq.y, c * a.y + s * bq.x);
    else {
      const double sg = t_flip(s, fw ^ (ebits << (31 - e)));
      w[e] = make_double2(c * a.x + sg * bq.x, c * a.y + sg * bq.y);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = w[e];
}

// Fused pass (F.partial != nullptr): in the LAST pass of a stream the groups of energy pass 0 (whose tile this is,
// k_t_plan_energy) are evaluated on every tile of this rank's slice before the tile is stored; one partial per
// workgroup in F.partial[b * F.stride + F.offset + blockIdx.x].  FUSED: the instantiation that runs those last passes
// (the other one then leaves them alone): the reduction's registers would cost every circuit pass a workgroup per CU.
struct FusedEnergy {
  double* partial;
  const ETilePass* epasses;
  const EGroupRec* egrp;
  const ETermRec* eterm;
  const double* ewi;
  int n_terms, tiles_rank, stride, offset;
};
template <bool FUSED>
__global__ void __launch_bounds__(kThreads, FUSED ? 4 : 1) k_t_ops(BatchArgs A, double2* states, const ChunkRec* __restrict__ chunks,
                                                    const double2* __restrict__ csop, const TilePass* passes,
                                                    const int32_t* npass, int pass, int max_pass, int tiles, FusedEnergy F) {
  constexpr int K = kTileK, E = 1 << K;
  constexpr int NE = kTileAmps / kThreads;
  static_assert(E == 8 && NE == 8, "k_t_ops: three ops per chunk, eight amplitudes per thread");
  __shared__ double2 tile[kTileAmps];
  const int b = blockIdx.y;
  if (pass >= npass[b]) return;
  const TilePass P = passes[(size_t)b * max_pass + pass];
  uint32_t basis[kTileBits];
#pragma unroll
  for (int i = 0; i < kTileBits; ++i) basis[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.basis[i]);
  const uint32_t pivmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.pivmask);
  const size_t dim = (size_t)1 << A.n;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane_off = tile_lane_offset<kTileLow>(basis, tid);
  double2* psi = states + (size_t)b * dim;
  const double2* src = pass == 0 ? A.init : psi;      // the first pass starts from the shared initial state
  const int o_begin = __builtin_amdgcn_readfirstlane(P.begin), o_end = __builtin_amdgcn_readfirstlane(P.end);
  const ChunkRec* __restrict__ CH = chunks + (size_t)b * A.max_ops;
  const double2* __restrict__ CS = csop + (size_t)b * A.max_ops;
  lds_cbyte* tile_rb = (lds_cbyte*)tile;
  typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
  lds_byte_t* tile_wb = (lds_byte_t*)tile;
  // tiles blockIdx.x, blockIdx.x + gridDim.x, ... (see k_t_energy); the next one is in flight while this one is worked on
  const int t_first = (int)blockIdx.x, t_step = (int)gridDim.x, t_last = tiles;
  // (F.partial set: the streams in their last pass belong to the FUSED instantiation, the others to the plain one)
  if (F.partial != nullptr && (pass == npass[b] - 1) != FUSED) return;
  constexpr bool fused = FUSED;
  ETileArgs EA{nullptr, nullptr, nullptr, 0, 0};
  int slice_lo = 0, slice_hi = 0;
  if (fused) {
    const ETilePass& EP = F.epasses[(size_t)b * kMaxEnergyPasses];
    const int g_begin = __builtin_amdgcn_readfirstlane(EP.begin), g_end = __builtin_amdgcn_readfirstlane(EP.end);
    const int r_begin = __builtin_amdgcn_readfirstlane(EP.rec_begin);
    EA.G = F.egrp + (size_t)b * A.ham.n_groups + g_begin;
    EA.T = F.eterm + (size_t)b * F.n_terms + r_begin;
    EA.WI = F.ewi + (size_t)b * F.n_terms + r_begin;
    EA.n_in_pass = g_end - g_begin;
    EA.r_count = __builtin_amdgcn_readfirstlane(EP.rec_count);
    slice_lo = A.amp_rank * F.tiles_rank;
    slice_hi = slice_lo + F.tiles_rank;
  }
  double acc = 0.0;
  TileRegs<NE> stage;
  uint32_t p0 = tile_origin((uint32_t)t_first, pivmask, A.n);
  stage.load([&](int k) { return src[(p0 ^ lane_off) ^ tile_k_offset(basis, k)]; });
  for (int tl = t_first; tl < t_last; tl += t_step) {
    if (tl != t_first) __syncthreads();                 // the previous tile has left the LDS
    stage.store([&](int k, const double2& v) { tile[tid + (uint32_t)k * kThreads] = v; });
    __syncthreads();
    const uint32_t pt = p0;
    if (tl + t_step < t_last) {
      p0 = tile_origin((uint32_t)(tl + t_step), pivmask, A.n);
      stage.load([&](int k) { return src[(p0 ^ lane_off) ^ tile_k_offset(basis, k)]; });
    }
    for (int o = o_begin, cnt = 0; o < o_end; o += cnt) {
      const ChunkRec& cr = CH[o];
      const uint32_t pvc = cr.pvc;
      cnt = (int)(pvc >> 16);
      // the thread's coset: zeros inserted into tid at the pivot positions, then the 2^K combinations of the slot masks
      uint32_t t0 = tid;
#pragma unroll
      for (int i = 0; i < K; ++i) t0 = insert0(t0, (int)((pvc >> (5 * i)) & 31u));
      const uint32_t a0 = t0 << 4;
      uint32_t off[E];
      double2 v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) if ((e >> i) & 1) x ^= cr.g16[i];
        off[e] = a0 ^ x;
        v[e] = lds_load_d2(tile_rb, off[e]);
      }
#pragma unroll 1
      for (int j = 0; j < cnt; ++j) {
        const uint32_t kfe = cr.op[j].kfe;
        const int kind = (int)(kfe & 0xfu), flip = (int)((kfe >> 4) & 0xfu);
        const uint32_t ebits = kfe >> 16;
        const double2 c = CS[o + j];
        // sign of element e: parity(t0 & cz) [vector] ^ ebits[e] ^ parity(tile origin & zm) ^ inversion [scalars]
        const uint32_t fw = ((uint32_t)__builtin_popcount(t0 & cr.op[j].cz) +
                             ((uint32_t)__builtin_popcount(pt & cr.op[j].zm) ^ ((kfe >> 8) & 1u))) << 31;
        if (kind == OP_RX) {
          switch (flip) {
            case 1: t_rot_pairs<E, 1, true>(v, c.x, c.y, fw, ebits); break;
            case 2: t_rot_pairs<E, 2, true>(v, c.x, c.y, fw, ebits); break;
            case 3: t_rot_pairs<E, 3, true>(v, c.x, c.y, fw, ebits); break;
            case 4: t_rot_pairs<E, 4, true>(v, c.x, c.y, fw, ebits); break;
            case 5: t_rot_pairs<E, 5, true>(v, c.x, c.y, fw, ebits); break;
            case 6: t_rot_pairs<E, 6, true>(v, c.x, c.y, fw, ebits); break;
            default: t_rot_pairs<E, 7, true>(v, c.x, c.y, fw, ebits); break;
          }
        } else if (kind == OP_RY) {
          switch (flip) {
            case 1: t_rot_pairs<E, 1, false>(v, c.x, c.y, fw, ebits); break;
            case 2: t_rot_pairs<E, 2, false>(v, c.x, c.y, fw, ebits); break;
            case 3: t_rot_pairs<E, 3, false>(v, c.x, c.y, fw, ebits); break;
            case 4: t_rot_pairs<E, 4, false>(v, c.x, c.y, fw, ebits); break;
            case 5: t_rot_pairs<E, 5, false>(v, c.x, c.y, fw, ebits); break;
            case 6: t_rot_pairs<E, 6, false>(v, c.x, c.y, fw, ebits); break;
            default: t_rot_pairs<E, 7, false>(v, c.x, c.y, fw, ebits); break;
          }
        } else if (kind == OP_RZ) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const double sg = t_flip(c.y, fw ^ (ebits << (31 - e)));
            const double2 a = v[e];
            v[e] = make_double2(c.x * a.x - sg * a.y, c.x * a.y + sg * a.x);
          }
        } else if (kind == OP_PZ) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const uint32_t f = fw ^ (ebits << (31 - e));
            v[e] = make_double2(t_flip(v[e].x, f), t_flip(v[e].y, f));
          }
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        d2v_t q; q.x = v[e].x; q.y = v[e].y;
        *(__attribute__((address_space(3))) d2v_t*)(tile_wb + off[e]) = q;
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) psi[(pt ^ lane_off) ^ tile_k_offset(basis, k)] = tile[tid + (uint32_t)k * kThreads];
    // the final state of this tile, evaluated while the stores above drain (the tile is complete: the last chunk
    // ended with a barrier, a pass without ops staged it behind one)
    if (fused && tl >= slice_lo && tl < slice_hi) e_tile_groups(tile, tid, pt, EA, acc);
  }
  if (fused) {
    __syncthreads();            // all reads of the tile are done: its first words become the reduction scratch
    const double tot = block_sum(acc, (double*)tile);
    if (threadIdx.x == 0) F.partial[(size_t)b * F.stride + F.offset + blockIdx.x] = tot;
  }
}

// <psi|H_shard|psi>: every pass stages each tile once (read only) and evaluates all its X-mask groups from LDS.
// grid = (tiles of this rank's amplitude slice, batch, passes); partial: [batch][passes][tiles].
// Dynamic LDS: the tile (64 KiB) + the term records of the pass with the sign of the tile origin folded in
// (per-term scalars fetched from global memory inside the group loops cost one L2 round trip per term).
#ifndef VQE_TILES_PER_BLOCK
#define VQE_TILES_PER_BLOCK 4
#endif
constexpr int kTilesPerBlock = VQE_TILES_PER_BLOCK;   // tiles a workgroup of k_t_energy walks through (the next one in flight while it computes)
__global__ void __launch_bounds__(kThreads) k_t_energy(BatchArgs A, const double2* states, int n_terms, const ETilePass* passes,
                                                       const int32_t* npass, const EGroupRec* __restrict__ egrp,
                                                       const ETermRec* __restrict__ eterm, const double* __restrict__ ewi,
                                                       double* partial, int tiles_rank, int stride, int skip0) {
  __shared__ double2 tile[kETileAmps];
  double* red = (double*)tile;      // reused for the block reduction after the last read of the tile
  const int b = blockIdx.y, pass = blockIdx.z;
  const size_t slot = (size_t)b * stride + (size_t)pass * gridDim.x + blockIdx.x;
  if (pass >= npass[b] || (skip0 && pass == 0)) {      // (skip0: pass 0 was evaluated by the last circuit pass)
    if (threadIdx.x == 0) partial[slot] = 0.0;
    return;
  }
  const ETilePass P = passes[(size_t)b * kMaxEnergyPasses + pass];
  uint32_t basis[kETileBits];
#pragma unroll
  for (int i = 0; i < kETileBits; ++i) basis[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.basis[i]);
  const uint32_t pivmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.pivmask);
  const size_t dim = (size_t)1 << A.n;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane_off = tile_lane_offset<kETileLow>(basis, tid);
  const double2* psi = states + (size_t)b * dim;
  constexpr int NE = kETileAmps / kThreads;           // elements per thread
  // this workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ... of this rank's slice.  Neighbouring tiles
  // share their DRAM pages (a tile is 128 runs of 256 B spread over the state): workgroups that run side by side
  // take neighbouring tiles, as a one-tile-per-workgroup grid does
  const int t_first = (int)blockIdx.x, t_step = (int)gridDim.x, t_last = tiles_rank;
  TileRegs<NE> stage;                                 // the next tile, on its way from HBM
  uint32_t p0 = tile_origin((uint32_t)t_first + (uint32_t)A.amp_rank * (uint32_t)tiles_rank, pivmask, A.n);
  stage.load([&](int k) { return psi[(p0 ^ lane_off) ^ tile_k_offset(basis, k)]; });
  const int g_begin = __builtin_amdgcn_readfirstlane(P.begin), g_end = __builtin_amdgcn_readfirstlane(P.end);
  const int r_begin = __builtin_amdgcn_readfirstlane(P.rec_begin), r_count = __builtin_amdgcn_readfirstlane(P.rec_count);
  const EGroupRec* __restrict__ G = egrp + (size_t)b * A.ham.n_groups + g_begin;
  const ETermRec* __restrict__ T = eterm + (size_t)b * n_terms + r_begin;
  const double* __restrict__ WI = ewi + (size_t)b * n_terms + r_begin;
  const int n_in_pass = g_end - g_begin;
  double acc = 0.0;
  for (int tl = t_first; tl < t_last; tl += t_step) {
    if (tl != t_first) __syncthreads();                 // the previous tile has been read to the end
    stage.store([&](int k, const double2& v) { tile[tid + (uint32_t)k * kThreads] = v; });
    __syncthreads();
    const uint32_t pt = p0;                             // origin of the tile being evaluated
    if (tl + t_step < t_last) {                         // request the next tile: it lands while this one is evaluated
      p0 = tile_origin((uint32_t)(tl + t_step) + (uint32_t)A.amp_rank * (uint32_t)tiles_rank, pivmask, A.n);
      stage.load([&](int k) { return psi[(p0 ^ lane_off) ^ tile_k_offset(basis, k)]; });
    }
    e_tile_groups(tile, tid, pt, ETileArgs{G, T, WI, n_in_pass, r_count}, acc);
  }
  __syncthreads();            // all reads of the tile are done: its first words become the reduction scratch
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) partial[slot] = tot;
}

}  // namespace vqe
