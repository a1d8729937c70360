// vqe_tile.h - LDS-tiled kernels of the streaming path (n >= 14; included by vqe_stream.h).
//
// One gate sweep over a 20-qubit state moves 32 MiB through HBM; a circuit of ~16 rotations and a
// Hamiltonian of ~20 X-mask groups would move it ~25 times if every op / group streamed the vector.
// Instead a workgroup stages a TILE of 2^11 amplitudes (32 KiB of LDS, four workgroups per CU: measured
// 31.5 k evaluations/s on the 20-qubit bench workload against 25.9 k with 64 KiB tiles and 28.4 k with
// 16 KiB ones - occupancy against passes) and does everything that closes inside it before the tile goes back:
//   * a tile is a coset p0 ^ V of an 11-dimensional subspace V of GF(2)^n that contains the unit vectors
//     e_0..e_3 (so the tile is made of aligned 256-byte runs: coalesced loads and stores) plus up to
//     seven independent pair masks - the physical partner masks A^-1 e_q of the rotations (vqe_device.h:
//     CNOTs never move data), or the physical X masks of Pauli groups; masks that DEPEND on the basis ride
//     along for free;
//   * a planner thread per stream cuts the op list into passes (maximal runs of ops whose masks fit one
//     V) and packs the X-mask groups into passes first-fit; V is kept as a fully reduced basis, so tile
//     coordinates of a mask are just its bits at the pivot positions, and the sign selector parity(p & z)
//     of a rotation / Pauli term splits into parity(p0 & z) (per tile, scalar) ^ parity(t & cz) with
//     cz_i = parity(basis_i & z);
//   * inside a tile three ops at a time are applied from registers (the coset trick of k_s_opk, in tile
//     coordinates: 256 threads x 8 amplitudes), so LDS sees one read + one write of the tile per three ops;
//   * the first pass of a stream reads the shared initial state instead of its own buffer (no separate
//     initialisation sweep).
// Every floating-point update is the per-element form of s_apply_k (vqe_stream.h), so states and
// energies are bit-identical to the one-sweep-per-four-ops kernels these replace up to the order in
// which the energy partials are summed.
#pragma once

namespace vqe {

#ifndef VQE_TILE_BITS
#define VQE_TILE_BITS 11
#endif
constexpr int kTileBits = VQE_TILE_BITS;
constexpr int kTileLow = 4;
constexpr int kTileAmps = 1 << kTileBits;
constexpr int kTileFree = kTileBits - kTileLow;   // independent masks a pass can take
constexpr int kMaxEnergyPasses = 32;

struct TilePass {
  uint32_t basis[kTileBits];  // fully reduced (every vector has a pivot = highest bit that no other vector has), ascending pivots:
                              // basis[i] = 1 << i below kTileLow, and tile coordinate i <-> basis[i]
  uint32_t pivmask;
  int32_t begin, end;         // ops [begin, end) of the stream / entries [begin, end) of its group order
  int32_t rec_begin, rec_count;   // energy passes: the term records of the pass (TermRec, pass order)
  uint32_t pad;
};
struct OpCoord { uint32_t cx, cz; };   // tile coordinates of an op's pair mask / sign mask
constexpr int kTileK = kTileBits - 8;   // ops applied per LDS round trip: 256 threads x 2^K amplitudes = one tile
// Coset bookkeeping of one chunk of kTileK ops (computed once by the planner; the tile kernel reads it with
// scalar loads): slot j of the basis is the partner mask of op j when that is independent of the earlier
// slots, else a filler unit vector; flip[j] = the op's partner as a combination of slots
struct ChunkRec { uint32_t g[4]; uint32_t flip; uint32_t pivots; uint32_t pad[2]; };   // 32 bytes, indexed by the chunk's first op
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
__global__ void k_t_plan_ops(BatchArgs A, const Op* ops, const int32_t* meta, TilePass* passes, OpCoord* opc,
                             ChunkRec* chunks, int32_t* npass, int max_pass) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.batch) return;
  const int nops = meta[(size_t)b * 8];
  const Op* op = ops + (size_t)b * A.max_ops;
  OpCoord* oc = opc + (size_t)b * A.max_ops;
  ChunkRec* ck = chunks + (size_t)b * A.max_ops;
  constexpr int K = kTileK, E = 1 << K;
  // the coset bookkeeping of k_s_opk for the ops [o, o + cnt) of a pass, in tile coordinates
  auto plan_chunk = [&](int o, int cnt) {
    uint32_t g[4] = {0u, 0u, 0u, 0u}, red[4] = {0u, 0u, 0u, 0u};
    int hbit[4] = {0, 0, 0, 0}, flip[4] = {0, 0, 0, 0};
    bool own[4] = {false, false, false, false};
    uint32_t pivots = 0;
    int nred = 0;
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
    for (int j = 0; j < K; ++j)
      if (j < cnt) {
        const int kd = op[o + j].kind & 0xff;
        if (kd == OP_RX || kd == OP_RY) {
          const uint32_t r = reduce(oc[o + j].cx);
          if (r) { push(r); g[j] = oc[o + j].cx; own[j] = true; flip[j] = 1 << j; }
        }
      }
    int q = 0;
    for (int j = 0; j < K; ++j)
      if (!own[j]) {
        uint32_t r = 0;
        while ((r = reduce(1u << q)) == 0) ++q;
        push(r);
        g[j] = 1u << q;
        ++q;
      }
    for (int j = 0; j < K; ++j)
      if (j < cnt && !own[j]) {
        const int kd = op[o + j].kind & 0xff;
        if (kd == OP_RX || kd == OP_RY)
          for (int f = 1; f < E; ++f) {
            uint32_t x = 0;
            for (int i = 0; i < K; ++i) if ((f >> i) & 1) x ^= g[i];
            if (x == oc[o + j].cx) flip[j] = f;
          }
      }
    ChunkRec c;
    for (int j = 0; j < 4; ++j) c.g[j] = g[j];
    c.flip = (uint32_t)flip[0] | ((uint32_t)flip[1] << 8) | ((uint32_t)flip[2] << 16) | ((uint32_t)flip[3] << 24);
    c.pivots = pivots; c.pad[0] = c.pad[1] = 0u;
    ck[o] = c;
  };
  TilePass* P = passes + (size_t)b * max_pass;
  TileBasis B;
  B.reset();
  int begin = 0, np = 0;
  auto close = [&](int end) {
    B.fill(A.n);
    B.sort();
    if (np < max_pass) {
      B.emit(P + np, begin, end);
      for (int o = begin; o < end; ++o) {
        const int kd = op[o].kind & 0xff;
        oc[o].cx = (kd == OP_RX || kd == OP_RY) ? B.coords(op[o].xm) : 0u;
        oc[o].cz = B.zcoords(op[o].zm);
      }
      for (int o = begin; o < end; o += K) plan_chunk(o, end - o < K ? end - o : K);
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
  close(nops);                                           // (a stream without ops still gets its copy pass)
  npass[b] = np < max_pass ? np : max_pass;
}

// X-mask groups of every stream -> passes (first fit) + tile coordinates of the groups + the term records of
// every pass in the order the tile kernel consumes them.
// order: [batch][n_groups] group ids pass by pass; gcx, grec: [batch][n_groups]; trec: [batch][n_terms]
__global__ void k_t_plan_energy(BatchArgs A, int n_terms, const uint32_t* gxp, const uint32_t* tzp, const double* tsg,
                                TilePass* passes, int32_t* npass, int32_t* order, uint32_t* gcx, int32_t* grec,
                                TermRec* trec, int32_t* gpass) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.batch) return;
  const int ng = A.ham.n_groups;
  const uint32_t* gx = gxp + (size_t)b * ng;
  int32_t* gp = gpass + (size_t)b * ng;
  TilePass* P = passes + (size_t)b * kMaxEnergyPasses;
  // pass bases live in the output records while they are built (fully reduced at every moment)
  int np = 0;
  int dims[kMaxEnergyPasses];
  int piv[kMaxEnergyPasses][kTileBits];
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
    for (int i = 0; i < kTileLow; ++i) { P[np].basis[i] = 1u << i; piv[np][i] = i; }
    dims[np] = kTileLow;
    return np++;
  };
  for (int g = 0; g < ng; ++g) {
    const uint32_t x = gx[g];
    int dst = -1;
    for (int k = 0; k < np && dst < 0; ++k) {
      const uint32_t r = reduce(k, x);
      if (!r) dst = k;
      else if (dims[k] < kTileBits) { add(k, r); dst = k; }
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
    for (int q = kTileLow; q < A.n && dims[k] < kTileBits; ++q) { const uint32_t r = reduce(k, 1u << q); if (r) add(k, r); }
    for (int i = 1; i < kTileBits; ++i)
      for (int j = i; j > 0 && piv[k][j - 1] > piv[k][j]; --j) {
        const uint32_t tv = P[k].basis[j]; P[k].basis[j] = P[k].basis[j - 1]; P[k].basis[j - 1] = tv;
        const int tp = piv[k][j]; piv[k][j] = piv[k][j - 1]; piv[k][j - 1] = tp;
      }
    uint32_t pm = 0;
    for (int i = 0; i < kTileBits; ++i) pm |= 1u << piv[k][i];
    P[k].pivmask = pm; P[k].begin = pos; P[k].rec_begin = rpos; P[k].pad = 0;
    for (int g = 0; g < ng; ++g)
      if (gp[g] == k) {
        ord[pos++] = g;
        uint32_t c = 0;
        for (int i = 0; i < kTileBits; ++i) c |= ((gx[g] >> piv[k][i]) & 1u) << i;
        gcx[(size_t)b * ng + g] = c;
        grec[(size_t)b * ng + g] = rpos - P[k].rec_begin;
        for (int t = A.ham.term_off[g]; t < A.ham.term_off[g + 1]; ++t) {
          const uint32_t z = tzp[(size_t)b * n_terms + t];
          uint32_t cz = 0;
          for (int i = 0; i < kTileBits; ++i) cz |= (uint32_t)parity32(P[k].basis[i] & z) << i;
          const double sg = tsg[(size_t)b * n_terms + t];
          rec[rpos++] = TermRec{sg * A.ham.term_cr[t], sg * A.ham.term_ci[t], z, cz};
        }
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
__device__ __forceinline__ uint32_t tile_lane_offset(const uint32_t (&basis)[kTileBits], uint32_t tid) {
  uint32_t x = tid & ((1u << kTileLow) - 1u);
#pragma unroll
  for (int i = 0; i < 4; ++i) if ((tid >> (kTileLow + i)) & 1u) x ^= basis[kTileLow + i];
  return x;
}
__device__ __forceinline__ uint32_t tile_k_offset(const uint32_t (&basis)[kTileBits], int k) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < kTileBits - kTileLow - 4; ++i) if ((k >> i) & 1) x ^= basis[kTileLow + 4 + i];
  return x;
}

// One pass of the circuit: stage the tile, apply the ops of the pass four at a time from registers, store.
__global__ void __launch_bounds__(kThreads) k_t_ops(BatchArgs A, double2* states, const Op* ops, const OpCoord* opc,
                                                    const ChunkRec* chunks, const double2* cs, const TilePass* passes,
                                                    const int32_t* npass, int pass, int max_pass) {
  constexpr int K = kTileK, E = 1 << K;
  __shared__ double2 tile[kTileAmps];
  const int b = blockIdx.y;
  if (pass >= npass[b]) return;
  const TilePass P = passes[(size_t)b * max_pass + pass];
  uint32_t basis[kTileBits];
#pragma unroll
  for (int i = 0; i < kTileBits; ++i) basis[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.basis[i]);
  const uint32_t pivmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.pivmask);
  const size_t dim = (size_t)1 << A.n;
  const uint32_t p0 = tile_origin(blockIdx.x, pivmask, A.n);
  const uint32_t tid = threadIdx.x;
  const uint32_t base = p0 ^ tile_lane_offset(basis, tid);
  double2* psi = states + (size_t)b * dim;
  const double2* src = pass == 0 ? A.init : psi;      // the first pass starts from the shared initial state
#pragma unroll
  for (int k = 0; k < kTileAmps / kThreads; ++k) tile[tid + (uint32_t)k * kThreads] = src[base ^ tile_k_offset(basis, k)];
  __syncthreads();
  const int o_begin = __builtin_amdgcn_readfirstlane(P.begin), o_end = __builtin_amdgcn_readfirstlane(P.end);
  const Op* sop = ops + (size_t)b * A.max_ops;
  const OpCoord* soc = opc + (size_t)b * A.max_ops;
  const ChunkRec* sck = chunks + (size_t)b * A.max_ops;
  const double2* csb = cs + (size_t)b * A.max_params;
  for (int o = o_begin; o < o_end; o += K) {
    const int cnt = o_end - o < K ? o_end - o : K;
    // coset bookkeeping of the chunk: made by the planner, wave-uniform here (scalar registers)
    const ChunkRec cr = sck[o];
    uint32_t g[K];
    int flip[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      g[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)cr.g[j]);
      flip[j] = (__builtin_amdgcn_readfirstlane((int)cr.flip) >> (8 * j)) & 0xff;
    }
    const uint32_t pivots = (uint32_t)__builtin_amdgcn_readfirstlane((int)cr.pivots);
    Op op[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      op[j] = Op{0u, 0u, 0, OP_NOP};
      if (j < cnt) {
        const Op raw = sop[o + j];
        const OpCoord c = soc[o + j];
        // sign selector of the op in tile coordinates; the part of parity(p & zm) that comes from the tile
        // origin is the same for the whole tile and goes into the op's inversion bit
        const int kd = __builtin_amdgcn_readfirstlane(raw.kind);
        const int inv = ((kd >> 8) & 1) ^ parity32(p0 & (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.zm));
        op[j] = Op{(uint32_t)__builtin_amdgcn_readfirstlane((int)c.cx), (uint32_t)__builtin_amdgcn_readfirstlane((int)c.cz),
                   __builtin_amdgcn_readfirstlane(raw.pidx), (kd & 0xff) | (inv << 8)};
      }
    }
    uint32_t t0 = tid;
    for (int q = 0; q < kTileBits; ++q) if ((pivots >> q) & 1u) t0 = insert0(t0, q);
    uint32_t idx[E];
    double2 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint32_t x = t0;
#pragma unroll
      for (int i = 0; i < K; ++i) if ((e >> i) & 1) x ^= g[i];
      idx[e] = x;
      v[e] = tile[x];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) if (j < cnt) s_apply_k<K>(v, idx, op[j], csb, j, flip[j]);
#pragma unroll
    for (int e = 0; e < E; ++e) tile[idx[e]] = v[e];
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kTileAmps / kThreads; ++k) psi[base ^ tile_k_offset(basis, k)] = tile[tid + (uint32_t)k * kThreads];
}

// <psi|H_shard|psi>: every pass stages each tile once (read only) and evaluates all its X-mask groups from LDS.
// grid = (tiles of this rank's amplitude slice, batch, passes); partial: [batch][passes][tiles].
// Dynamic LDS: the tile (64 KiB) + the term records of the pass with the sign of the tile origin folded in
// (per-term scalars fetched from global memory inside the group loops cost one L2 round trip per term).
struct TermLds { double wr, wi; uint32_t cz, pad; };
__global__ void __launch_bounds__(kThreads) k_t_energy(BatchArgs A, const double2* states, int n_terms, const TilePass* passes,
                                                       const int32_t* npass, const int32_t* order, const uint32_t* gcx,
                                                       const int32_t* grec, const TermRec* trec, double* partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  double2* tile = (double2*)tsm;
  TermLds* lrec = (TermLds*)(tsm + sizeof(double2) * kTileAmps);
  double* red = (double*)tile;      // reused for the block reduction after the last read of the tile
  const int b = blockIdx.y, pass = blockIdx.z;
  const size_t slot = ((size_t)b * gridDim.z + pass) * gridDim.x + blockIdx.x;
  if (pass >= npass[b]) {
    if (threadIdx.x == 0) partial[slot] = 0.0;
    return;
  }
  const TilePass P = passes[(size_t)b * kMaxEnergyPasses + pass];
  uint32_t basis[kTileBits];
#pragma unroll
  for (int i = 0; i < kTileBits; ++i) basis[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.basis[i]);
  const uint32_t pivmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.pivmask);
  const size_t dim = (size_t)1 << A.n;
  const uint32_t tile_id = blockIdx.x + (uint32_t)A.amp_rank * gridDim.x;     // this rank's slice of the tiles
  const uint32_t p0 = tile_origin(tile_id, pivmask, A.n);
  const uint32_t tid = threadIdx.x;
  const uint32_t base = p0 ^ tile_lane_offset(basis, tid);
  const double2* psi = states + (size_t)b * dim;
#pragma unroll
  for (int k = 0; k < kTileAmps / kThreads; ++k) tile[tid + (uint32_t)k * kThreads] = psi[base ^ tile_k_offset(basis, k)];
  const int nrec = __builtin_amdgcn_readfirstlane(P.rec_count);
  const TermRec* grecs = trec + (size_t)b * n_terms + __builtin_amdgcn_readfirstlane(P.rec_begin);
  for (int i = tid; i < nrec; i += kThreads) {
    const TermRec r = grecs[i];
    const bool neg = parity32(p0 & r.tz);
    lrec[i] = TermLds{neg ? -r.wr : r.wr, neg ? -r.wi : r.wi, r.cz, 0u};
  }
  __syncthreads();
  const int ng = A.ham.n_groups;
  const int32_t* ord = order + (size_t)b * ng;
  const uint32_t* cxs = gcx + (size_t)b * ng;
  const int32_t* grs = grec + (size_t)b * ng;
  const int g_begin = __builtin_amdgcn_readfirstlane(P.begin), g_end = __builtin_amdgcn_readfirstlane(P.end);
  double acc = 0.0;
  for (int gi = g_begin; gi < g_end; ++gi) {
    const int g = __builtin_amdgcn_readfirstlane(ord[gi]);
    const uint32_t cx = (uint32_t)__builtin_amdgcn_readfirstlane((int)cxs[g]);
    const int nt = A.ham.term_off[g + 1] - A.ham.term_off[g];
    const TermLds* lr = lrec + __builtin_amdgcn_readfirstlane(grs[g]);
    // Sign sums D(t) = sum_terms w (-1)^{parity(t & cz)}.  The tile elements (pair representatives) of a
    // thread are t_k = t_0 ^ U_k with U_k THE SAME FOR EVERY THREAD (k only moves index bits above the
    // thread id), so parity(t_k & cz) = parity(t_0 & cz) ^ parity(U_k & cz): one vector parity per term,
    // the k-dependence is a scalar +-1 folded into an FMA.
    if (cx == 0) {          // diagonal group
      constexpr int NE = kTileAmps / kThreads;
      double d[NE];
#pragma unroll
      for (int k = 0; k < NE; ++k) d[k] = 0.0;
      for (int t = 0; t < nt; ++t) {
        const TermLds r = lr[t];
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.cz);
        const double w = r.wr;
        const double v = parity32(tid & c) ? -w : w;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
          const double sg = (__builtin_popcount(((uint32_t)k * kThreads) & c) & 1) ? -1.0 : 1.0;   // uniform
          d[k] = fma(sg, v, d[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const double2 a = tile[tid + (uint32_t)k * kThreads];
        acc += (a.x * a.x + a.y * a.y) * d[k];
      }
    } else {
      const int hb = 31 - __clz((int)cx);
      constexpr int NPR = kTileAmps / 2 / kThreads;      // pairs per thread
      const uint32_t tr0 = insert0(tid, hb);
      double dr[NPR], di[NPR];
#pragma unroll
      for (int k = 0; k < NPR; ++k) { dr[k] = 0.0; di[k] = 0.0; }
      for (int t = 0; t < nt; ++t) {
        const TermLds r = lr[t];
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.cz);
        const double wr = r.wr, wi = r.wi;
        const bool neg = parity32(tr0 & c);
        const double vr = neg ? -wr : wr, vi = neg ? -wi : wi;
#pragma unroll
        for (int k = 0; k < NPR; ++k) {
          const double sg = (__builtin_popcount(insert0((uint32_t)k * kThreads, hb) & c) & 1) ? -1.0 : 1.0;   // uniform
          dr[k] = fma(sg, vr, dr[k]);
          di[k] = fma(sg, vi, di[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < NPR; ++k) {
        const uint32_t tr = tr0 ^ insert0((uint32_t)k * kThreads, hb);
        const double2 bb = tile[tr], a = tile[tr ^ cx];
        acc += 2.0 * ((a.x * bb.x + a.y * bb.y) * dr[k] - (a.x * bb.y - a.y * bb.x) * di[k]);
      }
    }
  }
  __syncthreads();            // all reads of the tile are done: its first words become the reduction scratch
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) partial[slot] = tot;
}

}  // namespace vqe
