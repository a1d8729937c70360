// CPU emulation of the DEVICE execution context of cobyla_m0.h (vqe_device.h: WaveCtx): 64 host threads
// play the 64 lanes of the one wavefront that runs the optimiser on the GPU, with the same row split
// (two lanes per row), zero padding, closed-form trust-region step and - lane for lane - the same
// reduction trees as the DPP / readlane code (wave_sum, arg_first, pair_sum).  Floating point is IEEE on
// both sides and the header forbids FMA contraction, so this build must reproduce the device's trial
// points BIT FOR BIT when it is told the device's function values.
//
// usage: cobyla_wave_emulation trace.txt [verbose]
//   trace.txt (tools/dump_cobyla_traces.py, hex floats):  "P nfev" / x0[P] / per evaluation: f x[P]
// For every evaluation k the program is told f_k and prints how far the next trial point of (a) this
// emulation, (b) the plain host context (HostCtx, the scipy-exact build) is from the device's.
// Exit code 0 iff the emulation matches the device bit for bit over the whole trace.
#include <pthread.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// decision log of cobyla_m0.h (compiled in only for this harness): argv[2] = "log" prints every branch
// decision of both contexts so that the step where they part can be read off
static bool cby_log_on = false;
static const char* cby_log_who = "";
#define CBY_LOG(...) do { if (ctx.tid == 0 && cby_log_on) { std::printf("      [%s] ", cby_log_who); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)
#include "cobyla_m0.h"

namespace {

constexpr int kLanes = 64;
pthread_barrier_t g_bar;
pthread_barrier_t g_pair[kLanes / 2];   // lanes l and l + 32 share a row: they exchange without the other lanes
double g_p[kLanes];
double g_d[kLanes];
int g_i[kLanes];

struct EmuWaveCtx {
  int tid = 0;
  static constexpr int nth = kLanes;
  static constexpr int kPad = 8;
  static constexpr bool kSplit = true;
  static constexpr bool kColumns = false;
  static constexpr bool kTile = false;   // (the device's LDS-tile walks are the plain row loops bit for bit: nothing to mirror)
  static constexpr bool kTileWalks = false;
  void sync() const { pthread_barrier_wait(&g_bar); }
  // value of v in lane `partner(tid)`
  template <class F>
  double xchg(double v, F partner) const {
    g_d[tid] = v;
    sync();
    const double o = g_d[partner(tid)];
    sync();
    return o;
  }
  template <class F>
  int xchg_i(int v, F partner) const {
    g_i[tid] = v;
    sync();
    const int o = g_i[partner(tid)];
    sync();
    return o;
  }
  static int p_b1(int t) { return t ^ 1; }                              // quad_perm [1,0,3,2]
  static int p_4e(int t) { return t ^ 2; }                              // quad_perm [2,3,0,1]
  static int p_141(int t) { return (t & ~7) | (7 - (t & 7)); }          // row_half_mirror
  static int p_140(int t) { return (t & ~15) | (15 - (t & 15)); }       // row_mirror
  double wave_sum(double v) const {
    v += xchg(v, p_b1);
    v += xchg(v, p_4e);
    v += xchg(v, p_141);
    v += xchg(v, p_140);
    g_d[tid] = v;
    sync();
    const double r = (g_d[0] + g_d[16]) + (g_d[32] + g_d[48]);
    sync();
    return r;
  }
  // (inside row loops only the lanes that own a row take part: a barrier of the two partner lanes)
  double pair_sum(double v) const {
    g_p[tid] = v;
    pthread_barrier_wait(&g_pair[tid & 31]);
    const double o = g_p[tid ^ 32];
    pthread_barrier_wait(&g_pair[tid & 31]);
    return v + o;
  }
  void lockstep() const { pthread_barrier_wait(&g_pair[tid & 31]); }
  int all_or(int v) const {
    g_i[tid] = v != 0;
    sync();
    int r = 0;
    for (int l = 0; l < kLanes; ++l) r |= g_i[l];
    sync();
    return r;
  }
  template <class F>
  double sum(int n, F f) const {
    double a = 0.0;
    for (int i = tid; i < n; i += kLanes) a += f(i);
    return wave_sum(a);
  }
  template <class F>
  int arg_first(int n, F f, double thresh, bool want_max, double* val) const {
    double best = thresh;
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += kLanes) {
      const double v = f(i);
      if (want_max ? (v > best) : (v < best)) { best = v; idx = i; }
    }
    auto merge = [&](double ob, int oi) {
      const bool better = want_max ? (ob > best) : (ob < best);
      if (better || (ob == best && oi < idx)) { best = ob; idx = oi; }
    };
    auto step = [&](int (*p)(int)) {
      const double ob = xchg(best, p);
      const int oi = xchg_i(idx, p);
      merge(ob, oi);
    };
    step(p_b1); step(p_4e); step(p_141); step(p_140);
    g_d[tid] = best; g_i[tid] = idx;
    sync();
    best = g_d[0]; idx = g_i[0];
    merge(g_d[16], g_i[16]); merge(g_d[32], g_i[32]); merge(g_d[48], g_i[48]);
    sync();
    *val = best;
    return idx == 0x7fffffff ? -1 : idx;
  }
};

struct Trace {
  int P = 0, nfev = 0;
  std::vector<double> x0, f;
  std::vector<std::vector<double>> x;
};

Trace read_trace(const char* path) {
  Trace t;
  FILE* fp = std::fopen(path, "r");
  if (!fp) { std::perror(path); std::exit(2); }
  if (std::fscanf(fp, "%d %d", &t.P, &t.nfev) != 2) std::exit(2);
  auto rd = [&]() { char buf[64]; if (std::fscanf(fp, "%63s", buf) != 1) std::exit(2); return std::strtod(buf, nullptr); };
  t.x0.resize(t.P);
  for (auto& v : t.x0) v = rd();
  t.f.resize(t.nfev);
  t.x.assign(t.nfev, std::vector<double>(t.P));
  for (int k = 0; k < t.nfev; ++k) { t.f[k] = rd(); for (auto& v : t.x[k]) v = rd(); }
  std::fclose(fp);
  return t;
}

// ---- emulated wave: 64 threads run start()/tell() in lock step on shared arrays ----------------
struct Shared {
  const Trace* tr;
  std::vector<double> mem;
  std::vector<std::vector<double>> xs;   // trial point of every evaluation (written by lane 0)
  int nfev = 0;
};

void* lane_main(void* arg) {
  auto* pr = (std::pair<Shared*, int>*)arg;
  Shared& S = *pr->first;
  cby::CobylaM0<EmuWaveCtx, false, double> c;
  c.ctx.tid = pr->second;
  const int P = S.tr->P;
  c.bind(S.mem.data(), P);
  if (c.ctx.tid == 0) for (int i = 0; i < P; ++i) c.x[i] = S.tr->x0[i];
  c.ctx.sync();
  int want = c.start(1.0, 1e-4, 1000);
  int k = 0;
  while (want && k < S.tr->nfev) {
    if (c.ctx.tid == 0) { S.xs.emplace_back(c.x, c.x + P); }
    c.ctx.sync();
    want = c.tell(S.tr->f[k]);
    ++k;
  }
  if (c.ctx.tid == 0) S.nfev = k;
  return nullptr;
}

double maxdiff(const std::vector<double>& a, const std::vector<double>& b) {
  double d = 0;
  for (size_t i = 0; i < a.size(); ++i) d = std::fmax(d, std::fabs(a[i] - b[i]));
  return d;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s trace.txt [verbose]\n", argv[0]); return 2; }
  const bool verbose = argc > 2;
  cby_log_on = argc > 2 && std::string(argv[2]) == "log";
  cby_log_who = "wave";
  const Trace tr = read_trace(argv[1]);
  // (a) emulated wave
  Shared S;
  S.tr = &tr;
  S.mem.assign(cby::scratch_doubles(tr.P, 8) + 16, 0.0);
  pthread_barrier_init(&g_bar, nullptr, kLanes);
  for (auto& b : g_pair) pthread_barrier_init(&b, nullptr, 2);
  std::vector<pthread_t> th(kLanes);
  std::vector<std::pair<Shared*, int>> args(kLanes);
  for (int l = 0; l < kLanes; ++l) { args[l] = {&S, l}; pthread_create(&th[l], nullptr, lane_main, &args[l]); }
  for (auto& t : th) pthread_join(t, nullptr);
  // (b) plain host context
  cby_log_who = "host";
  std::vector<std::vector<double>> hx;
  {
    cby::CobylaM0<cby::HostCtx, false> c;
    std::vector<double> mem(cby::scratch_doubles(tr.P, 1) + 8, 0.0);
    c.bind(mem.data(), tr.P);
    for (int i = 0; i < tr.P; ++i) c.x[i] = tr.x0[i];
    int want = c.start(1.0, 1e-4, 1000);
    int k = 0;
    while (want && k < tr.nfev) { hx.emplace_back(c.x, c.x + tr.P); want = c.tell(tr.f[k]); ++k; }
  }
  int first_emu = -1, first_host = -1;
  const int n = (int)std::min(S.xs.size(), (size_t)tr.nfev);
  for (int k = 0; k < n; ++k) {
    const double de = maxdiff(S.xs[k], tr.x[k]);
    const double dh = k < (int)hx.size() ? maxdiff(hx[k], tr.x[k]) : NAN;
    if (de != 0.0 && first_emu < 0) first_emu = k;
    if (!(dh <= 1e-9) && first_host < 0) first_host = k;
    if (verbose) std::printf("eval %4d  |x_emu - x_dev| %.3e   |x_host - x_dev| %.3e\n", k + 1, de, dh);
  }
  std::printf("%s: P=%d nfev=%d | emulated wave: %s", argv[1], tr.P, tr.nfev,
              first_emu < 0 ? "bit-identical to the device over the whole trace" : "DIFFERS");
  if (first_emu >= 0) std::printf(" from evaluation %d", first_emu + 1);
  std::printf(" | host context within 1e-9 of the device for the first %d evaluations\n", first_host < 0 ? n : first_host);
  return first_emu < 0 ? 0 : 1;
}
