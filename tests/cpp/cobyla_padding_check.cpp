// Host check of the zero-padded (batched inner loop) variant of cobyla_m0.h that the device
// context uses: with a serial context it must reproduce the unpadded run bit for bit.
#include <cstdio>
#include <vector>
#include <cmath>
#include "cobyla_m0.h"
struct PadCtx : cby::HostCtx { static constexpr int kPad = 8; };
static double fobj(const double* x, int n) { double s = 0; for (int i = 0; i < n; ++i) s += std::cos(x[i] * (1 + 0.1 * i)) + 0.3 * std::sin(x[i] * x[(i + 1) % n]); return s; }
template <class C> int run(int n, std::vector<double>& xs, double& fout) {
  cby::CobylaM0<C, false> c;
  std::vector<double> mem(cby::scratch_doubles(n, 8) + 8, 0.0);
  c.bind(mem.data(), n);
  for (int i = 0; i < n; ++i) c.x[i] = 0.3 * i - 1.0;
  int want = c.start(1.0, 1e-4, 1000);
  double f = 0;
  while (want) { f = fobj(c.x, n); want = c.tell(f); }
  xs.assign(c.x, c.x + n); fout = f;
  return c.nfvals;
}
int main() {
  for (int n : {3, 7, 8, 13, 15, 16, 21}) {
    std::vector<double> a, b; double fa, fb;
    int na = run<cby::HostCtx>(n, a, fa), nb = run<PadCtx>(n, b, fb);
    double d = 0; for (int i = 0; i < n; ++i) d = std::fmax(d, std::fabs(a[i] - b[i]));
    std::printf("n=%d nfev %d %d  f %.15g %.15g  max|dx| %.3g\n", n, na, nb, fa, fb, d);
    if (na != nb || fa != fb || d != 0.0) return 1;
  }
  return 0;
}
