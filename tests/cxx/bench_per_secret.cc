// tests/cxx/bench_per_secret.cc -- what a caller that keeps the reference's loop-over-secrets shape pays per call:
//   for each secret:  shares = scl::ss::shamirSecretShare(secret, t, n, prg);  secret' = scl::ss::shamirRecoverP(shares);
// (include/scl/ss/shamir.h:51-68,99-104; the reference spends ~0.5 us + ~1.4 us per (10,3) Mersenne61 call).
// The mirror runs these signatures on the host over FF's operators (detail/field.hpp) and the host AES (detail/aes_host.hpp).
//   bench_per_secret [count=100000] [n=10] [t=3] [--device]
// --device sets hip::setHostThreshold(0): Vector::random and innerProd inside the calls then go through their kernels --
// a device round trip per call, what round 2's mirror did -- for the comparison only (needs a GPU).
// Prints one line:  per_secret mode=<host|device> count=.. share_ns=.. recover_ns=.. mismatches=..
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <scl_hip/scl.h>

using namespace scl;
using F = math::Fp<61>;

int main(int argc, char** argv) {
  std::size_t count = 100000, n = 10, t = 3;
  bool device = false;
  int pos = 0;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--device")) device = true;
    else if (pos == 0) count = std::strtoull(argv[i], nullptr, 10), ++pos;
    else if (pos == 1) n = std::strtoull(argv[i], nullptr, 10), ++pos;
    else if (pos == 2) t = std::strtoull(argv[i], nullptr, 10), ++pos;
  }
  if (device) hip::setHostThreshold(0);
  using clk = std::chrono::steady_clock;
  auto prg = util::PRG::create("scl-bench");
  double ts = 0, tr = 0;
  std::size_t bad = 0;
  constexpr std::size_t CH = 4096;  // share a chunk, then recover it: the two halves are timed apart, as oracle/_ref does
  std::vector<math::Vector<F>> held;
  held.reserve(CH);
  for (std::size_t s0 = 0; s0 < count; s0 += CH) {
    const std::size_t c = std::min(CH, count - s0);
    held.clear();
    const auto a = clk::now();
    for (std::size_t i = 0; i < c; ++i) held.emplace_back(ss::shamirSecretShare(F((int)((s0 + i) & 0x7fffffff)), t, n, prg));
    const auto b = clk::now();
    for (std::size_t i = 0; i < c; ++i) bad += !(ss::shamirRecoverP(held[i]) == F((int)((s0 + i) & 0x7fffffff)));
    const auto e = clk::now();
    ts += std::chrono::duration<double>(b - a).count();
    tr += std::chrono::duration<double>(e - b).count();
  }
  std::printf("per_secret mode=%s count=%zu n=%zu t=%zu share_ns=%.1f recover_ns=%.1f mismatches=%zu\n", device ? "device" : "host", count,
              n, t, 1e9 * ts / (double)count, 1e9 * tr / (double)count, bad);
  return bad ? 1 : 0;
}
