// tests/cxx/bench_threshold.cc -- where a HOST-resident scl::math::Vector is better worked on: FF's operators on the host or
// upload + kernel + download (hip::hostThreshold(), include/scl_hip/hip/device.h).  For each element type and size: one
// multiplyEntryWise and one dot on the host (threshold above the size) and through the kernels (threshold 0), microseconds
// per call (best of several).  Needs a GPU.  The default threshold -- 2^20 limb products = elements x limbs^2 -- is read
// off this table (profiles/r3_host_threshold.txt).
#include <chrono>
#include <cstdio>
#include <vector>

#include <scl_hip/scl.h>

using namespace scl;

template <typename F>
static void sweep(const char* name) {
  using clk = std::chrono::steady_clock;
  auto prg = util::PRG::create("threshold");
  std::printf("%s (limbs %zu)\n  %10s %14s %14s %14s %14s\n", name, hip::limbsOf<F>(), "elements", "mul host us", "mul device us", "dot host us",
              "dot device us");
  for (std::size_t n = 64; n <= (1u << 20); n *= 4) {
    hip::setHostThreshold(~std::size_t(0) >> 8);
    const auto a = math::Vector<F>::random(n, prg), b = math::Vector<F>::random(n, prg);
    double t[4] = {1e30, 1e30, 1e30, 1e30};
    for (int mode = 0; mode < 2; ++mode) {
      hip::setHostThreshold(mode ? 0 : (~std::size_t(0) >> 8));
      for (int rep = 0; rep < 5; ++rep) {
        auto t0 = clk::now();
        const auto c = a.multiplyEntryWise(b);
        auto t1 = clk::now();
        const F d = a.dot(b);
        auto t2 = clk::now();
        if (c.size() != n || (d == F(0) && n == 0)) std::printf("?");
        t[mode] = std::min(t[mode], 1e6 * std::chrono::duration<double>(t1 - t0).count());
        t[2 + mode] = std::min(t[2 + mode], 1e6 * std::chrono::duration<double>(t2 - t1).count());
      }
    }
    std::printf("  %10zu %14.1f %14.1f %14.1f %14.1f   %s\n", n, t[0], t[1], t[2], t[3],
                n * hip::limbsOf<F>() * hip::limbsOf<F>() < (std::size_t(1) << 20) ? "host (default threshold)" : "device (default threshold)");
  }
}

int main() {
  sweep<math::Fp<61>>("Mersenne61");
  sweep<math::Fp<127>>("Mersenne127");
  sweep<math::FF<math::ff::Secp256k1Scalar>>("secp256k1_order");
  return 0;
}
