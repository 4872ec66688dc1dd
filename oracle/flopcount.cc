// flopcount.cc -- totals of the operation-counting build of the oracle (flopcount.h). TEST INFRASTRUCTURE ONLY.
#include "flopcount.h"
#undef double
namespace oflops {
thread_local Tally t_tally;
std::atomic<unsigned long long> g_total[5];
}
// totals since the last reset: [add/sub, mul, div, sqrt, other libm]; worker threads fold their tallies in when they exit
extern "C" void oracle_flops_read(unsigned long long* out5) {
  oflops::flush();
  for (int i = 0; i < 5; i++) out5[i] = oflops::g_total[i].load();
}
extern "C" void oracle_flops_reset(void) {
  oflops::flush();
  for (int i = 0; i < 5; i++) oflops::g_total[i] = 0;
}
