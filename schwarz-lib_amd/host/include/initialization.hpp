// The reference splits SchwarzBase into Initialize / Communicate / Solve mix-ins
// (include/initialization.hpp); here their work happens behind the C ABI, the header only has to exist for
// drivers that include it (benchmarking/bench_base.hpp:45-48).
#pragma once
#include <schwarz_base.hpp>
