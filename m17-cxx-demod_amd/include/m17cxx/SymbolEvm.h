// mobilinkd::SymbolEvm — the reference's error-vector-magnitude meter (include/m17cxx/SymbolEvm.h:10-52): the distance of every
// normalised symbol to the nearest of {-3, -1, +1, +3} (core::evm_error), exponentially averaged over ~184 symbols.
#pragma once

#include "StandardDeviation.h"
#include "detail/core.h"

namespace mobilinkd
{

template <typename FloatType>
struct SymbolEvm
{
    RunningStandardDeviation<FloatType, 184> stddev;

    void reset() { stddev.reset(); }
    FloatType evm() const { return stddev.stdev(); }
    void update(FloatType sample) { stddev.capture(core::evm_error(float(sample))); }
};

} // mobilinkd
