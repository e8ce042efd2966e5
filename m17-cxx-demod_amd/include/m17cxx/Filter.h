// Operator surface of the reference (include/m17cxx/Filter.h:8-12): FilterBase<T>, the abstract one-sample-in,
// one-sample-out stage every scalar filter of the chain derives from.  Here it is a name for detail::SampleStage so the
// batched (device) filters can share the same root without inheriting a per-sample virtual call.
#pragma once

namespace mobilinkd
{

namespace detail
{

template <typename T>
class SampleStage
{
public:
    virtual ~SampleStage() = default;
    virtual T operator()(T input) = 0;      // consume one sample, produce one sample
};

} // detail

template <typename NumericType> using FilterBase = detail::SampleStage<NumericType>;

} // mobilinkd
