// Operator surface of the reference (include/m17cxx/Filter.h:8-12): the abstract one-sample-in, one-sample-out filter.
#pragma once

namespace mobilinkd
{

template <typename NumericType>
struct FilterBase
{
    virtual NumericType operator()(NumericType input) = 0;
    virtual ~FilterBase() = default;
};

} // mobilinkd
