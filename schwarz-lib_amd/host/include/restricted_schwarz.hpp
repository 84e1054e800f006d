// schwz::SolverRAS -- Restricted Additive Schwarz (reference: include/restricted_schwarz.hpp:61-107).
#pragma once

#include <schwarz_base.hpp>

namespace schwz {

template <typename ValueType = gko::default_precision, typename IndexType = gko::int32,
          typename MixedValueType = gko::default_precision>
class SolverRAS : public SchwarzBase<ValueType, IndexType, MixedValueType> {
public:
    SolverRAS(Settings &settings, Metadata<ValueType, IndexType> &metadata)
        : SchwarzBase<ValueType, IndexType, MixedValueType>(settings, metadata)
    {}
};

}  // namespace schwz
