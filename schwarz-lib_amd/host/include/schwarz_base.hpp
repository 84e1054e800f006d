// schwz::SchwarzBase -- the solver base class the driver talks to
// (reference: include/schwarz_base.hpp:74-221, source/schwarz_base.cpp).  Same constructor,
// initialize() and run(solution) as the reference; the work is done by libschwz_hip.so
// (include/schwz_hip.h) on the GPU the rank is bound to.
#pragma once

#include <omp.h>

#include <memory>
#include <vector>

#include <schwarz/config.hpp>

#include <exception.hpp>
#include <settings.hpp>

namespace schwz {

template <typename ValueType = gko::default_precision, typename IndexType = gko::int32,
          typename MixedValueType = gko::default_precision>
class SchwarzBase {
public:
    SchwarzBase(Settings &settings, Metadata<ValueType, IndexType> &metadata);
    virtual ~SchwarzBase();
    SchwarzBase(const SchwarzBase &) = delete;
    SchwarzBase &operator=(const SchwarzBase &) = delete;

    // reads / generates the system, partitions it, builds this rank's subdomain, exchanges the
    // halo index lists and uploads everything (schwarz_base.cpp:128-271)
    void initialize();

    // Extension: the global system handed over as host CSR arrays (+ right-hand side, ones when
    // null), the same on every rank -- what the reference's deal.II overload does with
    // dealii::SparseMatrix / Vector (include/schwarz_base.hpp:96-97, source/schwarz_base.cpp:128-160)
    // without the deal.II types.  Rows must have ascending columns.  With a permuting partition
    // (regular2d / metis) rows are renumbered like for a matrix read from file;
    // metadata.permutation maps new to old.
    void initialize(IndexType num_rows, const IndexType *row_ptrs, const IndexType *col_idxs,
                    const ValueType *values, const ValueType *rhs = nullptr);

    // the outer RAS loop (schwarz_base.cpp:323-506); `solution` is allocated if null and filled
    // on rank 0
    void run(std::shared_ptr<gko::matrix::Dense<ValueType>> &solution);

    // host copies of this rank's vectors (public members of the reference class)
    std::shared_ptr<gko::matrix::Dense<ValueType>> local_rhs;
    std::shared_ptr<gko::matrix::Dense<ValueType>> local_solution;

protected:
    Settings &settings;
    Metadata<ValueType, IndexType> &metadata;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace schwz
