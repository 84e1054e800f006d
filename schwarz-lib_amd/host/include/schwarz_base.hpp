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

    // ---- the public data members of the reference class (include/schwarz_base.hpp:137-197), as HOST copies:
    // the working data lives in HBM behind the C ABI and is never read through these.
    // local_matrix / interface_matrix (interface columns are GLOBAL ids, like the reference stores them),
    // and for the direct local solver triangular_factor_l / _u (A(perm, perm) = L L^T, U = L^T) with
    // local_perm / local_inv_perm: filled by initialize() for subdomains of up to 2^24 nonzeros
    // (SCHWZ_PUBLIC_MEMBERS=1: any size, =0: never), or on demand by materialize_public_members().
    std::shared_ptr<gko::matrix::Csr<ValueType, IndexType>> local_matrix;
    std::shared_ptr<gko::matrix::Permutation<IndexType>> local_perm;
    std::shared_ptr<gko::matrix::Permutation<IndexType>> local_inv_perm;
    std::shared_ptr<gko::matrix::Csr<ValueType, IndexType>> triangular_factor_l;
    std::shared_ptr<gko::matrix::Csr<ValueType, IndexType>> triangular_factor_u;
    std::shared_ptr<gko::matrix::Csr<ValueType, IndexType>> interface_matrix;
    // Deliberately ABSENT (always null): nothing of global length exists per rank in this build (the reference
    // replicates the N x N matrix and the rhs on every rank, schwarz_base.cpp:142-147,169).
    std::shared_ptr<gko::matrix::Csr<ValueType, IndexType>> global_matrix;
    std::shared_ptr<gko::matrix::Dense<ValueType>> global_rhs;
    // host copies of this rank's vectors: rhs after initialize(), the last local solution after run()
    std::shared_ptr<gko::matrix::Dense<ValueType>> local_rhs;
    std::shared_ptr<gko::matrix::Dense<ValueType>> local_solution;
    // the assembled solution on rank 0 after run() (the object run() hands back), null elsewhere
    std::shared_ptr<gko::matrix::Dense<ValueType>> global_solution;
    // residual histories of the last run (also in metadata.post_process_data)
    std::vector<ValueType> local_residual_vector_out;
    std::vector<std::vector<ValueType>> global_residual_vector_out;

    // Extension: fills the matrix members above from the library's host-side subdomain (after initialize()).
    void materialize_public_members();

protected:
    Settings &settings;
    Metadata<ValueType, IndexType> &metadata;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace schwz
