// Test driver for the template instantiations of the C++ mirror that the reference's bench_ras
// does not reach (it fixes <double, int, double>): MixedValueType = float with
// settings.use_mixed_precision (fp32 halos on the wire, restricted_schwarz.cpp:898-903,929-933,
// 952-954) and IndexType = int64.  Usage under mpiexec:
//   ras_types_driver <types: d32d|d32f|d64d|d64f> <grid edge n> <mixed 0|1> <overlapped 0|1> <tol> <max_iters> [csr]
// Prints what SolverRAS::run prints plus one line "RESULT iters=<k> solnorm=<|x|_2>" on rank 0.
// With a seventh argument "csr" the 2-D Laplacian is assembled here and handed over through
// initialize(num_rows, row_ptrs, col_idxs, values, rhs) with rhs_i = 1 + (i mod 7) -- the
// deal.II-free analogue of the reference's initialize(dealii::SparseMatrix, dealii::Vector).
// With "members" instead the local solver is the direct one and every rank checks the public data members of
// the solver class (include/schwarz_base.hpp:137-197) after the run: one line "MEMBERS rank=.. ..." per rank.
#include <mpi.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>

#include <restricted_schwarz.hpp>

#include <vector>

// max |(L L^T - A(perm, perm))_ij| over all entries, dense (n of a few hundred)
template <typename Csr, typename Perm>
double factor_identity_error(const Csr &A, const Csr &L, const Perm &perm)
{
    const size_t n = A.get_size()[0];
    std::vector<double> D(n * n, 0.0), Ld(n * n, 0.0);
    std::vector<long> inv(n);
    for (size_t i = 0; i < n; ++i) inv[(size_t)perm.get_const_permutation()[i]] = (long)i;
    for (size_t i = 0; i < n; ++i)
        for (auto j = A.get_const_row_ptrs()[i]; j < A.get_const_row_ptrs()[i + 1]; ++j)
            D[(size_t)inv[i] * n + (size_t)inv[(size_t)A.get_const_col_idxs()[j]]] -= A.get_const_values()[j];
    for (size_t i = 0; i < n; ++i)
        for (auto j = L.get_const_row_ptrs()[i]; j < L.get_const_row_ptrs()[i + 1]; ++j)
            Ld[i * n + (size_t)L.get_const_col_idxs()[j]] = L.get_const_values()[j];
    double err = 0.0;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            double sum = D[i * n + j];
            for (size_t k = 0; k <= std::min(i, j); ++k) sum += Ld[i * n + k] * Ld[j * n + k];
            err = std::max(err, std::fabs(sum));
        }
    return err;
}

template <typename I, typename M>
int drive(int n, bool mixed, bool overlapped, double tol, int max_iters, bool from_csr, bool members = false)
{
    schwz::Settings settings("hip");
    schwz::Metadata<double, I> metadata;
    metadata.mpi_communicator = MPI_COMM_WORLD;
    MPI_Comm_rank(MPI_COMM_WORLD, &metadata.my_rank);
    MPI_Comm_size(MPI_COMM_WORLD, &metadata.comm_size);
    metadata.num_subdomains = metadata.comm_size;
    metadata.tolerance = tol;
    metadata.max_iters = max_iters;
    metadata.oned_laplacian_size = n;
    metadata.local_solver_tolerance = 1e-12;
    metadata.local_precond = "null";
    metadata.local_max_iters = -1;
    settings.explicit_laplacian = true;
    settings.use_mixed_precision = mixed;
    settings.convergence_settings.enable_global_check = true;
    settings.local_solver = schwz::Settings::local_solver_settings::iterative_solver_ginkgo;
    if (members) settings.local_solver = schwz::Settings::local_solver_settings::direct_solver_ginkgo;
    if (overlapped) {
        settings.comm_settings.enable_onesided = true;
        settings.comm_settings.enable_overlap = true;
        settings.convergence_settings.enable_decentralized_leader_election = true;
    }
    schwz::SolverRAS<double, I, M> solver(settings, metadata);
    if (from_csr) {
        // 5-point Laplacian, diagonal 4, natural ordering (initialization.cpp:227-265)
        const I N = (I)n * (I)n;
        std::vector<I> rp((size_t)N + 1, 0), col;
        std::vector<double> val, rhs((size_t)N);
        for (I r = 0; r < N; ++r) {
            const I x = r % n, y = r / n;
            if (y > 0) { col.push_back(r - n); val.push_back(-1.0); }
            if (x > 0) { col.push_back(r - 1); val.push_back(-1.0); }
            col.push_back(r); val.push_back(4.0);
            if (x < n - 1) { col.push_back(r + 1); val.push_back(-1.0); }
            if (y < n - 1) { col.push_back(r + n); val.push_back(-1.0); }
            rp[(size_t)r + 1] = (I)col.size();
            rhs[(size_t)r] = 1.0 + (double)(r % 7);
        }
        settings.explicit_laplacian = false;
        solver.initialize(N, rp.data(), col.data(), val.data(), rhs.data());
    } else {
        solver.initialize();
    }
    std::shared_ptr<gko::matrix::Dense<double>> solution;
    solver.run(solution);
    if (metadata.my_rank == 0) {
        double sq = 0.0;
        for (gko::size_type i = 0; i < solution->get_size()[0]; ++i) sq += solution->at(i) * solution->at(i);
        std::cout.precision(17);
        std::cout << "RESULT iters=" << metadata.iter_count << " solnorm=" << std::sqrt(sq) << std::endl;
    }
    if (members) {
        const auto &A = *solver.local_matrix;
        const auto &G = *solver.interface_matrix;
        const auto &L = *solver.triangular_factor_l;
        const auto &U = *solver.triangular_factor_u;
        const size_t nl = A.get_size()[0];
        bool perm_ok = solver.local_perm->get_permutation_size() == nl;
        for (size_t i = 0; i < nl && perm_ok; ++i)
            perm_ok = (size_t)solver.local_inv_perm->get_const_permutation()[(size_t)solver.local_perm->get_const_permutation()[i]] == i;
        // U = L^T: same number of entries, and entry (i, j) of U is entry (j, i) of L
        bool ut_ok = U.get_num_stored_elements() == L.get_num_stored_elements();
        for (size_t i = 0; i < nl && ut_ok; ++i)
            for (auto j = U.get_const_row_ptrs()[i]; j < U.get_const_row_ptrs()[i + 1] && ut_ok; ++j) {
                const size_t c = (size_t)U.get_const_col_idxs()[j];
                bool found = false;
                for (auto k = L.get_const_row_ptrs()[c]; k < L.get_const_row_ptrs()[c + 1]; ++k)
                    if ((size_t)L.get_const_col_idxs()[k] == i) found = L.get_const_values()[k] == U.get_const_values()[j];
                ut_ok = found;
            }
        // interior rows of local_solution against the assembled solution (rank 0 holds it: broadcast)
        std::vector<double> whole((size_t)metadata.global_size);
        if (metadata.my_rank == 0)
            for (size_t i = 0; i < whole.size(); ++i) whole[i] = solution->at(i);
        MPI_Bcast(whole.data(), (int)whole.size(), MPI_DOUBLE, 0, MPI_COMM_WORLD);
        double sol_err = 0.0;
        const size_t first = (size_t)metadata.first_row[(size_t)metadata.my_rank];
        for (size_t i = 0; i < (size_t)metadata.local_size; ++i)
            sol_err = std::max(sol_err, std::fabs(solver.local_solution->at(i) - whole[first + i]));
        // interface entries: columns are global ids outside this subdomain's rows, only overlap rows have any
        bool iface_ok = (size_t)G.get_const_row_ptrs()[(size_t)metadata.local_size] == 0;
 
        // one write per rank: the ranks share the terminal
        std::ostringstream line;
        line.precision(6);
        line << "MEMBERS rank=" << metadata.my_rank << " local_n=" << nl << " local_size_x=" << metadata.local_size_x
                  << " local_nnz=" << A.get_num_stored_elements() << " iface_nnz=" << G.get_num_stored_elements()
                  << " iface_ok=" << iface_ok << " factor_err=" << factor_identity_error(A, L, *solver.local_perm)
                  << " perm_ok=" << perm_ok << " ut_ok=" << ut_ok << " sol_err=" << sol_err
                  << " global_null=" << (!solver.global_matrix && !solver.global_rhs)
                  << " global_solution=" << (solver.global_solution ? 1 : 0)
                  << " hist=" << solver.local_residual_vector_out.size() << "/" << solver.global_residual_vector_out.size()
                  << " rhs0=" << solver.local_rhs->at(0) << "\n";
        std::cout << line.str() << std::flush;
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        std::cerr << "usage: ras_types_driver d32d|d32f|d64d|d64f n mixed overlapped tol max_iters" << std::endl;
        return 2;
    }
    MPI_Init(&argc, &argv);
    const std::string types = argv[1];
    const int n = std::atoi(argv[2]);
    const bool mixed = std::atoi(argv[3]) != 0, overlapped = std::atoi(argv[4]) != 0;
    const double tol = std::atof(argv[5]);
    const int max_iters = std::atoi(argv[6]);
    const bool from_csr = argc > 7 && std::string(argv[7]) == "csr";
    const bool members = argc > 7 && std::string(argv[7]) == "members";
    int rc = 2;
    try {
        if (types == "d32d") rc = drive<gko::int32, double>(n, mixed, overlapped, tol, max_iters, from_csr, members);
        if (types == "d32f") rc = drive<gko::int32, float>(n, mixed, overlapped, tol, max_iters, from_csr, members);
        if (types == "d64d") rc = drive<gko::int64, double>(n, mixed, overlapped, tol, max_iters, from_csr, members);
        if (types == "d64f") rc = drive<gko::int64, float>(n, mixed, overlapped, tol, max_iters, from_csr, members);
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        rc = 1;
    }
    MPI_Finalize();
    return rc;
}
