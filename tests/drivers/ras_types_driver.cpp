// Test driver for the template instantiations of the C++ mirror that the reference's bench_ras
// does not reach (it fixes <double, int, double>): MixedValueType = float with
// settings.use_mixed_precision (fp32 halos on the wire, restricted_schwarz.cpp:898-903,929-933,
// 952-954) and IndexType = int64.  Usage under mpiexec:
//   ras_types_driver <types: d32d|d32f|d64d|d64f> <grid edge n> <mixed 0|1> <overlapped 0|1> <tol> <max_iters> [csr]
// Prints what SolverRAS::run prints plus one line "RESULT iters=<k> solnorm=<|x|_2>" on rank 0.
// With a seventh argument "csr" the 2-D Laplacian is assembled here and handed over through
// initialize(num_rows, row_ptrs, col_idxs, values, rhs) with rhs_i = 1 + (i mod 7) -- the
// deal.II-free analogue of the reference's initialize(dealii::SparseMatrix, dealii::Vector).
#include <mpi.h>

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>

#include <restricted_schwarz.hpp>

#include <vector>

template <typename I, typename M>
int drive(int n, bool mixed, bool overlapped, double tol, int max_iters, bool from_csr)
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
    int rc = 2;
    try {
        if (types == "d32d") rc = drive<gko::int32, double>(n, mixed, overlapped, tol, max_iters, from_csr);
        if (types == "d32f") rc = drive<gko::int32, float>(n, mixed, overlapped, tol, max_iters, from_csr);
        if (types == "d64d") rc = drive<gko::int64, double>(n, mixed, overlapped, tol, max_iters, from_csr);
        if (types == "d64f") rc = drive<gko::int64, float>(n, mixed, overlapped, tol, max_iters, from_csr);
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        rc = 1;
    }
    MPI_Finalize();
    return rc;
}
