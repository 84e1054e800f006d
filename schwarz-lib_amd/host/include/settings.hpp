// schwz::Settings / schwz::Metadata -- the structs the driver fills from its flags
// (benchmarking/bench_ras.cpp:50-150).  Field names, types and defaults follow the reference's
// include/settings.hpp:77-496 so that driver code written against it compiles unchanged; members
// that only made sense for the replicated-global-matrix design (N-long index maps) are absent.
#pragma once

#include <mpi.h>

#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include <ginkgo/ginkgo.hpp>

#define MINIMAL_OVERLAP 2

namespace schwz {

struct Settings {
    std::string executor_string;
    std::shared_ptr<gko::Executor> executor = gko::ReferenceExecutor::create();

    enum partition_settings {
        partition_regular = 0x0,
        partition_regular2d = 0x4,
        partition_metis = 0x1,
        partition_zoltan = 0x2,
        partition_custom = 0x3
    };
    partition_settings partition = partition_settings::partition_regular;

    gko::int32 overlap = MINIMAL_OVERLAP;
    std::string matrix_filename = "null";
    bool explicit_laplacian = true;
    bool use_mixed_precision = false;
    bool enable_random_rhs = false;
    bool print_matrices = false;
    bool debug_print = false;

    enum local_solver_settings {
        direct_solver_cholmod = 0x0,
        direct_solver_umfpack = 0x5,
        direct_solver_ginkgo = 0x1,
        iterative_solver_ginkgo = 0x2,
        iterative_solver_dealii = 0x3,
        solver_custom = 0x4
    };
    local_solver_settings local_solver = local_solver_settings::iterative_solver_ginkgo;

    bool non_symmetric_matrix = false;
    unsigned int restart_iter = 1u;
    int reset_local_crit_iter = -1;
    bool naturally_ordered_factor = false;
    std::string metis_objtype;
    bool use_precond = false;
    bool write_debug_out = false;
    bool write_iters_and_residuals = false;
    bool enable_logging = false;
    bool write_perm_data = false;
    int shifted_iter = 1;

    struct comm_settings {
        bool enable_onesided = false;
        bool enable_overlap = false;
        bool enable_put = false;
        bool enable_get = true;
        bool stage_through_host = false;
        bool enable_one_by_one = false;
        bool enable_flush_local = false;
        bool enable_flush_all = true;
        bool enable_lock_local = false;
        bool enable_lock_all = true;
    };
    comm_settings comm_settings;

    struct convergence_settings {
        bool put_all_local_residual_norms = true;
        bool enable_global_simple_tree = false;
        bool enable_decentralized_leader_election = false;
        bool enable_global_check = true;
        bool enable_accumulate = false;
        bool enable_global_check_iter_offset = false;
        enum local_convergence_crit { residual_based = 0x0, solution_based = 0x1 };
        local_convergence_crit convergence_crit = local_convergence_crit::solution_based;
    };
    convergence_settings convergence_settings;

    std::string factorization = "cholmod";
    std::string reorder;

    Settings(std::string executor_string = "reference") : executor_string(std::move(executor_string)) {}
};


template <typename ValueType, typename IndexType>
struct Metadata {
    MPI_Comm mpi_communicator;
    gko::size_type global_size = 0;
    gko::size_type oned_laplacian_size = 0;
    gko::size_type local_size = 0;
    gko::size_type local_size_x = 0;
    gko::size_type local_size_o = 0;
    gko::size_type overlap_size = 0;
    gko::size_type num_subdomains = 1;
    int my_rank = 0;
    int my_local_rank = 0;
    int local_num_procs = 1;
    int comm_size = 1;
    int num_threads = 1;
    IndexType iter_count = 0;
    ValueType tolerance = 0;
    ValueType local_solver_tolerance = 0;
    IndexType max_iters = 0;
    IndexType local_max_iters = -1;
    IndexType updated_max_iters = -1;
    std::string local_precond = "null";
    unsigned int precond_max_block_size = 16;
    ValueType current_residual_norm = -1.0;
    ValueType min_residual_norm = -1.0;

    // (id, rank, count, name, samples): filled by MEASURE_ELAPSED_FUNC_TIME in the reference
    std::vector<std::tuple<int, int, int, std::string, std::vector<ValueType>>> time_struct;
    // per subdomain: (rank, [(from, n)], [(to, n)], num_in, num_out)
    std::vector<std::tuple<int, std::vector<std::tuple<int, int>>, std::vector<std::tuple<int, int>>, int, int>>
        comm_data_struct;

    struct post_process_data {
        std::vector<std::vector<ValueType>> global_residual_vector_out;
        std::vector<ValueType> local_residual_vector_out;
        std::vector<ValueType> local_converged_iter_count;
        std::vector<ValueType> local_converged_resnorm;
        std::vector<ValueType> local_timestamp;
    };
    post_process_data post_process_data;
    double init_mpi_wtime = 0.0;

    // contiguous ownership ranges, length num_subdomains+1 (the reference's first_row array)
    std::vector<long long> first_row;
    // new->old permutation when a partition vector was applied (empty otherwise)
    std::vector<long long> permutation;
};

}  // namespace schwz
