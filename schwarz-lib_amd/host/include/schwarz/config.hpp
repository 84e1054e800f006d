// Generated-file equivalent of include/config.hpp.in (the reference fills it with CMake).
#pragma once
#define SCHW_HAVE_METIS 0
#define SCHW_HAVE_CHOLMOD 0
#define SCHW_HAVE_UMFPACK 0
#define SCHW_HAVE_DEALII 0
#define SCHW_HAVE_CUDA 0
#define SCHW_HAVE_HWLOC 0
#define SCHW_HAVE_HIP 1
