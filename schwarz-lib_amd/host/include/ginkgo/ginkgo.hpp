// Minimal stand-in for <ginkgo/ginkgo.hpp>: the part of the gko:: surface that the mirrored
// public headers and benchmarking/bench_ras.cpp touch.  Ginkgo itself
// (un-pinned branch expt-develop, .github/workflows/main.yml:30-32) is not available here and
// none of its arithmetic is used: all device work goes through include/schwz_hip.h.
#pragma once

// the real ginkgo.hpp pulls these in transitively and driver code relies on it
// (bench_base.hpp uses std::sort / std::ostringstream without including their headers)
#include <algorithm>
#include <array>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <string>
#include <tuple>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <vector>

namespace gko {

using size_type = std::size_t;
using int32 = std::int32_t;
using int64 = std::int64_t;
using default_precision = double;

template <size_type N>
struct dim {
    std::array<size_type, N> v{};
    dim() = default;
    explicit dim(size_type n) { v.fill(n); }
    dim(size_type r, size_type c)
    {
        static_assert(N == 2, "two-argument constructor is for dim<2>");
        v[0] = r;
        v[1] = c;
    }
    size_type &operator[](size_type i) { return v[i]; }
    const size_type &operator[](size_type i) const { return v[i]; }
};

// Executors only tag where a host-side object lives; device memory is owned by libschwz_hip.
class Executor : public std::enable_shared_from_this<Executor> {
public:
    virtual ~Executor() = default;
    virtual std::shared_ptr<Executor> get_master() { return shared_from_this(); }
    virtual const char *name() const = 0;
};

class ReferenceExecutor : public Executor {
public:
    static std::shared_ptr<ReferenceExecutor> create() { return std::make_shared<ReferenceExecutor>(); }
    const char *name() const override { return "reference"; }
};

class OmpExecutor : public Executor {
public:
    static std::shared_ptr<OmpExecutor> create() { return std::make_shared<OmpExecutor>(); }
    const char *name() const override { return "omp"; }
};

// marks "--executor=hip": host-side handle of the GPU the rank is bound to
class HipExecutor : public Executor {
public:
    HipExecutor(int device, std::shared_ptr<Executor> master) : device_(device), master_(std::move(master)) {}
    static std::shared_ptr<HipExecutor> create(int device, std::shared_ptr<Executor> master)
    {
        return std::make_shared<HipExecutor>(device, std::move(master));
    }
    std::shared_ptr<Executor> get_master() override { return master_; }
    int get_device_id() const { return device_; }
    const char *name() const override { return "hip"; }

private:
    int device_;
    std::shared_ptr<Executor> master_;
};

namespace matrix {

// host-resident dense column vector / matrix (row-major, stride = columns)
template <typename T = default_precision>
class Dense {
public:
    using value_type = T;
    static std::unique_ptr<Dense> create(std::shared_ptr<Executor> exec, dim<2> size = dim<2>(0, 0))
    {
        return std::unique_ptr<Dense>(new Dense(std::move(exec), size));
    }
    dim<2> get_size() const { return size_; }
    T *get_values() { return data_.data(); }
    const T *get_const_values() const { return data_.data(); }
    size_type get_num_stored_elements() const { return data_.size(); }
    T &at(size_type r, size_type c = 0) { return data_[r * size_[1] + c]; }
    const T &at(size_type r, size_type c = 0) const { return data_[r * size_[1] + c]; }
    std::shared_ptr<Executor> get_executor() const { return exec_; }
    void copy_from(const Dense *other)
    {
        size_ = other->size_;
        data_ = other->data_;
    }

private:
    Dense(std::shared_ptr<Executor> exec, dim<2> size)
        : exec_(std::move(exec)), size_(size), data_(size[0] * size[1], T{})
    {}
    std::shared_ptr<Executor> exec_;
    dim<2> size_;
    std::vector<T> data_;
};

// host-resident CSR matrix: the arrays and the accessors of gko::matrix::Csr a caller of the solver class reads
// (the public matrix members of SchwarzBase, include/schwarz_base.hpp:137-167); no arithmetic
template <typename T = default_precision, typename I = int32>
class Csr {
public:
    using value_type = T;
    using index_type = I;
    static std::unique_ptr<Csr> create(std::shared_ptr<Executor> exec, dim<2> size = dim<2>(0, 0), size_type nnz = 0)
    {
        return std::unique_ptr<Csr>(new Csr(std::move(exec), size, nnz));
    }
    dim<2> get_size() const { return size_; }
    size_type get_num_stored_elements() const { return vals_.size(); }
    T *get_values() { return vals_.data(); }
    const T *get_const_values() const { return vals_.data(); }
    I *get_col_idxs() { return cols_.data(); }
    const I *get_const_col_idxs() const { return cols_.data(); }
    I *get_row_ptrs() { return rows_.data(); }
    const I *get_const_row_ptrs() const { return rows_.data(); }
    std::shared_ptr<Executor> get_executor() const { return exec_; }

private:
    Csr(std::shared_ptr<Executor> exec, dim<2> size, size_type nnz)
        : exec_(std::move(exec)), size_(size), rows_(size[0] + 1, I{}), cols_(nnz, I{}), vals_(nnz, T{})
    {}
    std::shared_ptr<Executor> exec_;
    dim<2> size_;
    std::vector<I> rows_, cols_;
    std::vector<T> vals_;
};

// row permutation as an index array (gko::matrix::Permutation)
template <typename I = int32>
class Permutation {
public:
    using index_type = I;
    static std::unique_ptr<Permutation> create(std::shared_ptr<Executor> exec, size_type n = 0)
    {
        return std::unique_ptr<Permutation>(new Permutation(std::move(exec), n));
    }
    dim<2> get_size() const { return dim<2>(perm_.size(), perm_.size()); }
    size_type get_permutation_size() const { return perm_.size(); }
    I *get_permutation() { return perm_.data(); }
    const I *get_const_permutation() const { return perm_.data(); }
    std::shared_ptr<Executor> get_executor() const { return exec_; }

private:
    Permutation(std::shared_ptr<Executor> exec, size_type n) : exec_(std::move(exec)), perm_(n, I{}) {}
    std::shared_ptr<Executor> exec_;
    std::vector<I> perm_;
};

}  // namespace matrix

template <typename T>
inline T *lend(const std::shared_ptr<T> &p)
{
    return p.get();
}
template <typename T>
inline T *lend(const std::unique_ptr<T> &p)
{
    return p.get();
}
template <typename T>
inline std::shared_ptr<T> share(std::unique_ptr<T> &&p)
{
    return std::shared_ptr<T>(std::move(p));
}

}  // namespace gko
