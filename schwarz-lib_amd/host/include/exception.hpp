// Error convention of the mirrored API: exceptions derived from the global-namespace
// `Error : std::exception` (reference: include/exception.hpp:42-210).  HipError takes the place
// of CudaError; the C ABI's status codes are turned into these by the host layer.
#pragma once

#include <exception>
#include <string>

class Error : public std::exception {
public:
    Error(const std::string &file, int line, const std::string &what)
        : what_(file + ":" + std::to_string(line) + ": " + what)
    {}
    const char *what() const noexcept override { return what_.c_str(); }

private:
    std::string what_;
};

class NotImplemented : public Error {
public:
    NotImplemented(const std::string &file, int line, const std::string &func)
        : Error(file, line, func + " is not implemented")
    {}
};

class ModuleNotImplemented : public Error {
public:
    ModuleNotImplemented(const std::string &file, int line, const std::string &module,
                         const std::string &func)
        : Error(file, line, func + " is not implemented for the module " + module)
    {}
};

class BadDimension : public Error {
public:
    BadDimension(const std::string &file, int line, const std::string &func, const std::string &what)
        : Error(file, line, func + ": " + what)
    {}
};

class HipError : public Error {
public:
    HipError(const std::string &file, int line, const std::string &func, const std::string &what)
        : Error(file, line, func + ": HIP error: " + what)
    {}
};

class MetisError : public Error {
public:
    MetisError(const std::string &file, int line, const std::string &func, int code)
        : Error(file, line, func + ": METIS error " + std::to_string(code))
    {}
};

#define SCHWARZ_NOT_IMPLEMENTED                               \
    {                                                         \
        throw ::NotImplemented(__FILE__, __LINE__, __func__); \
    }
