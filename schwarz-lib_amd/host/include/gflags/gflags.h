// Minimal stand-in for <gflags/gflags.h>: exactly the surface
// benchmarking/bench_base.hpp:50-153 uses (DEFINE_bool/int32/uint32/double/string,
// SetUsageMessage, ParseCommandLineFlags).  The reference downloads gflags from GitHub at
// configure time (third_party/gflags/CMakeLists.txt:1-7); there is no network here.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

namespace gflags {
namespace detail {

struct Flag {
    enum Kind { kBool, kInt32, kUint32, kDouble, kString } kind;
    void *ptr;
    std::string help;
};

inline std::map<std::string, Flag> &registry()
{
    static std::map<std::string, Flag> r;
    return r;
}

inline std::string &usage()
{
    static std::string u;
    return u;
}

struct Registerer {
    Registerer(const char *name, Flag::Kind kind, void *ptr, const char *help)
    {
        registry()[name] = Flag{kind, ptr, help};
    }
};

inline bool assign(const Flag &f, const std::string &name, const std::string &value)
{
    try {
        switch (f.kind) {
        case Flag::kBool: {
            bool v;
            if (value == "true" || value == "1" || value == "yes" || value == "t" || value == "y")
                v = true;
            else if (value == "false" || value == "0" || value == "no" || value == "f" || value == "n")
                v = false;
            else
                return false;
            *static_cast<bool *>(f.ptr) = v;
            return true;
        }
        case Flag::kInt32:
            *static_cast<std::int32_t *>(f.ptr) = static_cast<std::int32_t>(std::stol(value));
            return true;
        case Flag::kUint32:
            *static_cast<std::uint32_t *>(f.ptr) = static_cast<std::uint32_t>(std::stoul(value));
            return true;
        case Flag::kDouble:
            *static_cast<double *>(f.ptr) = std::stod(value);
            return true;
        case Flag::kString:
            *static_cast<std::string *>(f.ptr) = value;
            return true;
        }
    } catch (...) {
    }
    (void)name;
    return false;
}

}  // namespace detail

inline void SetUsageMessage(const std::string &msg) { detail::usage() = msg; }

// Accepts --name=value, --name value, -name..., --name / --noname for booleans, and stops at
// "--".  With remove_flags the parsed flags are removed from argv.  Unknown flags are fatal,
// as in gflags.
inline std::uint32_t ParseCommandLineFlags(int *argc, char ***argv, bool remove_flags)
{
    auto &reg = detail::registry();
    int out = 1;
    char **av = *argv;
    int i = 1;
    for (; i < *argc; ++i) {
        std::string arg = av[i];
        if (arg == "--") {
            ++i;
            break;
        }
        if (arg.size() < 2 || arg[0] != '-') {
            av[out++] = av[i];
            continue;
        }
        std::string body = arg.substr(arg[1] == '-' ? 2 : 1);
        std::string name = body, value;
        bool has_value = false;
        auto eq = body.find('=');
        if (eq != std::string::npos) {
            name = body.substr(0, eq);
            value = body.substr(eq + 1);
            has_value = true;
        }
        if (name == "help" || name == "helpshort") {
            std::cout << detail::usage() << "\n";
            for (auto &kv : reg) std::cout << "  --" << kv.first << "  " << kv.second.help << "\n";
            std::exit(0);
        }
        auto it = reg.find(name);
        if (it == reg.end() && name.compare(0, 2, "no") == 0) {
            auto it2 = reg.find(name.substr(2));
            if (it2 != reg.end() && it2->second.kind == detail::Flag::kBool && !has_value) {
                *static_cast<bool *>(it2->second.ptr) = false;
                continue;
            }
        }
        if (it == reg.end()) {
            std::cerr << "ERROR: unknown command line flag '" << name << "'" << std::endl;
            std::exit(1);
        }
        if (!has_value) {
            if (it->second.kind == detail::Flag::kBool) {
                *static_cast<bool *>(it->second.ptr) = true;
                continue;
            }
            if (i + 1 >= *argc) {
                std::cerr << "ERROR: flag '" << name << "' is missing its argument" << std::endl;
                std::exit(1);
            }
            value = av[++i];
        }
        if (!detail::assign(it->second, name, value)) {
            std::cerr << "ERROR: illegal value '" << value << "' specified for flag '" << name << "'"
                      << std::endl;
            std::exit(1);
        }
    }
    for (; i < *argc; ++i) av[out++] = av[i];
    if (remove_flags) {
        *argc = out;
        av[out] = nullptr;
    }
    return static_cast<std::uint32_t>(out);
}

}  // namespace gflags

#define SCHWZ_GFLAGS_DEFINE(type, kind, name, val, txt)                                         \
    type FLAGS_##name = val;                                                                   \
    static ::gflags::detail::Registerer schwz_gflags_reg_##name(#name, ::gflags::detail::Flag::kind, \
                                                                &FLAGS_##name, txt)

#define DEFINE_bool(name, val, txt) SCHWZ_GFLAGS_DEFINE(bool, kBool, name, val, txt)
#define DEFINE_int32(name, val, txt) SCHWZ_GFLAGS_DEFINE(std::int32_t, kInt32, name, val, txt)
#define DEFINE_uint32(name, val, txt) SCHWZ_GFLAGS_DEFINE(std::uint32_t, kUint32, name, val, txt)
#define DEFINE_double(name, val, txt) SCHWZ_GFLAGS_DEFINE(double, kDouble, name, val, txt)
#define DEFINE_string(name, val, txt) SCHWZ_GFLAGS_DEFINE(std::string, kString, name, val, txt)
