#!/bin/bash
# AddressSanitizer + UBSan over the CPU code of this repo (GPU sanitizers are not available on the
# pool): the C oracle and the host-side setup of libschwz_hip.so (host_setup.cpp compiled alone).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -fPIC -std=c99 -D_POSIX_C_SOURCE=200809L \
    -shared -o /tmp/libschwz_oracle_asan.so $ROOT/oracle/schwz_oracle.c -lm
g++ -O1 -g -std=c++17 -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include \
    -I$ROOT/schwarz-lib_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -shared -o /tmp/libhost_asan.so \
    $ROOT/schwarz-lib_amd/csrc/host_setup.cpp
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
python3 $ROOT/tools/sanitize/oracle_paths.py | tail -1
python3 $ROOT/tools/sanitize/host_paths.py | tail -1
