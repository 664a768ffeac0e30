#!/bin/bash
# Build the host test harness (g++, no GPU involved).
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off "$HERE/host_check.cpp" -o "$HERE/libhostcheck.so"
