#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side product library (libmasp_host: circuits, Jubjub / BLAKE2s /
# Pedersen natives, Groth16 host verification incl. the subgroup tests) and over the oracle, driven by the CPU tests that use
# them (SURVEY.md §5 suggested sanitizer runs for the host side).  usage: tools/sanitize_host.sh > profiles/rNN_sanitizers.txt
set -e
cd "$(dirname "$0")/.."
out=/tmp/masp_sanitize; mkdir -p $out
FLAGS="-O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ $FLAGS -Wall -Wno-unused-function masp_amd/csrc/host/host_api.cpp -o $out/libmasp_host.so
asan=$(g++ -print-file-name=libasan.so)
echo "# built $out/libmasp_host.so with: g++ $FLAGS ; preloading $asan"
# (leak checking off: CPython itself never frees everything; ASan still reports every invalid access and UBSan aborts on any UB)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$asan MASP_HOST_LIBRARY=$out/libmasp_host.so python -m pytest -q -m "not gpu" -p no:cacheprovider \
    tests/test_circuits.py tests/test_host_api.py tests/test_host_fast_merkle.py tests/test_binding_sig.py tests/test_subgroup_checks.py tests/test_pairing_program.py tests/test_params.py 2>&1 | tail -5
# the oracle (test infrastructure) the same way: its own build flags plus the sanitizers, into a scratch directory
mkdir -p $out/oracle
g++ -O1 -g -march=x86-64-v3 -std=c++17 -fPIC -pthread -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined oracle/groth16_oracle.cpp -o $out/oracle/liboracle.so
echo "# built $out/oracle/liboracle.so"
LD_PRELOAD=$asan MASP_ORACLE_LIBRARY=$out/oracle/liboracle.so python -m pytest -q -m "not gpu" -p no:cacheprovider tests/test_oracle_field_curve.py tests/test_oracle_groth16.py 2>&1 | tail -3
