#!/bin/bash
# Shows that the delay-injection cases (tests/test_gpu_parity.py::test_k8_free_counts_do_not_depend_on_timing,
# tools/fuzz_k8.py) CATCH the round-4 race of K8's free counts: the kernel built with one block of counts again
# (-DK8_SINGLE_FREE, make race) must produce mismatches under ASGART_TEST_K8_DELAY=20000, the shipped kernel none.
#     tools/k8_race.sh [cases=60]          (on a GPU box; the race build is made on the CPU side: make -C asgart_amd/csrc race)
cd "$(dirname "$0")/.."
cases=${1:-60}
echo "== shipped kernel (double-buffered counts), delay 20000 =="
ASGART_TEST_K8_DELAY=20000 FUZZ_STALL_S=300 python3 tools/fuzz_k8.py "$cases" 0 | tail -3
echo "== single block of counts (-DK8_SINGLE_FREE), delay 0 =="
ASGART_LIB=asgart_amd/libasgart_hip_k8race.so FUZZ_STALL_S=300 python3 tools/fuzz_k8.py "$cases" 0 | tail -3
echo "== single block of counts (-DK8_SINGLE_FREE), delay 20000: mismatches expected =="
ASGART_LIB=asgart_amd/libasgart_hip_k8race.so ASGART_TEST_K8_DELAY=20000 FUZZ_STALL_S=300 python3 tools/fuzz_k8.py "$cases" 0 | grep -c DIFFERS
ASGART_LIB=asgart_amd/libasgart_hip_k8race.so ASGART_TEST_K8_DELAY=20000 FUZZ_STALL_S=300 python3 tools/fuzz_k8.py "$cases" 0 | tail -2
