#!/bin/bash
# The CPU test suite against AddressSanitizer + UndefinedBehaviorSanitizer builds of the two CPU libraries (the oracle and the
# offline host library). CPU only: nothing here touches a GPU, and the HIP library is never built with a sanitizer.
# The reference's counterpart is -DUSE_SANITIZERS (CMakeLists.txt:28-30). usage: tools/run_sanitized_tests.sh [pytest args]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/oracle asan
make -C $R/dint_amd/csrc host-asan
export DINT_HOST_LIB=$R/dint_amd/libdint_host_asan.so DINT_ORACLE_LIB=$R/oracle/liboracle_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
# (python itself leaks by design at exit: leak checking off; everything else aborts the run)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd $R && exec python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
