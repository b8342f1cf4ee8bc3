#!/bin/bash
# tools/sanitize.sh — sanitizers on the CPU-side code (never on the GPU box: GPU ASan / XNACK are not available on the pool).
#  1. oracle/pt_oracle.c built with -fsanitize=address,undefined (both math modes) and the whole CPU test suite run against it;
#  1b. csrc/pt_objload.cpp (the native OBJ parser: untrusted input) with -fsanitize=address,undefined under tests/test_objloader.py and tools/fuzz_objloader.py;
#  2. the C++ facade (csrc/SampleRenderer.h through examples/facade_demo.cpp) compiled -fsanitize=undefined -Wall -Wextra -Werror;
#  3. include/pt_amd.h compiled as C99 and C++17 with -Wall -Wextra -pedantic -Werror.
# Log: profiles/r6_02_sanitizers.log (round 5: profiles/r5_04_sanitizers.log)
set -e
cd "$(dirname "$0")/.."
LOG=${LOG:-profiles/r6_02_sanitizers.log}
: > $LOG
make -C oracle asan 2>&1 | tee -a $LOG
ASAN_LIB=$(gcc -print-file-name=libasan.so)
echo "== pytest -m 'not gpu' against oracle/asan (LD_PRELOAD=$ASAN_LIB)" | tee -a $LOG
ORC_LIB_DIR=$PWD/oracle/asan LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15 | tee -a $LOG
echo "== csrc/pt_objload.cpp (pt_load_obj, the native OBJ parser) with -fsanitize=address,undefined: the pinned loader tests, then a mutation fuzz" | tee -a $LOG
g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wall -Wextra -Werror \
  optixpathtracer_amd/csrc/pt_objload.cpp -o /tmp/libptobj_asan.so 2>&1 | tee -a $LOG
PT_OBJ_LIB=/tmp/libptobj_asan.so LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_objloader.py -x -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $LOG
PT_OBJ_LIB=/tmp/libptobj_asan.so LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python tools/fuzz_objloader.py 600 2>&1 | tail -3 | tee -a $LOG
echo "== facade: g++ -fsanitize=undefined -Wall -Wextra -Werror" | tee -a $LOG
g++ -std=c++17 -O1 -g -fsanitize=undefined -Wall -Wextra -Werror -I. -Iinclude examples/facade_demo.cpp -Loptixpathtracer_amd -lptamd -Wl,-rpath,$PWD/optixpathtracer_amd -o /tmp/facade_demo_ubsan 2>&1 | tee -a $LOG
echo "facade_demo compiled and linked (it needs a GPU to run: tests/test_cabi.py::test_cxx_facade_demo_matches_python runs the plain build on the GPU box)" | tee -a $LOG
echo "== include/pt_amd.h as C99 and C++17, -Wall -Wextra -pedantic -Werror" | tee -a $LOG
echo '#include "pt_amd.h"' > /tmp/hdr_check.c
gcc -std=c99 -Wall -Wextra -pedantic -Werror -Iinclude -c /tmp/hdr_check.c -o /tmp/hdr_check_c.o 2>&1 | tee -a $LOG
g++ -std=c++17 -x c++ -Wall -Wextra -pedantic -Werror -Iinclude -c /tmp/hdr_check.c -o /tmp/hdr_check_cpp.o 2>&1 | tee -a $LOG
echo "headers ok" | tee -a $LOG
