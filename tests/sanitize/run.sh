#!/bin/bash
# Builds tests/sanitize/host_san.cpp + oracle/rt_oracle.c with AddressSanitizer and UndefinedBehaviorSanitizer (g++/gcc) and
# runs it; builds tests/sanitize/hostpar_san.cpp (the host threading of rt_tracks_create and of the pipelined fetch:
# raytracing.jl_amd/csrc/rt_hostpar.hpp, device copies replaced by memcpy) once with ASan + UBSan and once with ThreadSanitizer and
# runs both; runs host_san (and tests/sanitize/march_san.hip: the device geometry header on the host, hipcc host pass) over the fixtures, N seeded fuzz meshes and a set of malformed files.  usage: bash tests/sanitize/run.sh [N=200] [logfile]
set -eu
cd "$(dirname "$0")/../.."
N=${1:-200}
LOG=${2:-/dev/stdout}
OUT=tests/build/sanitize
mkdir -p $OUT/meshes
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -ffp-contract=off"
gcc $SAN -c oracle/rt_oracle.c -o $OUT/rt_oracle.o
g++ $SAN -std=c++17 -Wall -Wextra -Wno-unused-function -c tests/sanitize/host_san.cpp -o $OUT/host_san.o
g++ $SAN -o $OUT/host_san $OUT/host_san.o $OUT/rt_oracle.o -lm
# the device march's geometry header on the host under the same sanitizers (hipcc = clang, host pass only)
CLANG=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc --cuda-host-only $SAN -std=c++17 -Wno-unused-function -pthread -x hip -c tests/sanitize/march_san.hip -o $OUT/march_san.o
$CLANG/clang $SAN -c oracle/rt_oracle.c -o $OUT/rt_oracle_clang.o
$CLANG/clang++ $SAN -pthread -o $OUT/march_san $OUT/march_san.o $OUT/rt_oracle_clang.o -lm
g++ $SAN -std=c++17 -Wall -Wextra -pthread -o $OUT/hostpar_asan tests/sanitize/hostpar_san.cpp
g++ -fsanitize=thread -fno-omit-frame-pointer -g -O1 -std=c++17 -Wall -Wextra -pthread -o $OUT/hostpar_tsan tests/sanitize/hostpar_san.cpp
rm -f $OUT/meshes/*
python3 tests/sanitize/make_inputs.py $OUT/meshes $N
{
  echo "# $(g++ --version | head -1); flags: $SAN"
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $OUT/host_san raytracing.jl_amd/data/pincell.msh \
      raytracing.jl_amd/data/pincell.json raytracing.jl_amd/data/bwr_like.msh $OUT/meshes/*
  echo "exit code: $?"
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $OUT/march_san raytracing.jl_amd/data/pincell.msh \
      raytracing.jl_amd/data/bwr_like.msh $OUT/meshes/fuzz_*.json
  echo "exit code: $?"
  echo "# rt_hostpar.hpp under -fsanitize=address,undefined"
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $OUT/hostpar_asan
  echo "exit code: $?"
  echo "# rt_hostpar.hpp under -fsanitize=thread"
  TSAN_OPTIONS=halt_on_error=1 $OUT/hostpar_tsan
  echo "exit code: $?"
} > "$LOG" 2>&1
grep -E "host_san:|march_san:|hostpar_san:|exit code|ERROR|runtime error|WARNING: ThreadSanitizer" "$LOG"
