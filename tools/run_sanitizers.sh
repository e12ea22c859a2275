#!/bin/bash
# CPU-side AddressSanitizer + UBSan run of the oracle and the GPU-free host library, ThreadSanitizer on the chunked reader (GPU ASan is not
# available on this pool).  Usage: tools/run_sanitizers.sh
set -e
cd "$(dirname "$0")/.."
make -C oracle liboracle_asan.so
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -o /tmp/libsbwthost_asan.so \
    sbwt_amd/csrc/host/host_capi.cpp -lz
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  SBWT_ORACLE_LIB=$PWD/oracle/liboracle_asan.so SBWT_HOST_LIB=/tmp/libsbwthost_asan.so \
  python -m pytest tests/test_oracle_golden.py tests/test_host.py tests/test_fuzz.py -x -q -p no:cacheprovider
# ThreadSanitizer on the chunked reader (its pieces are parsed by several threads and emitted in order)
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -Isbwt_amd/csrc/host -o /tmp/tsan_chunked_reader tests/cpp/tsan_chunked_reader.cpp -lz
/tmp/tsan_chunked_reader
