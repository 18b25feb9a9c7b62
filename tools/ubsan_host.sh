#!/bin/bash
# The library's host side under UndefinedBehaviorSanitizer (+ float-cast-overflow) on the GPU box, like tools/asan_host.sh.
R=${GRAFT_REPO_ROOT:-$PWD}
LIB=$R/rafft_amd/libraffthip_ubsan.so
if [ ! -f $LIB ]; then
  (cd $R/rafft_amd/csrc && /opt/rocm/bin/hipcc -O3 -g --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-missing-braces \
     -fsanitize=undefined,float-cast-overflow -fno-gpu-sanitize -shared-libsan rafft_api.hip -o $LIB) || exit 1
fi
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
mkdir -p $R/gpurun_out
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$R/gpurun_out/ubsan
export RAFFT_LIB=$LIB
cd $R
LD_PRELOAD=$RT timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $R/gpurun_out/ubsan_suite.log 2>&1
echo "suite rc=$?" >> $R/gpurun_out/ubsan_suite.log
LD_PRELOAD=$RT timeout -k 10 300 python3 tools/stress_scheduler.py >> $R/gpurun_out/ubsan_suite.log 2>&1
echo "stress rc=$?" >> $R/gpurun_out/ubsan_suite.log
ls -la $R/gpurun_out/ubsan.* 2>/dev/null | head
